"""GPU: the library inside a host process that runs the device with hipDeviceScheduleBlockingSync.

Round 3 recorded one stall (profiles/r03e_host_wait_and_cu_mask.txt): with that device flag set, a bench with ten driver
threads waiting in hipStreamSynchronize did not finish within 300 s. A drop-in library cannot choose its host's device
flags, so amdzk_init reads them (hipGetDeviceFlags) and, when the runtime's waits are the non-spinning kind, every host
wait of the library polls a completion event instead of entering the runtime (amdzk_ctx::wait_forced). Run ONCE, in a
fresh child process that sets the flag before any other HIP call: two proving contexts in flight (threads), lanes keys,
a k = 6 lookup circuit, bytes compared with the protocol oracle — under a 60-second limit."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r'''
import ctypes, faulthandler, os, sys, threading
faulthandler.dump_traceback_later(50, exit=True)   # a hang names the call it sits in before the parent's limit hits
hip = ctypes.CDLL("libamdhip64.so.7", mode=ctypes.RTLD_GLOBAL)
rc = hip.hipSetDeviceFlags(ctypes.c_uint(0x4))     # hipDeviceScheduleBlockingSync, before anything else touches HIP
assert rc == 0, "hipSetDeviceFlags -> %d" % rc
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as ge
import zkutil as zu, circuits, plonk_ref as PR
pkg = ge.load_package()
O = zu.Oracle()
fl = ctypes.c_uint(0)
assert hip.hipGetDeviceFlags(ctypes.byref(fl)) == 0 and (fl.value & 0x7) == 0x4, "device flags %x" % fl.value
tau = 0x1234567890ABCDEF1234567
c = circuits.lookup_circuit(pkg.plonk, 6, seed=5)
ctxs = [pkg.Context(0), pkg.Context(0)]
params = pkg.kzg.ParamsKZG.setup(ctxs[0], c.k, zu.fr_from_int(tau))
fixed = np.stack([zu.ints_to_fr(O, col) for col in c.fixed])
pks = [pkg.plonk.ProvingKey(cx, params, c.desc, fixed, c.assembly.mapping, zu.fr_from_int(77)) for cx in ctxs]
adv = np.stack([zu.ints_to_fr(O, col) for col in c.advice])
inst = [zu.ints_to_fr(O, col) if col else np.zeros((0, 4), np.uint64) for col in c.instances]
d_adv = [cx.alloc(adv.nbytes).upload(adv) for cx in ctxs]
opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, tau, transcript_repr=77)
want = {s: PR.create_proof(opk, c.instances, c.advice, seed=s) for s in (3, 4)}
got, err = {}, []
def work(w):
    try:
        ctxs[w].set_host_wait(False)  # asks for spinning waits: under the device flag the library polls anyway
        for rep in range(6):
            s = 3 + (rep + w) % 2
            got[(w, rep)] = (s, pkg.plonk.create_proof(ctxs[w], pks[w], inst, d_adv[w], seed=s))
        ctxs[w].timer_start(); ctxs[w].timer_stop(); ctxs[w].sync()
    except BaseException as e:
        err.append(e)
th = [threading.Thread(target=work, args=(w,)) for w in range(2)]
[t.start() for t in th]; [t.join() for t in th]
assert not err, err
assert len(got) == 12 and all(p == want[s] for s, p in got.values()), "proof bytes differ from the oracle under the device flag"
for d in d_adv: d.free()
for q in pks: q.free()
params.free()
for cx in ctxs: cx.close()
print("ok blocking-sync host: 12 proofs on 2 contexts byte-equal to the oracle")
'''


def test_library_inside_a_blocking_sync_host():
    env = dict(os.environ)
    env.setdefault("GPU_MAX_HW_QUEUES", "16")
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 0, "child failed or hung (limit 60 s):\n%s\n%s" % (r.stdout[-1500:], r.stderr[-3000:])
    assert "ok blocking-sync host" in r.stdout
    assert "every host wait of this library polls" in r.stderr  # the library noticed the flag and said so, once
