"""GPU: device affinity of the C ABI. HIP's current device is per host thread; a ctx created on one
thread and used from another must still allocate and launch on ITS device. On a one-GPU box the device is 0 either
way, so the checks that can fail there are: the calling thread's device is restored after every call, a worker thread
can prove with a ctx / key made on the main thread, and every workspace pointer reports the ctx's device
(hipPointerGetAttributes). With two or more GPUs the same is done on device 1."""
import ctypes
import os
import sys
import threading

import numpy as np
import pytest
import zkutil as zu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import circuits  # noqa: E402
import plonk_ref as PR  # noqa: E402

pytestmark = pytest.mark.gpu
TAU = 0x1234567890ABCDEF1234567


def device_count():
    import torch

    return torch.cuda.device_count()


def prove_from_other_thread(pkg, oracle, dev):
    plonk = pkg.plonk
    c = circuits.lookup_circuit(plonk, 6, seed=5)
    ctx = pkg.Context(dev)
    assert ctx.device() == dev
    params = pkg.kzg.ParamsKZG.setup(ctx, c.k, zu.fr_from_int(TAU))
    fixed = np.stack([zu.ints_to_fr(oracle, col) for col in c.fixed])
    pk = plonk.ProvingKey(ctx, params, c.desc, fixed, c.assembly.mapping, zu.fr_from_int(77))
    adv = np.stack([zu.ints_to_fr(oracle, col) for col in c.advice])
    d_adv = ctx.alloc(adv.nbytes).upload(adv)
    inst = [zu.ints_to_fr(oracle, col) if col else np.zeros((0, 4), np.uint64) for col in c.instances]
    out = {}

    def work():
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            cur = ctypes.c_int(-1)
            # a fresh thread starts on device 0; point it at the LAST device so that "current device" != ctx device
            # whenever there is a choice, then check the call neither depends on it nor changes it
            other = device_count() - 1 if dev == 0 else 0
            assert hip.hipSetDevice(other) == 0
            out["proof"] = plonk.create_proof(ctx, pk, inst, d_adv, seed=3)
            assert hip.hipGetDevice(ctypes.byref(cur)) == 0
            out["device_after"] = (cur.value, other)
            ctx.check_affinity(pk=pk.h, ptr=d_adv.ptr.value)
        except BaseException as e:  # noqa: BLE001
            out["error"] = e

    t = threading.Thread(target=work)
    t.start()
    t.join()
    assert "error" not in out, out.get("error")
    assert out["device_after"][0] == out["device_after"][1], "the call changed the calling thread's device"
    opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=77)
    assert out["proof"] == PR.create_proof(opk, c.instances, c.advice, seed=3)
    ctx.check_affinity(pk=pk.h)
    d_adv.free(); pk.free(); params.free(); ctx.close()


def test_ctx_made_on_main_thread_proves_from_worker_thread(pkg, oracle):
    prove_from_other_thread(pkg, oracle, 0)


def test_same_on_a_non_zero_device(pkg, oracle):
    if device_count() < 2:
        pytest.skip("needs two GPUs (the driver's 8-GPU node exercises it through bench.py's per-rank affinity check)")
    prove_from_other_thread(pkg, oracle, 1)


def test_foreign_pointer_is_reported(pkg):
    ctx = pkg.Context(0)
    host = np.zeros(16, np.uint64)
    with pytest.raises(pkg.AmdzkError):
        ctx.check_affinity(ptr=host.ctypes.data)  # pageable host memory is not a device allocation of this ctx
    ctx.close()
