"""GPU: streaming witnesses (the device half). Different witnesses of one
layout, uploaded from pinned host memory on the copy stream while other proofs run — two proving contexts in flight,
double-buffered device staging per context — must give exactly the bytes of the sequential, resident-witness proofs,
which in turn equal the oracle's."""
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
SMALL = dict(num_advice=6, num_lookup_advice=2, lookup_bits=6, num_spread=2, spread_bits=4)


def test_streamed_witnesses_in_flight_equal_sequential(pkg, oracle):
    plonk, fd = pkg.plonk, pkg.feeder
    c = circuits.full_aadhaar_shape(plonk, k=9, **SMALL)
    nw = 5
    wit = [(c.advice, c.instances)] + [c.witness(100 + j) for j in range(1, nw)]
    for a, i in wit[1:]:
        circuits.check_satisfied(c, advice=a, instances=i)
    assert wit[1][0] != wit[0][0] and wit[1][1] != wit[0][1]  # really different witnesses and public inputs
    A, n = c.desc["num_advice"], c.n
    ctxs = [pkg.Context(0), pkg.Context(0)]
    params = pkg.kzg.ParamsKZG.setup(ctxs[0], c.k, zu.fr_from_int(TAU))
    fixed = np.stack([zu.ints_to_fr(oracle, col) for col in c.fixed])
    pks = [plonk.ProvingKey(cx, params, c.desc, fixed, c.assembly.mapping, zu.fr_from_int(5)) for cx in ctxs]
    adv = [np.stack([zu.ints_to_fr(oracle, col) for col in a]) for a, _ in wit]
    inst = [[zu.ints_to_fr(oracle, col) if col else np.zeros((0, 4), np.uint64) for col in i] for _, i in wit]
    # sequential, resident
    want = []
    for j in range(nw):
        d = ctxs[0].alloc(adv[j].nbytes).upload(adv[j])
        want.append(plonk.create_proof(ctxs[0], pks[0], inst[j], d, seed=40 + j))
        d.free()
    opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=5)
    assert want[1] == PR.create_proof(opk, wit[1][1], wit[1][0], seed=41)
    assert len(set(want)) == nw
    # streamed: pinned host copies, two contexts in flight, each with its own double buffer
    pinned = []
    for j in range(nw):
        pw = fd.PinnedWitness(ctxs[0], A, n)
        pw.array[...] = adv[j]
        pinned.append(pw)
    streams = [fd.WitnessStream(cx, A * n * 32) for cx in ctxs]
    jobs = [[(pinned[j], inst[j], 40 + j) for j in range(nw) if j % 2 == w] for w in range(2)]
    got = [None, None]
    err = []

    def work(w):
        try:
            got[w] = fd.prove_stream(plonk, ctxs[w], pks[w], streams[w], jobs[w])
        except BaseException as e:  # noqa: BLE001
            err.append(e)

    th = [threading.Thread(target=work, args=(w,)) for w in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    for w in range(2):
        assert got[w] == [want[j] for j in range(nw) if j % 2 == w]
    # a second pass through the same buffers (both halves of each double buffer have been used by now)
    assert fd.prove_stream(plonk, ctxs[1], pks[1], streams[1], jobs[0]) == got[0]
    for s in streams:
        s.free()
    for p in pinned:
        p.free()
    for q in pks:
        q.free()
    params.free()
    for cx in ctxs:
        cx.close()
