"""CPU: the C++ oracle (oracle/liboracle.so) against the pure-Python golden vectors
(tests/golden/bn254_golden.json) and against the constants /root/reference pins
(contract.sol:210-211, :82, :440 — restated as literals in oracle/pyref.py and the JSON)."""
import json
import os

import numpy as np
import zkutil as zu

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bn254_golden.json")))


def I(h):
    return int(h, 16)


def test_moduli_and_delta_match_contract_sol():
    # literals from solidity_verifier_contract/contract.sol:210,211,440
    assert I(G["q"]) == 21888242871839275222246405745257275088696311157297823662689037894645226208583
    assert I(G["r"]) == 21888242871839275222246405745257275088548364400416034343698204186575808495617
    assert I(G["delta"]) == 4131629893567559867359510883348571134090853742863529169391034518566172092834
    assert pow(7, 1 << 28, zu.R) == I(G["delta"])


def test_constants(oracle):
    root, zeta, delta = oracle.constants()
    assert zu.fr_to_int(root) == I(G["root_of_unity"])
    assert zu.fr_to_int(zeta) == I(G["zeta"])
    assert zu.fr_to_int(delta) == I(G["delta"])
    for k, h in G["omega"].items():
        assert zu.fr_to_int(oracle.omega(int(k))) == I(h)


def test_fr_arithmetic(oracle):
    a = zu.fr_array_from_ints([I(v["a"]) for v in G["fr"]])
    b = zu.fr_array_from_ints([I(v["b"]) for v in G["fr"]])
    assert zu.fr_array_to_ints(oracle.fr_mul(a, b)) == [I(v["mul"]) for v in G["fr"]]
    assert zu.fr_array_to_ints(oracle.fr_add(a, b)) == [I(v["add"]) for v in G["fr"]]
    assert zu.fr_array_to_ints(oracle.fr_sub(a, b)) == [I(v["sub"]) for v in G["fr"]]
    assert zu.fr_array_to_ints(oracle.fr_inv(a)) == [I(v["inv_a"]) for v in G["fr"]]
    raw = np.array([zu.limbs(I(v["a"])) for v in G["fr"]], dtype=np.uint64)
    assert np.array_equal(oracle.fr_from_raw(raw), a)
    assert np.array_equal(oracle.fr_to_raw(a), raw)


def test_ntt_vectors(oracle):
    for v in G["ntt"]:
        k = v["k"]
        a = zu.fr_array_from_ints([I(x) for x in v["a"]])
        want = [I(x) for x in v["ntt"]]
        w = oracle.omega(k)
        assert zu.fr_array_to_ints(oracle.best_fft(a.copy(), w, k, threads=1)) == want
        assert zu.fr_array_to_ints(oracle.best_fft(a.copy(), w, k, threads=4)) == want
        assert zu.fr_array_to_ints(oracle.dft_naive(a, w)) == want


def test_fft_vs_naive_larger(oracle):
    for k in (8, 10):
        a = zu.random_fr(1 << k, seed=k)
        w = oracle.omega(k)
        assert np.array_equal(oracle.best_fft(a.copy(), w, k), oracle.dft_naive(a, w))


def test_g1_and_srs(oracle):
    gen = oracle.generator()
    assert zu.point_to_ints(gen) == (1, 2)
    ks = zu.fr_array_from_ints([I(v["k"]) for v in G["g1_multiples"]])
    got = oracle.g1_mul_many(gen, ks)
    for p, v in zip(got, G["g1_multiples"]):
        assert zu.point_to_ints(p) == (I(v["p"][0]), I(v["p"][1]))
        assert oracle.on_curve(p)
    srs = oracle.srs_powers(zu.fr_from_int(I(G["srs"]["tau"])), len(G["srs"]["g"]))
    assert [zu.point_to_ints(p) for p in srs] == [(I(p[0]), I(p[1])) for p in G["srs"]["g"]]


def test_g_to_lagrange_against_closed_form(oracle):
    """pyref.g_to_lagrange (the group FFT of arithmetic.rs [UP]) on g_i = tau^i G gives L_i(tau) G, the
    closed-form Lagrange basis the C oracle builds from scalars (zkutil.test_srs)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as P

    tau = 0x1234567890ABCDEF1234567
    for k in (0, 1, 2, 4):
        g, gl = zu.test_srs(oracle, k, tau)
        got = P.g_to_lagrange([zu.point_to_ints(p) for p in g], k)
        assert got == [zu.point_to_ints(p) for p in gl]


def test_msm_vectors(oracle):
    srs = np.array([zu.point_from_ints((I(p[0]), I(p[1]))) for p in G["srs"]["g"]], dtype=np.uint64)
    for v in G["msm"]:
        s = zu.fr_array_from_ints([I(x) for x in v["scalars"]])
        want = None if v["result"] is None else (I(v["result"][0]), I(v["result"][1]))
        for threads in (1, 3, 8):
            assert zu.point_to_ints(oracle.best_multiexp(s, srs[: v["n"]], threads=threads)) == want
        assert zu.point_to_ints(oracle.msm_naive(s, srs[: v["n"]])) == want


def test_msm_pippenger_vs_naive_larger(oracle):
    n = 600
    g = oracle.srs_powers(zu.fr_from_int(99), n)
    s = zu.skewed_fr(n, 5, oracle)
    assert np.array_equal(oracle.best_multiexp(s, g), oracle.msm_naive(s, g))


def test_eval_and_kate(oracle):
    v = G["eval_polynomial"]
    poly = zu.fr_array_from_ints([I(c) for c in v["poly"]])
    x = zu.fr_from_int(I(v["x"]))
    assert zu.fr_to_int(oracle.eval_polynomial(poly, x)) == I(v["y"])
    q = oracle.kate_division(poly, x)
    assert zu.fr_array_to_ints(q) == [I(c) for c in G["kate_division"]["q"]]


def test_batch_invert(oracle):
    a = zu.random_fr(100, seed=9)
    a[7] = 0
    got = oracle.batch_invert(a)
    want = oracle.fr_inv(a)
    assert np.array_equal(got, want) and not got[7].any()


def test_evaluation_domain(oracle):
    v = G["domain"]
    d = zu.OracleDomain(oracle, v["j"], v["k"])
    assert d.extended_k == v["extended_k"]
    coeff = zu.fr_array_from_ints([I(x) for x in v["coeff"]])
    assert zu.fr_array_to_ints(d.coeff_to_lagrange(coeff)) == [I(x) for x in v["lagrange"]]
    assert np.array_equal(d.lagrange_to_coeff(d.coeff_to_lagrange(coeff)), coeff)
    ext = d.coeff_to_extended(coeff)
    assert zu.fr_array_to_ints(ext) == [I(x) for x in v["extended"]]
    assert zu.fr_array_to_ints(d.t_evaluations) == [I(x) for x in v["t_evaluations"]]
    assert zu.fr_array_to_ints(d.divide_by_vanishing_poly(ext)) == [I(x) for x in v["divided"]]
    back = d.extended_to_coeff(ext)
    assert np.array_equal(back[: d.n], coeff) and not back[d.n:].any()
