"""Generates tests/golden/bn254_golden.json from oracle/pyref.py (pure Python integers, no reference
import: the reference holds no vector for this path). Deterministic; re-run to
reproduce:  python tests/golden/gen_golden.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyref as P  # noqa: E402


def hx(x):
    return "%064x" % x


def pt(p):
    return None if p is None else [hx(p[0]), hx(p[1])]


def main():
    rng = P.SplitMix64(2024)
    out = {"source": "oracle/pyref.py (pure-Python integers)", "q": hx(P.Q), "r": hx(P.R),
           "root_of_unity": hx(P.ROOT_OF_UNITY), "zeta": hx(P.ZETA), "delta": hx(P.DELTA)}
    edge = [0, 1, 2, P.R - 1, P.R - 2, (1 << 253) % P.R, (1 << 64) - 1]
    xs = edge + [rng.fr() for _ in range(24)]
    ys = list(reversed(edge)) + [rng.fr() for _ in range(24)]
    out["fr"] = [{"a": hx(a), "b": hx(b), "mul": hx(a * b % P.R), "add": hx((a + b) % P.R),
                  "sub": hx((a - b) % P.R), "inv_a": hx(pow(a, -1, P.R) if a else 0)} for a, b in zip(xs, ys)]
    out["omega"] = {str(k): hx(P.omega(k)) for k in range(0, 29)}
    ntt = []
    for k in (0, 1, 2, 3, 5, 7):
        a = [rng.fr() for _ in range(1 << k)]
        w = P.omega(k)
        f = P.dft_naive(a, w)
        assert f == P.best_fft(a, w, k)
        winv = pow(w, -1, P.R)
        ninv = pow(1 << k, -1, P.R)
        assert [x * ninv % P.R for x in P.best_fft(f, winv, k)] == a
        ntt.append({"k": k, "a": [hx(x) for x in a], "ntt": [hx(x) for x in f]})
    out["ntt"] = ntt
    # G1: multiples of the generator, a small SRS, MSMs
    out["g1_multiples"] = [{"k": hx(k), "p": pt(P.g1_mul(P.G1_GEN, k))} for k in [1, 2, 3, 7, P.R - 1, rng.fr()]]
    tau = rng.fr()
    n = 64
    srs = [P.g1_mul(P.G1_GEN, pow(tau, i, P.R)) for i in range(n)]
    assert all(P.g1_on_curve(p) for p in srs)
    msm = []
    for m, kind in ((1, "uniform"), (3, "uniform"), (8, "uniform"), (40, "uniform"), (64, "uniform"),
                    (64, "small"), (64, "edge")):
        if kind == "uniform":
            s = [rng.fr() for _ in range(m)]
        elif kind == "small":
            s = [rng.next() % 5 for _ in range(m)]
        else:
            s = [[0, 1, P.R - 1, 2, (1 << 16) - 1, 1 << 16, (1 << 253) % P.R, rng.fr()][i % 8] for i in range(m)]
        res = P.msm_naive(s, srs[:m])
        assert res == P.msm_pippenger(s, srs[:m])
        msm.append({"n": m, "kind": kind, "scalars": [hx(x) for x in s], "result": pt(res)})
    out["srs"] = {"tau": hx(tau), "g": [pt(p) for p in srs]}
    out["msm"] = msm
    # EvaluationDomain(j=4, k=3): extended_k = 5
    d = P.EvaluationDomain(4, 3)
    a = [rng.fr() for _ in range(d.n)]
    ext = d.coeff_to_extended(a)
    # direct evaluation at zeta * extended_omega^i
    assert ext == [P.eval_polynomial(a, P.ZETA * pow(d.extended_omega, i, P.R) % P.R) for i in range(d.extended_len())]
    back = d.extended_to_coeff(ext)
    assert back[: d.n] == a and all(x == 0 for x in back[d.n:])
    lag = d.coeff_to_lagrange(a)
    assert d.lagrange_to_coeff(lag) == a
    out["domain"] = {"j": 4, "k": 3, "extended_k": d.extended_k, "coeff": [hx(x) for x in a],
                     "lagrange": [hx(x) for x in lag], "extended": [hx(x) for x in ext],
                     "t_evaluations": [hx(x) for x in d.t_evaluations],
                     "divided": [hx(x) for x in d.divide_by_vanishing_poly(ext)]}
    x = rng.fr()
    poly = [rng.fr() for _ in range(9)]
    out["eval_polynomial"] = {"poly": [hx(c) for c in poly], "x": hx(x), "y": hx(P.eval_polynomial(poly, x))}
    q = P.kate_division(poly, x)
    out["kate_division"] = {"b": hx(x), "q": [hx(c) for c in q]}
    with open(os.path.join(HERE, "bn254_golden.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote bn254_golden.json")


if __name__ == "__main__":
    main()
