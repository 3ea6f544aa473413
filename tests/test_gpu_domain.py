"""GPU parity: EvaluationDomain methods (fused coset/scale NTT steps) vs the oracle's restatement."""
import json
import os

import numpy as np
import pytest
import zkutil as zu

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bn254_golden.json")))


@pytest.mark.parametrize("j,k", [(4, 3), (3, 5), (4, 9), (5, 10), (4, 12), (4, 15), (9, 11), (4, 16)])
def test_domain_matches_oracle(ctx, pkg, oracle, j, k):
    od = zu.OracleDomain(oracle, j, k)
    d = pkg.domain.EvaluationDomain(ctx, j, k)
    assert d.extended_k == od.extended_k
    for nm in od.NAMES:
        assert np.array_equal(getattr(d, nm), getattr(od, nm)), nm
    a = zu.random_fr(1 << k, seed=1000 + k)
    assert np.array_equal(d.coeff_to_lagrange(a), od.coeff_to_lagrange(a))
    assert np.array_equal(d.lagrange_to_coeff(a), od.lagrange_to_coeff(a))
    ext = d.coeff_to_extended(a)
    assert np.array_equal(ext, od.coeff_to_extended(a))
    e = zu.random_fr(d.extended_len(), seed=2000 + k)
    assert np.array_equal(d.extended_to_coeff(e), od.extended_to_coeff(e))
    assert np.array_equal(d.divide_by_vanishing_poly(e), od.divide_by_vanishing_poly(e))
    back = d.extended_to_coeff(ext)  # round trip: degree < n survives, the rest is zero
    assert np.array_equal(back[: 1 << k], a) and not back[1 << k:].any()
    d.free()


def test_domain_golden(ctx, pkg):
    v = G["domain"]
    d = pkg.domain.EvaluationDomain(ctx, v["j"], v["k"])
    coeff = zu.fr_array_from_ints([int(x, 16) for x in v["coeff"]])
    assert zu.fr_array_to_ints(d.coeff_to_extended(coeff)) == [int(x, 16) for x in v["extended"]]
    assert zu.fr_array_to_ints(d.coeff_to_lagrange(coeff)) == [int(x, 16) for x in v["lagrange"]]
    ext = zu.fr_array_from_ints([int(x, 16) for x in v["extended"]])
    assert zu.fr_array_to_ints(d.divide_by_vanishing_poly(ext)) == [int(x, 16) for x in v["divided"]]
    d.free()


def test_domain_batched_columns(ctx, pkg, oracle):
    j, k, ncols = 4, 13, 6
    od = zu.OracleDomain(oracle, j, k)
    d = pkg.domain.EvaluationDomain(ctx, j, k)
    n, en = 1 << k, d.extended_len()
    cols = np.stack([zu.random_fr(n, seed=70 + c) for c in range(ncols)])
    src = ctx.alloc(cols.nbytes).upload(cols)
    dst = ctx.alloc(ncols * en * 32)
    d.coeff_to_extended_dev(src, dst, ncols=ncols)
    out = dst.download((ncols, en, 4))
    for c in range(ncols):
        assert np.array_equal(out[c], od.coeff_to_extended(cols[c]))
    d.lagrange_to_coeff_dev(src, ncols=ncols)
    out = src.download((ncols, n, 4))
    for c in range(ncols):
        assert np.array_equal(out[c], od.lagrange_to_coeff(cols[c]))
    src.free(); dst.free(); d.free()
