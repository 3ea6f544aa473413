"""GPU parity: amdzk_ntt_fr* (HIP, gfx950) vs the oracle's best_fft restatement, bit-exact."""
import numpy as np
import pytest
import zkutil as zu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [0, 1, 2, 3, 5, 8, 10, 11, 12, 13, 15, 16, 17, 18, 20])
def test_ntt_matches_oracle(ctx, pkg, oracle, k):
    a = zu.random_fr(1 << k, seed=100 + k)
    w = oracle.omega(k)
    want = oracle.best_fft(a.copy(), w, k)
    got = pkg.arithmetic.best_fft(ctx, a.copy(), w, k)
    assert np.array_equal(got, want)


def test_ntt_golden_vectors(ctx, pkg, oracle):
    import json, os
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bn254_golden.json")))
    for v in G["ntt"]:
        a = zu.fr_array_from_ints([int(x, 16) for x in v["a"]])
        got = pkg.arithmetic.best_fft(ctx, a, oracle.omega(v["k"]), v["k"])
        assert zu.fr_array_to_ints(got) == [int(x, 16) for x in v["ntt"]]


@pytest.mark.parametrize("k", [4, 12, 16, 19])
def test_intt_roundtrip_and_scale(ctx, pkg, oracle, k):
    a = zu.random_fr(1 << k, seed=7 * k)
    w = oracle.omega(k)
    winv = oracle.fr_inv(w.reshape(1, 4)).reshape(4)
    f = pkg.arithmetic.best_fft(ctx, a.copy(), w, k)
    back = pkg.arithmetic.best_fft(ctx, f.copy(), winv, k, flags=pkg.arithmetic.NTT_SCALE_NINV)
    assert np.array_equal(back, a)


def test_ntt_batch_device_columns(ctx, pkg, oracle):
    k, ncols = 14, 5
    n = 1 << k
    stride = n + 64  # columns need not be packed
    w = oracle.omega(k)
    host = np.zeros((ncols, stride, 4), dtype=np.uint64)
    cols = [zu.random_fr(n, seed=900 + c) for c in range(ncols)]
    for c in range(ncols):
        host[c, :n] = cols[c]
    buf = ctx.alloc(host.nbytes).upload(host)
    pkg.arithmetic.best_fft_dev(ctx, buf, w, k, ncols=ncols, col_stride=stride)
    out = buf.download(host.shape)
    buf.free()
    for c in range(ncols):
        assert np.array_equal(out[c, :n], oracle.best_fft(cols[c].copy(), w, k))
        assert not out[c, n:].any()


def test_ntt_linearity_at_full_size(ctx, pkg, oracle):
    """Size-independent property at the k=22 stress size: NTT(a+b) = NTT(a)+NTT(b), and
    iNTT(NTT(a)) = a; spot-check 64 outputs against direct evaluation sum_i a_i w^(ij)."""
    k = 22
    n = 1 << k
    a, b = zu.random_fr(n, seed=1), zu.random_fr(n, seed=2)
    w = oracle.omega(k)
    fa = pkg.arithmetic.best_fft(ctx, a.copy(), w, k)
    fb = pkg.arithmetic.best_fft(ctx, b.copy(), w, k)
    fab = pkg.arithmetic.best_fft(ctx, oracle.fr_add(a, b), w, k)
    assert np.array_equal(fab, oracle.fr_add(fa, fb))
    winv = oracle.fr_inv(w.reshape(1, 4)).reshape(4)
    back = pkg.arithmetic.best_fft(ctx, fa.copy(), winv, k, flags=pkg.arithmetic.NTT_SCALE_NINV)
    assert np.array_equal(back, a)
    # full-size cross-check against the oracle itself (CPU, a few seconds)
    assert np.array_equal(fa, oracle.best_fft(a.copy(), w, k))
