"""GPU parity: amdzk_msm_g1* (HIP, gfx950) vs the oracle's best_multiexp restatement. Equality is on
the affine point (the reference's projective coordinates depend on rayon's thread count)."""
import json
import os

import numpy as np
import pytest
import zkutil as zu

pytestmark = pytest.mark.gpu

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bn254_golden.json")))


@pytest.fixture(scope="module")
def srs14(oracle):
    return oracle.srs_powers(zu.fr_from_int(0xABCDEF12345), 1 << 14)


def check(ctx, pkg, oracle, params, bases, scalars, basis=0):
    got = pkg.arithmetic.best_multiexp(ctx, params.h, basis, scalars)
    want = oracle.best_multiexp(scalars, bases[: scalars.shape[0]])
    assert np.array_equal(zu.jac_to_affine_host(oracle, got), want)


def test_msm_golden_vectors(ctx, pkg, oracle):
    srs = np.array([zu.point_from_ints((int(p[0], 16), int(p[1], 16))) for p in G["srs"]["g"]], dtype=np.uint64)
    params = pkg.kzg.ParamsKZG(ctx, 6, g=srs)
    for v in G["msm"]:
        s = zu.fr_array_from_ints([int(x, 16) for x in v["scalars"]])
        got = zu.point_to_ints(zu.jac_to_affine_host(oracle, params.commit(s)))
        want = None if v["result"] is None else (int(v["result"][0], 16), int(v["result"][1], 16))
        assert got == want, v["kind"]
    params.free()


@pytest.mark.parametrize("k", [1, 4, 8, 10, 11, 12, 13, 14])
def test_msm_uniform_matches_oracle(ctx, pkg, oracle, srs14, k):
    n = 1 << k
    params = pkg.kzg.ParamsKZG(ctx, k, g=srs14[:n].copy())
    check(ctx, pkg, oracle, params, srs14, zu.random_fr(n, seed=300 + k))
    params.free()


@pytest.mark.parametrize("k", [15, 17, 18])
def test_msm_prover_sizes_match_oracle(ctx, pkg, oracle, k):
    """Every window width the prover selects for the BASELINE configurations (c = 13 at k = 15, c = 14 at k = 17 and
    18: msm.hip pick_window_bits) checked directly against the oracle's Pippenger: uniform scalars, a witness-like
    skewed column, and a short (ragged) column, in one batched call over the Lagrange basis."""
    n = 1 << k
    g = oracle.srs_powers(zu.fr_from_int(0x5EED0000 + k), n)
    params = pkg.kzg.ParamsKZG(ctx, k, g_lagrange=g)
    cols = [zu.random_fr(n, seed=900 + k), zu.skewed_fr(n, 910 + k, oracle)]
    got = pkg.arithmetic.best_multiexp_batch(ctx, params.h, 1, cols)
    for c in range(2):
        assert np.array_equal(zu.jac_to_affine_host(oracle, got[c]), oracle.best_multiexp(cols[c], g))
    short = zu.random_fr(n - 12345, seed=920 + k)
    check(ctx, pkg, oracle, params, g, short, basis=1)
    params.free()


def test_msm_edge_cases(ctx, pkg, oracle, srs14):
    k = 10
    n = 1 << k
    params = pkg.kzg.ParamsKZG(ctx, k, g=srs14[:n].copy(), g_lagrange=srs14[n:2 * n].copy())
    one = zu.fr_from_int(1)
    zeros = np.zeros((n, 4), np.uint64)
    # all zero -> identity in normal form (0,1,0)
    got = params.commit(zeros)
    assert zu.point_to_ints(zu.jac_to_affine_host(oracle, got)) is None
    # empty and ragged lengths (bases[..len])
    for m in (0, 1, 2, 3, 31, 33, 1000):
        check(ctx, pkg, oracle, params, srs14, zu.random_fr(m, seed=m + 1) if m else np.zeros((0, 4), np.uint64))
    # all ones: every digit lands in one bucket (heavy-bucket path)
    check(ctx, pkg, oracle, params, srs14, np.tile(one, (n, 1)))
    # r-1 (= -1), 2^k-boundaries of the window digits, skewed witness-like column
    check(ctx, pkg, oracle, params, srs14, np.tile(zu.fr_from_int(zu.R - 1), (n, 1)))
    edge = zu.fr_array_from_ints([[0, 1, zu.R - 1, 127, 128, 129, 255, 256, (1 << 253) % zu.R, (1 << 64) - 1][i % 10] for i in range(n)])
    check(ctx, pkg, oracle, params, srs14, edge)
    check(ctx, pkg, oracle, params, srs14, zu.skewed_fr(n, 77, oracle))
    # second basis
    check(ctx, pkg, oracle, params, srs14[n:], zu.random_fr(n, seed=5), basis=1)
    params.free()


def test_msm_repeated_and_identity_bases(ctx, pkg, oracle):
    """Bases that collide (same point many times, P and -P, identity): exercises the doubling and
    cancellation branches of the mixed addition."""
    k = 8
    n = 1 << k
    gen = oracle.generator()
    neg = zu.point_from_ints((1, zu.Q - 2))
    bases = np.tile(gen, (n, 1))
    bases[1::3] = neg
    bases[2::7] = 0
    params = pkg.kzg.ParamsKZG(ctx, k, g=bases)
    for seed, s in ((1, zu.random_fr(n, seed=41)), (2, np.tile(zu.fr_from_int(1), (n, 1))), (3, zu.skewed_fr(n, 3, oracle))):
        check(ctx, pkg, oracle, params, bases, s)
    params.free()


def test_msm_batch_columns(ctx, pkg, oracle, srs14):
    k, ncols = 12, 7
    n = 1 << k
    params = pkg.kzg.ParamsKZG(ctx, k, g_lagrange=srs14[:n].copy())
    cols = [zu.skewed_fr(n, 50 + c, oracle) if c % 2 else zu.random_fr(n, seed=60 + c) for c in range(ncols)]
    cols[3] = np.zeros((n, 4), np.uint64)
    got = pkg.arithmetic.best_multiexp_batch(ctx, params.h, 1, cols)
    for c in range(ncols):
        assert np.array_equal(zu.jac_to_affine_host(oracle, got[c]), oracle.best_multiexp(cols[c], srs14[:n]))
    params.free()


def test_msm_linearity_large(ctx, pkg, oracle):
    """Size-independent property at a size the naive oracle cannot reach quickly: with bases
    g_i = t^i G (known t), MSM(s, g) = (sum s_i t^i) G = eval_polynomial(s, t) * G."""
    k = 16
    n = 1 << k
    tau = zu.fr_from_int(123456789)
    g = oracle.srs_powers(tau, n)
    params = pkg.kzg.ParamsKZG(ctx, k, g=g)
    s = zu.random_fr(n, seed=8)
    got = zu.jac_to_affine_host(oracle, params.commit(s))
    e = oracle.eval_polynomial(s, tau)
    want = oracle.g1_mul_many(oracle.generator(), e.reshape(1, 4))[0]
    assert np.array_equal(got, want)
    assert np.array_equal(got, oracle.best_multiexp(s, g))
    params.free()


def test_srs_setup_on_device_matches_oracle(ctx, pkg, oracle):
    """amdzk_srs_setup (ParamsKZG::setup with an explicit trapdoor) against the oracle's tau-powers and
    closed-form Lagrange bases; then commit_lagrange(p) == commit(lagrange_to_coeff(p))."""
    k, tau = 10, 0x1234567890ABCDEF1234567
    n = 1 << k
    params = pkg.kzg.ParamsKZG.setup(ctx, k, zu.fr_from_int(tau), want_host_copy=True)
    g, gl = zu.test_srs(oracle, k, tau)
    assert np.array_equal(params._g, g)
    assert np.array_equal(params._gl, gl)
    p = zu.random_fr(n, seed=31)
    od = zu.OracleDomain(oracle, 3, k)
    a = zu.jac_to_affine_host(oracle, params.commit_lagrange(p))
    b = zu.jac_to_affine_host(oracle, params.commit(od.lagrange_to_coeff(p)))
    assert np.array_equal(a, b) and np.array_equal(a, oracle.best_multiexp(p, gl))
    params.free()


def test_g_to_lagrange_matches_oracle(ctx, pkg, oracle):
    """arithmetic::g_to_lagrange on the device (FFT over G1): against the pure-Python restatement on
    small domains, including points at infinity and repeated points, and against the closed-form
    Lagrange basis of a tau-powers SRS at k = 10."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as P

    tau = 0x1234567890ABCDEF1234567
    for k in (0, 1, 3, 5):
        g, _ = zu.test_srs(oracle, k, tau)
        if k == 3:
            g[2] = 0          # identity
            g[5] = g[4]       # u == v -> u - v = identity, u + v = doubling
            x6, y6 = zu.point_to_ints(g[6])
            g[7] = zu.point_from_ints((x6, (-y6) % P.Q))  # v == -u
        got = pkg.arithmetic.g_to_lagrange(ctx, g, k)
        want = P.g_to_lagrange([zu.point_to_ints(p) for p in g], k)
        assert [zu.point_to_ints(p) for p in got] == want, k
    g, gl = zu.test_srs(oracle, 10, tau)
    assert np.array_equal(pkg.arithmetic.g_to_lagrange(ctx, g, 10), gl)


def test_params_downsize(ctx, pkg, oracle):
    """ParamsKZG::downsize: g truncated, g_lagrange recomputed for the smaller domain (equal to a fresh
    setup at that k with the same trapdoor), commitments agree across bases, growing is refused."""
    tau = 0x1234567890ABCDEF1234567
    params = pkg.kzg.ParamsKZG.setup(ctx, 10, zu.fr_from_int(tau), want_host_copy=True)
    with pytest.raises(pkg.AmdzkError):
        params.downsize(11)
    params.downsize(7)
    g, gl = zu.test_srs(oracle, 7, tau)
    assert params.k == 7 and params.n == 128
    assert np.array_equal(params.get_g(), g) and np.array_equal(params.get_g_lagrange(), gl)
    assert np.array_equal(params._g, g) and np.array_equal(params._gl, gl)
    p = zu.random_fr(128, seed=5)
    od = zu.OracleDomain(oracle, 3, 7)
    a = zu.jac_to_affine_host(oracle, params.commit_lagrange(p))
    assert np.array_equal(a, zu.jac_to_affine_host(oracle, params.commit(od.lagrange_to_coeff(p))))
    assert np.array_equal(a, oracle.best_multiexp(p, gl))
    fresh = pkg.kzg.ParamsKZG.setup(ctx, 7, zu.fr_from_int(tau))
    assert fresh.write() == params.write()
    fresh.free(); params.free()


def test_msm_k22_full_size_tau_identity(ctx, pkg, oracle):
    """BASELINE config 5 size. With bases g_i = s^i G (device-built SRS), MSM(c, g) must equal
    eval_polynomial(c, s) * G — a size-independent identity checked at n = 2^22 against the oracle's
    Horner evaluation and one oracle scalar multiplication."""
    k = 22
    n = 1 << k
    s = zu.fr_from_int(7 ** 20)
    params = pkg.kzg.ParamsKZG.setup(ctx, k, s)
    for seed, kind in ((42, "uniform"), (43, "skewed")):
        c = zu.random_fr(n, seed=seed)
        if kind == "skewed":
            sel = zu.splitmix64(seed + 9, n) % np.uint64(4)
            c[sel < 2] = 0
        got = zu.jac_to_affine_host(oracle, params.commit(c))
        e = oracle.eval_polynomial(c, s)
        want = oracle.g1_mul_many(oracle.generator(), e.reshape(1, 4))[0]
        assert np.array_equal(got, want), kind
    params.free()


def test_params_write_read_roundtrip(ctx, pkg, oracle):
    """ParamsKZG::write / ::read: device compression matches the pure-Python encoder on every point,
    read(write(p)) commits identically, and a corrupted encoding is rejected."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import plonk_ref as PR

    k, tau = 8, 0x1234567890ABCDEF1234567
    n = 1 << k
    params = pkg.kzg.ParamsKZG.setup(ctx, k, zu.fr_from_int(tau), want_host_copy=True)
    g2, s_g2 = bytes(range(64)), bytes(range(64, 128))
    blob = params.write(g2, s_g2)
    assert len(blob) == 4 + 2 * n * 32 + 128 and int.from_bytes(blob[:4], "little") == k
    for i in (0, 1, 17, n - 1):
        assert blob[4 + 32 * i:4 + 32 * (i + 1)] == PR.g1_compress(zu.point_to_ints(params._g[i]))
        assert blob[4 + 32 * (n + i):4 + 32 * (n + i + 1)] == PR.g1_compress(zu.point_to_ints(params._gl[i]))
    p2, g2b, sg2b = pkg.kzg.ParamsKZG.read(ctx, blob)
    assert (g2b, sg2b) == (g2, s_g2)
    c = zu.random_fr(n, seed=4)
    assert np.array_equal(p2.commit(c), params.commit(c)) and np.array_equal(p2.commit_lagrange(c), params.commit_lagrange(c))
    bad = bytearray(blob)
    bad[4 + 32 * 5] ^= 1  # x of g[5] -> almost surely not on the curve (or a different point: then commits differ)
    try:
        p3, _, _ = pkg.kzg.ParamsKZG.read(ctx, bytes(bad))
        assert not np.array_equal(p3.commit(c), params.commit(c))
        p3.free()
    except pkg.AmdzkError as e:
        assert "invalid point" in str(e)
    with pytest.raises(pkg.AmdzkError):
        pkg.kzg.ParamsKZG.read(ctx, blob[:100])
    p2.free(); params.free()


@pytest.mark.parametrize("env", [{"AMDZK_TAIL_QUAD": "1"}, {"AMDZK_TAIL_QUAD": "0", "AMDZK_TAIL_TREE": "1"}, {"AMDZK_MSM_NLEV": "2"}, {"AMDZK_L1_LDS": "4"}, {"AMDZK_L1_LDS": "9"}])
def test_alternative_kernels_give_the_same_points(env):
    """The kernel variants that only run in a proof's latency mode (the bucket reduction with quad-lane point additions, or
    with shuffle-tree row / column sums) and the measured-and-rejected ones that stay in the tree behind switches (one fold
    level, level 1 with the accumulator in LDS, level 1 as a persistent grid) must give the oracle's points too: the parity
    tests above, again, in a child process with the switch forced (the switches are read once per process)."""
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "golden_vectors or uniform_matches_oracle or edge or batch"], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600)
    assert r.returncode == 0, "%r:\n%s" % (env, r.stdout[-3000:])
    assert " passed" in r.stdout and "failed" not in r.stdout
