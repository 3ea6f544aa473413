"""GPU: the PLONK layer kernel by kernel against the oracle (rows a7-a12 used to be pinned
only through whole-proof bytes). Every call goes through the function-level C ABI (include/amdzk.h, "function by
function"), i.e. the very kernels create_proof runs:

  a8/a9  amdzk_batch_invert_dev, amdzk_grand_product_dev     vs oracle batch_invert / Python running products
  a9     amdzk_permute_expression_pair_dev                   vs plonk_ref.permute_expression_pair (adversarial multisets)
  a11    amdzk_eval_poly_dev                                 vs oracle eval_polynomial
  a12    amdzk_poly_axpy_dev, amdzk_kate_div_dev             vs Python sums / oracle kate_division
  a7     evaluate_h + quotient (amdzk_quotient_eval_dev and the in-prover path): the vanishing identity
         sum_j y^(..) constraint_j(x) = h(x) (x^n - 1) at random x with plonk_ref.h_constraints, on the prover's own
         committed polynomials; plus the permutation / lookup product columns recomputed from their definition.
"""
import os
import sys

import numpy as np
import pytest
import zkutil as zu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import circuits  # noqa: E402
import plonk_ref as PR  # noqa: E402
import pyref as P  # noqa: E402

pytestmark = pytest.mark.gpu
R = zu.R
TAU = 0x1234567890ABCDEF1234567


def rand_ints(n, seed, zeros_every=0):
    v = zu.fr_array_to_ints(zu.random_fr(n, seed=seed)) if n else []
    if zeros_every:
        for i in range(0, n, zeros_every):
            v[i] = 0
    return v


@pytest.fixture(scope="module")
def ar(pkg):
    return pkg.arithmetic


@pytest.mark.parametrize("n", [1, 2, 127, 128, 129, 1000, 4097])
def test_batch_invert(ctx, ar, oracle, n):
    a = zu.random_fr(n, seed=70 + n)
    a[::7] = 0  # zeros stay zero, and do not poison their chunk (one inversion per 128 elements)
    if n > 200:
        a[128:256] = 0  # a whole chunk of zeros
    got = ar.batch_invert(ctx, a)
    assert np.array_equal(got, oracle.batch_invert(a))
    assert not got[::7].any()


@pytest.mark.parametrize("n", [0, 1, 129, 4097, 70001])
def test_batch_invert_assigned(ctx, ar, oracle, n):
    """poly::batch_invert_assigned: Rational cells with random, unit (Trivial) and ZERO
    denominators against Python integers — num * den^-1 mod r, zero for a zero denominator — out of place, in place, and with
    no denominators at all."""
    import random
    rnd = random.Random(300 + n)
    nums = [rnd.randrange(zu.R) for _ in range(n)]
    dens = [rnd.randrange(1, zu.R) for _ in range(n)]
    for i in range(n):
        if i % 5 == 0:
            dens[i] = 1
        elif i % 11 == 3:
            dens[i] = 0
        if i % 13 == 7:
            nums[i] = 0
    want = [a * pow(d, zu.R - 2, zu.R) % zu.R for a, d in zip(nums, dens)]
    fa, fd, fw = (zu.ints_to_fr(oracle, v).reshape(-1, 4) for v in (nums, dens, want))
    assert np.array_equal(ar.batch_invert_assigned(ctx, fa, fd), fw)
    assert np.array_equal(ar.batch_invert_assigned(ctx, fa, fd, in_place=True), fw)
    assert np.array_equal(ar.batch_invert_assigned(ctx, fa, None), fa)


@pytest.mark.parametrize("n,ncols", [(1, 1), (7, 3), (2047, 2), (2048, 2), (2049, 3), (5000, 3)])
def test_grand_product_plain_and_chained(ctx, ar, oracle, n, ncols):
    """SCAN_BLOCK is 2048 elements: sizes below, at, just above and not a multiple of it; with and without the
    permutation argument's chaining through last_z."""
    cols = [rand_ints(n, 900 + 10 * n + c) for c in range(ncols)]
    dev = [zu.ints_to_fr(oracle, c) for c in cols]

    def running(col, start):
        z, acc = [], start
        for v in col:
            z.append(acc)
            acc = acc * v % R
        return z

    got = ar.grand_product(ctx, dev)
    for c in range(ncols):
        assert zu.fr_array_to_ints(got[c]) == running(cols[c], 1)
    u = n - 1 if n < 10 else n - 6  # the row whose value the next column starts from (usable rows)
    got = ar.grand_product(ctx, dev, chain=True, chain_row=u)
    start = 1
    for c in range(ncols):
        want = running(cols[c], start)
        assert zu.fr_array_to_ints(got[c]) == want
        start = want[u]


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 4096, 65536, 65537, 70000, 131072, 262144 + 77])
def test_eval_polynomial(ctx, ar, oracle, n):
    """Above 2^16 coefficients a query is cut into segments of 2^16 evaluated by separate workgroups and folded with x^(2^16)
    (poly_eval_combine_kernel): sizes at, just above and well above one, two and four segments, ragged last segments."""
    polys = [zu.random_fr(n, seed=300 + n + i) for i in range(5)]
    pts = zu.random_fr(5, seed=77 + n)
    pts[0] = 0
    pts[1] = zu.fr_from_int(1)
    pts[2] = zu.fr_from_int(R - 1)
    got = ar.eval_polynomial(ctx, polys, pts)
    for i in range(5):
        assert np.array_equal(got[i], oracle.eval_polynomial(polys[i], pts[i]))


@pytest.mark.parametrize("m,n", [(1, 1000), (3, 1000), (300, 513)])
def test_poly_axpy(ctx, ar, oracle, m, n):
    """m = 300 crosses the kernel's 256-coefficient LDS chunk."""
    polys = [rand_ints(n, 5000 + 7 * j) for j in range(m)]
    coefs = rand_ints(m, 4242 + m)
    got = ar.poly_axpy(ctx, [zu.ints_to_fr(oracle, p) for p in polys], zu.ints_to_fr(oracle, coefs))
    want = [sum(coefs[j] * polys[j][i] for j in range(m)) % R for i in range(n)]
    assert zu.fr_array_to_ints(got) == want


@pytest.mark.parametrize("n", [2, 8, 2048, 2049, 6000])
def test_kate_division(ctx, ar, oracle, n):
    """KD_BLOCK is 2048 coefficients: one block, exactly one, and the multi-block path with a ragged tail."""
    polys = [zu.random_fr(n, seed=600 + n + i) for i in range(3)]
    roots = zu.random_fr(3, seed=9 + n)
    roots[1] = 0
    roots[2] = zu.fr_from_int(1)
    got = ar.kate_division(ctx, polys, roots)
    for i in range(3):
        assert np.array_equal(got[i], oracle.kate_division(polys[i], roots[i]))


class FakeRng:
    def fr(self):
        return 0


def ref_permute(inp, tab, usable):
    a, s = PR.permute_expression_pair(inp, tab, usable, -1, FakeRng())  # bf + 1 = 0 blinding rows appended
    return a, s


MULTISETS = {
    # (input, table) generators over `u` usable rows
    "all_equal": lambda u: ([5] * u, [5] + list(range(100, 100 + u - 1))),
    "all_distinct": lambda u: (list(range(u, 0, -1)), list(range(1, u + 1))),
    "many_leftovers": lambda u: ([3, 3, 3, 9, 9, 1] * (u // 6) + [1] * (u % 6), [1, 3, 9] + list(range(1000, 1000 + u - 3))),
    "big_values": lambda u: ([(R - 1 - (i % 4)) for i in range(u)], [R - 1, R - 2, R - 3, R - 4] + [(1 << 200) + i for i in range(u - 4)]),
    "table_with_repeats": lambda u: ([i % 3 for i in range(u)], [0, 1, 2] + [0] * (u - 3)),
}


@pytest.mark.parametrize("name", sorted(MULTISETS))
def test_permute_expression_pair(ctx, ar, oracle, name):
    n, usable = 64, 58
    inp, tab = MULTISETS[name](usable)
    assert len(inp) == usable and len(tab) == usable
    pad = [12345] * (n - usable)  # whatever sits in the blinding rows must not matter
    a_want, s_want = ref_permute(inp, tab, usable)
    a_got, s_got = ar.permute_expression_pair(ctx, [zu.ints_to_fr(oracle, inp + pad)], [zu.ints_to_fr(oracle, tab + pad)], usable)
    assert zu.fr_array_to_ints(a_got[0]) == a_want + [0] * (n - usable)
    assert zu.fr_array_to_ints(s_got[0]) == s_want + [0] * (n - usable)


@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192])
def test_permute_expression_pair_around_the_sort_tile(ctx, ar, oracle, n):
    """The bitonic sort keeps a 2048-key tile in LDS and drops the workgroup barrier between stages whose stride is at
    most 64 (a wavefront then only hands data to itself: csrc/plonk_kernels.hip bitonic_lds_kernel). Sizes just below, at
    and above the tile — half a tile, one tile, two and four tiles with global passes between them — on the orders that
    move the most keys: strictly descending input (every compare-exchange swaps), and a saw-tooth with many repeats."""
    usable = n - 6
    for name in ("all_distinct", "many_leftovers", "big_values"):
        inp, tab = MULTISETS[name](usable)
        pad = [777] * (n - usable)
        a_want, s_want = ref_permute(inp, tab, usable)
        a_got, s_got = ar.permute_expression_pair(ctx, [zu.ints_to_fr(oracle, inp + pad)], [zu.ints_to_fr(oracle, tab + pad)], usable)
        assert zu.fr_array_to_ints(a_got[0])[:usable] == a_want, (name, n)
        assert zu.fr_array_to_ints(s_got[0])[:usable] == s_want, (name, n)


def test_permute_expression_pair_batch_and_failure(ctx, pkg, ar, oracle):
    n, usable = 128, 122
    pairs = [MULTISETS[k](usable) for k in sorted(MULTISETS)]
    a_got, s_got = ar.permute_expression_pair(ctx, [zu.ints_to_fr(oracle, i + [0] * (n - usable)) for i, _ in pairs],
                                              [zu.ints_to_fr(oracle, t + [0] * (n - usable)) for _, t in pairs], usable)
    for j, (i, t) in enumerate(pairs):
        a_want, s_want = ref_permute(i, t, usable)
        assert zu.fr_array_to_ints(a_got[j])[:usable] == a_want
        assert zu.fr_array_to_ints(s_got[j])[:usable] == s_want
    bad_in = [1, 2, 3, 777] + [1] * (usable - 4) + [0] * (n - usable)
    tab = list(range(usable)) + [0] * (n - usable)
    with pytest.raises(pkg.AmdzkError) as e:
        ar.permute_expression_pair(ctx, [zu.ints_to_fr(oracle, bad_in)], [zu.ints_to_fr(oracle, tab)], usable)
    assert "not in table" in str(e.value)


# ------------------------------------------------------------------------------ a7 / a8 / a9 on the prover's own columns
def prove_and_inspect(ctx, pkg, oracle, c, seed):
    plonk = pkg.plonk
    g, gl = zu.test_srs(oracle, c.k, TAU)
    params = pkg.kzg.ParamsKZG(ctx, c.k, g=g, g_lagrange=gl)
    fixed = np.stack([zu.ints_to_fr(oracle, col) for col in c.fixed])
    pk = plonk.ProvingKey(ctx, params, c.desc, fixed, c.assembly.mapping, zu.fr_from_int(99))
    adv = np.stack([zu.ints_to_fr(oracle, col) for col in c.advice])
    d_adv = ctx.alloc(adv.nbytes).upload(adv)
    inst = [zu.ints_to_fr(oracle, col) if col else np.zeros((0, 4), np.uint64) for col in c.instances]
    proof = plonk.create_proof(ctx, pk, inst, d_adv, seed=seed)
    polys = pk.inspect(0)
    chal = pk.inspect(1)
    pieces = pk.inspect(2)
    to_ints = lambda a: zu.fr_array_to_ints(np.ascontiguousarray(a))
    out = dict(proof=proof, polys_raw=polys, chal_raw=chal, pieces_raw=pieces, polys=[to_ints(p) for p in polys], chal=to_ints(chal),
               pieces=[to_ints(p) for p in pieces], pk=pk)
    d_adv.free()
    return out, params


def circuits_under_test(plonk):
    return [("lookup5", circuits.lookup_circuit(plonk, 5, seed=11)), ("lookup7", circuits.lookup_circuit(plonk, 7, seed=12)),
            ("rsa7", circuits.full_aadhaar_shape(plonk, k=7, num_advice=5, num_lookup_advice=2, lookup_bits=5, num_spread=2, spread_bits=3))]


@pytest.mark.parametrize("which", [0, 1, 2])
def test_quotient_identity_and_product_columns(ctx, pkg, oracle, which):
    plonk = pkg.plonk
    name, c = circuits_under_test(plonk)[which]
    res, params = prove_and_inspect(ctx, pkg, oracle, c, seed=31 + which)
    desc, n = c.desc, c.n
    A, I, L = desc["num_advice"], desc["num_instance"], len(desc["lookups"])
    S = len(desc["permutation_columns"])
    chunk = desc["cs_degree"] - 2
    nsets = (S + chunk - 1) // chunk
    bf = desc["blinding_factors"]
    usable = n - (bf + 1)
    polys = res["polys"]
    assert len(polys) == A + I + 2 * L + nsets + L
    adv, ins = polys[:A], polys[A:A + I]
    la, ls = polys[A + I:A + I + L], polys[A + I + L:A + I + 2 * L]
    zp, zl = polys[A + I + 2 * L:A + I + 2 * L + nsets], polys[A + I + 2 * L + nsets:]
    theta, beta, gamma, y = res["chal"]
    opk = PR.keygen(desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=99)
    d = opk.domain
    # the challenges are the ones the oracle prover derives from the same transcript (byte-equal proofs imply it; this
    # pins the inspection hook itself)
    trace = []
    assert PR.create_proof(opk, c.instances, c.advice, seed=31 + which, trace=trace) == res["proof"]
    tr = {t[0]: t[1] for t in trace if len(t) == 2}
    assert tr["theta"] == theta
    # ---- a7: the vanishing identity at random points, constraint list from the oracle, polynomials from the device
    ev = P.eval_polynomial
    qdeg = desc["cs_degree"] - 1
    assert len(res["pieces"]) == qdeg
    for t in range(3):
        x = (0x9E3779B97F4A7C15 * (t + 3) + (which << 40) + 12345) % R
        lb = PR.lagrange_basis_at(d, n, [0, n - bf - 1] + list(range(n - bf, n)), x)
        l0, l_last = lb[0], lb[n - bf - 1]
        l_blind = sum(lb[i] for i in range(n - bf, n)) % R

        def get(kind, i, rot):
            xr = d.rotate_omega(x, rot)
            if kind == "advice":
                return ev(adv[i], xr)
            if kind == "fixed":
                return ev(opk.fixed_polys[i], xr)
            if kind == "instance":
                return ev(ins[i], xr)
            if kind == "sigma":
                return ev(opk.permutation_polys[i], xr)
            if kind == "z":
                return ev(zp[i], xr)
            if kind == "lz":
                return ev(zl[i], xr)
            if kind == "la":
                return ev(la[i], xr)
            if kind == "ls":
                return ev(ls[i], xr)
            return {"l0": l0, "l_last": l_last, "l_active": (1 - (l_last + l_blind)) % R}[kind]

        vals = PR.h_constraints(desc, d, get, beta, gamma, theta, x, nsets, L)
        acc = 0
        for v in vals:
            acc = (acc * y + v) % R
        xn = pow(x, n, R)
        hx = 0
        for piece in reversed(res["pieces"]):
            hx = (hx * xn + ev(piece, x)) % R
        assert acc == hx * (xn - 1) % R, "%s: h(x)(x^n - 1) != folded constraints at point %d" % (name, t)
    # ---- the function-level entry point gives the same pieces from the same polynomials and challenges
    again = res["pk"].quotient_eval(res["polys_raw"], res["chal_raw"][0], res["chal_raw"][1], res["chal_raw"][2], res["chal_raw"][3])
    assert np.array_equal(again, res["pieces_raw"])
    # and different challenges give different pieces (the call really recomputes)
    other = res["pk"].quotient_eval(res["polys_raw"], res["chal_raw"][0], res["chal_raw"][1], res["chal_raw"][2], res["chal_raw"][0])
    assert not np.array_equal(other, res["pieces_raw"])
    # ---- a8: the permutation product columns from their definition (permutation::prover::Argument::commit)
    lag = lambda poly: d.coeff_to_lagrange(poly)
    colvals = {}
    inst_vals = [list(v) + [0] * (n - len(v)) for v in c.instances]

    def column(kc):
        if kc not in colvals:
            kind, idx = kc
            colvals[kc] = lag(adv[idx]) if kind == 0 else (c.fixed[idx] if kind == 1 else inst_vals[idx])
        return colvals[kc]

    omega_pow = [pow(d.omega, i, R) for i in range(n)]
    last = 1
    for s in range(nsets):
        cols = desc["permutation_columns"][s * chunk:(s + 1) * chunk]
        z, acc = [], last
        for row in range(usable + 1):
            z.append(acc)
            num = den = 1
            for j, kc in enumerate(cols):
                v = column(tuple(kc))[row]
                gi = s * chunk + j
                num = num * ((v + beta * pow(PR.DELTA, gi, R) % R * omega_pow[row] + gamma) % R) % R
                den = den * ((v + beta * opk.permutations[gi][row] + gamma) % R) % R
            acc = acc * num % R * pow(den, -1, R) % R
        assert lag(zp[s])[:usable + 1] == z, "%s: permutation product %d" % (name, s)
        last = z[usable]
    assert last == 1, "%s: the permutation argument must close" % name
    # ---- a9: the lookup product columns (lookup::prover::commit_product) from the committed A', S' and the
    # theta-compressed expressions
    fixed_rows = c.fixed
    adv_rows = [lag(p) for p in adv]
    for li, lk in enumerate(desc["lookups"]):
        a_rows, s_rows = lag(la[li]), lag(ls[li])

        def compressed(exprs, row):
            acc = 0
            for e in exprs:
                v = PR.evaluate_expr(e, lambda cc, r: fixed_rows[cc][(row + r) % n], lambda cc, r: adv_rows[cc][(row + r) % n],
                                     lambda cc, r: inst_vals[cc][(row + r) % n])
                acc = (acc * theta + v) % R
            return acc

        z, acc = [], 1
        for row in range(usable + 1):
            z.append(acc)
            num = (compressed(lk["inputs"], row) + beta) * (compressed(lk["tables"], row) + gamma) % R
            den = (a_rows[row] + beta) * (s_rows[row] + gamma) % R
            acc = acc * num % R * pow(den, -1, R) % R
        assert lag(zl[li])[:usable + 1] == z, "%s: lookup product %d" % (name, li)
        assert z[usable] == 1
        # A' is sorted and S' aligned on the usable rows
        assert a_rows[:usable] == sorted(a_rows[:usable])
        assert all(a_rows[r] == s_rows[r] or a_rows[r] == a_rows[r - 1] for r in range(1, usable)) and a_rows[0] == s_rows[0]
    res["pk"].free()
    params.free()


def test_h_program_of_the_metric_shape_respects_every_bound(ctx, pkg, oracle):
    """The finalised h(X) program of the metric's constraint system (141 advice, 24 lookups, 118 permutation columns —
    the program depends on the shape, not on k, so k = 10 keeps keygen short) walked by the independent bound checker:
    every product within a*b < 169 p^2, every difference with an adequate K*p, nothing stored above 2p; every term
    added to the wide accumulator with its own power of y, and one reduction per hot-column group."""
    import ctypes as C

    import limb_program_check as LC

    plonk = pkg.plonk
    shape = dict(circuits.SHAPES["full"])
    shape.pop("composite")
    shape["k"] = 10
    c = circuits.full_aadhaar_shape(plonk, **shape)
    assert c.desc["num_advice"] == 141 and len(c.desc["lookups"]) == 24 and len(c.desc["permutation_columns"]) == 118
    params = pkg.kzg.ParamsKZG.setup(ctx, c.k, zu.fr_from_int(TAU))
    fixed = np.stack([zu.ints_to_fr(oracle, col) for col in c.fixed])
    pk = plonk.ProvingKey(ctx, params, c.desc, fixed, c.assembly.mapping, zu.fr_from_int(99))
    n = C.c_size_t(0)
    assert pkg.lib().amdzk_pk_h_program(pk.h, None, 0, C.byref(n)) == 0
    words = np.zeros(n.value, dtype=np.uint32)
    assert pkg.lib().amdzk_pk_h_program(pk.h, words.ctypes.data, n.value, C.byref(n)) == 0
    words = [int(w) for w in words]
    depth, nred, nfused = LC.check(words)
    names = [LC.NAME[w >> 24] for w in words]
    nterms = nfused
    # gates: 80 vertical + 12 identity polynomials + 1 square; permutation: 2 + (nsets - 1) + nsets terms with
    # nsets = 118 / (degree - 2) = 59; lookups: 5 per lookup
    nsets = (118 + c.desc["cs_degree"] - 3) // (c.desc["cs_degree"] - 2)
    assert nterms == 93 + (2 + nsets - 1 + nsets) + 5 * 24
    # l_0, l_last, l_active groups + the gates; the program is cut into six pieces (one more flush wherever a cut falls
    # inside a group) of about equal length
    assert LC.check.pieces == 6 and 4 <= names.count("WFLUSH") <= 4 + 5 and "MUL_HOT" not in names
    starts, prev_end = [], 0  # a piece begins behind the previous flush and its first flush carries bit 4
    for i, w in enumerate(words):
        if (w >> 24) == LC.OPS["WFLUSH"]:
            if w & 16:
                starts.append(prev_end)
            prev_end = i + 1
    lens = [b - a for a, b in zip(starts, starts[1:] + [len(words)])]
    assert len(lens) == 6 and max(lens) < 1.5 * len(words) / 6 and min(lens) > 0.5 * len(words) / 6, lens
    # stack: the set's two shared values w_j, the two running products and one copy in flight (round 3: the factored
    # permutation terms); 4 products per 2-column set instead of 8 multiplications by constants and running products
    assert depth <= 5
    assert names.count("PICK") == 2 * 118 and names.count("NIP") == nsets
    assert names.count("MUL_CONST") + names.count("MUL") + names.count("MUL_COL") + names.count("SQR") <= 812 - 118 + 8
    pk.free()
    params.free()
