"""CPU: the host pass that prepares quotient-domain programs for the limb-resident interpreter (prover.hip
finalize_limb_program, reached through amdzk_debug_limb_program — pure host code) on synthetic programs that force each
of its decisions, checked by the independent bound walker in limb_program_check.py."""
import ctypes as C

import numpy as np
import pytest

import limb_program_check as LC

W = LC.word


def finalize(pkg, words):
    L = pkg.lib()
    src = np.array(words, dtype=np.uint32)
    n = C.c_size_t(0)
    depth = C.c_uint32(0)
    assert L.amdzk_debug_limb_program(src.ctypes.data, len(words), None, 0, C.byref(n), C.byref(depth)) == 0
    out = np.zeros(n.value, dtype=np.uint32)
    assert L.amdzk_debug_limb_program(src.ctypes.data, len(words), out.ctypes.data, n.value, C.byref(n), C.byref(depth)) == 0
    return [int(x) for x in out], depth.value


def ops(words):
    return [LC.NAME[w >> 24] for w in words]


def test_gate_term_goes_to_the_wide_accumulator(pkg):
    # q * (a + b*c - d), then h = h*y + that: the term is added as term * y^0, the group flushed without a factor
    prog = [W("PUSH_COL", 1), W("MUL_COL", 2), W("ADD_COL", 3), W("SUB_COL", 4), W("MUL_COL", 5), W("ACC")]
    out, depth = finalize(pkg, prog)
    assert ops(out) == ["PUSH_COL", "MUL_COL", "ADD_COL", "SUB_COL", "MUL_COL", "WACC", "WFLUSH"]
    assert out[-2] & 0xFFFFFF == 0 and out[-1] & 0xFFFFFF == 4 | 16  # power 0; no factor, first (and only) flush
    assert LC.check(out) == (1, 0, 1) and depth >= 1


def test_terms_are_grouped_by_their_hot_factor_and_keep_their_power(pkg):
    # five terms: plain, *hot0, *hot2, *hot0, plain -> groups hot0 {1, 3}, hot2 {2}, plain {0, 4}; the factor is dropped
    # from the term and applied once per group
    t = lambda col, hot=None: [W("PUSH_COL", col), W("SUB_COL", col + 1)] + ([W("MUL_HOT", hot)] if hot is not None else []) + [W("ACC")]
    out, _ = finalize(pkg, t(0) + t(10, 0) + t(20, 2) + t(30, 0) + t(40))
    o = ops(out)
    assert "MUL_HOT" not in o and o.count("WACC") == 5 and o.count("WFLUSH") == 3
    seq = [(LC.NAME[w >> 24], w & 0xFFFFFF) for w in out if LC.NAME[w >> 24] in ("PUSH_COL", "WACC", "WFLUSH")]
    assert seq == [("PUSH_COL", 10), ("WACC", 1), ("PUSH_COL", 30), ("WACC", 3), ("WFLUSH", 0 | 16),
                   ("PUSH_COL", 20), ("WACC", 2), ("WFLUSH", 2),
                   ("PUSH_COL", 0), ("WACC", 0), ("PUSH_COL", 40), ("WACC", 4), ("WFLUSH", 4)]
    assert LC.check(out) == (1, 0, 5)


def test_a_hot_factor_inside_a_term_stays_a_product(pkg):
    # hot * x + c is not `term * hot`: the product must stay in the term
    prog = [W("PUSH_COL", 0), W("MUL_HOT", 1), W("ADD_COL", 2), W("ACC")]
    out, _ = finalize(pkg, prog)
    assert ops(out) == ["PUSH_COL", "MUL_HOT", "ADD_COL", "WACC", "WFLUSH"] and out[-1] & 7 == 4
    LC.check(out)


def test_a_large_group_is_split_and_carries_move_every_sixth_term(pkg):
    # 1500 plain terms of bound ~38: the group is flushed every ~130 terms so the reduced sum stays small (h is kept
    # canonical in its output row between flushes); every sixth term of a group carries
    term = lambda i: [W("PUSH_COL", i)] + [W("ADD_COL", 1)] * 37 + [W("ACC")]
    prog = []
    for i in range(1500):
        prog += term(i)
    out, _ = finalize(pkg, prog)
    flushes = [w & 0xFFFFFF for w in out if LC.NAME[w >> 24] == "WFLUSH"]
    assert len(flushes) >= 10 and all(f & 7 == 4 for f in flushes) and [bool(f & 16) for f in flushes] == [True] + [False] * (len(flushes) - 1)
    since, carries = 0, 0
    for w in out:
        if LC.NAME[w >> 24] == "WACC":
            since += 1
            if w & (1 << 23):
                assert since == 6
                since, carries = 0, carries + 1
        elif LC.NAME[w >> 24] == "WFLUSH":
            since = 0
    assert carries >= 1500 // 6 - len(flushes)
    assert LC.check(out)[2] == 1500


def test_long_sum_is_reduced_before_it_leaves_range(pkg):
    prog = [W("PUSH_COL", 0)] + [W("ADD_COL", i) for i in range(200)] + [W("MUL_COL", 7), W("ACC")]
    out, _ = finalize(pkg, prog)
    assert "REDUCE" in ops(out)
    depth, nred, nfused = LC.check(out)
    assert nred >= 4 and nfused == 1


def test_store_needs_a_value_below_2p(pkg):
    out, _ = finalize(pkg, [W("PUSH_CONST", 0), W("SUB_COL", 1), W("SUB_COL", 2), W("STORE", 0)])
    assert ops(out) == ["PUSH_CONST", "SUB_COL", "SUB_COL", "REDUCE", "STORE"]
    LC.check(out)


def test_subtrahend_bounds_pick_the_constant(pkg):
    small = [W("PUSH_COL", 0), W("PUSH_COL", 1), W("SUB"), W("MUL_COL", 2), W("ACC")]                       # t below 2p: K = 3
    big = [W("PUSH_COL", 0), W("PUSH_COL", 1), W("ADD_COL", 2), W("ADD_COL", 3), W("SUB"), W("MUL_COL", 2), W("ACC")]  # below 9p: K = 10
    huge = [W("PUSH_COL", 0), W("PUSH_COL", 1)] + [W("ADD_COL", 2)] * 12 + [W("SUB"), W("MUL_COL", 2), W("ACC")]       # reduced first
    assert "SUB" in ops(finalize(pkg, small)[0])
    assert "SUB_BIG" in ops(finalize(pkg, big)[0])
    o = ops(finalize(pkg, huge)[0])
    assert o.index("REDUCE") < o.index("SUB")
    for p in (small, big, huge):
        LC.check(finalize(pkg, p)[0])
    neg = [W("PUSH_COL", 0), W("ADD_COL", 1), W("ADD_COL", 1), W("NEG"), W("MUL_COL", 2), W("ACC")]
    assert "NEG_BIG" in ops(finalize(pkg, neg)[0])
    LC.check(finalize(pkg, neg)[0])


def test_values_sinking_into_the_stack_are_kept_small(pkg):
    # a value with a bound above 8 is reduced before another push buries it; the stack product then stays in range
    prog = [W("PUSH_COL", 0)] + [W("ADD_COL", 1)] * 12 + [W("PUSH_COL", 2)] + [W("ADD_COL", 3)] * 12 + [W("MUL"), W("ACC")]
    out, depth = finalize(pkg, prog)
    o = ops(out)
    assert o.count("REDUCE") >= 1 and o[-3:] == ["MUL", "WACC", "WFLUSH"]
    assert LC.check(out)[0] == 2 and depth >= 2


def test_squares_and_hot_terms(pkg):
    prog = [W("PUSH_COL", 0)] + [W("ADD_COL", 1)] * 13 + [W("SQR"), W("SUB_COL", 0), W("MUL_HOT", 1), W("ACC"),
                                                        W("PUSH_COL", 3), W("ACC")]
    out, _ = finalize(pkg, prog)
    o = ops(out)
    assert o.index("REDUCE") < o.index("SQR")       # 14^2 > 169
    assert "MUL_HOT" not in o and o.count("WFLUSH") == 2   # group hot1 {0}, then the plain term
    LC.check(out)


def test_checker_rejects_an_unreduced_program():
    bad = [W("PUSH_COL", 0)] + [W("ADD_COL", 1)] * 14 + [W("SQR"), W("STORE", 0)]
    with pytest.raises(AssertionError):
        LC.check(bad)


def test_shared_values_are_picked_from_the_stack(pkg):
    """The factored permutation term of a 2-column set, l_active * (z(wX) (s0 + w0)(s1 + w1) - z(X) (d0 + w0)(d1 + w1)) with
    w_j = (v_j + gamma) / beta computed once and copied (OP_PICK) into both products, then dropped from under the result
    (OP_NIP): the pass keeps the structure, tracks the copies' bounds, and the closing `* hot 2` still leaves the term."""
    v0, v1, s0, s1, d0, d1, z = 1, 2, 3, 4, 5, 6, 7
    prog = [W("PUSH_COL", v0), W("ADD_CONST", 1), W("MUL_CONST", 2),
            W("PUSH_COL", v1), W("ADD_CONST", 1), W("MUL_CONST", 2),
            W("PUSH_COL", z),
            W("PICK", 2), W("ADD_COL", s0), W("MUL"),
            W("PICK", 1), W("ADD_COL", s1), W("MUL"),
            W("PUSH_COL", z),
            W("PICK", 3), W("ADD_COL", d0), W("MUL"),
            W("PICK", 2), W("ADD_COL", d1), W("MUL"),
            W("SUB"), W("NIP", 2), W("MUL_HOT", 2), W("ACC")]
    out, depth = finalize(pkg, prog)
    o = ops(out)
    assert o.count("PICK") == 4 and o.count("NIP") == 1 and "MUL_HOT" not in o and o[-2:] == ["WACC", "WFLUSH"]
    assert out[-1] & 7 == 2                       # the group of l_active
    assert o.count("MUL_CONST") == 2 and o.count("MUL") == 4   # 6 products for 2 columns (8 before the factoring)
    d, nred, nterms = LC.check(out)
    assert d == 5 and nterms == 1 and depth >= 5


def test_checker_rejects_a_pick_below_the_stack():
    with pytest.raises(AssertionError):
        LC.check([W("PUSH_COL", 1), W("PICK", 1), W("ADD"), W("WACC", 0), W("WFLUSH", 4 | 16)])
