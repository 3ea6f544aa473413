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


def test_gate_shape_fuses_the_accumulate(pkg):
    # q * (a + b*c - d), then h = h*y + that: the closing product joins the accumulate
    prog = [W("PUSH_COL", 1), W("MUL_COL", 2), W("ADD_COL", 3), W("SUB_COL", 4), W("MUL_COL", 5), W("ACC")]
    out, depth = finalize(pkg, prog)
    assert ops(out) == ["PUSH_COL", "MUL_COL", "ADD_COL", "SUB_COL", "ACC_MUL_COL"]
    assert out[-1] & 0xFFFFFF == 5  # the operand of the fused product is kept
    assert LC.check(out) == (1, 0, 1) and depth >= 1


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
    assert o.count("REDUCE") >= 1 and o[-1] == "ACC_MUL"
    assert LC.check(out)[0] == 2 and depth >= 2


def test_squares_and_unfusable_accumulates(pkg):
    prog = [W("PUSH_COL", 0)] + [W("ADD_COL", 1)] * 13 + [W("SQR"), W("SUB_COL", 0), W("MUL_HOT", 1), W("ACC"),
                                                        W("PUSH_COL", 3), W("ACC")]
    out, _ = finalize(pkg, prog)
    o = ops(out)
    assert o.index("REDUCE") < o.index("SQR")       # 14^2 > 169
    assert "ACC_MUL_HOT" in o and o[-1] == "ACC"    # the second accumulate has no product in front of it
    LC.check(out)


def test_checker_rejects_an_unreduced_program():
    bad = [W("PUSH_COL", 0)] + [W("ADD_COL", 1)] * 14 + [W("SQR"), W("STORE", 0)]
    with pytest.raises(AssertionError):
        LC.check(bad)
