"""Independent checker for programs finalised for the limb-resident interpreter (csrc/plonk_kernels.hip
expr_eval_limbs_kernel): walks the words with every value's bound in units of p, exactly by the rules fp29.cuh
documents for each function, and asserts every precondition — the safety net behind the host pass
(prover.hip finalize_limb_program) that places the weak reductions."""

OPS = dict(END=0, PUSH_COL=1, PUSH_CONST=2, ADD=3, SUB=4, MUL=5, NEG=6, MUL_CONST=7, ADD_CONST=8, MUL_COL=9, ADD_COL=10, SUB_COL=11, ACC=12,
           STORE=13, SQR=14, PUSH_HOT=15, MUL_HOT=16, REDUCE=17, SUB_BIG=18, NEG_BIG=19, ACC_MUL_COL=20, ACC_MUL_CONST=21, ACC_MUL_HOT=22,
           ACC_MUL=23)
NAME = {v: k for k, v in OPS.items()}
MUL_RANGE = 169.0      # f29_mul / f29_sqr / f29_mul2: a*b (+ c*d) < 169 p^2
VALUE_RANGE = 169.0    # a normalised value must stay below 2^261 = 169.3 p


def word(op, arg=0):
    return (OPS[op] << 24) | (arg & 0xFFFFFF)


def check(words):
    """Returns (max stack depth, number of reductions, number of fused accumulates). Raises AssertionError."""
    st, h, depth, nred, nfused = [], 0.0, 0, 0, 0
    for pc, w in enumerate(words):
        op = NAME.get(w >> 24)
        where = "pc %d %s" % (pc, op)
        assert op is not None, "unknown opcode at pc %d" % pc
        if op in ("PUSH_COL", "PUSH_CONST", "PUSH_HOT"):
            st.append(1.0)                                   # canonical value
        elif op in ("MUL_COL", "MUL_CONST", "MUL_HOT"):
            assert st[-1] * 1.0 < MUL_RANGE, where
            st[-1] = 2.0
        elif op in ("ADD_COL", "ADD_CONST"):
            st[-1] += 1.0
        elif op == "SUB_COL":                                # f29_sub3: subtrahend (canonical) below 2p
            st[-1] += 3.0
        elif op == "ADD":
            b = st.pop()
            st[-1] += b
        elif op in ("SUB", "SUB_BIG"):                       # f29_sub3 / f29_sub10: subtrahend below 2p / 9p
            b = st.pop()
            assert b < (2.0 if op == "SUB" else 9.0), where
            st[-1] += 3.0 if op == "SUB" else 10.0
        elif op == "MUL":
            b = st.pop()
            assert st[-1] * b < MUL_RANGE, where
            st[-1] = 2.0
        elif op in ("NEG", "NEG_BIG"):                       # f29_neg3 / f29_neg10
            assert st[-1] < (2.0 if op == "NEG" else 9.0), where
            st[-1] = 3.0 if op == "NEG" else 10.0
        elif op == "SQR":
            assert st[-1] * st[-1] < MUL_RANGE, where
            st[-1] = 2.0
        elif op == "REDUCE":                                 # f29_reduce_weak: any normalised value -> below 1.0002 p
            assert st[-1] < VALUE_RANGE, where
            st[-1] = 1.0002
            nred += 1
        elif op == "ACC":                                    # h = h*y + t
            assert h * 1.0 < MUL_RANGE, where
            h = 2.0 + st.pop()
        elif op in ("ACC_MUL_COL", "ACC_MUL_CONST", "ACC_MUL_HOT"):
            assert h * 1.0 + st[-1] * 1.0 < MUL_RANGE, where  # f29_mul2(h, y, t, x)
            st.pop()
            h = 2.0
            nfused += 1
        elif op == "ACC_MUL":
            b = st.pop()
            a = st.pop()
            assert h * 1.0 + a * b < MUL_RANGE, where
            h = 2.0
            nfused += 1
        elif op == "STORE":                                  # f29_pack_canonical: below 2p
            assert st.pop() < 2.0, where
        elif op == "END":
            pass
        for v in st:
            assert v < VALUE_RANGE, where
        assert h < VALUE_RANGE, where
        depth = max(depth, len(st))
    assert not st, "values left on the stack"
    return depth, nred, nfused
