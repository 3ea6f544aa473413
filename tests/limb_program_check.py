"""Independent checker for programs finalised for the limb-resident interpreter (csrc/plonk_kernels.hip
expr_eval_limbs_kernel): walks the words with every value's bound in units of p, exactly by the rules fp29.cuh
documents for each function, and asserts every precondition — the safety net behind the host pass
(prover.hip finalize_limb_program) that places the weak reductions."""

OPS = dict(END=0, PUSH_COL=1, PUSH_CONST=2, ADD=3, SUB=4, MUL=5, NEG=6, MUL_CONST=7, ADD_CONST=8, MUL_COL=9, ADD_COL=10, SUB_COL=11, ACC=12,
           STORE=13, SQR=14, PUSH_HOT=15, MUL_HOT=16, REDUCE=17, SUB_BIG=18, NEG_BIG=19, WACC=20, WFLUSH=21, PICK=22, NIP=23)
NAME = {v: k for k, v in OPS.items()}
MUL_RANGE = 169.0      # f29_mul / f29_sqr / f29_mul2: a*b (+ c*d) < 169 p^2
VALUE_RANGE = 169.0    # a normalised value must stay below 2^261 = 169.3 p


def word(op, arg=0):
    return (OPS[op] << 24) | (arg & 0xFFFFFF)


def check(words):
    """Returns (max stack depth, number of reductions, number of terms added to the wide accumulator). Raises
    AssertionError. Every power of y a WACC names must be used exactly once (a term dropped or doubled by the grouping
    would go unnoticed by a bound walk alone)."""
    st, depth, nred, nfused = [], 0, 0, 0
    wide, seen, uncarried, flushes, pieces = 0.0, set(), 0, 0, 0
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
        elif op == "ACC":                                    # Horner fold: not an instruction of finalised programs
            raise AssertionError("ACC left in a finalised program at pc %d" % pc)
        elif op == "WACC":                                   # wide += t * y^e: 81 multiply-adds, operands normalised
            j = w & 0x7FFFFF
            assert j not in seen, where + ": power %d used twice" % j
            seen.add(j)
            wide += st.pop() * 1.0                           # in units of p^2 (the power is canonical)
            uncarried = 0 if w & (1 << 23) else uncarried + 1  # bit 23: the carries move up with this term
            assert uncarried < 6, where + ": a column holds six terms of 9 * 2^58 at most"
            nfused += 1
        elif op == "WFLUSH":                                 # g = redc(wide) [* hot]; h = (h +) g, stored canonical
            k = w & 7
            assert k <= 4 and (w & 0xFFFFFF) & ~0x17 == 0, where
            # bit 4: this flush overwrites h — the first flush of the program, or of a further piece of a cut program
            # (pieces run side by side into their own h and are summed afterwards: prover.hip finalize_limb_program)
            assert (w & 16) or flushes > 0, where + ": the first flush must overwrite h"
            pieces += 1 if w & 16 else 0
            flushes += 1
            g = wide / 169.3 + 1.0                           # f29_wide_redc: T / 2^261 + p
            assert g < VALUE_RANGE, where
            if k < 4:
                assert g * 1.0 < MUL_RANGE, where
                g = 2.0
            assert g + 1.0 < VALUE_RANGE, where              # + h (canonical), then f29_reduce_weak
            wide, uncarried = 0.0, 0
        elif op == "PICK":                                   # copy of the entry `arg` below the top
            k = w & 0xFFFFFF
            assert k < len(st), where
            st.append(st[-1 - k])
        elif op == "NIP":                                    # the `arg` entries below the top are discarded
            k = w & 0xFFFFFF
            assert k < len(st), where
            del st[len(st) - 1 - k:len(st) - 1]
        elif op == "STORE":                                  # f29_pack_canonical: below 2p
            assert st.pop() < 2.0, where
        elif op == "END":
            pass
        for v in st:
            assert v < VALUE_RANGE, where
        depth = max(depth, len(st))
    assert not st, "values left on the stack"
    assert wide == 0.0, "terms left in the wide accumulator"
    assert seen == set(range(len(seen))), "the powers of y are not 0..K-1"
    check.pieces = pieces  # how many flushes overwrite h (1, or the pieces of a cut program)
    return depth, nred, nfused
