#!/usr/bin/env python3
"""Dynamic instruction counts of the expression interpreters WITHOUT a GPU: walks the gfx950 assembly of
expr_eval_limbs_kernel (hipcc -save-temps) for ONE wave, executing only what decides the control flow — the scalar
ALU, the wave-uniform VGPRs (counters, flags) and the loads of the instruction stream — and counts every instruction
that is issued on the way. The interpreter's control flow depends on the program alone, never on the data, so the walk
is exact for a given program.

Usage:
  python tools/isa_walk.py <file.s> <kernel-name-substring> per-op          cost of every opcode (synthetic programs)
  python tools/isa_walk.py <file.s> <kernel-name-substring> words.txt       a finalised program, one word per line (hex)

Scope: the handful of SALU / VALU forms these kernels use; an instruction the walker would need but does not know
raises, it never guesses. Data-dependent VALU results are 'unknown' and must not reach a branch."""
import collections
import re
import sys

FULL = (1 << 64) - 1
M32 = 0xFFFFFFFF
KA, PB, CB, OUTS, HOUT, COLDATA = 0x10000, 0x200000, 0x300000, 0x310000, 0x400000, 0x1000000


def s32(x):
    x &= M32
    return x - (1 << 32) if x & 0x80000000 else x


class Walker:
    def __init__(self, asm, kernel):
        m = re.search(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)^\.Lfunc_end\d+:" % re.escape(kernel), asm, re.S | re.M)
        if not m:
            raise SystemExit("kernel %s not found" % kernel)
        self.ins, self.labels = [], {}
        for ln in m.group(2).splitlines():
            t = ln.split(";")[0].strip()
            if not t:
                continue
            lm = re.match(r"^(\.LBB\d+_\d+):", t)
            if lm:
                self.labels[lm.group(1)] = len(self.ins)
                continue
            if t.startswith("."):
                continue
            parts = t.split(None, 1)
            ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
            self.ins.append((parts[0], ops, t))

    # ---- operands
    def rd(self, o, wide=False):
        o = o.strip()
        if o == "vcc":
            return self.vcc
        if o == "exec":
            return self.exec
        if o == "scc":
            return self.scc
        m = re.match(r"^([sv])\[(\d+):(\d+)\]$", o)
        if m:
            f = self.s if m.group(1) == "s" else self.v
            lo, hi = f.get(int(m.group(2))), f.get(int(m.group(2)) + 1)
            if lo is None or hi is None:
                return None
            return (lo & M32) | ((hi & M32) << 32)
        m = re.match(r"^([sv])(\d+)$", o)
        if m:
            return (self.s if m.group(1) == "s" else self.v).get(int(m.group(2)))
        try:
            v = int(o, 0)
        except ValueError:
            raise NotImplementedError("operand " + o)
        return v & (FULL if wide else M32)

    def wr(self, o, val):
        o = o.strip()
        if o == "vcc":
            self.vcc = val
            return
        if o == "exec":
            self.exec = val
            return
        m = re.match(r"^([sv])\[(\d+):(\d+)\]$", o)
        if m:
            f = self.s if m.group(1) == "s" else self.v
            a, b = int(m.group(2)), int(m.group(3))
            for i in range(a, b + 1):
                f[i] = None if val is None else (val >> (32 * (i - a))) & M32
            return
        m = re.match(r"^([sv])(\d+)$", o)
        f = self.s if m.group(1) == "s" else self.v
        f[int(m.group(2))] = None if val is None else val & M32

    def mem(self, addr, ndw):
        out = 0
        for i in range(ndw):
            w = self.memory.get(addr + 4 * i)
            if w is None:
                return None
            out |= (w & M32) << (32 * i)
        return out

    def run(self, words, limit=50_000_000):
        self.s, self.v = collections.defaultdict(lambda: None), collections.defaultdict(lambda: None)
        self.scc, self.vcc, self.exec = 0, 0, FULL
        self.memory = {}
        put = lambda a, v, n=1: [self.memory.__setitem__(a + 4 * i, (v >> (32 * i)) & M32) for i in range(n)]
        put(KA + 0x00, PB, 2); put(KA + 0x08, len(words)); put(KA + 0x10, CB, 2); put(KA + 0x18, OUTS, 2)
        # ExprArgs (plonk_kernels.hpp): prog, prog_len, cols, outs, h_out, mask, nrows, hot[4], radix261, nparts, part_start[8], part_len[8]
        put(KA + 0x20, HOUT, 2); put(KA + 0x28, (1 << 15) - 1, 2); put(KA + 0x30, 3 << 15, 2)
        for i in range(4):
            put(KA + 0x38 + 4 * i, i)
        put(KA + 0x48, 1)
        for i in range(17):
            put(KA + 0x4C + 4 * i, 0)  # nparts = 0 (the whole program), part_start[8], part_len[8]
        for i in range(64):
            put(CB + 8 * i, COLDATA + (i << 24), 2)
            put(OUTS + 8 * i, COLDATA + ((64 + i) << 24), 2)
        for i, w in enumerate(list(words) + [0, 0]):  # the host closes every program with two END instructions
            put(PB + 16 * i, w); put(PB + 16 * i + 4, 0); put(PB + 16 * i + 8, COLDATA, 2)
        self.s[0], self.s[1] = KA & M32, KA >> 32
        for r in range(2, 16):
            self.s[r] = 0  # workgroup id etc.
        self.v[0] = None  # thread id: divergent
        cnt = collections.Counter()
        pc, steps = 0, 0
        while True:
            steps += 1
            if steps > limit:
                raise RuntimeError("walk does not terminate")
            op, o, text = self.ins[pc]
            pc += 1
            if op.startswith("v_"):
                cnt["valu"] += 1
                if op == "v_mad_u64_u32":
                    cnt["mad64"] += 1
            elif op.startswith("s_"):
                cnt["salu"] += 1
            elif op.startswith("ds_"):
                cnt["lds"] += 1
            else:
                cnt["vmem"] += 1
            try:
                tgt = self.step(op, o)
            except NotImplementedError as e:
                raise NotImplementedError("%s   (%s)" % (text, e))
            if tgt == "END":
                return cnt
            if tgt is not None:
                pc = self.labels[tgt]

    def need(self, *vals):
        for v in vals:
            if v is None:
                raise NotImplementedError("a data-dependent value reaches the control flow")
        return vals if len(vals) > 1 else vals[0]

    def step(self, op, o):
        R, W = self.rd, self.wr
        if op == "s_endpgm":
            return "END"
        if op in ("s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_sleep"):
            return None
        if op == "s_branch":
            return o[0]
        if op == "s_cbranch_scc1":
            return o[0] if self.scc else None
        if op == "s_cbranch_scc0":
            return None if self.scc else o[0]
        if op == "s_cbranch_vccnz":
            return o[0] if self.need(self.vcc) & self.exec else None  # uniform masks: 0 or the exec mask
        if op == "s_cbranch_vccz":
            return None if self.need(self.vcc) & self.exec else o[0]
        if op == "s_cbranch_execz":
            return o[0] if self.exec == 0 else None
        if op == "s_cbranch_execnz":
            return o[0] if self.exec != 0 else None
        if op in ("s_mov_b32", "s_mov_b64"):
            W(o[0], R(o[1], op.endswith("64")) if not (op.endswith("64") and o[1] == "-1") else FULL)
            return None
        if op == "s_brev_b32":
            W(o[0], int("{:032b}".format(R(o[1]) & M32)[::-1], 2))
            return None
        if op.startswith("s_cmp_"):
            a, b = R(o[0], op.endswith("u64")), R(o[1], op.endswith("u64"))
            self.need(a, b)
            k = op[6:]
            if k.endswith("i32"):
                a, b = s32(a), s32(b)
            self.scc = int({"eq": a == b, "lg": a != b, "lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b}[k.split("_")[0]])
            return None
        if op == "s_bitcmp0_b32":
            self.scc = int(not (self.need(R(o[0])) >> (R(o[1]) & 31)) & 1)
            return None
        if op == "s_bitcmp1_b32":
            self.scc = int((self.need(R(o[0])) >> (R(o[1]) & 31)) & 1)
            return None
        if op == "s_cselect_b64":
            a, b = (FULL if o[1] == "-1" else R(o[1], True)), (FULL if o[2] == "-1" else R(o[2], True))
            W(o[0], a if self.scc else b)
            return None
        if op == "s_cselect_b32":
            W(o[0], R(o[1]) if self.scc else R(o[2]))
            return None
        if op in ("s_and_b64", "s_or_b64", "s_andn2_b64", "s_xor_b64", "s_and_b32", "s_or_b32", "s_andn2_b32", "s_xor_b32"):
            w = op.endswith("64")
            a = FULL if o[1] == "-1" and w else R(o[1], w)
            b = FULL if o[2] == "-1" and w else R(o[2], w)
            if a is None or b is None:
                # masks of divergent compares only matter where they reach a branch
                W(o[0], None)
                self.scc = 1
                return None
            k = op.split("_")[1]
            r = {"and": a & b, "or": a | b, "andn2": a & ~b, "xor": a ^ b}[k] & (FULL if w else M32)
            W(o[0], r)
            self.scc = int(r != 0)
            return None
        if op == "s_and_saveexec_b64":
            src = R(o[1], True)
            W(o[0], self.exec)
            self.exec = self.exec if src is None else self.exec & src  # divergent guard: this wave's rows are all live
            self.scc = int(self.exec != 0)
            return None
        if op in ("s_add_i32", "s_add_u32", "s_sub_i32", "s_sub_u32", "s_addc_u32", "s_mul_i32", "s_mul_hi_u32", "s_lshl_b32", "s_lshr_b32", "s_ashr_i32"):
            a, b = R(o[1]), R(o[2])
            if a is None or b is None:
                W(o[0], None)
                return None
            k = op[2:]
            if k in ("add_i32", "add_u32"):
                r = a + b
                self.scc = int(r > M32) if k == "add_u32" else 0
            elif k == "addc_u32":
                r = a + b + self.scc
                self.scc = int(r > M32)
            elif k in ("sub_i32", "sub_u32"):
                r = a - b
                self.scc = int(r < 0)
            elif k == "mul_i32":
                r = a * b
            elif k == "mul_hi_u32":
                r = ((a & M32) * (b & M32)) >> 32
            elif k == "lshl_b32":
                r = a << (b & 31)
            elif k == "lshr_b32":
                r = (a & M32) >> (b & 31)
            else:
                r = s32(a) >> (b & 31)
            W(o[0], r & M32)
            return None
        if op == "s_lshl_b64":
            a = R(o[1], True)
            W(o[0], None if a is None else (a << (R(o[2]) & 63)) & FULL)
            return None
        if op.startswith("s_load_dword"):
            ndw = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8}[op]
            base = R(o[1], True)
            off = 0
            for extra in o[2:]:
                for tok in extra.split():
                    if tok.startswith("offset:"):
                        off += int(tok[7:], 0)
                    else:
                        off += self.need(R(tok))
            val = None if base is None else self.mem(base + off, ndw)
            if ndw == 1:
                W(o[0], val)
            else:
                m = re.match(r"s\[(\d+):(\d+)\]", o[0])
                for i in range(ndw):
                    self.s[int(m.group(1)) + i] = None if val is None else (val >> (32 * i)) & M32
            return None
        if op.startswith("s_"):
            raise NotImplementedError("scalar op")
        # ---- vector side: only what feeds the control flow is tracked
        if op in ("global_load_dword", "global_load_dwordx2") and len(o) >= 3 and o[2].split()[0].startswith("s["):
            base = R(o[2].split()[0], True)
            voff = R(o[1])
            off = 0
            for tok in o[2].split()[1:] + sum([x.split() for x in o[3:]], []):
                if tok.startswith("offset:"):
                    off = int(tok[7:], 0)
            ndw = 1 if op.endswith("dword") else 2
            val = None
            if base is not None and voff is not None:
                val = self.mem(base + voff + off, ndw)
            W(o[0], val)
            return None
        if op.startswith(("global_", "flat_", "ds_", "scratch_", "buffer_")):
            if op.startswith(("global_load", "flat_load", "ds_read", "ds_load")):
                W(o[0], None)
            return None
        if op == "v_readfirstlane_b32":
            W(o[0], R(o[1]))  # unknown stays unknown: it only matters if it reaches a compare or a branch
            return None
        if op in ("v_mov_b32_e32", "v_mov_b64_e32"):
            W(o[0], R(o[1], op.startswith("v_mov_b64")))
            return None
        if op in ("v_add_u32_e32", "v_sub_u32_e32"):
            a, b = R(o[1]), R(o[2])
            W(o[0], None if a is None or b is None else ((a + b) if "add" in op else (a - b)) & M32)
            return None
        if op in ("v_cndmask_b32_e64", "v_cndmask_b32_e32"):
            mask = R(o[3], True) if op.endswith("e64") else self.vcc
            a, b = R(o[1]), R(o[2])
            if mask is None or a is None or b is None:
                W(o[0], None)
            else:
                W(o[0], b if mask & self.exec else a)
            return None
        if op.startswith("v_cmp_"):
            dst, a, b = (o[0], o[1], o[2]) if len(o) == 3 else ("vcc", o[0], o[1])
            w = "64" in op.split("_")[3]
            x, y = R(a, w), R(b, w)
            if x is None or y is None:
                W(dst, None)
                return None
            k = op.split("_")[2]
            if "i32" in op:
                x, y = s32(x), s32(y)
            r = {"eq": x == y, "ne": x != y, "lg": x != y, "lt": x < y, "le": x <= y, "gt": x > y, "ge": x >= y}[k]
            val = self.exec if r else 0
            W(dst, val)
            return None
        # any other VALU instruction: data
        if o:
            d = o[0]
            if re.match(r"^[sv](\d+|\[\d+:\d+\])$", d) or d == "vcc":
                if d == "vcc":
                    self.vcc = None
                else:
                    W(d, None)
            if op in ("v_mad_u64_u32", "v_mad_i64_i32") and len(o) > 1 and re.match(r"^s\[\d+:\d+\]$|^vcc$", o[1]):
                W(o[1], None) if o[1] != "vcc" else setattr(self, "vcc", None)
        return None


OPS = dict(END=0, PUSH_COL=1, PUSH_CONST=2, ADD=3, SUB=4, MUL=5, NEG=6, MUL_CONST=7, ADD_CONST=8, MUL_COL=9, ADD_COL=10, SUB_COL=11, ACC=12,
           STORE=13, SQR=14, PUSH_HOT=15, MUL_HOT=16, REDUCE=17, SUB_BIG=18, NEG_BIG=19, WACC=20, WFLUSH=21, PICK=22, NIP=23)
NAME = {v: k for k, v in OPS.items()}


def W_(op, arg=0):
    return (OPS[op] << 24) | arg


def per_op(wk, names):
    """Marginal cost of one more instance of each opcode inside a program that keeps the stack balanced."""
    reps = 24
    unit = {  # (prefix, repeated body, suffix): the body leaves the stack as it found it
        "PUSH_COL+WACC": ([], [W_("PUSH_COL", 1), W_("WACC", 0)], []),
        "PUSH_CONST+WACC": ([], [W_("PUSH_CONST", 1), W_("WACC", 0)], []),
        "PUSH_HOT3+WACC": ([], [W_("PUSH_HOT", 3), W_("WACC", 0)], []),
        "ADD_COL": ([W_("PUSH_COL", 1)], [W_("ADD_COL", 2)], [W_("WACC", 0)]),
        "ADD_CONST": ([W_("PUSH_COL", 1)], [W_("ADD_CONST", 2)], [W_("WACC", 0)]),
        "SUB_COL": ([W_("PUSH_COL", 1)], [W_("SUB_COL", 2)], [W_("WACC", 0)]),
        "MUL_COL": ([W_("PUSH_COL", 1)], [W_("MUL_COL", 2)], [W_("WACC", 0)]),
        "MUL_CONST": ([W_("PUSH_COL", 1)], [W_("MUL_CONST", 2)], [W_("WACC", 0)]),
        "MUL_HOT3": ([W_("PUSH_COL", 1)], [W_("MUL_HOT", 3)], [W_("WACC", 0)]),
        "SQR": ([W_("PUSH_COL", 1)], [W_("SQR")], [W_("WACC", 0)]),
        "REDUCE": ([W_("PUSH_COL", 1)], [W_("REDUCE")], [W_("WACC", 0)]),
        "NEG": ([W_("PUSH_COL", 1)], [W_("NEG")], [W_("WACC", 0)]),
        "NEG_BIG": ([W_("PUSH_COL", 1)], [W_("NEG_BIG")], [W_("WACC", 0)]),
        "PUSH_COL+MUL": ([W_("PUSH_COL", 1)], [W_("PUSH_COL", 2), W_("MUL")], [W_("WACC", 0)]),
        "PUSH_COL+ADD": ([W_("PUSH_COL", 1)], [W_("PUSH_COL", 2), W_("ADD")], [W_("WACC", 0)]),
        "PUSH_COL+SUB": ([W_("PUSH_COL", 1)], [W_("PUSH_COL", 2), W_("SUB")], [W_("WACC", 0)]),
        "PUSH_COL+SUB_BIG": ([W_("PUSH_COL", 1)], [W_("PUSH_COL", 2), W_("SUB_BIG")], [W_("WACC", 0)]),
        "WFLUSH(hot0)": ([], [W_("WFLUSH", 0)], []),
        "PUSH_COL+WACC(carry)": ([], [W_("PUSH_COL", 1), W_("WACC", 1 << 23)], []),
        "WFLUSH(plain)": ([], [W_("WFLUSH", 4)], []),
    }
    print("%-20s %8s %8s %8s %6s %6s" % ("opcode(s)", "VALU", "mad64", "SALU", "LDS", "VMEM"))
    for name, (pre, body, suf) in unit.items():
        if any(NAME[w >> 24] not in names for w in pre + body + suf):
            continue
        a = wk.run(pre + body * reps + suf)
        b = wk.run(pre + body * (2 * reps) + suf)
        d = {k: (b[k] - a[k]) / reps for k in ("valu", "mad64", "salu", "lds", "vmem")}
        print("%-20s %8.1f %8.1f %8.1f %6.1f %6.1f" % (name, d["valu"], d["mad64"], d["salu"], d["lds"], d["vmem"]))
    e = wk.run([])
    print("%-20s %8d %8d %8d %6d %6d   (empty program: prologue + epilogue)" % ("-", e["valu"], e["mad64"], e["salu"], e["lds"], e["vmem"]))


def main():
    asm = open(sys.argv[1]).read()
    wk = Walker(asm, sys.argv[2])
    if sys.argv[3] == "per-op":
        per_op(wk, set(OPS))
        return
    words = [int(x, 0) for x in open(sys.argv[3]).read().split()]
    c = wk.run(words)
    hist = collections.Counter(NAME[w >> 24] for w in words)
    print("%d instructions: %s" % (len(words), dict(hist)))
    print("one wave: VALU %d (v_mad_u64_u32 %d), SALU %d, LDS %d, VMEM %d" % (c["valu"], c["mad64"], c["salu"], c["lds"], c["vmem"]))
    print("per program instruction: %.1f VALU" % (c["valu"] / max(1, len(words))))


if __name__ == "__main__":
    main()
