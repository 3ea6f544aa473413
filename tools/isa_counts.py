#!/usr/bin/env python3
"""Static instruction mix per kernel from a `hipcc -save-temps` gfx950 assembly file.
Usage: python tools/isa_counts.py <file.s> [name-substring ...]
For each kernel: VGPRs, spills, and counts of v_mad_u64_u32 / other VALU / SALU / VMEM / LDS instructions."""
import collections
import re
import sys


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2:]
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
        blk = m.group(2)
        g = lambda k: int(re.search(k + r":\s+(\d+)", blk).group(1)) if re.search(k + r":\s+(\d+)", blk) else -1
        meta[m.group(1)] = (g(r"\.vgpr_count"), g(r"\.vgpr_spill_count"), g(r"\.sgpr_count"), g(r"\.group_segment_fixed_size"))
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if name not in meta or (want and not any(w in name for w in want)):
            continue
        c = collections.Counter()
        ops = collections.Counter()
        for ln in body.splitlines():
            ln = ln.strip()
            mm = re.match(r"([vsd]\w*|global_\w+|buffer_\w+|flat_\w+|scratch_\w+)\b", ln)
            if not mm:
                continue
            op = mm.group(1)
            ops[op] += 1
            if op == "v_mad_u64_u32":
                c["mad64"] += 1
            elif op.startswith("v_"):
                c["valu_other"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            else:
                c["vmem"] += 1
        v, sp, sg, lds = meta[name]
        print("%s\n   vgpr %d spill %d sgpr %d lds %d | mad64 %d  other VALU %d  SALU %d  VMEM %d  LDS %d" %
              (name, v, sp, sg, lds, c["mad64"], c["valu_other"], c["salu"], c["vmem"], c["lds"]))
        print("   top VALU:", ", ".join("%s %d" % kv for kv in ops.most_common(12)))


if __name__ == "__main__":
    main()
