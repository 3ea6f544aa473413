#!/usr/bin/env python3
"""Per-kernel instruction totals from a rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
SQ_WAVES pass (counter_collection.csv): how many wave-instructions one run issued, and how long they take at a
given issue cost — the check of whether throughput is bound by instruction issue.
Usage: python tools/summarize_insts.py <counter_collection.csv> [proofs_in_run] > profiles/<tag>_valu_instruction_counts.txt"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:44]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    proofs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for r in rows:
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            calls[k] += 1
    startup = ("table_next_kernel", "table_to_r261_kernel", "fixed_base_mul_kernel", "powers_kernel", "twiddle_gen_kernel",
               "sub_from_const_kernel", "mul2_kernel")
    per_proof = sum(v.get("SQ_INSTS_VALU", 0) for k, v in agg.items() if k not in startup)
    print("%-46s %6s %12s %10s %10s %10s %10s" % ("kernel", "calls", "VALU", "SALU", "LDS", "VMEM_RD", "waves"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:22]:
        print("%-46s %6d %12.3e %10.2e %10.2e %10.2e %10.2e%s" % (k, calls[k], v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0),
                                                                v.get("SQ_INSTS_LDS", 0), v.get("SQ_INSTS_VMEM_RD", 0), v.get("SQ_WAVES", 0),
                                                                "   (start-up)" if k in startup else ""))
    print()
    print("VALU wave-instructions outside the start-up kernels: %.3e over %.2f proofs = %.3e per proof" % (per_proof, proofs, per_proof / proofs))
    for cpi in (4.0, 4.5, 5.0):
        t = per_proof / proofs * cpi / (1024 * 2.4e9) * 1e3
        print("  at %.1f cycles per VALU wave-instruction on 1024 SIMDs at 2.4 GHz: %.1f ms per proof" % (cpi, t))


if __name__ == "__main__":
    main()
