#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof/{stats,fetch,write}) into profiles/<tag>_kernel_summary.csv.
Usage: python tools/summarize_prof.py gpurun_out/prof r01"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:60]


def main():
    src, tag = sys.argv[1], sys.argv[2]
    os.makedirs("profiles", exist_ok=True)
    stats = {}
    f = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
    if f:
        for r in csv.DictReader(open(f[0])):
            stats[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["Percentage"]))
    pmc = {}
    for kind in ("fetch", "write"):
        f = glob.glob(os.path.join(src, kind, "*", "*_counter_collection.csv"))
        agg = collections.defaultdict(lambda: [0, 0.0])
        if f:
            for r in csv.DictReader(open(f[0])):
                k = short(r["Kernel_Name"])
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
        pmc[kind] = agg
    rows = []
    for k, (calls, avg_ms, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
        if k.startswith("at::") or "rocclr" in k:
            continue
        fe = pmc["fetch"].get(k)
        wr = pmc["write"].get(k)
        rows.append((k, calls, avg_ms, pct, fe[1] / fe[0] if fe else None, wr[1] / wr[0] if wr else None))
    path = os.path.join("profiles", tag + "_kernel_summary.csv")
    with open(path, "w") as o:
        o.write("kernel,calls,avg_ms,pct_of_gpu_time,FETCH_SIZE_KiB_per_launch_raw,WRITE_SIZE_KiB_per_launch_raw,"
                "hbm_MB_per_launch_corrected(2*FETCH+WRITE)\n")
        for k, calls, avg_ms, pct, fe, wr in rows:
            corr = "" if fe is None or wr is None else "%.2f" % ((2 * fe + wr) * 1024 / 1e6)
            o.write("%s,%d,%.4f,%.2f,%s,%s,%s\n" % (k, calls, avg_ms, pct, "" if fe is None else "%.1f" % fe,
                                                   "" if wr is None else "%.1f" % wr, corr))
    print(open(path).read())


if __name__ == "__main__":
    main()
