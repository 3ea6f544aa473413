#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/profile_k22.sh (BASELINE config 5: tools/stress_k22.py) into one small table,
ON THE GPU BOX (the per-dispatch databases are too large to travel back): per kernel the launches, the average and the
LAST launch's duration (the last multi-scalar multiplication of the run is the uniform-scalar one of the per-kernel
pass), SQ_INSTS_VALU / SQ_WAVES / SQ_INSTS_VMEM_RD and raw FETCH_SIZE / WRITE_SIZE per launch, and cycles per VALU
wave-instruction on 1024 SIMDs at 2.4 GHz.
Usage: python tools/summarize_k22.py gpurun_out/prof_k22 <out.csv>"""
import collections
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SKIP = ("at::", "rocclr", "elementwise", "distribution", "vectorized", "Cijk", "index_")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:60]


def db(src, kind):
    p = os.path.join(src, kind, "run_results.db")
    return sqlite3.connect(p) if os.path.exists(p) else None


def main():
    src, out = sys.argv[1], sys.argv[2]
    import bench
    stats = collections.OrderedDict()
    c = db(src, "stats")
    if c:
        for name, calls, tot, avg in c.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc"):
            k = short(name)
            if any(k.startswith(s) or s in k for s in SKIP):
                continue
            last = c.execute("select duration from kernels where name = ? order by start desc limit 1", (name,)).fetchone()[0]
            stats[k] = [calls, avg / 1e6, last / 1e6]
    ctr = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))  # kernel -> counter -> [launches, sum, last]
    for kind in ("insts", "fetch", "write"):
        c = db(src, kind)
        if not c:
            continue
        for d, name, cname, val in c.execute("select dispatch_id, name, counter_name, sum(counter_value) from pmc_events "
                                             "group by dispatch_id, name, counter_name order by dispatch_id"):
            e = ctr[short(name)][cname]
            e[0] += 1
            e[1] += float(val)
            e[2] = float(val)
    cols = ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_INSTS_VMEM_RD", "FETCH_SIZE", "WRITE_SIZE")
    with open(out, "w") as o:
        o.write("# kernel_src_sha256=%s\n" % bench.kernel_src_hash())
        o.write("# rocprofv3 passes of tools/profile_k22.sh over tools/stress_k22.py (2^22-point MSM: uniform x4, skewed x4, uniform x1; 2^22 transforms); "
                "*_last = the kernel's last launch of the run (for msm_*: the uniform-scalar MSM); FETCH/WRITE raw KiB\n")
        o.write("kernel,launches,avg_ms,last_ms," + ",".join("%s_last" % x for x in cols) + ",cycles_per_valu_inst_last\n")
        for k, (calls, avg, last) in stats.items():
            vals = [ctr[k][x][2] if ctr[k][x][0] else None for x in cols]
            cpi = last * 1e-3 * 1024 * 2.4e9 / vals[0] if vals[0] else None
            o.write('"%s",%d,%.4f,%.4f,%s,%s\n' % (k, calls, avg, last, ",".join("" if v is None else "%.0f" % v for v in vals),
                                                   "" if cpi is None else "%.2f" % cpi))
    print(open(out).read())


if __name__ == "__main__":
    main()
