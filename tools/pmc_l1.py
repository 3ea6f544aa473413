#!/usr/bin/env python3
"""Sum the SQ counters of one rocprofv3 --pmc pass per kernel and print those of the level-1 accumulation kernels
(run on the GPU box). Usage: python tools/pmc_l1.py <dir with run_results.db> <label>"""
import collections
import os
import sqlite3
import sys

src, label = sys.argv[1], sys.argv[2]
db = None
for root, _, files in os.walk(src):
    for f in files:
        if f.endswith(".db"):
            db = os.path.join(root, f)
c = sqlite3.connect(db)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for d, name, cname, val in c.execute("select dispatch_id, name, counter_name, sum(counter_value) from pmc_events group by dispatch_id, name, counter_name"):
    agg[name][cname] += val
    if cname == "SQ_WAVES":
        n[name] += 1
for name, v in agg.items():
    if "accum_l1" in name or "accum_seg_kernel<true>" in name:
        short = name.split("(")[0][-60:]
        wc, busy, valu = v.get("SQ_WAVE_CYCLES", 0), v.get("SQ_BUSY_CYCLES", 0), v.get("SQ_INSTS_VALU", 0)
        print("%s %s launches %d" % (label, short, n[name]))
        for k in sorted(v):
            print("   %-24s %.4e" % (k, v[k]))
        if wc and valu:
            print("   wave-cycles per VALU instruction %.3f; active %.1f %%, waiting to issue %.1f %%, s_waitcnt %.1f %%" % (
                wc / valu, 100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * v.get("SQ_WAIT_ANY", 0) / wc))
