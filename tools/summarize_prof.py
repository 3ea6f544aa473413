#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/profile_gpu.sh (gpurun_out/prof/{stats,fetch,write,insts}/run_results.db —
ROCm 7.2 writes rocpd SQLite databases) into tracked files:
  profiles/<tag>_rocprofv3_kernel_stats.csv   rocprofv3 --kernel-trace --stats: calls, total / average ns, share
  profiles/<tag>_kernel_summary.csv           the same plus FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU per launch (--pmc passes)
  profiles/<tag>_kernel_stats_steady.csv      the same trace cut down to ONE steady-state proof (the dispatches between the
                                              last two random-polynomial kernels, one per proof, of a run whose keys are
                                              serial — AMDZK_SERIAL=1 in tools/profile_gpu.sh — so that kernels do not
                                              overlap): launches and average duration per kernel. avg x launches of the
                                              dominant kernel is what bench.py's live HIP-event figure measures
  profiles/<tag>_valu_instruction_counts.txt  instruction totals per kernel and per proof
The first line of the summary stamps the hash of the kernel sources the run was made with (bench.kernel_src_hash):
bench.py uses a summary's counters only when that hash is its own build's.
Usage: python tools/summarize_prof.py gpurun_out/prof r02d [proofs_in_run]"""
import collections
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STARTUP = ("table_next_kernel", "table_to_r261_kernel", "fixed_base_mul_kernel", "powers_kernel", "twiddle_gen_kernel",
           "sub_from_const_kernel", "mul2_kernel", "coset_table_kernel")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:60]


def db(src, kind):
    p = os.path.join(src, kind, "run_results.db")
    return sqlite3.connect(p) if os.path.exists(p) else None


def mad_share_of_hot_loop():
    """Share of v_mad_u64_u32 among the VALU instructions of msm_accum_seg_kernel<true>'s mixed-addition block (the
    basic block with the most multiply-adds), from hipcc's gfx950 assembly of this tree's msm.hip. None without hipcc."""
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        return None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "anon-aadhaar-halo2_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "msm.s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-DAMDZK_ASM_PRODUCT", "-I" + os.path.join(root, "include"), "-I" + csrc,
                            "--cuda-device-only", "-S", os.path.join(csrc, "msm.hip"), "-o", out], capture_output=True, timeout=900)
        if r.returncode != 0:
            return None
        text = open(out).read()
    m = re.search(r"^(_Z\w*msm_accum_seg_kernelILb1E\w*):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M)
    if not m:
        return None
    best = (0, 0)
    mad = valu = 0
    for ln in m.group(2).splitlines() + [".LBB_end:"]:
        t = ln.split(";")[0].strip()
        if re.match(r"^\.LBB\w+:", t):
            if mad > best[0]:
                best = (mad, valu)
            mad = valu = 0
        elif t.startswith("v_"):
            valu += 1
            mad += t.startswith("v_mad_u64_u32")
    return (best[0] / best[1], best[0], best[1]) if best[1] else None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    proofs = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0  # warm-up + timed + the per-kernel-timed proof + the 3 latency proofs of bench.py
    os.makedirs("profiles", exist_ok=True)
    stats = collections.OrderedDict()
    c = db(src, "stats")
    if c:
        rows = c.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc").fetchall()
        total = sum(r[2] for r in rows) or 1
        with open(os.path.join("profiles", tag + "_rocprofv3_kernel_stats.csv"), "w") as o:
            o.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage"\n')
            for name, calls, tot, avg in rows:
                o.write('"%s",%d,%d,%.1f,%.2f\n' % (name, calls, tot, avg, 100.0 * tot / total))
                k = short(name)
                if k.startswith("at::") or "rocclr" in k or "elementwise" in k:
                    continue
                stats[k] = (calls, avg / 1e6, 100.0 * tot / total)
        # one steady-state proof: from the start of the last-but-one random-polynomial kernel (the first kernel of a
        # proof) to the start of the last one
        marks = [r[0] for r in c.execute("select start from kernels where name like '%chacha20_fr_random%' order by start")]
        if len(marks) >= 2:
            import bench as _b
            a, b = marks[-2], marks[-1]
            srows = c.execute("select name, count(*), sum(duration), avg(duration) from kernels where start >= ? and start < ? "
                              "group by name order by sum(duration) desc", (a, b)).fetchall()
            # the same proof in the instruction-count pass (same command, same dispatch sequence): VALU per launch there —
            # the whole-run averages of the summary also hold keygen's larger commitment batches
            steady_valu = {}
            ci = db(src, "insts")
            if ci:
                seq = ci.execute("select dispatch_id, name, sum(counter_value) from pmc_events where counter_name = 'SQ_INSTS_VALU' "
                                 "group by dispatch_id, name order by dispatch_id").fetchall()
                im = [d for d, name, _ in seq if "chacha20_fr_random" in name]
                if len(im) >= 2:
                    tot, cnt = collections.Counter(), collections.Counter()
                    for d, name, v in seq:
                        if im[-2] <= d < im[-1]:
                            tot[short(name)] += v
                            cnt[short(name)] += 1
                    steady_valu = {k: (tot[k] / cnt[k], cnt[k]) for k in tot}
            with open(os.path.join("profiles", tag + "_kernel_stats_steady.csv"), "w") as o:
                o.write("# kernel_src_sha256=%s\n" % _b.kernel_src_hash())
                o.write("# one steady-state proof of the rocprofv3 --kernel-trace pass (serial keys: no overlapping kernels), wall %.3f ms, "
                        "sum of kernel durations %.3f ms; SQ_INSTS_VALU per launch from the same proof of the --pmc pass\n" % ((b - a) / 1e6, sum(r[2] for r in srows) / 1e6))
                o.write("kernel,launches,total_ms,avg_ms,SQ_INSTS_VALU_per_launch,cycles_per_valu_inst_1024_simds_2p4GHz\n")
                for name, calls, tot_ns, avg in srows:
                    k = short(name)
                    sv = steady_valu.get(k)
                    ok = sv is not None and sv[1] == calls and sv[0] > 0
                    o.write('"%s",%d,%.4f,%.4f,%s,%s\n' % (k, calls, tot_ns / 1e6, avg / 1e6, "%.0f" % sv[0] if ok else "",
                                                        "%.2f" % (avg * 1e-9 * 1024 * 2.4e9 / sv[0]) if ok else ""))
            print(open(os.path.join("profiles", tag + "_kernel_stats_steady.csv")).read())
    pmc = {}
    per_counter = collections.defaultdict(lambda: collections.defaultdict(float))
    calls_insts = collections.Counter()
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("insts", "SQ_INSTS_VALU")):
        agg = collections.defaultdict(lambda: [0, 0.0])
        c = db(src, kind)
        if c:
            for name, cname, val in c.execute("select name, counter_name, sum(counter_value) from pmc_events group by dispatch_id, name, counter_name"):
                k = short(name)
                if kind == "insts":
                    per_counter[k][cname] += val
                    if cname == "SQ_WAVES":
                        calls_insts[k] += 1
                if cname == counter:
                    agg[k][0] += 1
                    agg[k][1] += float(val)
        pmc[kind] = agg
    # ONE proof in the steady state = the dispatches between the last two random-polynomial kernels (one per proof):
    # totals over the whole run also hold keygen's commitments and transforms, which are not per-proof work
    one_proof = None
    c = db(src, "insts")
    if c:
        seq = c.execute("select dispatch_id, name, sum(counter_value) from pmc_events where counter_name = 'SQ_INSTS_VALU' "
                        "group by dispatch_id, name order by dispatch_id").fetchall()
        marks = [d for d, name, _ in seq if "chacha20_fr_random" in name]
        if len(marks) >= 2:
            a, b = marks[-2], marks[-1]
            one_proof = collections.Counter()
            for d, name, v in seq:
                if a <= d < b:  # the random-polynomial kernel is the first kernel of a proof
                    one_proof[short(name)] += v
    import bench
    path = os.path.join("profiles", tag + "_kernel_summary.csv")
    with open(path, "w") as o:
        o.write("# kernel_src_sha256=%s\n" % bench.kernel_src_hash())
        share = mad_share_of_hot_loop()
        if share:
            o.write("# msm_accum_l1_mad_share=%.4f  (%d v_mad_u64_u32 of %d VALU instructions in the mixed-addition block, hipcc -S of this tree)\n" % share)
        o.write("kernel,calls,avg_ms,pct_of_gpu_time,FETCH_SIZE_KiB_per_launch_raw,WRITE_SIZE_KiB_per_launch_raw,"
                "hbm_MB_per_launch_corrected(2*FETCH+WRITE),SQ_INSTS_VALU_per_launch\n")
        for k, (calls, avg_ms, pct) in stats.items():
            fe, wr, va = (pmc[x].get(k) for x in ("fetch", "write", "insts"))
            fe = fe[1] / fe[0] if fe else None
            wr = wr[1] / wr[0] if wr else None
            va = va[1] / va[0] if va else None
            corr = "" if fe is None or wr is None else "%.2f" % ((2 * fe + wr) * 1024 / 1e6)
            o.write('"%s",%d,%.4f,%.2f,%s,%s,%s,%s\n' % (k, calls, avg_ms, pct, "" if fe is None else "%.1f" % fe, "" if wr is None else "%.1f" % wr,
                                                     corr, "" if va is None else "%.0f" % va))
    print(open(path).read())
    if per_counter:
        per_proof = sum(v.get("SQ_INSTS_VALU", 0) for k, v in per_counter.items() if k not in STARTUP)
        with open(os.path.join("profiles", tag + "_valu_instruction_counts.txt"), "w") as o:
            o.write("rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES -- python3 bench.py --steps 1 --warmup 1 "
                    "--concurrency 1 --regions 1 --no-cpu-baseline --no-stream-pass --no-serial-latency --no-k22 (AMDZK_SERIAL=1)\n(shape full, k = 15, kernel sources %s; keygen + %.2f proofs)\n\n" % (bench.kernel_src_hash(), proofs))
            o.write("%-46s %6s %12s %10s %10s %10s %10s\n" % ("kernel", "calls", "VALU", "SALU", "LDS", "VMEM_RD", "waves"))
            for k, v in sorted(per_counter.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:24]:
                o.write("%-46s %6d %12.3e %10.2e %10.2e %10.2e %10.2e%s\n" % (k[:46], calls_insts[k], v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0),
                                                                      v.get("SQ_INSTS_LDS", 0), v.get("SQ_INSTS_VMEM_RD", 0), v.get("SQ_WAVES", 0),
                                                                      "   (start-up)" if k in STARTUP else ""))
            o.write("\nVALU wave-instructions outside the start-up kernels: %.3e over keygen + %.2f proofs\n" % (per_proof, proofs))
            if one_proof:
                tot = sum(one_proof.values())
                o.write("\nONE proof (the dispatches between the last two random-polynomial kernels): %.3e VALU wave-instructions\n" % tot)
                for k, v in one_proof.most_common(12):
                    o.write("  %-44s %10.3e  %5.1f %%\n" % (k[:44], v, 100.0 * v / tot))
                for cpi in (4.0, 4.5, 5.0):
                    o.write("  at %.1f cycles per VALU wave-instruction on 1024 SIMDs at 2.4 GHz: %.1f ms per proof\n" % (cpi, tot * cpi / (1024 * 2.4e9) * 1e3))
        print(open(os.path.join("profiles", tag + "_valu_instruction_counts.txt")).read())


if __name__ == "__main__":
    main()
