#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: everything the round's final numbers come from, condensed on the
# box (the per-dispatch databases are too large to travel back) into gpurun_out/final/, to be copied into profiles/<tag>_*:
#   1. tools/profile_gpu.sh (rocprofv3 stats + FETCH / WRITE / SQ_INSTS passes of one and the same bench command) ->
#      tools/summarize_prof.py <tag>: kernel stats, PMC summary, steady-state proof, instruction counts
#   2. the single-proof timeline on lanes (tools/timeline_single_proof.py)
#   3. tools/profile_k22.sh -> tools/summarize_k22.py (BASELINE config 5)
#   4. the default bench line (python bench.py --steps 20 --warmup 5), read AFTER the summaries exist so that its roofline
#      carries this build's PMC counters
# Usage: bash tools/final_profiles.sh <tag>      e.g. r04z
set -o pipefail
TAG=${1:-r04z}
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/final
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 700 bash tools/profile_gpu.sh > "$OUT/profile_gpu.log" 2>&1 || { echo "profile_gpu failed"; tail -5 "$OUT/profile_gpu.log"; exit 1; }
python tools/summarize_prof.py gpurun_out/prof "$TAG" > "$OUT/summarize.log" 2>&1 || { echo "summarize failed"; tail -5 "$OUT/summarize.log"; exit 1; }
cp profiles/${TAG}_* "$OUT"/
rm -rf gpurun_out/prof
echo "1/4 rocprofv3 passes summarised" >> "$OUT/progress.log"
CMD="python3 $ROOT/bench.py --steps 1 --warmup 1 --concurrency 1 --regions 1 --no-cpu-baseline --no-stream-pass --no-serial-latency --no-k22"
(cd /tmp && rocprofv3 --kernel-trace -d "$OUT/tl" -o run -- $CMD > "$OUT/tl.json" 2> "$OUT/tl.err") || { echo "timeline trace failed"; tail -3 "$OUT/tl.err"; exit 1; }
python tools/timeline_single_proof.py "$OUT/tl" -v > "$OUT/${TAG}_timeline_single_proof.txt" 2>> "$OUT/tl.err" || { echo "timeline failed"; exit 1; }
rm -rf "$OUT/tl"
echo "2/4 timeline" >> "$OUT/progress.log"
timeout -k 10 500 bash tools/profile_k22.sh > "$OUT/profile_k22.log" 2>&1 || { echo "profile_k22 failed"; tail -5 "$OUT/profile_k22.log"; exit 1; }
python tools/summarize_k22.py gpurun_out/prof_k22 "$OUT/${TAG}_k22_kernel_summary.csv" > "$OUT/summ_k22.log" 2>&1
cp gpurun_out/prof_k22/stats.json "$OUT/${TAG}_k22_stress_profiled_run.json"
rm -rf gpurun_out/prof_k22
echo "3/4 k22" >> "$OUT/progress.log"
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > "$OUT/${TAG}_bench_default.json" 2> "$OUT/bench_default.err" || { echo "bench failed"; tail -5 "$OUT/bench_default.err"; exit 1; }
python tools/show_bench.py "$OUT/${TAG}_bench_default.json" 2>/dev/null | head -12
echo "4/4 bench" >> "$OUT/progress.log"
