#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: the rocprofv3 passes behind profiles/<tag>_*:
#   stats  : --kernel-trace --stats                       per-kernel time
#   fetch  : --pmc FETCH_SIZE        } separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes
#   write  : --pmc WRITE_SIZE        } (TCC counters do not fit one pass); counters only, no extra trace domains
#   insts  : --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES
# of one and the same command: one warm-up proof, one timed proof, the per-kernel-timed proof and the 3 latency proofs (6) on a single context (so that every launch in
# the trace belongs to keygen or to a whole proof). Output under gpurun_out/prof/<pass>; condense it afterwards with
#   python tools/summarize_prof.py gpurun_out/prof <tag>
set -o pipefail
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
# AMDZK_SERIAL=1: every key keeps one proof on one stream, so no two kernels of the trace overlap and a kernel's duration
# is its own (the default keys spread a proof over three streams: tools/timeline_single_proof.py looks at those)
export AMDZK_SERIAL=1
CMD="python3 $ROOT/bench.py --steps 1 --warmup 1 --concurrency 1 --regions 1 --no-cpu-baseline --no-stream-pass --no-serial-latency --no-k22"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run -- $CMD > "$OUT/stats.json" 2> "$OUT/stats.err" && echo "stats ok" &&
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o run -- $CMD > "$OUT/fetch.json" 2> "$OUT/fetch.err" && echo "fetch ok" &&
rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o run -- $CMD > "$OUT/write.json" 2> "$OUT/write.err" && echo "write ok" &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES -d "$OUT/insts" -o run -- $CMD > "$OUT/insts.json" 2> "$OUT/insts.err" && echo "insts ok"
rc=$?
cd "$ROOT"
# keep only what the summarizer reads (the traces are large)
find "$OUT" -name "*.csv" | head -50
du -sh "$OUT"
exit $rc
