#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: rocprofv3 passes over BASELINE config 5
# (tools/stress_k22.py: one 2^22-point MSM, uniform and skewed scalars, and one 2^22 transform both ways).
#   stats : --kernel-trace --stats      per-kernel time
#   insts : --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD
#   fetch : --pmc FETCH_SIZE            } separate passes (TCC counters do not fit one pass with the SQ ones),
#   write : --pmc WRITE_SIZE            } counters only, no extra trace domains
# The program itself follows `--` (no env / bash -c hop). Output under gpurun_out/prof_k22/<pass>; condense with
#   python tools/summarize_k22.py gpurun_out/prof_k22 <tag>
set -o pipefail
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_k22
rm -rf "$OUT"; mkdir -p "$OUT"
K=${K22_K:-22}
CMD="python3 $ROOT/tools/stress_k22.py --k $K --reps 3"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run -- $CMD > "$OUT/stats.json" 2> "$OUT/stats.err" && echo "stats ok" &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD -d "$OUT/insts" -o run -- $CMD > "$OUT/insts.json" 2> "$OUT/insts.err" && echo "insts ok" &&
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o run -- $CMD > "$OUT/fetch.json" 2> "$OUT/fetch.err" && echo "fetch ok" &&
rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o run -- $CMD > "$OUT/write.json" 2> "$OUT/write.err" && echo "write ok"
rc=$?
cd "$ROOT"
du -sh "$OUT"
exit $rc
