#!/bin/bash
# Run ON THE GPU BOX: A/B of two builds of libamdzk.so on the SAME box (box-to-box variance of the 4-in-flight
# throughput is several percent, more than most single changes). ab/libamdzk_old.so and ab/libamdzk_new.so are
# swapped into place alternately; same python, same bench flags.
set -o pipefail
cd "$(dirname "$0")/.."
LIB=anon-aadhaar-halo2_amd/libamdzk.so
cp $LIB /tmp/libamdzk_keep.so
# whatever happens (a failed or timed-out bench included), the tree gets its own build back
trap 'cp /tmp/libamdzk_keep.so "$LIB"' EXIT
for round in 1 2 3; do
  for v in old new; do
    cp ab/libamdzk_$v.so $LIB
    timeout -k 10 200 python bench.py --steps 48 --warmup 4 --regions 3 --no-cpu-baseline --no-k22 --no-serial-latency ${AB_FLAGS:-} 2>/dev/null | tail -1 > gpurun_out/ab_${v}_$round.json || exit 1
    python - "$v" "$round" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_%s_%s.json" % (sys.argv[1], sys.argv[2])))
print(sys.argv[1], sys.argv[2], "proofs/s %.2f  ms/step %.3f  latency %.3f  gpu_busy %.3f" % (d["value"], d["ms_per_step"], d["config"]["single_proof_latency_ms"], d["roofline"]["gpu_busy_ms_per_step"]), flush=True)
PY
  done
done
