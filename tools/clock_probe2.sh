#!/bin/bash
# Run ON THE GPU BOX: sample shader clock / power while bench.py's streamed then resident regions run (8 proofs in
# flight, 2-s regions), with wall-clock stamps that tools line up with AMDZK_BENCH_TRACE's region markers.
# sysfs at 20 Hz when readable, rocm-smi at ~1 Hz otherwise.
set -o pipefail
export TMPDIR=/tmp PYTHONUNBUFFERED=1
OUT=gpurun_out/clk
mkdir -p $OUT
ls -d /sys/class/drm/card*/device/hwmon/hwmon* > $OUT/hwmon_dirs.txt 2>&1
for d in $(cat $OUT/hwmon_dirs.txt 2>/dev/null); do ls $d; done > $OUT/hwmon_files.txt 2>&1
probe_sysfs() {
  while true; do
    t=$(date +%s.%N)
    for d in /sys/class/drm/card*/device/hwmon/hwmon*; do
      echo "$t $d f=$(cat $d/freq1_input 2>/dev/null) p=$(cat $d/power1_average 2>/dev/null)$(cat $d/power1_input 2>/dev/null) T=$(cat $d/temp2_input 2>/dev/null)"
    done
    sleep 0.05
  done
}
probe_smi() {
  while true; do
    echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|fclk|Power' | tr -s ' ' | tr '\n' '|')"
    sleep 0.3
  done
}
probe_sysfs > $OUT/sysfs.log 2>/dev/null &
P1=$!
probe_smi > $OUT/smi.log 2>/dev/null &
P2=$!
trap 'kill $P1 $P2 2>/dev/null' EXIT
AMDZK_BENCH_TRACE=$OUT/trace.json timeout -k 10 400 python bench.py --steps 160 --concurrency 8 --regions 10 --warmup 2 --no-cpu-baseline --no-k22 --no-serial-latency > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -3 $OUT/bench.err; exit 1; }
kill $P1 $P2 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/clk/bench.json").read().strip().splitlines()[-1]); c=d["config"]
print("value", d["value"], "resident", c["resident_proofs_per_s"]); print(c["value_samples"]); print(c["resident_samples"])
PY
wc -l $OUT/sysfs.log $OUT/smi.log; tail -2 $OUT/smi.log; tail -2 $OUT/sysfs.log
