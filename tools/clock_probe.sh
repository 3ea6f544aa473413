#!/bin/bash
# Run ON THE GPU BOX: samples the GPU's shader clock and power (rocm-smi) every 0.5 s while bench.py runs with the
# default 4 proofs in flight, then with 1 — is the chip at its nominal 2.4 GHz under this load, or power-limited?
cd "$(dirname "$0")/.."
probe() {
  while true; do
    rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (edge|junction|hotspot)" | tr -s ' ' | tr '\n' '|'
    echo
    sleep 0.5
  done
}
for conc in 4 1; do
  echo "== concurrency $conc"
  probe > gpurun_out/clock_probe_$conc.log &
  PROBE=$!
  timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-stream-pass --concurrency $conc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('proofs/s', d['value'], 'ms/step', d['ms_per_step'])"
  kill $PROBE
  wait $PROBE 2>/dev/null
  # the busiest samples = highest power
  sort -t'|' -k2 gpurun_out/clock_probe_$conc.log | tail -40 | awk 'NR%6==0'
done
