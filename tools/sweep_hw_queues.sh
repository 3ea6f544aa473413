#!/bin/bash
# Run ON THE GPU BOX: proofs in flight x GPU_MAX_HW_QUEUES (the HIP runtime maps a process's streams onto that many
# hardware queues, 4 by default — the kernel timeline never shows more than 4 kernels resident).
cd "$(dirname "$0")/.."
for Q in ${QS:-4 8}; do
  for P in ${PS:-4 6 8}; do
    GPU_MAX_HW_QUEUES=$Q timeout -k 10 200 python bench.py --steps 48 --warmup 4 --concurrency $P --no-cpu-baseline --no-stream-pass 2>/dev/null | tail -1 > gpurun_out/hwq_${Q}_$P.json || exit 1
    python - <<PY
import json
d=json.load(open('gpurun_out/hwq_${Q}_$P.json'))
print('GPU_MAX_HW_QUEUES=$Q in_flight=$P  %.2f proofs/s  %.3f ms/step' % (d['value'], d['ms_per_step']), flush=True)
PY
  done
done
