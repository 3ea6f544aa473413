for P in 3 4 5 6 8; do
  python bench.py --steps 24 --warmup 2 --concurrency $P --no-cpu-baseline --no-stream-pass > gpurun_out/r2_sweep_P$P.json 2>> gpurun_out/r2_sweep.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r2_sweep_P$P.json').read().strip().splitlines()[-1])
print('P=$P', d['value'], 'proofs/s', d['ms_per_step'], 'ms/step')
PY
done
