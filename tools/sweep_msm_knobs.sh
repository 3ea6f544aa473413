# MSM knob sweep on the bench workload (env overrides read by msm.hip): task size T1 and window bits c
for T1 in 16 32 64 128; do
  AMDZK_MSM_T1=$T1 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-stream-pass 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['per_kernel_ms']
print('T1=$T1', d['value'], 'proofs/s; l1', k.get('msm_accum_l1'), 'fold', k.get('msm_accum_fold'), 'final', k.get('msm_accum_final'))"
done
for C in 12 14; do
  AMDZK_MSM_C=$C python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-stream-pass 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['per_kernel_ms']
print('c=$C', d['value'], 'proofs/s; l1', k.get('msm_accum_l1'), 'fold', k.get('msm_accum_fold'), 'rowcol', k.get('msm_rowcol'), 'final', k.get('msm_accum_final'))"
done
