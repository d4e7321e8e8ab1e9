# same-box A/B of the attention backward forms inside the step: SPLICE_ATTN_BWD_VARIANT=1 (16x16x32 kernels) against the default (32x32x16)
for rep in 1 2; do for v in 1 0; do for cfg in "" "--size 448 --steps 60 --warmup 10" "--pairs 8 --steps 60 --warmup 10"; do
  SPLICE_ATTN_BWD_VARIANT=$v timeout 300 python bench.py $cfg --no-cpu-baseline --no-train-regime --pairs-sweep "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bwd_variant', '$v', '$cfg', d['value'], d['ms_per_step'], d['config'].get('pair_steps_per_s'))"
done; done; done
