#!/bin/bash
# same-box alternating comparison of the batched convolution's output-channel fragments per workgroup (SPLICE_CONV_BATCH_FN / _MIN)
run() { env "$1" python bench.py --pairs $2 --steps 60 --warmup 10 --no-cpu-baseline --pairs-sweep "" --no-train-regime --prof-kernels "" --allow-dev-env 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['config']['pair_steps_per_s'])"; }
for r in 1 2 3; do for P in 4 8; do
  echo "P$P fn1 $(run SPLICE_CONV_BATCH_MIN=9999 $P)"
  echo "P$P fn2 $(run SPLICE_CONV_BATCH_FN=2 $P)"
  echo "P$P fn4 $(run SPLICE_CONV_BATCH_FN=4 $P)"
done; done | sort
