run() { python bench.py "$@" --no-cpu-baseline --pairs-sweep "" --no-train-regime --prof-kernels "" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['config']['pair_steps_per_s'])"; }
for r in 1 2 3; do
 for m in on off; do
  echo "P1 attn8=$m $(run --fp8 --fp8-attention $m --steps 150 --warmup 20)"
  echo "P8 attn8=$m $(run --fp8 --fp8-attention $m --pairs 8 --steps 60 --warmup 10)"
  echo "SC attn8=$m $(run --fp8 --fp8-attention $m --scales 224,320,448 --steps 40 --warmup 8)"
 done
done | sort
