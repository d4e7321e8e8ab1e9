# usage (GPU box): bash tools/attn_thresh_sweep.sh   -- in-step sweep of the attention launch policies at 2 / 4 / 8 pairs per GPU
run() { P=$1; shift; echo -n "P=$P $* -> "; env "$@" python bench.py --pairs $P --steps 60 --warmup 10 --no-cpu-baseline --prof-kernels "" --pairs-sweep "" --no-train-regime 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['config']['pair_steps_per_s'])" 2>/dev/null || echo FAIL; }
for r in 1 2; do for P in 2 4 8; do
run $P X=0
run $P SPLICE_ATTN_QB2_TASKS=2000
run $P SPLICE_ATTN_QB2_TASKS=100000
run $P SPLICE_ATTN_MERGE_MAX=1300
run $P SPLICE_ATTN_MERGE_MAX=100000
run $P SPLICE_ATTN_MERGE_MAX=500
done; done
