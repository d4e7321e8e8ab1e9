# usage (GPU box): bash tools/gemm_thresh_sweep.sh <pairs>   -- in-step sweep of the GEMM tile thresholds at P pairs per GPU
P=${1:-8}
run() { echo -n "$* -> "; env "$@" python bench.py --pairs $P --steps 60 --warmup 10 --no-cpu-baseline --prof-kernels "" --pairs-sweep "" --no-train-regime 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['config']['pair_steps_per_s'])" 2>/dev/null || echo FAIL; }
for r in 1 2; do
run X=0
run SPLICE_GEMM_BIGM=2400
run SPLICE_GEMM_BIGM=9600
run SPLICE_GEMM_BIGM=100000
run SPLICE_GEMM_T1MIN=300
run SPLICE_GEMM_T1MIN=640
run SPLICE_GEMM_T1MIN=100000
run SPLICE_GEMM_T2MIN=200
run SPLICE_GEMM_T2MIN=100000
done
