#!/bin/bash
# in-step sweep of the remaining GEMM dispatch knobs at P pairs per GPU (alternating, 3 rounds): ms per step
P=${1:-1}
run() { echo -n "$1 -> "; env $1 python bench.py --pairs $P --steps 100 --warmup 15 --no-cpu-baseline --prof-kernels "" --pairs-sweep "" --no-train-regime 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for r in 1 2 3; do for v in X=0 SPLICE_GEMM_T2RING=0 SPLICE_GEMM_T2RING=4 SPLICE_GEMM_RINGWG=320 SPLICE_GEMM_RINGWG=1280 SPLICE_GEMM_SHORTNS=4 SPLICE_GEMM_T96=0 SPLICE_GEMM_T96=3 SPLICE_GEMM_T96=5 SPLICE_GEMM_RINGK=1536 SPLICE_GEMM_T2MIN=200 SPLICE_GEMM_T2MIN=800 SPLICE_GEMM_T1MIN=180; do run $v; done; done | sort | awk '{k=$1; s[k]+=$3; n[k]++} END {for (k in s) printf "%s -> %.4f\n", k, s[k]/n[k]}' | sort -t'>' -k2 -n
