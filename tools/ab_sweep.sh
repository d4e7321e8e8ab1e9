#!/bin/bash
# Alternating same-box sweep of environment settings over the benchmark step (run inside ONE gpurun call: boxes differ by
# +-0.7 %, effects of 0.3-0.5 % only show when the variants alternate A B C A B C A B C on the same box).
# usage: bash tools/ab_sweep.sh "VAR1 VAR2" "a1,a2 b1,b2 c1,c2" [reps] [steps]
#   e.g. bash tools/ab_sweep.sh "SPLICE_GEMM_T96 SPLICE_GEMM_SHORTNS" "6,3 0,3 3,4" 3
VARS=($1); SETS=($2); REPS=${3:-3}; STEPS=${4:-400}
for r in $(seq 1 $REPS); do
  for set in "${SETS[@]}"; do
    IFS=, read -ra VALS <<< "$set"
    envs=()
    for i in "${!VARS[@]}"; do envs+=("${VARS[$i]}=${VALS[$i]}"); done
    ms=$(env "${envs[@]}" python bench.py --steps $STEPS --warmup $((STEPS / 10)) --no-cpu-baseline --prof-kernels "" --pairs-sweep "" --no-train-regime 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null || echo FAIL)
    echo "${envs[*]} -> $ms"
  done
done | sort | awk '{k=$0; sub(/ -> .*/, "", k); v=$NF; if (v != "FAIL") {s[k]+=v; n[k]++}; print} END {print "-- means"; for (k in s) printf "%s -> %.4f ms (n=%d)\n", k, s[k]/n[k], n[k]}'
