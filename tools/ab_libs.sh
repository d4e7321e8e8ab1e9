#!/bin/bash
# usage (GPU box): bash tools/ab_libs.sh "<tag>=<lib.so>[,ENV=V...] ..." [rounds] [pairs list]   -- alternating same-box comparison of
# library builds / environment settings over the benchmark step (boxes differ by +-0.7 %: only alternating runs on ONE box resolve < 1.5 %)
R=${2:-3}; PAIRS=${3:-"1"}
cp splice_amd/libsplice_hip.so /tmp/keep.so
for r in $(seq $R); do for spec in $1; do
  tag=${spec%%=*}; rest=${spec#*=}; IFS=, read -ra parts <<< "$rest"
  cp "${parts[0]}" splice_amd/libsplice_hip.so
  for P in $PAIRS; do
    env "${parts[@]:1}" python bench.py --pairs $P --steps 100 --warmup 20 --no-cpu-baseline --pairs-sweep "" --no-train-regime --prof-kernels "" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag P=$P', j['ms_per_step'], j['config']['pair_steps_per_s'])"
  done
done; done | sort | awk '{k=$1" "$2; s[k]+=$3; n[k]++; print} END {print "-- means (ms per step)"; for (k in s) printf "%s -> %.4f (n=%d)\n", k, s[k]/n[k], n[k]}'
cp /tmp/keep.so splice_amd/libsplice_hip.so
