#!/bin/bash
# in-step sweep of SPLICE_ATTN_MERGE_MAX (dQ and dK/dV halves of the attention backward in one launch up to that many workgroups)
run() { env SPLICE_ATTN_MERGE_MAX=$1 python bench.py --pairs $2 --steps 80 --warmup 15 --no-cpu-baseline --pairs-sweep "" --no-train-regime --prof-kernels "" --allow-dev-env 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for r in 1 2 3; do for P in 1 2 4 8; do for M in 0 768 1600 4000; do echo "P$P merge_max=$M $(run $M $P)"; done; done; done | sort | awk '{k=$1" "$2; s[k]+=$3; n[k]++} END {for (k in s) printf "%s -> %.4f\n", k, s[k]/n[k]}' | sort
