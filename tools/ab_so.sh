# usage (GPU box): bash tools/ab_so.sh <a.so> <b.so> [rounds]   -- alternating A/B of two builds of the library on the same box
A=$1; B=$2; R=${3:-2}
cp splice_amd/libsplice_hip.so /tmp/keep.so
for r in $(seq $R); do for so in $A $B; do cp $so splice_amd/libsplice_hip.so; for P in 1 8; do python bench.py --pairs $P --steps 80 --warmup 15 --no-cpu-baseline --pairs-sweep "" --no-train-regime --prof-kernels "" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$so P=$P', j['value'], j['ms_per_step'], j['config']['pair_steps_per_s'])"; done; done; done
cp /tmp/keep.so splice_amd/libsplice_hip.so
