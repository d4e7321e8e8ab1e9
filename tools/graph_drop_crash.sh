# usage (GPU box): bash tools/graph_drop_crash.sh <tag> <iterations> [ENV=VALUE ...]  -- loops tools/dbg_e2e.py (4 train_model runs of 2000 steps per
# process, random crop sizes) until a process dies; the native backtrace of the faulting thread lands in gpurun_out/<tag>_crash.txt
TAG=$1; N=$2; shift 2
for i in $(seq 1 $N); do
  env "$@" E2E_RUNS=4 E2E_STEPS=2000 timeout 150 python -u tools/dbg_e2e.py > gpurun_out/${TAG}_$i.txt 2>&1; rc=$?
  echo "$TAG run $i rc=$rc" >> gpurun_out/${TAG}_rc.txt
  if [ $rc -ne 0 ]; then cp gpurun_out/${TAG}_$i.txt gpurun_out/${TAG}_crash.txt; rm gpurun_out/${TAG}_$i.txt; break; fi
  rm gpurun_out/${TAG}_$i.txt
done
