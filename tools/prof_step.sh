# usage: bash tools/prof_step.sh <tag> [bench args...]   -- rocprofv3 kernel-trace statistics of bench.py's step (no PMC).
# Writes gpurun_out/<tag>_kernel_stats.csv (copy the ones to keep into profiles/).
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --prof-kernels "" --pairs-sweep "" --no-train-regime "$@" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench.err
f=$(ls /tmp/prof_$TAG/*/*kernel_stats.csv | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv
head -30 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
