cd /tmp && export TMPDIR=/tmp
for sh in 4x785 2x785; do
  ATTN_SHAPES=$sh rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/attn_$sh -- python $GRAFT_REPO_ROOT/tools/attn_bench.py $1 > /dev/null 2>&1
  echo "== $sh"; f=$(ls $GRAFT_REPO_ROOT/gpurun_out/attn_$sh/*/*kernel_stats.csv | head -1); grep attn $f | cut -d, -f1-4,6-8
done
