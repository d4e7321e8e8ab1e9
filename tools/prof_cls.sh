# per-kernel durations of the [CLS]-tail kernels inside the step (rocprofv3 kernel trace, P = 1)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_cls
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cls -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --prof-kernels "" --pairs-sweep "" --no-train-regime --steps 40 --warmup 5 > /dev/null 2>&1
f=$(ls /tmp/prof_cls/*/*kernel_stats.csv | head -1)
grep -i "cls\|rows_\|ln_rows\|Name" "$f" | cut -c1-160
grep "gemm_nt_kernel<64, 64, 4u, 2" "$f" | cut -c1-200
