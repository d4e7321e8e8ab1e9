mkdir -p gpurun_out
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python bench.py --pairs 4 --no-cpu-baseline --pairs-sweep "" > gpurun_out/bench_pairs4.json 2>gpurun_out/bench_pairs4.err
python bench.py --pairs 8 --no-cpu-baseline --pairs-sweep "" > gpurun_out/bench_pairs8.json 2>gpurun_out/bench_pairs8.err
bash tools/prof_step.sh p1 --steps 60 --warmup 10 > /dev/null
bash tools/prof_step.sh p4 --pairs 4 --steps 35 --warmup 10 > /dev/null
bash tools/prof_step.sh p8 --pairs 8 --steps 25 --warmup 10 > /dev/null
bash tools/pmc_traffic.sh p1 'gemm_nt_kernel' --pairs 1 > /dev/null
bash tools/pmc_traffic.sh p8 'gemm_nt_kernel' --pairs 8 > /dev/null
tail -c 600 gpurun_out/bench_default.json; wc -l gpurun_out/traffic_*.txt
