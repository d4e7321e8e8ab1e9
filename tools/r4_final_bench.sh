timeout 500 python bench.py > gpurun_out/r04y_bench_default.json 2> gpurun_out/r04y_bench_default.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04y_bench_driver_form.json 2> gpurun_out/r04y_bench_driver_form.err
timeout 300 python bench.py --size 448 --steps 60 --warmup 10 --no-cpu-baseline --no-train-regime --pairs-sweep "" > gpurun_out/r04y_bench_448.json 2> gpurun_out/r04y_bench_448.err
timeout 300 python bench.py --pairs 8 --steps 60 --warmup 10 --no-cpu-baseline --no-train-regime --pairs-sweep "" --prof-kernels 4,1,2,9,5,3,6,7,8 > gpurun_out/r04y_bench_pairs8.json 2> gpurun_out/r04y_bench_pairs8.err
tail -c 400 gpurun_out/r04y_bench_default.json
