# usage (GPU box): bash tools/micro/run_l2_stream.sh   -- builds and runs the L2 / LDS-DMA streaming micro-benchmark
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -o /tmp/l2_stream tools/micro/l2_stream.hip && timeout 300 /tmp/l2_stream
