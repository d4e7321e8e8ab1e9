#!/bin/bash
# Timing-only ablations of the generator's implicit-GEMM convolution (results are garbage): builds gen_kernels.o with -DCONV_ABL=<m> for each m
# (0 = as shipped, 1 = no MFMA, 2 = no gather after the first channel tile, 3 = one k step per tile), links it against the other objects of the
# current build and prints the kernel-trace averages of the conv kernels for bench.py's --image 900x1200 leg with each library.
#   here (no GPU):   bash tools/conv_ablate.sh build "0 1 2 3"
#   on the GPU box:  bash tools/conv_ablate.sh run "0 1 2 3" ["--image 900x1200 --steps 10 --warmup 3"]
mode=$1; masks=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=16 -I$ROOT/include -Wno-unused-result"
if [ "$mode" = build ]; then
  mkdir -p $ROOT/build/abl
  others=$(ls $ROOT/build/csrc/*.o | grep -v gen_kernels.o)
  for m in $masks; do
    ( /opt/rocm/bin/hipcc $FLAGS -DCONV_ABL=$m -c $ROOT/splice_amd/csrc/gen_kernels.hip -o $ROOT/build/abl/gen_kernels_$m.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/abl/libconv_$m.so $others $ROOT/build/abl/gen_kernels_$m.o ) &
  done
  wait; ls -la $ROOT/build/abl/libconv_*.so
else
  ARGS=${3:---image 900x1200 --steps 10 --warmup 3}
  cp $ROOT/splice_amd/libsplice_hip.so /tmp/keep_convabl.so
  cd /tmp && export TMPDIR=/tmp
  for m in $masks; do
    cp $ROOT/build/abl/libconv_$m.so $ROOT/splice_amd/libsplice_hip.so
    rm -rf /tmp/prof_cabl
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cabl -- python $ROOT/bench.py --no-cpu-baseline --prof-kernels "" --pairs-sweep "" --no-train-regime $ARGS > /tmp/cabl.json 2> /tmp/cabl.err
    echo "== CONV_ABL=$m  ms_per_step $(python -c "import json;print(json.loads(open('/tmp/cabl.json').read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null)"
    python - <<PY
import csv, glob
f = glob.glob("/tmp/prof_cabl/*/*kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "conv_" in r["Name"]]
steps = None
tot = 0.0
for r in rows:
    tot += float(r["TotalDurationNs"])
    if "wgrad_batched_kernel<false>" in r["Name"]: steps = int(r["Calls"]) / 2
for r in rows[:12]:
    print("   %-62s calls/step %5.1f avg %8.1f us" % (r["Name"][:62], int(r["Calls"]) / (steps or 1), float(r["AverageNs"]) / 1e3))
print("   all conv kernels: %.2f ms per step" % (tot / 1e6 / (steps or 1)))
PY
  done
  cp /tmp/keep_convabl.so $ROOT/splice_amd/libsplice_hip.so
fi
