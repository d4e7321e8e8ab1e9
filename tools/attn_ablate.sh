#!/bin/bash
# Timing-only ablations of the attention kernels (results are garbage): builds attention.o with -DATTN_ABL=<mask> for each mask,
# links it against the other objects of the current build and times tools/attn_bench.py with each library.
#   here (no GPU):   bash tools/attn_ablate.sh build "0 1 18 4 12 31"
#   on the GPU box:  bash tools/attn_ablate.sh run "0 1 18 4 12 31" "16x785,2x3137" [variants]
mode=$1; masks=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=16 -I$ROOT/include -Wno-unused-result"
if [ "$mode" = build ]; then
  mkdir -p $ROOT/build/abl
  others=$(ls $ROOT/build/csrc/*.o | grep -v attention.o)
  for m in $masks; do
    ( /opt/rocm/bin/hipcc $FLAGS -D${3:-ATTN_ABL}=$m -c $ROOT/splice_amd/csrc/attention.hip -o $ROOT/build/abl/attention_$m.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/abl/lib_$m.so $others $ROOT/build/abl/attention_$m.o ) &
  done
  wait; ls -la $ROOT/build/abl/*.so
else
  cp $ROOT/splice_amd/libsplice_hip.so /tmp/keep_abl.so
  for m in $masks; do
    cp $ROOT/build/abl/lib_$m.so $ROOT/splice_amd/libsplice_hip.so
    echo "== ATTN_ABL=$m"
    ATTN_SHAPES=$3 ATTN_NOCHECK=1 python $ROOT/tools/attn_bench.py ${4:-0} 2>&1 | sed 's/ err .*//'
  done
  cp /tmp/keep_abl.so $ROOT/splice_amd/libsplice_hip.so
fi
# round 5, the ping-pong forward (attn_pp.h):  bash tools/attn_ablate.sh build "0 1 2" PP_ABL   -- the third argument names the macro (default ATTN_ABL)
