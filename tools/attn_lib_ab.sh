python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -3
cp splice_amd/libsplice_hip.so /tmp/keep.so
for r in 1 2; do for L in Head TR; do cp build/lib$L.so splice_amd/libsplice_hip.so; echo "== $L"; ATTN_SHAPES=2x785,8x785,16x785,2x3137,2x8193 python tools/attn_bench.py 0 2>&1 | sed 's/ err fwd.*dq/ dq/'; done; done
cp /tmp/keep.so splice_amd/libsplice_hip.so
