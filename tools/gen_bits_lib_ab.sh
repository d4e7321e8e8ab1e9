#!/bin/bash
# usage (GPU box): bash tools/gen_bits_lib_ab.sh <libA.so> <libB.so> ["N h w" ...]  -- digests of the generator's output and every gradient tensor (tools/gen_bits.py)
# under two BUILDS of the library; prints IDENTICAL / DIFFERENT per size.  (Two builds that claim the same bits, e.g. -DCONV_INTERIOR=0 against the default.)
A=$1; B=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cp $ROOT/splice_amd/libsplice_hip.so /tmp/keep_bits_ab.so
for sz in "${@:-1 224 224}"; do
  cp $A $ROOT/splice_amd/libsplice_hip.so; python $ROOT/tools/gen_bits.py $sz > /tmp/bits_a.txt 2>/dev/null
  cp $B $ROOT/splice_amd/libsplice_hip.so; python $ROOT/tools/gen_bits.py $sz > /tmp/bits_b.txt 2>/dev/null
  if [ -s /tmp/bits_a.txt ] && cmp -s /tmp/bits_a.txt /tmp/bits_b.txt; then echo "size [$sz]: IDENTICAL ($(wc -l < /tmp/bits_a.txt) digests)"; else echo "size [$sz]: DIFFERENT"; diff /tmp/bits_a.txt /tmp/bits_b.txt | head -6; fi
done
cp /tmp/keep_bits_ab.so $ROOT/splice_amd/libsplice_hip.so
