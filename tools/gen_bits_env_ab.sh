#!/bin/bash
# usage (GPU box): bash tools/gen_bits_env_ab.sh VAR=a VAR=b ["N h w" ...]  -- digests of the generator's output, statistics and every gradient tensor
# (tools/gen_bits.py) under two settings of an environment switch; prints IDENTICAL / DIFFERENT per size.
A=$1; B=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for sz in "${@:-1 224 224}"; do
  env $A python $ROOT/tools/gen_bits.py $sz > /tmp/bits_a.txt 2>/tmp/bits_a.err
  env $B python $ROOT/tools/gen_bits.py $sz > /tmp/bits_b.txt 2>/tmp/bits_b.err
  if [ -s /tmp/bits_a.txt ] && cmp -s /tmp/bits_a.txt /tmp/bits_b.txt; then echo "size [$sz] $A | $B: IDENTICAL ($(wc -l < /tmp/bits_a.txt) digests)"; else echo "size [$sz] $A | $B: DIFFERENT ($(diff /tmp/bits_a.txt /tmp/bits_b.txt | grep -c '^<') of $(wc -l < /tmp/bits_a.txt))"; diff /tmp/bits_a.txt /tmp/bits_b.txt | head -8; tail -2 /tmp/bits_b.err; fi
done
