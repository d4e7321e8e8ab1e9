#!/bin/bash
# per shape: one rocprofv3 run, keep the top kernel name + average duration
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for tag in qkv_2 proj_2 fc1_2 fc2_2 qkv_16 proj_16 fc1_16 fc2_16; do
  rm -rf /tmp/yn_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/yn_$tag -- python $R/tools/yardstick_names.py $tag > /dev/null 2>&1
  f=$(ls /tmp/yn_$tag/*/*kernel_stats.csv | head -1)
  echo "== $tag"; head -4 "$f" | cut -c1-400
done
