#!/bin/bash
# on the GPU box: time every variant library under tools/ablate/ (and the in-tree build first)
echo "== in-tree"
timeout 300 python bench.py --steps ${STEPS:-30} --warmup 2 --no-cpu-baseline --no-secondary | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/launch', d['roofline']['avg_launch_ms'], 'value %.3e'%d['value'])"
for f in tools/ablate/*.so; do
  echo "== $f"
  RRRMC_HIP_LIB=$PWD/$f timeout 300 python bench.py --steps ${STEPS:-30} --warmup 2 --no-cpu-baseline --no-secondary | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/launch', d['roofline']['avg_launch_ms'], 'value %.3e'%d['value'])"
done
