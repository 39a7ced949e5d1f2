#!/bin/bash
# on the GPU box: time every ablation variant
for f in tools/ablate/*.so; do
  echo "== $f"
  RRRMC_HIP_LIB=$PWD/$f timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/launch', d['roofline']['avg_launch_ms'], 'value %.3e'%d['value'])"
done
