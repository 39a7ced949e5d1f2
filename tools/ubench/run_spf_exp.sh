#!/bin/bash
# GPU box: timing builds of spf_team_kernel side by side (tools/ubench/spf_exp_<name>.out built with -DSPF_TEAM_...); args: names
O=gpurun_out/spf_fused_exp.txt; : > $O
for b in spf_team_bench "$@"; do
  echo "#### $b" >> $O
  timeout -k 10 120 ./tools/ubench/$b.out 3 4096 8192 32768 1.0 2 16 4096 0 32 >> $O 2>&1
  timeout -k 10 120 ./tools/ubench/$b.out 3 4096 65536 16384 1.0 2 8 4096 0 64 >> $O 2>&1
done
grep -E "^####|launch|identical|FAILED" $O
