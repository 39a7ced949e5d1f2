#!/bin/bash
O=gpurun_out/spf_team_$1.txt
mkdir -p gpurun_out
: > $O
for n in nopair behind30 batch1; do
  echo "## $n M=38" >> $O; timeout -k 10 120 ./tools/ubench/spf_team_bench_$n.out 3 4096 8192 32768 1.0 1 16 4096 38 2>&1 | grep -v "^    \[" >> $O
done
cat $O
