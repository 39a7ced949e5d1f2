#!/bin/bash
# quick timing + identity: default shape at the team widths
B=./tools/ubench/spf_team_bench.out
O=gpurun_out/spf_team_$1.txt
mkdir -p gpurun_out
: > $O
run() { echo "## $*" >> $O; timeout -k 10 120 $B "$@" 2>&1 | grep -v "^    \[" >> $O; }
run 3 4096 8192 32768 1.0 2 16 4096 0 64
run 3 4096 8192 32768 1.0 2 16 4096 0 32
run 3 4096 8192 32768 1.0 2 16 4096 0 16
run 3 4096 8192 32768 1.0 2 16 4096 30 32
run 3 4096 4096 32768 1.0 2 16 4096 0 16
run 3 4096 16384 32768 1.0 2 16 4096 0 64
cat $O
