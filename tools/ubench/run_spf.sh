#!/bin/bash
# GPU box: the stand-alone harness of spf_team_kernel over a few shapes (every field / spin / undo record / count compared with spf_sweep_kernel)
# args of the harness: K N R iters beta launches NW step M TW
B=./tools/ubench/spf_team_bench.out
O=gpurun_out/spf_team_$1.txt
mkdir -p gpurun_out
: > $O
run() { echo "## $*" >> $O; timeout -k 10 120 $B "$@" 2>&1 | grep -v "^    \[" >> $O; }
run 3 4096 8192 32768 1.0 3 16 4096 0 64
run 3 4096 8192 32768 1.0 3 16 4096 0 32
run 3 4096 8192 32768 1.0 2 8 4096 0 64
run 3 32 256 20000 0.0 3 16 7 0 64
run 3 32 256 20000 0.0 3 16 7 0 16
run 3 256 1024 50000 0.3 3 16 100 0 32
run 4 1024 2048 30000 0.5 3 16 1 0 64
run 4 1024 2048 30000 0.5 3 16 1 0 16
run 6 4096 2048 30000 0.7 3 8 333 0 64
run 6 4096 2048 30000 0.7 3 16 333 0 32
run 3 4096 65536 16384 1.0 2 8 4096 0 64
grep -c identical $O; grep -n "FAILED\|ABORT\|no such" $O
grep -E "^##|launch 0|identical|FAILED" $O
