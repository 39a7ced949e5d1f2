#!/bin/bash
# GPU box: the 2.5e10-attempt soak of spf_team_kernel (every field, spin, undo record, move_last, energy, sample and count against spf_sweep_kernel)
# harness args: K N R iters beta launches NW step M TW
B=./tools/ubench/spf_team_bench.out
O=gpurun_out/spf_team_soak.txt
mkdir -p gpurun_out
: > $O
run() { echo "## $*" >> $O; timeout -k 10 300 $B "$@" 2>&1 | grep -v "^    \[" >> $O; }
run 3 4096 8192 262144 1.0 4 16 4096 0 32
run 3 4096 8192 262144 1.0 2 16 4096 0 64
run 3 4096 65536 65536 1.0 3 8 4096 0 64
run 3 256 8192 262144 0.3 3 16 1000 0 32
run 3 32 2048 262144 0.1 3 16 77 0 16
run 4 4096 4096 262144 0.7 3 16 4096 0 16
run 6 4096 4096 200000 1.0 3 8 4096 0 64
run 6 4096 8192 100000 1.0 2 16 4096 0 32
grep -c identical $O; grep -n "FAILED\|ABORT\|no such\|MISMATCH" $O | head
grep -E "^##|launch 0|acceptance" $O
