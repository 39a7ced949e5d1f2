#!/bin/bash
# GPU box: fused pairs reported at once (SPF_DEFER=0) or one pair late (SPF_DEFER=1) over replica counts and graph sizes
B=./tools/ubench/spf_team_bench.out
O=gpurun_out/spf_team_defer.txt; : > $O
run() { echo "## defer=$SPF_DEFER $*" >> $O; timeout -k 10 120 $B "$@" 2>&1 | grep -E "launch 1|identical|FAILED" >> $O; }
for d in 0 1; do export SPF_DEFER=$d
run 3 4096 256 32768 1.0 2 16 4096 0 16
run 3 4096 1024 32768 1.0 2 16 4096 0 16
run 3 4096 4096 32768 1.0 2 16 4096 0 16
run 3 4096 8192 32768 1.0 2 16 4096 0 32
run 3 10000 8192 32768 1.0 2 16 4096 0 32
run 3 256 1024 50000 0.3 2 16 100 0 16
run 3 1024 1024 50000 0.5 2 16 100 0 16
run 6 4096 2048 30000 0.7 2 16 333 0 32
done
cat $O
