#!/bin/bash
# GPU box: dense-SK block kernel, whole-group (h8) against split (h4: two workgroups of four replicas per compute unit) builds over N
O=gpurun_out/skh_split.txt; : > $O
for n in 512 768 1024 1536 2048; do
  for b in h8 h4; do echo "## $b $n 2048 65536" >> $O; timeout -k 10 120 ./tools/ubench/skh_bench.out $b $n 2048 65536 2>&1 | tail -1 >> $O; done
done
cat $O
