#!/bin/bash
# GPU box: dense-SK block kernel at N > 2048 — shipped build, always-hot rows (timing only), stamped phases
O=gpurun_out/skh_bign.txt; : > $O
for n in 1024 1536 2048 3072 4096; do
  for b in skh_bench skh_bench_hot; do echo "## $b h8 $n 2048 32768" >> $O; timeout -k 10 120 ./tools/ubench/$b.out h8 $n 2048 32768 >> $O 2>&1; done
done
echo "## stamps 4096" >> $O; timeout -k 10 120 ./tools/ubench/skh_bench_stamps.out h8 4096 2048 32768 >> $O 2>&1
cat $O
