#!/bin/bash
# GPU box: A/B of harness builds on one box, alternating (box-to-box noise is +-3 %): run_spf_ab.sh <buildA> <buildB> ...
O=gpurun_out/spf_ab.txt; : > $O
for rep in 1 2 3 4; do for b in "$@"; do
  echo "## $b" >> $O; timeout -k 10 120 ./tools/ubench/$b.out 3 4096 8192 32768 1.0 2 16 4096 0 32 2>&1 | grep -E "launch 1|FAILED" >> $O
done; done
grep -E "^##|launch" $O | paste - - | awk '{print $2, $(NF-3)}' | sort | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}'
