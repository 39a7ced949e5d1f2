#!/bin/bash
# GPU box: dense-SK block kernel A/B on one box, alternating: run_skh.sh <buildA> <buildB> ... (tools/ubench/<build>.out h8 N 2048 65536)
O=gpurun_out/skh_ab.txt; : > $O
for n in ${SKH_NS:-512 1024 2048 3072 4096}; do for rep in 1 2 3 4 5; do for b in "$@"; do
  echo "## $b $n" >> $O; timeout -k 10 120 ./tools/ubench/$b.out h8 $n 2048 65536 2>&1 | tail -1 >> $O
done; done; done
grep -E "^##|attempts" $O | paste - - | sed 's/  */ /g' | awk '{print $2, $3, $8}' | sort | awk '{a[$1" "$2]=a[$1" "$2]" "$3} END{for(k in a) print k, a[k]}' | sort -k2n -k1
