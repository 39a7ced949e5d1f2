O=gpurun_out/spf_fused_exp.txt; : > $O
for b in spf_team_bench spf_exp_NOMEM spf_exp_NOPROTO spf_exp_NOPROTO2 spf_exp_STAMPS; do echo "#### $b" >> $O; timeout -k 10 120 ./tools/ubench/$b.out 3 4096 8192 32768 1.0 2 16 4096 0 32 >> $O 2>&1; done
cat $O
