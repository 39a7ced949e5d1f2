#!/bin/bash
# build tools/ubench/spf_team_bench.out and extract one instantiation's ISA: build_spf.sh [extra hipcc flags]
cd /root/repo/tools/ubench || exit 1
mkdir -p tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off "$@" spf_team_bench.hip -o spf_team_bench.out -save-temps=obj 2>&1 | grep -v "^$" | head -30
S=spf_team_bench-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^_ZN5rrrmc15spf_team_kernelILi3ELi16ELi60ELi32EEEvNS_13SpfTeamParamsE:/{p=1} p{print} /s_endpgm/{if(p) exit}' $S > tmp/k3_16.s
grep -A30 "^    .name:           _ZN5rrrmc15spf_team_kernelILi3ELi16ELi60ELi32EEEvNS_13SpfTeamParamsE" $S | grep -E "vgpr_count|sgpr_count|private_segment|group_segment|spill" 
rm -f spf_team_bench-h* spf_team_bench.hip-hip*
wc -l tmp/k3_16.s
