#!/bin/bash
# Run ON THE GPU BOX (gpurun): round-3 evidence — ubench (measured clock), SK kernel (config 3) trace + SQ counters, §8f kernels trace + SQ.
# Usage: bash tools/profile_r03.sh  -> gpurun_out/prof_r03x/*
set -u
OUT=$PWD/gpurun_out/prof_r03x
mkdir -p "$OUT"
export TMPDIR=/tmp
./tools/ubench/valu_rates.out > "$OUT/ubench_valu_rates.txt" 2>&1
[ -x tools/ubench/producer_task.out ] && ./tools/ubench/producer_task.out > "$OUT/ubench_producer_task.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/sk_trace" -- python3 tools/bench_models.py sk > "$OUT/sk_trace.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/sk_pmc" -- python3 tools/bench_models.py sk > "$OUT/sk_pmc.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d "$OUT/sk_pmc2" -- python3 tools/bench_models.py sk > "$OUT/sk_pmc2.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/f8_trace" -- python3 tools/bench_8f.py > "$OUT/f8_trace.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/f8_pmc" -- python3 tools/bench_8f.py > "$OUT/f8_pmc.log" 2>&1
for d in sk_trace f8_trace; do python3 tools/summarize_prof.py "$OUT/$d/.." > /dev/null 2>&1; done
ls "$OUT"
tail -3 "$OUT/sk_trace.log"; cat "$OUT/f8_trace.log" | tail -9; head -4 "$OUT/ubench_valu_rates.txt"
