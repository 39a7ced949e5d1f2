#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + PMC passes of the default bench.py workload.
# Usage: bash tools/profile_gpu.sh <tag>     -> gpurun_out/prof_<tag>/*
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --steps ${PROFILE_STEPS:-3} --warmup ${PROFILE_WARMUP:-1} --no-cpu-baseline --no-secondary --no-verify"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/bench_trace.log" 2>&1
# PMC passes: separate runs, no trace domains (gpurun rule); TCC slots: FETCH_SIZE=3, WRITE_SIZE=2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > "$OUT/bench_pmc1.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > "$OUT/bench_pmc2.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_sq" -- python3 $ARGS > "$OUT/bench_pmc3.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_lds" -- python3 $ARGS > "$OUT/bench_pmc4.log" 2>&1
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
