#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 kernel trace + PMC passes (each its own run, program directly after `--`) of ONE
# tools/bench_models.py workload, condensed by tools/summarize_prof.py.
# Usage: bash tools/profile_model.sh <tag> <bench_models.py arguments...>     -> gpurun_out/prof_<tag>/summary.txt
#        PROFILE_SCRIPT=tools/bench_8f.py bash tools/profile_model.sh f8 4096   (another workload script)
set -u
TAG=$1; shift
SCRIPT=${PROFILE_SCRIPT:-tools/bench_models.py}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$SCRIPT" "$@" > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$SCRIPT" "$@" > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$SCRIPT" "$@" > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$SCRIPT" "$@" > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_sq2" -- python3 "$SCRIPT" "$@" > "$OUT/pmc_sq2.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_sq3" -- python3 "$SCRIPT" "$@" > "$OUT/pmc_sq3.log" 2>&1
# (the summary is the counters and the trace table only: no rocprofv3 log lines, no absolute paths of the box)
{ echo "# tools/profile_model.sh $TAG $* ($SCRIPT)  git $(cat .git_head 2>/dev/null)"; grep -E "^\{" "$OUT/trace.log" | tail -4; echo; python3 tools/summarize_prof.py "$OUT"; } > "$OUT/summary.txt" 2>&1
cp "$OUT"/trace/*/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
tail -2 "$OUT/trace.log"
