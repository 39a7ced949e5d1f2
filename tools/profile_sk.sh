#!/bin/bash
# Run ON THE GPU BOX (gpurun): config 3's kernel (sk_block_kernel) — kernel trace + two SQ counter passes.
# Usage: bash tools/profile_sk.sh  -> gpurun_out/prof_sk/*   (summarised by tools/summarize_prof.py)
set -u
OUT=$PWD/gpurun_out/prof_sk
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/bench_models.py sk > "$OUT/sk_trace.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_a" -- python3 tools/bench_models.py sk > "$OUT/sk_pmc_a.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d "$OUT/pmc_b" -- python3 tools/bench_models.py sk > "$OUT/sk_pmc_b.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 tools/bench_models.py sk > "$OUT/sk_pmc_f.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 tools/bench_models.py sk > "$OUT/sk_pmc_w.log" 2>&1
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
tail -40 "$OUT/summary.txt"
