#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + stats, and the two HBM-traffic PMC passes, of the default bench.py workload.
# Usage: bash tools/profile_trace_only.sh <tag>     -> gpurun_out/prof_<tag>/*
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/bench_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > "$OUT/bench_pmc1.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > "$OUT/bench_pmc2.log" 2>&1
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
