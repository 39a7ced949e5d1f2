#!/bin/bash
# Build timing-only variants of the library with one role of the sweep kernel removed (results are WRONG in these
# builds; they only show which role bounds the chunk time).  Run here (hipcc cross-compiles), then bench on the GPU box:
#   RRRMC_HIP_LIB=tools/ablate/lib_no_tally.so python bench.py --no-cpu-baseline
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden"
for v in TALLY CONSUME PRODUCE; do
  lc=$(echo $v | tr A-Z a-z)
  hipcc $F -DRRRMC_ABLATE_$v rrrmc.jl_amd/csrc/rrrmc_hip.hip -o tools/ablate/lib_no_$lc.so &
done
hipcc $F -DRRRMC_ABLATE_TALLY -DRRRMC_ABLATE_CONSUME rrrmc.jl_amd/csrc/rrrmc_hip.hip -o tools/ablate/lib_only_produce.so &
hipcc $F -DRRRMC_ABLATE_TALLY -DRRRMC_ABLATE_PRODUCE rrrmc.jl_amd/csrc/rrrmc_hip.hip -o tools/ablate/lib_only_consume.so &
hipcc $F -DRRRMC_ABLATE_CONSUME -DRRRMC_ABLATE_PRODUCE rrrmc.jl_amd/csrc/rrrmc_hip.hip -o tools/ablate/lib_only_tally.so &
wait
ls -la tools/ablate
