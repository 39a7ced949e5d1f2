#!/bin/bash
# tools/variant.sh NAME "-DFLAG ..." : build a variant of the library into tools/ablate/NAME.so (same flags as rrrmc.jl_amd/build.py);
# select it on the GPU box with RRRMC_HIP_LIB=$PWD/tools/ablate/NAME.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ablate
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off $2 rrrmc.jl_amd/csrc/rrrmc_hip.hip -o tools/ablate/$1.so
