#!/bin/bash
# GPU box, round 5: the profiles bench.py and DESIGN.md quote, taken on HEAD.  gpurun_out/prof_<tag>/summary.txt each.
set -u
./tools/ubench/skh_bench_stamps.out h8 1024 2048 65536 > gpurun_out/skh_stamps.txt 2>&1
bash tools/profile_model.sh spf8192 spf 8192
bash tools/profile_model.sh spf262144 spf 262144
bash tools/profile_model.sh c3 sk 1024 2048
ls gpurun_out/prof_spf8192 gpurun_out/prof_c3 | head -30
cat gpurun_out/skh_stamps.txt
