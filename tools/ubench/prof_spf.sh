#!/bin/bash
# GPU box: SQ counters of spf_team_kernel in the stand-alone harness (one launch of 32768 iterations, 8192 replicas)
# usage: tools/ubench/prof_spf.sh <tag> [harness args]
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_spf_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="${@:-3 4096 8192 32768 1.0 1 16 4096 0 64}"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/sq1" -- ./tools/ubench/spf_team_bench.out $ARGS > "$OUT/sq1.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/sq2" -- ./tools/ubench/spf_team_bench.out $ARGS > "$OUT/sq2.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_I8 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS --output-format csv -d "$OUT/sq3" -- ./tools/ubench/spf_team_bench.out $ARGS > "$OUT/sq3.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("sq1", "sq2", "sq3"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
        for k, v in acc.items():
            if "team_kernel" in k or "sweep_kernel" in k:
                print(d, k); [print("    %-28s %.6g" % (c, x)) for c, x in sorted(v.items())]
PY
