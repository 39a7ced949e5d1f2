#!/usr/bin/env python3
"""profiles/<round>/spf_traffic.json (round = $RRRMC_PROF_ROUND, default r06) from the rocprofv3 passes of `tools/profile_model.sh <tag> spf <R>` (gpurun_out/prof_<tag>): HBM-side bytes of
spf_team_kernel per attempt, stamped with the sources it was measured on (bench.py drops a stale or unstamped file).

  python tools/make_spf_traffic.py gpurun_out/prof_spf8192 [gpurun_out/prof_spf262144]

FETCH_SIZE / WRITE_SIZE are per-dispatch values in KiB (separate --pmc passes); FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies
128-byte requests at 64 bytes).  tools/bench_models.py spf R makes one call of 16384 and one of 65536 iterations: the bytes of BOTH
dispatches over the attempts of BOTH."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ITERS_BOTH = 16384 + 65536


def passes(d):
    """{pass: {counter: [per-dispatch values]}} of the spf_team_kernel dispatches, the kernel's name, its total trace time in ms"""
    out, name = {}, None
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_sq3"):
        # (gpurun merges every run's files into the same directory: the newest file of a pass is this run's)
        for f in sorted(glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
            for row in csv.DictReader(open(f)):
                if "spf_team_kernel" in row["Kernel_Name"]:
                    out.setdefault(sub, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                    name = row["Kernel_Name"].replace("void rrrmc::", "").split("(")[0]
    ms = 0.0
    for f in sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1:]:
        for row in csv.DictReader(open(f)):
            if "spf_team_kernel" in row["Kernel_Name"]:
                ms += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
    return out, name, ms


def entry(d, R):
    c, name, ms = passes(d)
    fetch = 2.0 * sum(c["pmc_fetch"]["FETCH_SIZE"]) * 1024.0
    write = sum(c["pmc_write"]["WRITE_SIZE"]) * 1024.0
    attempts = R * ITERS_BOTH
    e = {"kernel": name, "replicas": R, "fetch_bytes": fetch, "write_bytes": write, "attempts": attempts,
         "measured_bytes_per_attempt": (fetch + write) / attempts, "kernel_ms_both_dispatches": ms,
         "traffic_TBps": (fetch + write) / (ms * 1e-3) / 1e12 if ms else None}
    if e["traffic_TBps"]:
        e["frac_of_8TBps"] = e["traffic_TBps"] / 8.0
    sq3 = c.get("pmc_sq3", {})
    if sq3.get("SQ_WAVE_CYCLES") and sq3.get("SQ_ACTIVE_INST_ANY"):
        e["wave_issue_frac"] = sum(sq3["SQ_ACTIVE_INST_ANY"]) / sum(sq3["SQ_WAVE_CYCLES"])
    sq = c.get("pmc_sq", {})
    if sq.get("SQ_INSTS_VALU"):
        per = attempts / 64.0                       # attempts of 64-replica groups: instructions per attempt OF A GROUP (teams x their replicas / 64)
        e["instructions_per_group_attempt"] = {k[9:].lower(): sum(sq[k]) / per for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS") if k in sq}
    return e


def main():
    d8 = sys.argv[1]
    e8 = entry(d8, 8192)
    a = 0.1936                                       # acceptance of this workload (tools/bench_models.py spf), for the algorithmic bytes
    alg = 8 + a * (10 + 17 * 3)
    out = dict(e8)
    out.update({"workload": "GraphRRGNormal(N=4096,K=3) standardMC beta=1.0, 8192 replicas: one call of 16384 and one of 65536 iterations (tools/bench_models.py spf 8192)",
                "algorithmic_bytes_per_attempt": alg, "traffic_ratio": e8["measured_bytes_per_attempt"] / alg,
                "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) through tools/profile_model.sh; counters in KiB, FETCH_SIZE doubled per "
                          "MI355X_MICROARCH.md; bytes of BOTH dispatches over the attempts of BOTH; memory side of the L2s (Infinity Cache hits included)",
                "source_files": bench.SPF_TEAM_SOURCES, "source_stamp": bench.source_stamp(bench.SPF_TEAM_SOURCES), "git_commit": bench.git_head()})
    if len(sys.argv) > 2:
        out["at_262144_replicas"] = entry(sys.argv[2], 262144)
    dst = os.path.join(ROOT, "profiles", os.environ.get("RRRMC_PROF_ROUND", "r06"), "spf_traffic.json")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
