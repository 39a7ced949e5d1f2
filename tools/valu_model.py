#!/usr/bin/env python3
"""Cycle-weighted VALU utilisation model of sweep_kernel<3, 1> (the roofline that binds: the replica state never leaves LDS).

  python tools/valu_model.py         (in the build container: hipcc cross-compiles; reads the committed rocprofv3 / ubench outputs)

Inputs
  * the kernel's ISA (hipcc -S of a one-kernel translation unit) -> opcode histogram of the whole kernel and of ONE PRODUCER TASK
    (64 slots: 3 Philox4x32-10 blocks + the bit-sliced threshold refinement) — the producers execute ~85 % of the kernel's VALU
    wave-instructions, so their mix is taken as the dynamic mix;
  * profiles/r02/ubench_valu_rates.txt (tools/ubench/valu_rates.hip on the MI355X: ns per wave-instruction per SIMD with 8 waves/SIMD);
  * profiles/r02/r02e_summary.txt (or the file named on the command line; rocprofv3 --pmc passes of `bench.py --steps 3 --warmup 1`): SQ_INSTS_VALU per launch, GRBM_GUI_ACTIVE.
Output: profiles/r02/sweep31_isa_hist.txt and profiles/r02/valu_model.json (read by bench.py for roofline.valu)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles", os.environ.get("RRRMC_PROF_ROUND", "r03"))
PROF_OLD = os.path.join(ROOT, "profiles", "r02")
SIMDS = 256 * 4
SUMMARY = sys.argv[1] if len(sys.argv) > 1 else "r03b_summary.txt"      # the rocprofv3 summary of the build being modelled
sys.path.insert(0, ROOT)


def prof_file(name):
    """newest copy of a ubench / profile output: this round's, else round 2's"""
    p = os.path.join(PROF, name)
    return p if os.path.exists(p) else os.path.join(PROF_OLD, name)


def provenance():
    import bench
    return {"source_stamp": bench.source_stamp(), "git_commit": bench.git_head(), "summary": os.path.relpath(prof_file(SUMMARY), ROOT)}


def kernel_isa():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "sw.hip")
        open(src, "w").write('#include "%s/rrrmc.jl_amd/csrc/sparse_kernels.hpp"\n'
                             'template __global__ void rrrmc::sweep_kernel<3, 1>(rrrmc::SweepParams);\n' % ROOT)
        out = os.path.join(d, "sw.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-w", "-o", out, src])
        lines = open(out).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN5rrrmc12sweep_kernelILi3ELi1EEEvNS_11SweepParamsE:"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return [l.strip() for l in lines[start:end + 1]]


def is_inst(l):
    return bool(l) and not l.startswith((";", ".", "_")) and not l.endswith(":")


def ubench_ns():
    ns = {}
    for l in open(prof_file("ubench_valu_rates.txt")):
        m = re.match(r"(\S+)\s+[\d.]+ ms\s+-> ([\d.]+) ns", l)
        if m:
            ns[m.group(1)] = float(m.group(2))
    return ns


def ubench_clock_ghz():
    """the shader clock the ubench measured for itself (s_memtime over s_memrealtime; round 3), None for an older output"""
    g = [float(m.group(1)) for m in re.finditer(r"measured ([\d.]+) GHz", open(prof_file("ubench_valu_rates.txt")).read())]
    return sum(g) / len(g) if g else None


def cost_ns(op, ns):
    """ns per wave-instruction per SIMD of one opcode: measured where the ubench has it, else the class it issues like."""
    base = re.sub(r"_e(32|64)$", "", op)
    if base in ns:
        return ns[base]
    if base.startswith("v_mad_u64") or base.startswith("v_mad_i64"):
        return ns["v_mad_u64_u32"]
    if base.startswith(("v_mul_lo", "v_mul_hi")):
        return ns["v_mul_lo_u32"]
    three_src = ("v_or3", "v_and_or", "v_bfi", "v_alignbit", "v_lshl_or", "v_lshl_add", "v_add3", "v_xad", "v_add_lshl", "v_perm", "v_bfe", "v_mad_u32_u24", "v_cndmask_b32_e64", "v_lshl_add_u64")
    if op == "v_cndmask_b32_e64" or base.startswith(three_src):
        return ns["v_and_or_b32"]
    if base.startswith("v_bcnt"):
        return ns["v_bcnt_u32_b32"]
    return ns["v_xor_b32"]          # one- and two-operand integer ops, moves, compares


def pmc():
    txt = open(prof_file(SUMMARY)).read()
    blk = lambda tag: txt[txt.index("[%s] void rrrmc::sweep_kernel<3, 1>" % tag):]
    val = lambda tag, name: float(re.search(r"%s\s+avg=([\d.e+]+)" % name, blk(tag)).group(1))
    ns = float(re.search(r"sweep_kernel<3, 1>\(rrrmc::SweepParams\)\s+n=\d+ avg_ns=(\d+)", txt).group(1))
    return {"SQ_INSTS_VALU": val("pmc_sq", "SQ_INSTS_VALU"), "SQ_INSTS_SALU": val("pmc_sq", "SQ_INSTS_SALU"), "SQ_INSTS_LDS": val("pmc_sq", "SQ_INSTS_LDS"),
            "GRBM_GUI_ACTIVE": val("pmc_lds", "GRBM_GUI_ACTIVE"), "FETCH_SIZE_KB": val("pmc_fetch", "FETCH_SIZE"), "WRITE_SIZE_KB": val("pmc_write", "WRITE_SIZE"),
            "rocprof_avg_ns": ns}


def main():
    isa = kernel_isa()
    insts = [l.split()[0] for l in isa if is_inst(l)]
    whole = collections.Counter(insts)
    # one producer task = the longest stretch of code between two labels / branches that holds the Philox multiplies
    blocks, cur = [], []
    for l in isa:
        if not is_inst(l) or l.split()[0].startswith(("s_cbranch", "s_branch", "s_barrier")):
            if cur:
                blocks.append(cur)
            cur = []
        else:
            cur.append(l.split()[0])
    task = max(blocks, key=lambda b: sum(o.startswith("v_mad_u64") for o in b))
    tvalu = collections.Counter(o for o in task if o.startswith("v_"))
    ns = ubench_ns()
    n_valu = sum(tvalu.values())
    task_ns = sum(c * cost_ns(o, ns) for o, c in tvalu.items())
    P = pmc()
    clock_hz = P["GRBM_GUI_ACTIVE"] / 8.0 / (P["rocprof_avg_ns"] * 1e-9)          # GRBM_GUI_ACTIVE sums the 8 XCDs
    mean_ns = task_ns / n_valu
    model = {"kernel": "sweep_kernel<3, 1>", "simds": SIMDS, "valu_insts_per_launch": P["SQ_INSTS_VALU"], "clock_hz": clock_hz,
             "mean_issue_ns": mean_ns, "mean_issue_cycles": mean_ns * 1e-9 * clock_hz,
             "producer_task": {"valu_wave_insts": n_valu, "issue_ns_per_simd": task_ns, "mix": dict(tvalu.most_common())},
             "pmc": P,
             # the same count at the guide's 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md), whatever the opcode
             "valu_busy_frac_vs_guide_2cycle_under_rocprof": P["SQ_INSTS_VALU"] * 2.0 / (SIMDS * P["rocprof_avg_ns"] * 1e-9 * clock_hz),
             "ubench_measured_clock_ghz": ubench_clock_ghz(),
             "valu_busy_frac_under_rocprof": P["SQ_INSTS_VALU"] * mean_ns * 1e-9 / (SIMDS * P["rocprof_avg_ns"] * 1e-9),
             "note": "mean issue cost = the producer task's VALU mix (ISA histogram) x tools/ubench/valu_rates.hip (8 waves per SIMD); the producers execute "
                     "~85 % of the kernel's VALU wave-instructions; SQ_ACTIVE_INST_VALU is not used: on gfx950 it counts one quad-cycle per instruction"}
    # the same task measured directly (tools/ubench/producer_task.hip: the kernel's own Philox + refinement code) at 8 waves per SIMD and
    # at the 4 waves per SIMD sweep_kernel runs with (16 waves per workgroup, one workgroup per CU): the SIMD's real throughput on this
    # dependent multiply / bit-op mix is lower at the kernel's occupancy
    pt = prof_file("ubench_producer_task.txt")
    if os.path.exists(pt):
        meas = {}
        for l in open(pt):
            m = re.match(r"NT=2 blocks=3\s+(\d+) waves/SIMD:.*-> ([\d.]+) ns per task per SIMD", l)
            if m:
                meas[int(m.group(1))] = float(m.group(2))
        if 4 in meas and 8 in meas:
            model["producer_task_ubench_ns"] = {"waves_per_simd_4": meas[4], "waves_per_simd_8": meas[8], "waves_per_simd_1": meas.get(1)}
            model["occupancy_factor_4_waves"] = meas[4] / meas[8]
    model.update(provenance())
    os.makedirs(PROF, exist_ok=True)
    json.dump(model, open(os.path.join(PROF, "valu_model.json"), "w"), indent=1)
    traffic = {"hbm_bytes_per_launch": (2 * P["FETCH_SIZE_KB"] + P["WRITE_SIZE_KB"]) * 1024.0,
               "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `bench.py --no-cpu-baseline --no-secondary --no-verify` through tools/profile_gpu.sh (round 4: --steps 20 --warmup 5, the driver's counts; earlier rounds --steps 3 --warmup 1), " \
                         + os.path.relpath(prof_file(SUMMARY), ROOT) + "; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); counters are in KiB"}
    traffic.update(provenance())
    json.dump(traffic, open(os.path.join(PROF, "traffic.json"), "w"), indent=1)
    with open(os.path.join(PROF, "sweep31_isa_hist.txt"), "w") as f:
        f.write("# opcode histogram of sweep_kernel<3, 1> (hipcc -O3 --offload-arch=gfx950 -S), static counts\n")
        f.write("# whole kernel: %d instructions\n" % len(insts))
        for o, c in whole.most_common():
            f.write("%6d %s\n" % (c, o))
        f.write("\n# one producer task (64 slots x 32 replicas: 3 Philox4x32-10 blocks + threshold refinement): %d instructions, %d VALU, %.0f ns of SIMD issue\n"
                % (len(task), n_valu, task_ns))
        for o, c in collections.Counter(task).most_common():
            f.write("%6d %s   %s\n" % (c, o, ("%.2f ns" % cost_ns(o, ns)) if o.startswith("v_") else ""))
    print(json.dumps({k: model[k] for k in ("valu_insts_per_launch", "clock_hz", "mean_issue_ns", "mean_issue_cycles", "valu_busy_frac_under_rocprof")}, indent=1))
    print("traffic per launch: %.1f MB" % (traffic["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    sys.exit(main())
