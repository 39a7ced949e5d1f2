"""Build the gfx950 C-ABI library in-tree (rrrmc.jl_amd/lib/librrrmc_hip.so) with hipcc.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "rrrmc_hip.hip")
DEPS = [SRC, os.path.join(HERE, "..", "include", "rrrmc_hip.h")] + \
       [os.path.join(HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(HERE, "csrc"))) if f.endswith(".hpp")]
OUT = os.path.join(HERE, "lib", "librrrmc_hip.so")
# -ffp-contract=off: the Float64 kernels promise the reference's sequence of IEEE operations (no a*b+c fusing); the oracle is
# built the same way, so Float64 trajectories agree bit for bit by construction (IEEE division still expands to FMAs: exact).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-ffp-contract=off", "-Wall", "-Wextra"]


def lib_path():
    # RRRMC_HIP_LIB: point the binding at another build of the same ABI (timing experiments, tools/ablate.sh)
    return os.environ.get("RRRMC_HIP_LIB", OUT)


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build librrrmc_hip.so")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    tmp = "%s.tmp%d" % (OUT, os.getpid())          # several processes may build at once: each writes its own file, rename is atomic
    cmd = [hipcc] + FLAGS + [SRC, "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, OUT)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
