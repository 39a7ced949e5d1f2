"""Build the gfx950 C-ABI library in-tree (rrrmc.jl_amd/lib/librrrmc_hip.so) with hipcc.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the snapshot.
The library is a few translation units (csrc/*.hip) compiled side by side and linked into one shared object.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# translation units: the C ABI with most kernels, and the builds of spf_team_kernel (a long kernel in thirty-two builds)
SRCS = [os.path.join(CSRC, "rrrmc_hip.hip"), os.path.join(CSRC, "spf_team_tu.hip")]
HDRS = [os.path.join(HERE, "..", "include", "rrrmc_hip.h")] + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hpp")]
DEPS = SRCS + HDRS
OUT = os.path.join(HERE, "lib", "librrrmc_hip.so")
OBJ_DIR = os.path.join(HERE, "lib", "obj")
# -ffp-contract=off: the Float64 kernels promise the reference's sequence of IEEE operations (no a*b+c fusing); the oracle is
# built the same way, so Float64 trajectories agree bit for bit by construction (IEEE division still expands to FMAs: exact).
# No -mtgsplit: spf_team_kernel's protocol needs a workgroup's wavefronts on ONE compute unit (threadgroup-split mode off, the default).
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-Wall", "-Wextra"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden"]
FLAGS = CFLAGS + ["-shared"]          # one-step form, kept for tools that compile a single file against the headers


def lib_path():
    # RRRMC_HIP_LIB: point the binding at another build of the same ABI (timing experiments, tools/ablate.sh)
    return os.environ.get("RRRMC_HIP_LIB", OUT)


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")


def _deps(src):
    """src and the headers it includes (transitively; #include "..." only), so that touching one kernel header rebuilds only its users."""
    import re
    seen, todo = set(), [src]
    while todo:
        f = os.path.normpath(todo.pop())
        if f in seen or not os.path.exists(f):
            continue
        seen.add(f)
        with open(f) as fh:
            for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', fh.read(), re.M):
                todo.append(os.path.join(os.path.dirname(f), m.group(1)))
    return sorted(seen)


def _newer(path, than):
    return os.path.exists(path) and os.path.getmtime(path) > than


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(_newer(d, t) for d in DEPS)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build librrrmc_hip.so")
    os.makedirs(OBJ_DIR, exist_ok=True)
    pid = os.getpid()          # several processes may build at once: each writes its own files, the renames are atomic
    jobs = []
    for src in SRCS:
        obj = _obj(src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(d) for d in _deps(src)):
            continue
        tmp = "%s.tmp%d" % (obj, pid)
        cmd = [hipcc] + CFLAGS + ["-c", src, "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        jobs.append((subprocess.Popen(cmd), cmd, tmp, obj))
    failed = None
    for proc, cmd, tmp, obj in jobs:
        rc = proc.wait()
        if rc == 0:
            os.replace(tmp, obj)
        else:
            failed = failed or (rc, cmd)
            if os.path.exists(tmp):
                os.remove(tmp)
    if failed:
        raise subprocess.CalledProcessError(failed[0], failed[1])
    tmp = "%s.tmp%d" % (OUT, pid)
    cmd = [hipcc] + LDFLAGS + [_obj(s) for s in SRCS] + ["-o", tmp]
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
