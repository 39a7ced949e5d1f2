"""Short runs of the randomised soak / fuzz tools (tools/*_soak.py, tests/soak/resume_fuzz.py) as part of the GPU suite: every tool prints one JSON
line per case and exits with 1 on the first kind of mismatch it is built to find.  The long runs' logs are under profiles/r06/.  One child
process at a time (the tools initialise the GPU themselves)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,cases,seed", [
    ("hook_soak.py", 60, 101),          # a run cut into resumed calls at random points == the run made in one call (all five samplers)
    ("resume_fuzz.py", 80, 102),        # read-only calls change nothing between resumed calls; run-ending ones make the next call a fresh run
    ("stop_soak.py", 60, 103),          # hooks that stop replicas at random samples, against the oracle
    ("gather_soak.py", 80, 104),        # the staged-gather kernels (rrrMC / bklMC / extremal_opt / rrrMC(DoubleGraph)) against the oracle
    ("queue_soak.py", 30, 105),         # random sequences of queued asynchronous standardMC calls against synchronised ones and the oracle
    ("qeat_soak.py", 40, 106),          # GraphQEAT, all five samplers, against the oracle
    ("misc_soak.py", 80, 109),          # colour-parallel sweeps, the fast Float64 mode, snapshot overlaps against the oracle
    ("std_family_soak.py", 120, 108),   # standardMC across the model kinds (levels, Float64 sparse, DoubleGraphs, SK, GraphQuant) against the oracle
    ("cont_family_soak.py", 60, 107),   # the continuous-energy samplers on GraphRRGNormal / GraphSKNormal / GraphQuant against the oracle
])
def test_soak_tool_short_run(tool, cases, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak", tool), str(cases), str(seed)], capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stderr[-2000:]
    last = json.loads(lines[-1])
    assert r.returncode == 0 and last["cases"] == cases and last["mismatches"] == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert sum('"same": true' in ln for ln in lines) >= cases // 2          # (a tool may skip combinations the library refuses)
