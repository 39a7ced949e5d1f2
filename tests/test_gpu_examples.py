"""The two experiment drivers under examples/ (the reference's scripts/scripts.jl test_RRG / test_RRGCont / test_QIsing) run end to
end at toy sizes: log files in the script's format, BitMatrix dumps, sensible physics."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_example_qising(tmp_path):
    out = str(tmp_path / "q")
    s = _load("test_qising").main(["--N", "64", "--M", "8", "--samples", "6", "--step", "400", "--replicas", "5", "--out", out])
    assert set(s) == {"met", "rrr"}
    # RRR accepts most proposals where Metropolis rejects most (the point of the experiment, scripts.jl:778)
    assert s["rrr"][3] > 5 * s["met"][3] and s["met"][1] < 0 and s["rrr"][1] < 0
    files = sorted(os.listdir(out))
    assert sum(f.endswith(".txt") for f in files) == 10 and sum(f.endswith(".npy") for f in files) == 2
    lines = open(os.path.join(out, [f for f in files if f.startswith("output_rrr") and f.endswith("_r0.txt")][0])).read().splitlines()
    assert lines[0] == "#mctime acc QE clocktime" and len(lines) == 7 and lines[1].split()[0] == "400"
    Cs = np.load(os.path.join(out, [f for f in files if f.startswith("Cs_rrr")][0]))
    assert Cs.size == 6 * ((64 * 8 + 63) // 64)            # 6 sampled configurations of N = 512 spins, as BitVector chunks


@pytest.mark.parametrize("cont", [False, True])
def test_example_rrg(tmp_path, cont):
    out = str(tmp_path / "r")
    argv = ["--N", "200", "--samples", "6", "--step", "300", "--replicas", "4", "--out", out] + (["--cont"] if cont else [])
    curves = _load("test_rrg").main(argv)
    assert set(curves) == {"met", "bkl", "rrr", "wtm"}
    assert sum(f.endswith(".txt") for f in os.listdir(out)) == 16
