"""CPU tests of bench.py's stand-alone multi-GPU launcher (`python bench.py --gpus N` without torchrun): rank environment, relay
of rank 0's line, failure propagation, and the refusal to run a smaller job than the one asked for.  The sampling itself needs a
GPU (there is no CPU fallback): the hardware end-to-end run is tests/test_gpu_async_and_contexts.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_CHILD = r"""
import json, os, sys
rank = int(os.environ["RANK"])
if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
    sys.exit(7)
if "--hang-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--hang-rank") + 1]):
    import time
    time.sleep(600)
print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "EXTRA")} | {"argv": sys.argv[1:]}))
"""


def _child(tmp_path):
    p = tmp_path / "child.py"
    p.write_text(_CHILD)
    return [sys.executable, str(p)]


def test_spawn_ranks_sets_the_rank_environment_and_relays_rank0(tmp_path):
    import bench
    rc, out0 = bench.spawn_ranks(3, ["--steps", "4"], child_cmd=_child(tmp_path), env_extra={"EXTRA": "x"}, timeout=120)
    assert rc == 0
    got = json.loads(out0)
    assert got["RANK"] == "0" and got["LOCAL_RANK"] == "0" and got["WORLD_SIZE"] == "3" and got["MASTER_ADDR"] == "127.0.0.1"
    assert int(got["MASTER_PORT"]) > 0 and got["EXTRA"] == "x" and got["argv"] == ["--steps", "4"]


def test_spawn_ranks_propagates_a_failing_rank_and_ends_the_others(tmp_path):
    import bench
    rc, _ = bench.spawn_ranks(2, ["--fail-rank", "1", "--hang-rank", "0"], child_cmd=_child(tmp_path), timeout=120)
    assert rc == 7


def test_gpus_more_than_visible_is_refused_not_downgraded():
    """No GPU in this container: `--gpus 2` must exit non-zero with a message and print NO bench line (round 1 printed n_gpus: 1)."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has two GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 2
    assert "--gpus 2 requested" in r.stderr and not r.stdout.strip()


def test_ranks_without_a_device_fail_loudly():
    """The debug switch that lets ranks share a device does not conjure one up: without a HIP device every rank exits non-zero,
    the launcher reports the failure and prints no JSON (the product has no CPU fallback)."""
    import torch
    if torch.cuda.device_count() >= 1:
        import pytest
        pytest.skip("this box has a GPU")
    env = dict(os.environ, RRRMC_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--iters", "4096",
                        "--replicas", "64", "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "multi-rank run failed" in r.stderr


def test_gpus_flag_must_match_world_size_under_torchrun(tmp_path):
    """Launched by torch.distributed.run the rank count comes from WORLD_SIZE; a larger --gpus is an error on rank 0, never a relabel."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_the_printed_line_is_short_and_carries_all_five_configs():
    """a driver keeps the tail of stdout — the line must hold BASELINE's five configs in a few KB.  The full record of a
    real run (the committed round-5 line, 14 KB of numbers and prose) is reduced by bench.compact_line: numbers and short keys only."""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05", "bench_steps20_e.json")))
    line = json.dumps(bench.compact_line(full))
    assert len(line) <= bench.LINE_LIMIT_BYTES == 5000
    got = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in got, k
    assert got["config"]["workload"].startswith("GraphRRG(N=4096,K=3")
    assert set(got["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and set(got["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    for name in ("c3_sk_normal", "c4_ea_checkerboard", "c4_ea_random_site", "c5_quant_rrr"):          # configs 3, 4, 5 (config 2 is the headline, 1 the CPU leg)
        assert got["secondary"][name]["value"] > 0 and got["secondary"][name]["unit"]
    assert abs(got["value"] / full["value"] - 1) < 1e-5 and abs(got["roofline"]["frac"] / full["roofline"]["frac"] - 1) < 1e-4
