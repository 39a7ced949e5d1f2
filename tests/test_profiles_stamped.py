"""Every committed profile JSON that bench.py reads carries the stamp of the sources it was measured on: bench.py drops the
figures of a file whose stamp no longer matches, and this test fails when a file has no stamp at all (or names sources that do not exist)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_every_profile_json_read_by_bench_is_stamped():
    src = open(os.path.join(ROOT, "bench.py")).read()
    names = sorted(set(re.findall(r'load_profile_json\("([\w.]+)"\)', src)))
    assert names and "spf_traffic.json" in names and "traffic.json" in names
    for name in names:
        j, path, state = bench.load_profile_json(name)
        assert j is not None, "bench.py reads %s, no profiles/r*/%s is committed" % (name, name)
        assert state in ("match", "stale"), "%s has no source_stamp" % path
        for f in j.get("source_files") or bench.SWEEP_KERNEL_SOURCES:
            assert os.path.exists(os.path.join(ROOT, f)), "%s names a source that does not exist: %s" % (path, f)
        assert j.get("git_commit"), "%s does not say which commit it was measured on" % path


def test_a_stale_or_unstamped_traffic_file_is_dropped(tmp_path, monkeypatch):
    """bench.py's spf_traffic(): figures only from a file whose stamp matches the kernel sources of this tree"""
    import json
    good, _, state = bench.load_profile_json("spf_traffic.json")
    d = tmp_path / "rXX"
    d.mkdir()
    monkeypatch.setattr(bench, "PROFILE_DIRS", [str(d)])
    for mutate, why in ((lambda j: j.pop("source_stamp"), "unstamped"), (lambda j: j.update(source_stamp="0" * 16), "stale")):
        j = dict(good)
        mutate(j)
        (d / "spf_traffic.json").write_text(json.dumps(j))
        got, note = bench.spf_traffic()
        assert got is None and why in note
    (d / "spf_traffic.json").write_text(json.dumps(dict(good, source_stamp=bench.source_stamp(good["source_files"]))))
    got, note = bench.spf_traffic()
    assert got is not None and note is None
