"""Committed counter (PMC) records carry a hash of the kernel sources they describe; the benches flag a
record that no longer matches the tree instead of quoting it silently."""

import json
import shutil
from pathlib import Path

from differt_amd import _srchash

ROOT = Path(__file__).resolve().parents[1]


def test_hash_changes_with_any_source_of_the_group(tmp_path):
    for name in set(sum(_srchash.GROUPS.values(), ())):
        shutil.copy(_srchash.CSRC / name, tmp_path / name)
    for kind, names in _srchash.GROUPS.items():
        base = _srchash.source_hash(kind, tmp_path)
        assert base == _srchash.source_hash(kind)
        for name in names:
            f = tmp_path / name
            orig = f.read_bytes()
            f.write_bytes(orig + b"\n// edit\n")
            assert _srchash.source_hash(kind, tmp_path) != base, (kind, name)
            f.write_bytes(orig)
        assert _srchash.source_hash(kind, tmp_path) == base


def test_stale_flag():
    rec = {"source_hash": _srchash.source_hash("dense")}
    assert not _srchash.is_stale(rec, "dense")
    assert _srchash.is_stale({"source_hash": "0" * 16}, "dense")
    assert _srchash.is_stale({}, "dense")  # records from before the hash existed are stale by definition


def test_committed_records_carry_a_hash():
    recs = [ROOT / "profiles" / "pmc_traffic.json", *sorted((ROOT / "profiles").glob("r*/pmc_trace_filter.json"))]
    assert recs
    for p in recs:
        rec = json.loads(p.read_text())
        assert isinstance(rec.get("source_hash"), str) and len(rec["source_hash"]) == 16, p


def test_exhaustive_record_is_quoted_only_while_its_hash_matches_the_tree(monkeypatch):
    """the bench line's `same_as_exhaustive` quoted round 4's record of a kernel that was no longer the default
    mapping.  bench_paths.exhaustive_record returns the latest committed record of scratch/exhaustive_pairs.py only while the
    source hashes stamped into it describe the kernels in the tree."""
    import sys

    sys.path.insert(0, str(ROOT))
    import bench_paths

    recs = sorted((ROOT / "profiles").glob("r*/stress/exhaustive_pairs.json"))
    if not recs:
        return
    data = json.loads(recs[-1].read_text())
    stamp = data.get("source_hash") or {}
    fresh = all(stamp.get(k) == _srchash.source_hash(k) for k in ("beam", "trace_filter"))
    got = bench_paths.exhaustive_record("configs[3]")
    assert (got is not None) == fresh
    if fresh:
        assert got["all_equal"] is True and got["checked_pairs"] >= 16 and set(got["kappas"]) == {64.0, 1.0}
        assert bench_paths.exhaustive_record("bruxelles order 3")["checked_pairs"] >= 8
    monkeypatch.setattr(_srchash, "source_hash", lambda kind, root=None: "0" * 16)
    assert bench_paths.exhaustive_record("configs[3]") is None


def test_round6_stress_records_are_clean_and_stamped():
    """The stress records DESIGN.md section 9.6 quotes: zero lost / extra paths, vertex-bit differences and mapping row mismatches
    at every unit >= 1/4 (the run at 1/64 is the one that is SUPPOSED to lose paths: it shows where the margins stop being loose),
    short-segment artifacts counted apart, every record stamped with the hash of the kernels it was taken on."""
    recs = sorted((ROOT / "profiles" / "r06" / "stress").glob("beam_stress*.json"))
    assert len(recs) >= 5
    scenes = 0
    for p in recs:
        rec = json.loads(p.read_text())
        assert set(rec["source_hash"]) >= {"beam", "trace_filter"} and len(rec["source_hash"]["beam"]) == 16, p
        assert rec["extra"] == 0 and rec["vertex_mismatch"] == 0 and rec["mapping_row_mismatch"] == 0, p
        if rec["kappa"] >= 0.25:
            assert rec["missed"] == 0, p
            scenes += rec["cases"]
        else:
            assert rec["missed"] > 0, p
        assert rec["short_segment_paths_lost"] <= rec["short_segment_paths_seen"] < 1e-3 * rec["valid_paths"]
    assert scenes > 800_000
    ex = json.loads((ROOT / "profiles" / "r06" / "stress" / "exhaustive_pairs.json").read_text())
    assert ex["all_equal"] and {r["config"]: r["checked_pairs"] for r in ex["records"]} == {
        "configs[3]": 16, "configs[4]": 16, "bruxelles order 3": 8, "manhattan order 3": 8}
    assert ex["source_hash"]["beam"] == json.loads(recs[0].read_text())["source_hash"]["beam"]  # one tree for all of them
