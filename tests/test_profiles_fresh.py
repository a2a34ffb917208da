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
