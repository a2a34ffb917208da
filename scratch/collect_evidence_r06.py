"""gpurun_out/evidence_r06 -> profiles/r06 (run here after scratch/evidence_r06.sh ran on the GPU box): the stress records get the
hash of the kernel sources they were taken on (the tree must not have changed in between: the hash is computed HERE)."""
import json
import shutil
import sys
from pathlib import Path

sys.path.insert(0, ".")
from differt_amd._srchash import source_hash  # noqa: E402

src, dst = Path("gpurun_out/evidence_r06"), Path("profiles/r06")
(dst / "stress").mkdir(parents=True, exist_ok=True)
stamp = {k: source_hash(k) for k in ("beam", "trace_filter", "dense")}
if (src / "r06_pytest_gpu.txt").exists():
    shutil.copy(src / "r06_pytest_gpu.txt", dst / "r06_pytest_gpu.txt")
summary = {}
for f in sorted(src.glob("*.json")):
    try:
        rec = json.loads([ln for ln in f.read_text().splitlines() if ln.startswith("{")][-1]) if f.name != "exhaustive_pairs.json" else json.loads(f.read_text())
    except (IndexError, ValueError):
        print("unreadable:", f)
        continue
    if f.name == "exhaustive_pairs.json":
        assert rec["source_hash"] == {k: stamp[k] for k in rec["source_hash"]}, "the tree changed since the record was taken"
    else:
        rec["source_hash"] = stamp
    (dst / "stress" / f.name).write_text(json.dumps(rec, indent=1) + "\n")
    summary[f.name] = {k: v for k, v in rec.items() if k in ("cases", "missed", "extra", "vertex_mismatch", "mapping_row_mismatch", "mapping_checks",
                                                           "valid_paths", "exhaustive_candidates", "soups", "short_segment_paths_seen",
                                                           "short_segment_paths_lost", "kappa", "seconds", "all_equal", "candidate_evals",
                                                           "mask_mismatch", "object_mismatch", "compact_mismatch", "tests", "hit_mismatch", "t_mismatch")}
print(json.dumps(summary, indent=1))
