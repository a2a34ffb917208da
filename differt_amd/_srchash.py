"""Content hash of kernel sources, stored next to committed counter (PMC) records.

A counter record under ``profiles/`` describes the kernel it was collected on.  ``bench.py`` /
``bench_paths.py`` quote such records (HBM traffic per launch, executed VALU instructions per
candidate) beside numbers measured live, so a record must be recognisable as stale once the kernel
it describes has changed: every record carries ``source_hash`` = :func:`source_hash` of the files
listed in :data:`GROUPS`, and the bench emits ``"pmc_stale": true`` when it differs from the sources
in the tree (``tests/test_profiles_fresh.py``).
"""

from __future__ import annotations

import hashlib
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"

# record kind -> the sources whose change invalidates it
GROUPS = {
    "dense": ("ray_ops.hip", "geom.hpp", "common.hpp"),
    "trace_filter": ("trace.hip", "trace_stages.hpp", "trace_common.hpp", "image_chain.hpp", "geom.hpp", "common.hpp"),
    "trace_dense": ("trace_dense.hip", "stores.hpp", "trace_stages.hpp", "trace_common.hpp", "image_chain.hpp", "geom.hpp",
                    "common.hpp"),
    "image_method": ("image_method.hip", "stores.hpp", "image_chain.hpp", "geom.hpp", "common.hpp"),
    "beam": ("beam.hip", "beam_margins.hpp", "mesh.hpp", "geom.hpp", "common.hpp"),
}


def source_hash(kind: str, root: Path | None = None) -> str:
    """sha256 over the named group's files (name + content, in the listed order), first 16 hex digits."""
    h = hashlib.sha256()
    for name in GROUPS[kind]:
        h.update(name.encode())
        h.update(b"\0")
        h.update(((root or CSRC) / name).read_bytes())
    return h.hexdigest()[:16]


def is_stale(record: dict, kind: str, root: Path | None = None) -> bool:
    """True when ``record`` (a parsed profiles/*.json) does not describe the sources in the tree."""
    return record.get("source_hash") != source_hash(kind, root)
