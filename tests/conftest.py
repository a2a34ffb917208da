"""Shared test configuration.

Markers: ``gpu`` = needs a real MI355X (run with ``-m gpu`` on the GPU box); everything else runs
on CPU.  Nothing here (or in any test) reads /root/reference: goldens live in tests/golden/.
"""

from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU (HIP extension loaded)")


@pytest.fixture(scope="session")
def goldens() -> dict:
    return json.loads((GOLDEN / "reference_goldens.json").read_text())


@pytest.fixture(scope="session")
def two_buildings() -> dict:
    d = json.loads((GOLDEN / "two_buildings.json").read_text())
    return {
        "vertices": np.asarray(d["vertices"], dtype=np.float32),
        "triangles": np.asarray(d["triangles"], dtype=np.int32),
    }


@pytest.fixture()
def rng() -> np.random.Generator:
    return np.random.default_rng(1234)


def canyon_scene(rng: np.random.Generator, nextra: int = 3):
    """Street canyon of boxes (two facing walls, a ground slab, a few obstacles): yields many valid
    order-1..3 paths AND many occlusion rejections.  Returns (vertices f32[Nv,3], triangles i32[T,3])."""
    import oracle as orc

    verts, tris = [], []

    def add(length, width, height, centre):
        v, t = orc.box_mesh(length, width, height, with_top=True)
        tris.append(t + 8 * len(verts))
        verts.append((v + np.asarray(centre, np.float32)).astype(np.float32))

    add(60, 8, 20, (0, -10, 10))
    add(60, 8, 20, (0, 10, 10))
    add(70, 30, 1, (0, 0, -0.5))
    for _ in range(nextra):
        add(*rng.uniform(2, 5, 3), (rng.uniform(-25, 25), rng.uniform(-4, 4), rng.uniform(1, 3)))
    return np.concatenate(verts), np.concatenate(tris).astype(np.int32)


def canyon_case(rng: np.random.Generator, order: int, assume_quads: bool):
    """(V, Tr, mask, tx[3,3], rx[5,3], candidates i32[C,order]) on the canyon scene."""
    import oracle as orc

    V, Tr = canyon_scene(rng)
    mask = rng.random(Tr.shape[0]) > 0.1
    mask[:36] = True
    if assume_quads:
        mask[1::2] = mask[0::2]
    tx = np.stack([rng.uniform(-20, 20, 3), rng.uniform(-3.5, 3.5, 3), rng.uniform(5, 15, 3)], -1).astype(np.float32)
    rx = np.stack([rng.uniform(-20, 20, 5), rng.uniform(-3.5, 3.5, 5), rng.uniform(1, 8, 5)], -1).astype(np.float32)
    n_prim = Tr.shape[0] // 2 if assume_quads else Tr.shape[0]
    if order == 3:  # keep the oracle's brute force small: the reflecting faces + random others
        key = np.array([0, 1, 16, 17, 34, 35])
        if assume_quads:
            key = np.unique(key // 2)
        others = np.setdiff1d(np.arange(n_prim), key)
        sub = np.sort(np.concatenate([key, rng.choice(others, (14 if assume_quads else 20) - len(key), replace=False)]))
        cand = sub[orc.generate_all_path_candidates(len(sub), order)]
    else:
        cand = orc.generate_all_path_candidates(n_prim, order)
    cand = (cand * (2 if assume_quads else 1)).astype(np.int32)
    return V, Tr, mask, tx, rx, cand
