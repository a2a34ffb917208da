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
