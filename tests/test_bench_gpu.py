"""bench.py contract (one JSON line with `roofline`; N > 1 under torch.distributed.run) on the GPU box.

The two-rank case uses the DRT_BENCH_SHARE_GPU test hook (both ranks on cuda:0 over gloo): it checks
the sharding / barrier / all-reduce / rank-0 JSON plumbing, not performance.
"""

from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]
COMMON = ["--steps", "2", "--warmup", "1", "--rays", "2048", "--no-cpu-baseline", "--no-paths"]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _last_json(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, "bench.py", *COMMON], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0
    rf = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rf) and rf["bound"] == "hbm"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"]


def test_two_ranks_share_gpu():
    env = dict(os.environ, DRT_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "2", *COMMON]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)  # rank 0 only
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "TEST HOOK" in d["data"]
    assert d["value"] > 0 and d["config"]["rays_per_gpu"] == 2048
