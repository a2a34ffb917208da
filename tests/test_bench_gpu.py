"""bench.py contract (one COMPACT JSON line with `roofline`, every leg in full in a sidecar file; N > 1 under
torch.distributed.run) on the GPU box.

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
COMMON = ["--steps", "2", "--warmup", "1", "--rays", "2048", "--no-cpu-baseline", "--no-paths",
          # configs[4] strong-scaling legs on a small city (400 boxes = 4000 triangles, 4 x 4 receivers, whole space)
          "--cfg5-boxes", "400", "--cfg5-rx-side", "4"]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _last_json(out: str) -> dict:
    """The line as the driver reads it (the last line of an 8 KB stdout tail) + the sidecar's legs merged in."""
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    assert len(lines[0]) < 4000, len(lines[0])
    d = json.loads(out[-8000:].splitlines()[-1])
    assert d == json.loads(lines[0])
    full = json.loads((ROOT / d["full"]).read_text() if not os.path.isabs(d["full"]) else Path(d["full"]).read_text())
    for k in ("metric", "value", "n_gpus", "steps", "warmup"):
        assert full[k] == pytest.approx(d[k], rel=1e-5), k
    d["strong_scaling"] = full.get("strong_scaling")
    d["prewarm_ms"] = full["prewarm_ms"]
    d["full_record"] = full
    return d


def test_single_gpu_line():
    r = subprocess.run([sys.executable, "bench.py", "--full-json", "/tmp/drt_bench_full_n1.json", *COMMON], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0
    rf = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "valu_frac", "valu_frac_of_157TF"} <= set(rf)
    assert rf["bound"] == "hbm+valu" and 0 < rf["valu_frac"] < 1 and abs(rf["valu_frac_of_157TF"] * 2 / rf["valu_frac"] - 1) < 0.01
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5  # (the line carries 6 significant digits)
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"]
    assert "strong_headline" not in d  # N = 1: the line is the plain single-GPU bench
    assert d["prewarm_ms"] > 0 and d["roofline"]["pmc_stale"] in (True, False)
    _check_strong(d["strong_scaling"], 1)
    Path(os.environ.get("DRT_BENCH_N1_JSON", "/tmp/drt_bench_n1.json")).write_text(json.dumps(d["strong_scaling"]))


def _check_strong(sc: dict, world: int):
    assert sc["scaling"] == "strong" and sc["n_gpus"] == world, sc
    cs, tb = sc["candidate_sharded"], sc["triangle_block"]
    assert "error" not in cs and "error" not in tb, sc
    assert cs["path_candidates_per_s"] > 0 and cs["valid_paths"] > 0 and cs["keys_sorted"] and cs["grad_tx_finite"]
    assert tb["rays_per_s"] > 0 and 0 < tb["hit_fraction"] <= 1
    bs = sc["beam_sharded"]
    assert "error" not in bs, sc
    # the window of the candidate-sharded leg is the WHOLE space on this small city: same valid paths
    assert bs["valid_paths"] == cs["valid_paths"] and bs["grad_tx_finite"] and bs["s_per_step"] > 0
    if world > 1:
        assert sc["ranks_seen_by_backend"] == world


def test_two_ranks_share_gpu():
    env = dict(os.environ, DRT_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "2", "--full-json", "/tmp/drt_bench_full_n2.json",
           *COMMON]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)  # rank 0 only
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "TEST HOOK" in d["data"]
    assert d["value"] > 0 and d["config"]["rays_per_gpu"] == 2048
    _check_strong(d["strong_scaling"], 2)
    # N > 1: the fixed-total-work legs are surfaced at top level (the north-star scaling claim), `value` stays weak
    heads = {h["leg"]: h for h in d["strong_headline"]}
    assert set(heads) == {"triangle_block", "beam_sharded", "candidate_sharded"}  # triangle_block: the leg configs[4] names
    for leg, h in heads.items():
        assert h["n_gpus"] == 2 and h["scaling"] == "strong"
        assert h["s_per_step"] == pytest.approx(d["strong_scaling"][leg]["s_per_step"], rel=1e-4)  # (5 significant digits in the line)
    ref = Path(os.environ.get("DRT_BENCH_N1_JSON", "/tmp/drt_bench_n1.json"))
    if ref.exists():  # same fixed work as the single-rank run: same valid paths, same first hits, same gradient
        one = json.loads(ref.read_text())
        two = d["strong_scaling"]
        assert two["candidate_sharded"]["valid_paths"] == one["candidate_sharded"]["valid_paths"]
        assert two["candidate_sharded"]["path_candidates_per_step"] == one["candidate_sharded"]["path_candidates_per_step"]
        assert two["triangle_block"]["checksum_idx"] == one["triangle_block"]["checksum_idx"]
        g1, g2 = one["candidate_sharded"]["grad_tx_absmax"], two["candidate_sharded"]["grad_tx_absmax"]
        assert abs(g1 - g2) <= 1e-5 * max(abs(g1), 1e-30)
        assert two["beam_sharded"]["valid_paths"] == one["beam_sharded"]["valid_paths"]
        assert two["beam_sharded"]["checksum_keys"] == one["beam_sharded"]["checksum_keys"]
        g1, g2 = one["beam_sharded"]["grad_tx_absmax"], two["beam_sharded"]["grad_tx_absmax"]
        assert abs(g1 - g2) <= 1e-5 * max(abs(g1), 1e-30)


def test_default_command_line_is_parseable_by_the_driver():
    """The DEFAULT command as the driver runs it (`--steps 20 --warmup 5`, paths ON): ONE JSON line under 4 000
    characters that the last line of an 8 KB stdout tail parses to, carrying `roofline`, `cpu_baseline`, the paths
    half of the BASELINE metric, and the sidecar with every leg (the r04 line was 20 KB)."""
    side = Path("/tmp/drt_bench_full_default.json")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--full-json", str(side)],
                       cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout[-8000:].splitlines()[-1]
    assert len(line) < 4000, len(line)
    d = json.loads(line)
    assert KEYS <= set(d) and d["full"] == str(side)
    assert {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"} <= set(d["roofline"])
    assert {"value", "unit", "cores", "kind", "reference_installed"} <= set(d["cpu_baseline"]) and d["cpu_baseline"]["value"] > 0
    assert d["cpu_baseline"]["reference_installed"] is False  # (no jax / differt on these boxes: reported, not hidden)
    assert d["roofline"]["bound"] == "hbm+valu" and d["cfg2_literal_us"] > 0 and 0 < d["cfg2_literal_hbm_frac"] < 1
    pm = d["paths_metric"]
    assert pm["unit"] == "valid order-2 paths/s (fwd+grad)" and pm["config"] == "configs[2]" and pm["value"] > 0
    assert pm["same_as_exhaustive"] is True
    ps = d["paths"]
    for k in ("cfg2_fwdgrad_s", "cfg3_s", "cfg4_s", "dense_api_frac", "cfg3_expand_valu_frac", "cfg3_expand_frac_of_157TF"):
        assert ps[k] is not None and ps[k] > 0, (k, ps)
    assert ps["same_as_exhaustive"] is True
    for v in d.values():  # numbers and short identifiers only: no prose in the line
        assert not isinstance(v, str) or len(v) < 100
    full = json.loads(side.read_text())
    # every VALU roofline block states both denominators, and no fraction is null
    def walk(o):
        if isinstance(o, dict):
            if o.get("bound") == "valu":
                assert o.get("frac") is not None and o.get("frac_of_157TF") is not None, o
            for v in o.values():
                walk(v)
    walk(full)
