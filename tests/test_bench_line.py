"""bench.py's ONE line (CPU): whatever the legs hold, the line the driver parses stays under 4 000 characters, carries
the contract keys + roofline + cpu_baseline + the paths half of the BASELINE metric, and holds no prose (VERDICT r04:
round 4's line had grown to 20 KB and the driver's 8 KB tail cut the head off)."""

import json
from pathlib import Path

import numpy as np

import bench
import synthetic_scenes as S

ROOT = Path(__file__).resolve().parents[1]


def test_compact_line_of_the_round4_record():
    full = json.loads((ROOT / "profiles" / "r04" / "bench_driver.json").read_text())  # the 20 KB line of round 4
    assert len(json.dumps(full)) > 15000
    full["cpu_baseline"]["sample_short"] = "154 x (8192 rays x 10000 triangles), 12.0 s"
    line = json.dumps(bench.compact_line(full, "gpurun_out/bench_full.json"), separators=(",", ":"))
    assert len(line) < bench.LINE_LIMIT
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "paths_metric", "paths", "full"):
        assert k in d, k
    assert d["value"] == float(f"{full['value']:.7g}") and d["vs_baseline"] is None and "workload" in d["config"]
    assert {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"} <= set(d["roofline"])
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-5
    assert {"value", "unit", "cores", "kind"} <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    pm = d["paths_metric"]
    assert pm["unit"] == "valid order-2 paths/s (fwd+grad)" and pm["config"] == "configs[2]" and pm["valid_paths"] == 54
    assert pm["value"] == float(f"{54 / full['paths']['beam_pruned']['s_per_step']:.5g}")
    ps = d["paths"]
    assert ps["cfg3_s"] > 0 and ps["cfg4_s"] > 0 and ps["cfg2_fwdgrad_s"] > 0 and 0 < ps["dense_api_frac"] <= 1
    assert ps["same_as_exhaustive"] is True
    for v in list(d.values()) + list(ps.values()):
        assert not isinstance(v, str) or len(v) < 100  # identifiers, not prose


def test_compact_line_survives_failed_legs():
    full = json.loads((ROOT / "profiles" / "r04" / "bench_driver.json").read_text())
    full["paths"] = {"error": "RuntimeError('x' * 5000)" + "x" * 5000}
    full["strong_scaling"] = {"error": "boom"}
    full.pop("cpu_baseline")
    line = json.dumps(bench.compact_line(full, None), separators=(",", ":"))
    assert len(line) < bench.LINE_LIMIT
    d = json.loads(line)
    assert "error" in d["paths"] and "paths_metric" not in d and d["full"] is None


def test_random_rotation_is_a_rotation_with_bounded_tilt():
    rng = np.random.default_rng(0)
    for _ in range(200):
        R = S.random_rotation(rng, 10.0)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
        assert R[2, 2] >= np.cos(np.deg2rad(10.0)) - 1e-12  # the z axis tilts by at most 10 degrees
    (p,) = S.rotate_points(np.eye(3), np.array([[1.5, 2.5, 3.5]], np.float32))
    assert p.dtype == np.float32 and np.array_equal(p, np.array([[1.5, 2.5, 3.5]], np.float32))


def test_a_dead_legs_process_costs_its_legs_not_the_line(monkeypatch, tmp_path):
    """On one GPU the extra legs run in a child process (bench.legs_in_child): a child that dies -- a GPU memory fault
    aborts the process -- yields `legs_error`, and the line with the headline, roofline and cpu_baseline is still built."""
    import subprocess
    import types

    args = types.SimpleNamespace(cfg5_boxes=400, cfg5_rx_side=4, cfg5_window=None, no_cpu_baseline=True, no_paths=False,
                                 no_scaling=False)

    def dead(cmd, **kw):
        assert "--legs-child" in cmd and "--no-cpu-baseline" in cmd
        return subprocess.CompletedProcess(cmd, -6)  # SIGABRT

    monkeypatch.setattr(subprocess, "run", dead)
    legs = bench.legs_in_child(args)
    assert "legs_error" in legs and "-6" in legs["legs_error"]
    full = json.loads((ROOT / "profiles" / "r04" / "bench_driver.json").read_text())
    for k in ("paths", "strong_scaling", "queries"):
        full.pop(k)
    full.update(legs)
    d = json.loads(json.dumps(bench.compact_line(full, None)))
    assert d["legs_error"].startswith("the legs process exited") and d["value"] > 0 and "roofline" in d and "paths" not in d
