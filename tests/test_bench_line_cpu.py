"""bench.py's ONE line (CPU part of the contract): the compact summary of a full record parses, carries the contract keys plus
`roofline` / `cpu_baseline`, and a line that outgrows the driver's stdout tail sheds optional blocks instead of killing the run."""

import json
from pathlib import Path

import bench

ROOT = Path(__file__).resolve().parents[1]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"}


def _full():
    recs = sorted((ROOT / "profiles").glob("r*/bench_driver_full.json"))
    return json.loads(recs[-1].read_text())


def test_compact_line_of_the_committed_full_record():
    comp = bench.compact_line(_full(), "gpurun_out/bench_full.json")
    line = bench.fit_line(dict(comp))
    assert len(line) < bench.LINE_LIMIT and "line_shed" not in json.loads(line)
    d = json.loads(line)
    assert KEYS <= set(d) and d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm+valu" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5 and 0 < rf["valu_frac"] < 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["reference_installed"] is False
    assert d["cfg2_literal_us"] > 0 and d["paths_metric"]["entry"] == "drt_trace_paths_beam"


def test_a_line_that_outgrows_the_limit_sheds_optional_blocks_only():
    comp = bench.compact_line(_full(), "x" * 50)
    comp["strong_headline"] = [{"leg": "beam_sharded", "s_per_step": 0.02, "n_gpus": 8, "note": "y" * 3000}]
    comp["paths"] = {**comp["paths"], "pad": "z" * 3000}
    d = json.loads(bench.fit_line(comp))
    assert d["line_shed"] == ["strong_headline", "paths"] and "paths" not in d and "strong_headline" not in d
    assert KEYS <= set(d) and "roofline" in d and "cpu_baseline" in d and "paths_metric" in d
    assert len(json.dumps(d, separators=(",", ":"))) < bench.LINE_LIMIT
