"""gpurun_out/prof_r06 -> profiles/r06 (run here, after scratch/profile_r06.sh ran on the GPU box)."""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

sys.path.insert(0, ".")
from differt_amd._srchash import source_hash  # noqa: E402

src, dst = Path("gpurun_out/prof_r06"), Path("profiles/r06")
ONLY_TRAFFIC = "--traffic-only" in sys.argv
dst.mkdir(parents=True, exist_ok=True)
(dst / "raw").mkdir(exist_ok=True)


def find(pat):
    return sorted(glob.glob(str(src / "**" / pat), recursive=True))


def copy(pat, name):
    f = find(pat)
    if f:
        shutil.copy(f[0], dst / name)
        return True
    return False


copy("r06_kernel_stats.csv", "r06_kernel_stats.csv")
copy("r06_kernel_trace.csv", "r06_kernel_trace.csv")
copy("dense_kernel_stats.csv", "dense_kernel_stats.csv")
for leg in ("cfg4", "cfg4quads", "cfg3", "bruxelles3"):
    copy(f"beam_{leg}_kernel_stats.csv", f"beam_{leg}_kernel_stats.csv")
for n in ("strong_n1.json", "bench_default.json", "bench_driver.json", "bench_traced.json", "dense_traced.json", "emulate_shards8.json",
          "bench_default_full.json", "bench_driver_full.json", "bench_traced_full.json"):
    if (src / n).exists() and (src / n).stat().st_size:
        shutil.copy(src / n, dst / n)
for f in find("*counter_collection.csv"):
    shutil.copy(f, dst / "raw" / Path(f).name)


def counters(pattern, kernel_substr, grid=None):
    """mean per launch and launch count of every counter of the kernels whose name contains kernel_substr"""
    vals = collections.defaultdict(list)
    for f in find(pattern):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"] and (grid is None or int(r["Grid_Size"]) == grid):
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v), sum(v)) for k, v in vals.items()}


out = {}
# ---- headline kernel (dense Moller-Trumbore): HBM traffic per launch
f = counters("pmcmt_fetch_size_counter_collection.csv", "mt_dense_aligned_kernel", None)
w = counters("pmcmt_write_size_counter_collection.csv", "mt_dense_aligned_kernel", None)
def _big(pattern, name):
    vals = []
    for fn in find(pattern):
        for r in csv.DictReader(open(fn)):
            if "mt_dense_aligned_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name and int(r["Grid_Size"]) == 5242880:
                vals.append(float(r["Counter_Value"]))
    return vals
fv, wv = _big("pmcmt_fetch_size_counter_collection.csv", "FETCH_SIZE"), _big("pmcmt_write_size_counter_collection.csv", "WRITE_SIZE")
if fv and wv:
    fa, wa = sum(fv) / len(fv), sum(wv) / len(wv)
    rec = {"source": "scratch/profile_r06.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, counters only) -- python bench.py --steps 3 --warmup 1 --no-paths; full-size launches (65 536 rays x 10 000 triangles)",
           "FETCH_SIZE_KiB_avg": fa, "WRITE_SIZE_KiB_avg": wa, "launches": len(wv),
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE reads 1/2)",
           "mt_dense_kernel_bytes_per_launch": (2 * fa + wa) * 1024,
           "source_hash": source_hash("dense"), "source_hash_of": "differt_amd/_srchash.py GROUPS[\"dense\"]"}
    Path("profiles/pmc_traffic.json").write_text(json.dumps(rec, indent=1) + "\n")
    out["dense_mt"] = rec["mt_dense_kernel_bytes_per_launch"]
# ---- dense tracer: HBM traffic per launch
rows = 64 * (1 << 20)
rec = {"kernel": "drt::trace_dense_kernel<K, false>", "rows_per_launch": rows, "bytes_per_launch": {}, "detail": {},
       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half, MI355X_MICROARCH.md)",
       "source": "scratch/profile_r06.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each, counters only) -- "
                 "python bench_dense.py [--order 3] --max-chunks 6; full-size launches (1 TX x 64 RX x 2^20 candidates)",
       "source_hash": source_hash("trace_dense"), "source_hash_of": "differt_amd/_srchash.py GROUPS[\"trace_dense\"]"}
for order, tag in ((2, "pmc"), (3, "pmc3")):
    f = counters(f"{tag}_fetch_size_counter_collection.csv", "trace_dense_kernel", 524288).get("FETCH_SIZE")
    w = counters(f"{tag}_write_size_counter_collection.csv", "trace_dense_kernel", 524288).get("WRITE_SIZE")
    if not f or not w:
        continue
    written = 12 * (order + 2) + 4 * (order + 2) + 1 + 4 * order
    rec["bytes_per_launch"][str(order)] = (2 * f[0] + w[0]) * 1024
    rec["detail"][str(order)] = {"FETCH_SIZE_KiB_avg": f[0], "WRITE_SIZE_KiB_avg": w[0], "launches": w[1],
                                 "written_bytes_expected": written * rows,
                                 "write_traffic_over_expected": w[0] * 1024 / (written * rows),
                                 "survey_algorithmic_bytes": (written + 4 * order) * rows}
if rec["bytes_per_launch"]:
    (dst / "pmc_trace_dense.json").write_text(json.dumps(rec, indent=1) + "\n")
    out["trace_dense"] = rec["detail"]

# ---- exhaustive filter kernel
tr = {}
for f in find("tr_*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "trace_filter_kernel" in r["Kernel_Name"]:
            tr.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if tr:
    frec = {k: sum(v) / len(v) for k, v in tr.items()}
    cands = 20_000_000 * 16 * 64
    frec.update({"kernel": "drt::trace_filter_kernel<2, false>", "candidates_per_launch": cands,
                 "source": "scratch/profile_r06.sh: rocprofv3 --pmc <set> -- python bench_paths.py --ranks 20000000 --steps 1 --no-cpu",
                 "source_hash": source_hash("trace_filter")})
    if "SQ_INSTS_VALU" in frec:
        frec["executed_valu_per_candidate"] = {"2": frec["SQ_INSTS_VALU"] * 64 / cands}
    (dst / "pmc_trace_filter.json").write_text(json.dumps(frec, indent=1) + "\n")
    out["trace_filter"] = frec.get("executed_valu_per_candidate")

# ---- pruned search: the last expansion (two steps per run: warm-up + timed)
legs = {}
for leg, key, cfg in (("cfg4", "order3", "configs[3], triangle mesh (coplanar-pair mode)"),
                      ("cfg4quads", "order3_quads", "configs[3], assume_quads"),
                      ("cfg3", "order2", "configs[2], triangle mesh (coplanar-pair mode)"),
                      ("bruxelles3", "bruxelles_order3", "bruxelles.obj (14 206 triangles -> 8 376 primitives of the pairing pass), 16 TX x 64 RX, order 3")):
    # the last expansion = the fused kernel (orders 1-2) or, at order 3, the box-stage kernel + the per-primitive kernel
    LAST = ("beam_expand_clustered_last_kernel", "beam_boxes_kernel", "beam_expand_pairs_kernel")
    c = {}
    for pat in LAST:
        for k, (mean, cnt, tot) in counters(f"pmcbeam_{leg}_*counter_collection.csv", pat).items():
            m0, c0, t0 = c.get(k, (0.0, 0, 0.0))
            c[k] = (0.0, max(c0, cnt), t0 + tot)
    if not c:
        continue
    steps = 2
    names = set()
    for f in find(f"pmcbeam_{leg}_*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if any(pat in r["Kernel_Name"] for pat in LAST):
                names.add(r["Kernel_Name"].split("(")[0].replace("void ", ""))
    d = {"config": cfg, "kernel": " + ".join(sorted(names)) if names else None, "launches_per_step": c["SQ_INSTS_VALU"][1] / steps}
    for k, (_, _, tot) in c.items():
        d[f"{k}_per_step"] = tot / steps
    if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
        d["SQ_WAIT_INST_ANY_over_SQ_WAVE_CYCLES"] = c["SQ_WAIT_INST_ANY"][2] / c["SQ_WAVE_CYCLES"][2]
    if "GRBM_GUI_ACTIVE" in c and "SQ_INSTS_VALU" in c:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; a SIMD issues one wave instruction per 2 cycles; 1024 SIMDs
        cycles = c["GRBM_GUI_ACTIVE"][2] / 8
        d["kernel_cycles_per_step"] = cycles / steps
        d["valu_issue_frac_from_counters"] = c["SQ_INSTS_VALU"][2] / 1024 * 2 / cycles
    # kernel time of the same step from the kernel trace
    for f in find(f"beam_{leg}_kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if any(pat in r["Name"] for pat in LAST):
                d["kernel_ms_per_step_from_trace"] = d.get("kernel_ms_per_step_from_trace", 0.0) + float(r["TotalDurationNs"]) / 1e6 / steps
                d.setdefault("kernel_ms_per_step_by_kernel", {})[r["Name"].split("(")[0].replace("void ", "")] = float(r["TotalDurationNs"]) / 1e6 / steps
    legs[key] = d
if legs:
    brec = {"legs": legs, "source": "scratch/profile_r06.sh: rocprofv3 --pmc <set> -- python scratch/cfg_beam.py cfg4 [--quads] | cfg3 "
                                   "(two steps per run; one pass per counter set, counters only)",
            "source_hash": source_hash("beam"), "source_hash_of": "differt_amd/_srchash.py GROUPS[\"beam\"]"}
    (dst / "pmc_beam_expand.json").write_text(json.dumps(brec, indent=1) + "\n")
    out["beam"] = {k: {kk: vv for kk, vv in v.items() if "frac" in kk or "ms" in kk or "over" in kk} for k, v in legs.items()}
print(json.dumps(out, indent=1))
