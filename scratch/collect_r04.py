"""gpurun_out/prof_r04_dense -> profiles/r04 (run here, after scratch/profile_r04_dense.sh ran on the GPU box)."""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

sys.path.insert(0, ".")
from differt_amd._srchash import source_hash  # noqa: E402

src, dst = Path("gpurun_out/prof_r04_dense"), Path("profiles/r04")
dst.mkdir(parents=True, exist_ok=True)
(dst / "raw").mkdir(exist_ok=True)

for n in ("dense_kernel_stats.csv", "dense_kernel_trace.csv", "dense_legacy_kernel_stats.csv", "dense3_kernel_stats.csv",
          "dense_all.json", "dense_traced.json"):
    if (src / n).exists():
        shutil.copy(src / n, dst / n)
for n in src.glob("pmc*_counter_collection.csv"):
    shutil.copy(n, dst / "raw" / n.name)


def counter(tag, c):
    vals = collections.defaultdict(list)
    f = src / f"{tag}_{c}_counter_collection.csv"
    if not f.exists():
        return None
    for r in csv.DictReader(open(f)):
        if "trace_dense_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == 524288:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    v = next(iter(vals.values()), None)
    return None if not v else sum(v) / len(v)


rows = 64 * (1 << 20)
rec = {"kernel": "drt::trace_dense_kernel<K, false>", "rows_per_launch": rows, "bytes_per_launch": {}, "detail": {},
       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half, MI355X_MICROARCH.md)",
       "source": "scratch/profile_r04_dense.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each, counters only) -- "
                 "python bench_dense.py [--order 3] --max-chunks 6; full-size launches (1 TX x 64 RX x 2^20 candidates)",
       "source_hash": source_hash("trace_dense"),
       "source_hash_of": "differt_amd/_srchash.py GROUPS[\"trace_dense\"]"}
for order, tag in ((2, "pmc"), (3, "pmc3")):
    f, w = counter(tag, "fetch_size"), counter(tag, "write_size")
    if f is None or w is None:
        continue
    written = 12 * (order + 2) + 4 * (order + 2) + 1 + 4 * order
    rec["bytes_per_launch"][str(order)] = (2 * f + w) * 1024
    rec["detail"][str(order)] = {"FETCH_SIZE_KiB_avg": f, "WRITE_SIZE_KiB_avg": w,
                                 "written_bytes_expected": written * rows,
                                 "write_traffic_over_expected": w * 1024 / (written * rows),
                                 "survey_algorithmic_bytes": (written + 4 * order) * rows}
(dst / "pmc_trace_dense.json").write_text(json.dumps(rec, indent=1) + "\n")
print(json.dumps(rec, indent=1))
