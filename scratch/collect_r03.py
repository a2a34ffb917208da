"""gpurun_out/prof_r03 -> profiles/r03 (run here, after scratch/profile_r03.sh ran on the GPU box)."""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

sys.path.insert(0, ".")
from differt_amd._srchash import source_hash  # noqa: E402

src, dst = Path("gpurun_out/prof_r03"), Path("profiles/r03")
dst.mkdir(parents=True, exist_ok=True)
(dst / "raw").mkdir(exist_ok=True)


def find(pat):
    return sorted(glob.glob(str(src / "**" / pat), recursive=True))


for pat, name in (("r03_kernel_trace.csv", "r03_kernel_trace.csv"), ("r03_kernel_stats.csv", "r03_kernel_stats.csv"),
                  ("r03_pmc_fetch_size_counter_collection.csv", "r03_pmc_fetch_size.csv"),
                  ("r03_pmc_write_size_counter_collection.csv", "r03_pmc_write_size.csv")):
    f = find(pat)
    if f:
        shutil.copy(f[0], dst / name)
for f in find("sq_*counter_collection.csv") + find("tr_*counter_collection.csv"):
    shutil.copy(f, dst / "raw" / Path(f).name)
for n in ("bench_default.json", "bench_driver.json"):
    if (src / n).exists():
        shutil.copy(src / n, dst / n)

# dense SQ counters (the timed launches = the largest grid)
dense = {}
for f in find("sq_*counter_collection.csv"):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mt_dense" in r["Kernel_Name"] and int(r["Grid_Size"]) > 1_000_000:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in vals.items():
        dense[k] = sum(v) / len(v)
dense["source_hash"] = source_hash("dense")
dense["source"] = "scratch/profile_r03.sh: rocprofv3 --pmc <set> -- python bench.py --steps 3 --warmup 1 ... (one pass per set), 65536 rays x 10000 triangles"
(dst / "pmc_dense_sq.json").write_text(json.dumps(dense, indent=1) + "\n")

# trace filter counters
tr = collections.defaultdict(list)
for f in find("tr_*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "trace_filter_kernel" in r["Kernel_Name"]:
            tr[r["Counter_Name"]].append(float(r["Counter_Value"]))
rec = {k: sum(v) / len(v) for k, v in tr.items()}
cands = 20_000_000 * 16 * 64
rec.update({"kernel": "drt::trace_filter_kernel<2, false, false>", "candidates_per_launch": cands,
            "source": "scratch/profile_r03.sh: rocprofv3 --pmc <set> -- python bench_paths.py --ranks 20000000 --steps 1 --no-cpu",
            "source_hash": source_hash("trace_filter")})
if "SQ_INSTS_VALU" in rec:
    rec["executed_valu_per_candidate"] = {"2": rec["SQ_INSTS_VALU"] * 64 / cands}
(dst / "pmc_trace_filter.json").write_text(json.dumps(rec, indent=1) + "\n")
print(json.dumps({"dense": dense, "filter": rec}, indent=1)[:2000])
