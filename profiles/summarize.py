#!/usr/bin/env python3
"""Summarise rocprofv3 outputs copied from gpurun_out/ into profiles/rNN/.

    python profiles/summarize.py profiles/r01

Reads <dir>/*_kernel_trace.csv (rocprofv3 --kernel-trace --stats --output-format csv) and, when
present, <dir>/*_pmc_fetch_size.csv / *_pmc_write_size.csv (two separate `--pmc` passes).  Writes
<dir>/summary.md and profiles/pmc_traffic.json = HBM bytes per launch of the dominant kernel:
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md
prescribes (gfx950 tallies 128-B read requests as 64 B).
"""

from __future__ import annotations

import csv
import json
import sys
from collections import defaultdict
from pathlib import Path


def short(name: str) -> str:
    return name.replace("void ", "").split("(")[0][:70]


def main() -> None:
    d = Path(sys.argv[1])
    tag = d.name
    trace = next(d.glob("*_kernel_trace.csv"))
    groups = defaultdict(list)
    for r in csv.DictReader(trace.open()):
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        key = short(r["Kernel_Name"])
        if "mt_dense" in key:
            key += f" [grid={r.get('Grid_Size_X', r.get('Grid_Size', '?'))}x{r.get('Grid_Size_Y', '')}]"
        groups[key].append(dur)
    lines = [f"# rocprofv3 --kernel-trace --stats summary ({tag})", "",
             "Commands (on the MI355X box, `cd /tmp && export TMPDIR=/tmp` first):", "",
             "    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o " + tag +
             " -- python bench.py --steps 10 --no-cpu-baseline",
             "    rocprofv3 --pmc FETCH_SIZE --output-format csv ... -- python bench.py --steps 3 --warmup 1 "
             "--no-cpu-baseline --no-paths",
             "    rocprofv3 --pmc WRITE_SIZE --output-format csv ... -- python bench.py --steps 3 --warmup 1 "
             "--no-cpu-baseline --no-paths", "",
             "`mt_dense_aligned_kernel [grid=524288x10]` are the timed launches of `bench.py` "
             "(65 536 rays x 10 000 triangles); `[grid=32768x10]` the literal 256-ray configs[1] launches.", "",
             "| kernel | calls | avg us | min us | max us | total ms |", "|---|---|---|---|---|---|"]
    # the timed launches of the driver-style run are the LAST `--steps` launches of the big grid (a time-based
    # pre-warm and the counted warm-up precede them)
    big = [k for k in groups if "mt_dense" in k and "524288" in k]
    if big:
        last20 = groups[big[0]][-20:]
        lines += [f"Timed launches of the bench (last 20 of `{big[0]}`): avg {sum(last20) / len(last20) / 1e3:.2f} us, "
                  f"min {min(last20) / 1e3:.2f}, max {max(last20) / 1e3:.2f}.", ""]
        lines += ["| kernel | calls | avg us | min us | max us | total ms |", "|---|---|---|---|---|---|"]
        del lines[lines.index("| kernel | calls | avg us | min us | max us | total ms |"):lines.index("| kernel | calls | avg us | min us | max us | total ms |") + 2]
    for k, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"| `{k}` | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | {min(v) / 1e3:.2f} | "
                     f"{max(v) / 1e3:.2f} | {sum(v) / 1e6:.3f} |")
    fetch, write = list(d.glob("*_pmc_fetch_size.csv")), list(d.glob("*_pmc_write_size.csv"))
    if fetch and write:
        per = {}
        for f, ctr in ((fetch[0], "FETCH_SIZE"), (write[0], "WRITE_SIZE")):
            vals = defaultdict(list)
            for r in csv.DictReader(f.open()):
                if r["Counter_Name"] == ctr and "mt_dense" in r["Kernel_Name"]:
                    vals[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
            big = max(vals)  # the timed launches (largest grid)
            per[ctr] = sum(vals[big]) / len(vals[big])
            per[ctr + "_launches"] = len(vals[big])
        traffic = (2 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024
        out = {
            "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), profiles/{tag}",
            "FETCH_SIZE_KiB_avg": per["FETCH_SIZE"], "WRITE_SIZE_KiB_avg": per["WRITE_SIZE"],
            "launches": per["FETCH_SIZE_launches"],
            "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE reads 1/2)",
            "mt_dense_kernel_bytes_per_launch": traffic,
        }
        try:  # the record describes the kernel sources in the tree at collection time (differt_amd/_srchash.py)
            sys.path.insert(0, str(d.resolve().parent.parent))
            from differt_amd._srchash import source_hash

            out["source_hash"] = source_hash("dense")
            out["source_hash_of"] = 'differt_amd/_srchash.py GROUPS["dense"]'
        except Exception:  # noqa: BLE001
            pass
        (d.parent / "pmc_traffic.json").write_text(json.dumps(out, indent=1))
        lines += ["", "## HBM traffic of `mt_dense_aligned_kernel` (PMC, per launch of 65536 rays x 10000 triangles)",
                  "", f"FETCH_SIZE {per['FETCH_SIZE']:.1f} KiB, WRITE_SIZE {per['WRITE_SIZE']:.1f} KiB -> "
                  f"{traffic / 1e9:.4f} GB per launch (algorithmic 3.2787 GB)."]
    (d / "summary.md").write_text("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
