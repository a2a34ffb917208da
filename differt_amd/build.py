"""Build libdiffert_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m differt_amd.build [--force]

Flags that matter for parity: -ffp-contract=off (hipcc's device default is `fast`) and correctly
rounded fp32 divide/sqrt, so every mask decision is one-rounding-per-operation like the oracle.
"""

from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT = Path(os.environ.get("DIFFERT_AMD_LIB", HERE / "lib" / "libdiffert_amd.so"))
EXTRA = os.environ.get("DRT_EXTRA_FLAGS", "").split()

FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-fno-fast-math",
    # measured on MI355X: SLP-packed v_pk_mul/add_f32 run at half the issue rate of the scalar
    # forms and need v_mov shuffles -> 8 % slower dense kernel
    "-fno-slp-vectorize",
    "-fno-gpu-rdc",
    "-Wall",
    "-Wno-unused-function",
]


def sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.cpp"))


def needs_build() -> bool:
    if not OUT.exists():
        return True
    mtime = OUT.stat().st_mtime
    deps = list(CSRC.glob("*")) + [HERE.parent / "include" / "differt_amd.h", Path(__file__)]
    return any(p.stat().st_mtime > mtime for p in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    OUT.parent.mkdir(parents=True, exist_ok=True)
    objs = []
    procs = []
    objdir = OUT.parent / ("obj_" + OUT.stem)
    objdir.mkdir(parents=True, exist_ok=True)
    compile_flags = [f for f in FLAGS if f != "-shared"] + EXTRA
    for src in sources():
        obj = objdir / (src.stem + ".o")
        objs.append(obj)
        if src.suffix == ".cpp":  # host-only translation units (candidate enumeration)
            cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-fopenmp", "-Wall", "-c", str(src), "-o", str(obj)]
        else:
            cmd = [hipcc, *compile_flags, "-x", "hip", "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src.name} ---\n{out.decode()}\n")
        elif verbose and out:
            sys.stderr.write(out.decode())
    if failed:
        raise RuntimeError("hipcc failed")
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT), *map(str, objs), "-lgomp"]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        sys.stderr.write(r.stdout.decode())
        raise RuntimeError("link failed")
    return OUT


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(path)
