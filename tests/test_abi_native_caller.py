"""The C ABI is usable from compiled host code with no Python / torch in the process: build
tests/abi/abi_host_example.cpp against include/differt_amd.h + libdiffert_amd.so and run it."""

from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "abi" / "abi_host_example.cpp"
LIBDIR = ROOT / "differt_amd" / "lib"


def _build(tmp: Path) -> Path:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp / "abi_host_example"
    subprocess.run(
        [hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", str(ROOT / "include"), str(SRC), "-L",
         str(LIBDIR), "-ldiffert_amd", f"-Wl,-rpath,{LIBDIR}", "-o", str(exe)],
        check=True, capture_output=True,
    )
    return exe


def test_native_caller_compiles_against_the_header(tmp_path):
    """CPU: the example compiles and links (header and exported symbols are consistent)."""
    assert _build(tmp_path).exists()


@pytest.mark.gpu
def test_native_caller_runs(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                       env={**os.environ, "LD_LIBRARY_PATH": f"{LIBDIR}:{os.environ.get('LD_LIBRARY_PATH', '')}"})
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("OK"), r.stdout
