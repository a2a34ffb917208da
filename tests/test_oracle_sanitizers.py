"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers row): the
golden-vector suite is re-run in a subprocess against `make -C oracle sanitize` with libasan preloaded.  Any
out-of-bounds access, use of an uninitialised shift / overflow / misaligned load in the oracle aborts the run."""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_golden_suite_under_asan_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not Path(asan).exists():
        pytest.skip("libasan not installed")
    r = subprocess.run(["make", "-C", str(ROOT / "oracle"), "sanitize"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    so = ROOT / "oracle" / "_build" / "libdiffert_oracle_san.so"
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", DRT_ORACLE_LIB=str(so))
    probe = subprocess.run([sys.executable, "-c", "import oracle; print(oracle.lib()._name)"], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=300)
    assert probe.returncode == 0 and probe.stdout.strip().endswith("libdiffert_oracle_san.so"), probe.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_oracle_golden.py", "tests/test_oracle_twin.py", "-x", "-q",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-3000:]
    assert " passed" in out
