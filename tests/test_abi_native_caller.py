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
BEAM_SRC = ROOT / "tests" / "abi" / "abi_beam_example.cpp"
COMM_SRC = ROOT / "tests" / "abi" / "abi_comm_two_rank.cpp"
THREADS_SRC = ROOT / "tests" / "abi" / "abi_threads.cpp"
LIBDIR = ROOT / "differt_amd" / "lib"


def _build(tmp: Path, src: Path = SRC) -> Path:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp / src.stem
    subprocess.run(
        [hipcc, "-O2", "-std=c++17", "-pthread", "--offload-arch=gfx950", "-I", str(ROOT / "include"), str(src), "-L",
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


def test_native_beam_caller_compiles_against_the_header(tmp_path):
    assert _build(tmp_path, BEAM_SRC).exists()


def _run_beam(exe, tmp, V, Tr, tx, rx, order, quads=False, max_paths=4096):
    import numpy as np

    scene, out = tmp / "scene.bin", tmp / "out.bin"
    with scene.open("wb") as f:
        f.write(np.asarray([len(V), len(Tr), int(quads), len(tx), len(rx), order, max_paths], np.int64).tobytes())
        for a, dt in ((V, np.float32), (Tr, np.int32), (tx, np.float32), (rx, np.float32)):
            f.write(np.ascontiguousarray(a, dt).tobytes())
    r = subprocess.run([str(exe), str(scene), str(out)], capture_output=True, text=True, timeout=600,
                       env={**os.environ, "LD_LIBRARY_PATH": f"{LIBDIR}:{os.environ.get('LD_LIBRARY_PATH', '')}"})
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr
    raw = out.read_bytes()
    head = np.frombuffer(raw[:48], np.int64)
    nv, k2, off = int(head[0]), order + 2, 48
    keys = np.frombuffer(raw[off:off + 8 * nv], np.int64); off += 8 * nv
    objs = np.frombuffer(raw[off:off + 4 * nv * k2], np.int32).reshape(nv, k2); off += 4 * nv * k2
    verts = np.frombuffer(raw[off:off + 12 * nv * k2], np.float32).reshape(nv, k2, 3); off += 12 * nv * k2
    gtx = np.frombuffer(raw[off:off + 12 * len(tx)], np.float32).reshape(len(tx), 3)
    return head, keys, objs, verts, gtx


@pytest.mark.gpu
def test_native_beam_caller_reproduces_the_reference_goldens(tmp_path, two_buildings, goldens):
    """No torch, no Python in the traced process: drt_trace_paths_beam on the reference's two-buildings scene returns
    the valid paths its own test holds (differt/tests/geometry/test_scene.py:116-160), orders 0..3."""
    import numpy as np

    exe = _build(tmp_path, BEAM_SRC)
    ex = goldens["advanced_path_tracing_example"]
    V, Tr = two_buildings["vertices"], two_buildings["triangles"]
    for order in (0, 1, 2, 3):
        g = ex["orders"][str(order)]
        _, keys, objs, verts, gtx = _run_beam(exe, tmp_path, V, Tr, [ex["tx"]], [ex["rx"]], order)
        assert objs.tolist() == g["objects"]
        # the goldens hold the interior vertices (reflection points); the end points are the inputs themselves
        np.testing.assert_allclose(verts[:, 1:-1], np.asarray(g["path_vertices"], np.float32).reshape(1, order, 3),
                                   rtol=ex["rtol"])
        assert np.array_equal(verts[:, 0], np.asarray([ex["tx"]], np.float32)) and np.array_equal(verts[:, -1], np.asarray([ex["rx"]], np.float32))
        n = len(Tr)
        assert keys.tolist() == [sum(int(m) * n ** (order - 1 - j) for j, m in enumerate(o[1:-1])) for o in objs]
        assert np.isfinite(gtx).all()


@pytest.mark.gpu
def test_native_beam_caller_cfg3_matches_the_exhaustive_keys(tmp_path):
    """BASELINE configs[2] (16 TX x 64 RX, 10 000 triangles, order 2: 1.0e11 candidates) through the native caller:
    the 54 valid paths of the exhaustive tracer -- same keys, objects, vertex bits -- and the same grad(TX)."""
    import numpy as np
    import torch

    import differt_amd.geometry as G
    import synthetic_scenes as S

    V, Tr, c, h = S.manhattan(1000)
    tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
    exe = _build(tmp_path, BEAM_SRC)
    head, keys, objs, verts, gtx = _run_beam(exe, tmp_path, V, Tr, tx, rx, 2)
    txg = torch.tensor(tx, device="cuda", requires_grad=True)
    scene = G.Scene(txg, torch.tensor(rx, device="cuda"), G.Mesh(V, Tr))
    ex = G.ExhaustivePathTracer().trace_rank_range_literal(scene, 2, max_survivors=1 << 24, max_paths=1 << 16)
    assert ex.objects.shape[0] == 54 and objs.shape[0] == 54
    assert np.array_equal(objs, ex.objects.cpu().numpy())
    assert np.array_equal(verts.view(np.uint32), ex.vertices.detach().cpu().numpy().view(np.uint32))
    n, o = len(Tr), ex.objects.cpu().numpy().astype(np.int64)
    assert np.array_equal(keys, (o[:, 0] * 64 + o[:, 3]) * n * n + o[:, 1] * n + o[:, 2])
    ex.vertices.sum().backward()
    np.testing.assert_allclose(gtx, txg.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)
    assert head[1] < 1e-4 * 1024 * n * (n - 1)  # rows traced vs candidates


def test_two_rank_comm_example_compiles_against_the_header(tmp_path):
    """CPU: the two-process RCCL example (drt_comm_* + triangle-block first-hit reduce, no torch / MPI) builds."""
    assert _build(tmp_path, COMM_SRC).exists()


@pytest.mark.gpu
def test_two_rank_comm_example_runs_when_two_gpus_are_visible(tmp_path):
    """Two processes, two GPUs, the id through a file: MIN all-reduce of the packed first-hit keys == the unsharded
    operator bit for bit.  RCCL refuses two ranks on one device, so this runs on multi-GPU nodes only (the 1-GPU boxes
    of the round: skipped; world size 1 runs in tests/test_comm.py)."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL: one rank per device)")
    exe = _build(tmp_path, COMM_SRC)
    idf = tmp_path / "rccl.id"
    env = {**os.environ, "LD_LIBRARY_PATH": f"{LIBDIR}:{os.environ.get('LD_LIBRARY_PATH', '')}",
           "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    procs = [subprocess.Popen([str(exe), str(r), "2", str(idf), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    oks = []
    for p, (so, se) in zip(procs, outs):
        ok = [ln for ln in so.splitlines() if ln.startswith("OK ")]  # (RCCL prints its banner on stdout first)
        assert p.returncode == 0 and ok, so + se
        oks.append(ok[-1])
    assert oks[0] == oks[1]  # same ray and hit counts on both ranks


@pytest.mark.gpu
def test_comm_example_world_size_1(tmp_path):
    """The same binary as ONE rank (the whole mesh is its block, the all-reduce runs through RCCL with a world of 1):
    everything but the second GPU -- rendezvous file, key kernel, collective, decode, bit comparison."""
    exe = _build(tmp_path, COMM_SRC)
    env = {**os.environ, "LD_LIBRARY_PATH": f"{LIBDIR}:{os.environ.get('LD_LIBRARY_PATH', '')}",
           "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    r = subprocess.run([str(exe), "0", "1", str(tmp_path / "rccl.id"), "0"], capture_output=True, text=True, timeout=300,
                       env=env)
    ok = [ln for ln in r.stdout.splitlines() if ln.startswith("OK 65536")]  # (RCCL prints its banner on stdout first)
    assert r.returncode == 0 and ok, r.stdout + r.stderr
    assert int(ok[-1].split()[2]) > 1000  # most rays hit something


def test_threads_example_compiles_against_the_header(tmp_path):
    assert _build(tmp_path, THREADS_SRC).exists()


@pytest.mark.gpu
def test_two_host_threads_two_streams_equal_serial(tmp_path):
    """SURVEY section 8b, re-entrancy: two host threads on two streams -- each with its own drt_mesh_t (LBVH and clusters
    built lazily inside the concurrent calls), then sharing ONE handle whose LBVH / clusters were prebuilt -- run dense
    Moller-Trumbore + the compact trace + the beam-pruned trace concurrently; every result equals the serial run bit
    for bit (tests/abi/abi_threads.cpp; the calls that write to a handle are named in include/differt_amd.h)."""
    exe = _build(tmp_path, THREADS_SRC)
    r = subprocess.run([str(exe), "4"], capture_output=True, text=True, timeout=600,
                       env={**os.environ, "LD_LIBRARY_PATH": f"{LIBDIR}:{os.environ.get('LD_LIBRARY_PATH', '')}"})
    assert r.returncode == 0 and r.stdout.startswith("OK 4 rounds"), r.stdout + r.stderr
    assert int(r.stdout.split()[3]) > 0  # the scene has valid paths


SHARDED_SRC = ROOT / "tests" / "abi" / "abi_beam_sharded.cpp"


def _write_scene(path, V, Tr, tx, rx, order, quads=False, max_paths=4096):
    import numpy as np

    with path.open("wb") as f:
        f.write(np.asarray([len(V), len(Tr), int(quads), len(tx), len(rx), order, max_paths], np.int64).tobytes())
        for a, dt in ((V, np.float32), (Tr, np.int32), (tx, np.float32), (rx, np.float32)):
            f.write(np.ascontiguousarray(a, dt).tobytes())


def test_sharded_beam_caller_compiles_against_the_header(tmp_path):
    """CPU: BASELINE configs[4] end to end behind the C ABI (prefix-sharded pruned search + count / record all-gathers + key
    sort + SUM all-reduce of grad(TX), no torch / Python / MPI in the process) builds against the header."""
    assert _build(tmp_path, SHARDED_SRC).exists()


def _run_sharded(exe, tmp_path, world, devices):
    env = {**os.environ, "LD_LIBRARY_PATH": f"{LIBDIR}:{os.environ.get('LD_LIBRARY_PATH', '')}", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    idf = tmp_path / f"rccl{world}.id"
    procs = [subprocess.Popen([str(exe), str(r), str(world), str(idf), str(tmp_path / "scene.bin"), str(devices[r])],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    oks = []
    for p, (so, se) in zip(procs, outs):
        ok = [ln for ln in so.splitlines() if ln.startswith("OK ")]  # (RCCL prints its banner on stdout first)
        assert p.returncode == 0 and ok, so + se
        oks.append([int(x) for x in ok[-1].split()[1:]])
    return oks


@pytest.mark.gpu
def test_sharded_beam_caller_configs4_world_of_one(tmp_path):
    """configs[4] itself (1 TX x 1024 RX, 200 000 triangles, order 2, fwd + grad) through tests/abi/abi_beam_sharded.cpp as ONE
    rank: search, both all-gathers and the gradient all-reduce on RCCL, union == the unsharded call bit for bit (the
    binary checks), 122 paths."""
    import synthetic_scenes as S

    V, Tr, tx, rx = S.cfg5_scene()
    _write_scene(tmp_path / "scene.bin", V, Tr, tx, rx, 2)
    exe = _build(tmp_path, SHARDED_SRC)
    (total, mine), = _run_sharded(exe, tmp_path, 1, [0])
    assert total == mine == 122


@pytest.mark.gpu
def test_sharded_beam_caller_two_ranks_when_two_gpus_are_visible(tmp_path):
    """The same binary as two processes on two GPUs (RCCL refuses two ranks per device: 1-GPU boxes skip): the shards
    partition the 122 paths, the sorted union equals the unsharded search, grad(TX) sums over the ranks."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL: one rank per device)")
    import synthetic_scenes as S

    V, Tr, tx, rx = S.cfg5_scene()
    _write_scene(tmp_path / "scene.bin", V, Tr, tx, rx, 2)
    exe = _build(tmp_path, SHARDED_SRC)
    (t0, m0), (t1, m1) = _run_sharded(exe, tmp_path, 2, [0, 1])
    assert t0 == t1 == 122 and m0 + m1 == 122 and 0 < m0 < 122
