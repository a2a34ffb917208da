"""Synthetic inputs of the BASELINE configs, NumPy `default_rng` seeded.

Shared by bench.py / bench_paths.py and the tests so that CPU oracle and GPU see identical bytes.
"""

from __future__ import annotations

import numpy as np

_BOX_TRIS_TOP_NO_BOTTOM = np.array(
    # Mesh.box(with_top=True, with_bottom=False): reference _mesh.py:2186-2208 triangle table
    [[0, 1, 2], [0, 2, 3], [3, 2, 4], [3, 4, 5], [5, 4, 6], [5, 6, 7], [7, 6, 1], [7, 1, 0],
     [0, 3, 5], [0, 5, 7]],
    dtype=np.int32,
)


def _box_vertices(length, width, height):
    f = np.float32
    dx = np.array([f(length) * f(0.5), 0, 0], dtype=f)
    dy = np.array([0, f(width) * f(0.5), 0], dtype=f)
    dz = np.array([0, 0, f(height) * f(0.5)], dtype=f)
    return np.stack((+dx + dy + dz, +dx + dy - dz, -dx + dy - dz, -dx + dy + dz,
                     -dx - dy - dz, -dx - dy + dz, +dx - dy - dz, +dx - dy + dz)).astype(f)


def manhattan(num_boxes: int = 1000, pitch: float = 40.0, seed: int = 1234):
    """cfg3/cfg4 (num_boxes=1000 -> 10 000 triangles) and cfg5 (num_boxes=20000 -> 200 000):
    boxes on a `pitch`-metre grid, footprint U(10,30) m, height U(10,80) m, 10 triangles each
    (walls + roof, no floor), quads = consecutive triangle pairs."""
    rng = np.random.default_rng(seed)
    nx = int(np.ceil(np.sqrt(num_boxes)))
    verts = np.empty((num_boxes, 8, 3), np.float32)
    tris = np.empty((num_boxes, 10, 3), np.int32)
    heights = np.empty(num_boxes, np.float32)
    centres = np.empty((num_boxes, 2), np.float32)
    for b in range(num_boxes):
        ix, iy = b % nx, b // nx
        l, w = rng.uniform(10, 30, 2)
        h = rng.uniform(10, 80)
        cx, cy = (ix - nx / 2 + 0.5) * pitch, (iy - nx / 2 + 0.5) * pitch
        verts[b] = _box_vertices(l, w, h) + np.array([cx, cy, h / 2], np.float32)
        tris[b] = _BOX_TRIS_TOP_NO_BOTTOM + 8 * b
        heights[b] = h
        centres[b] = (cx, cy)
    return verts.reshape(-1, 3), tris.reshape(-1, 3), centres, heights


def manhattan_tx_rx(centres, heights, num_tx: int, num_rx: int, pitch: float = 40.0, seed: int = 99):
    """TX 5 m above randomly chosen roofs, RX at 1.5 m on street crossings (between boxes)."""
    rng = np.random.default_rng(seed)
    sel = rng.choice(len(heights), num_tx, replace=False)
    tx = np.column_stack((centres[sel], heights[sel] + 5.0)).astype(np.float32)
    lo, hi = centres.min(axis=0), centres.max(axis=0)
    nx = int(round((hi[0] - lo[0]) / pitch))
    ny = int(round((hi[1] - lo[1]) / pitch))
    gx = rng.integers(0, max(nx, 1), num_rx)
    gy = rng.integers(0, max(ny, 1), num_rx)
    jitter = rng.uniform(-3, 3, (num_rx, 2))
    rx = np.column_stack((lo[0] + (gx + 0.5) * pitch + jitter[:, 0], lo[1] + (gy + 0.5) * pitch + jitter[:, 1],
                          np.full(num_rx, 1.5))).astype(np.float32)
    return tx, rx


def cfg5_scene(num_boxes: int = 20000, rx_side: int = 32, pitch: float = 40.0):
    """BASELINE configs[4]: one transmitter on a mast above the middle of a `num_boxes`-box Manhattan
    city (20 000 boxes = 200 000 triangles), receivers on a `rx_side` x `rx_side` street-level grid
    (32 x 32 = 1024).  Returns (vertices, triangles, tx[1,3], rx[rx_side^2,3])."""
    V, Tr, centres, heights = manhattan(num_boxes, pitch)
    c0 = centres.mean(axis=0)
    tx = np.array([[c0[0], c0[1], heights.max() + 10.0]], dtype=np.float32)
    g = (np.arange(rx_side) - (rx_side - 1) / 2) * pitch
    rx = np.stack(np.meshgrid(c0[0] + g + pitch / 2, c0[1] + g + pitch / 2, indexing="ij"), -1).reshape(-1, 2)
    rx = np.column_stack((rx, np.full(len(rx), 1.5))).astype(np.float32)
    return V, Tr, tx, rx


def load_real_mesh(name: str):
    """The reference's in-tree meshes as committed geometry arrays (tests/golden/<name>.npz, written by
    tests/golden/make_golden.py from docs/source/notebooks/<name>.obj with the reader rule of
    differt-core/src/geometry/mesh.rs:399-429): vertices f32[V,3], triangles i32[T,3]."""
    from pathlib import Path

    d = np.load(Path(__file__).resolve().parent / "tests" / "golden" / f"{name}.npz")
    return d["vertices"].astype(np.float32), d["triangles"].astype(np.int32)


def outdoor_end_points(G, V, Tr, num_tx: int, num_rx: int, seed: int = 7):
    """End points for a real city mesh (no street grid to place them on): transmitters 5 m above the highest roof and
    at 40 % of that height, receivers at 1.5 m, all in the open -- random points of the central 90 % of the footprint
    whose upward ray hits nothing (`G` = differt_amd.geometry: the first-hit query runs on the GPU)."""
    rng = np.random.default_rng(seed)
    lo, hi = V.min(0), V.max(0)
    c, e = (lo + hi) / 2, (hi - lo) / 2
    mesh = G.Mesh(V, Tr)

    def outdoor(n, z):
        out = []
        while len(out) < n:
            p = np.concatenate([c[:2] + rng.uniform(-0.45, 0.45, (256, 2)) * 2 * e[:2], np.full((256, 1), z)], 1).astype(np.float32)
            up = np.tile(np.array([[0, 0, 1]], np.float32), (256, 1))
            idx, _ = mesh.first_triangle_hit_by_ray(p, up)
            out.extend(p[idx.cpu().numpy() < 0].tolist())
        return np.asarray(out[:n], np.float32)

    ntop = max(num_tx // 2, 1)
    tx = np.concatenate([outdoor(ntop, float(hi[2]) + 5.0), outdoor(num_tx - ntop, 0.4 * float(hi[2]))])[:num_tx]
    return tx, outdoor(num_rx, 1.5)


def random_rotation(rng, max_tilt_deg: float = 10.0) -> np.ndarray:
    """A rotation (float64 3x3): any yaw about z, then a tilt of at most `max_tilt_deg` about a horizontal axis --
    walls are no longer axis-aligned, roofs no longer horizontal: dot products keep all three terms, images are
    inexact, association order and contraction become visible (every scene-level parity case so far was
    axis-aligned)."""
    yaw, tilt, azim = rng.uniform(0, 2 * np.pi), np.deg2rad(rng.uniform(0, max_tilt_deg)), rng.uniform(0, 2 * np.pi)
    cz, sz = np.cos(yaw), np.sin(yaw)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]])
    a = np.array([np.cos(azim), np.sin(azim), 0.0])  # tilt axis (Rodrigues)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    Rt = np.eye(3) + np.sin(tilt) * K + (1 - np.cos(tilt)) * (K @ K)
    return Rt @ Rz


def rotate_points(R: np.ndarray, *arrays):
    """Apply R (float64) to float32 point arrays, rounding the result to float32 (the rotated scene is a NEW float32 scene)."""
    return tuple((np.asarray(a, np.float64) @ R.T).astype(np.float32) for a in arrays)
