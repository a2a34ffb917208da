"""How many triangles of the reference's real meshes can be paired into coplanar convex quads, under which criterion?
(CPU study for the pairing pass of csrc/beam.hip; float32 arithmetic as mesh_prepare_kernel: no FMA.)"""
import sys
from collections import defaultdict
from pathlib import Path
import numpy as np

G = Path(__file__).resolve().parents[1] / "tests" / "golden"
f32 = np.float32

def normals(tv):
    a = tv[:, 1] - tv[:, 0]; b = tv[:, 2] - tv[:, 1]
    c = np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], 1).astype(f32)
    l = np.sqrt(((c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]).astype(f32) + c[:, 2] * c[:, 2]).astype(f32)).astype(f32)
    den = np.where(l == 0, f32(1), l)
    return (c / den[:, None]).astype(f32), l

for name in ("bruxelles", "manhattan", "manhattan_small"):
    d = np.load(G / f"{name}.npz")
    V, T = d["vertices"], d["triangles"]
    tv = V[T]
    n, area2 = normals(tv)
    edges = defaultdict(list)
    for t, (a, b, c) in enumerate(T):
        for e in ((a, b), (b, c), (c, a)):
            edges[(min(e), max(e))].append(t)
    adj = [(ts[0], ts[1]) for ts in edges.values() if len(ts) == 2]
    multi = sum(1 for ts in edges.values() if len(ts) > 2)
    print(f"{name}: {len(T)} triangles, {len(V)} vertices, {len(adj)} interior edges, {multi} edges with >2 triangles, degenerate {int((area2 == 0).sum())}")
    # vertex positions may be duplicated (same coordinates, different index): adjacency by coordinates as well
    key = {}
    Vk = np.array([key.setdefault(tuple(v), len(key)) for v in V])
    Tk = Vk[T]
    edges2 = defaultdict(list)
    for t, (a, b, c) in enumerate(Tk):
        for e in ((a, b), (b, c), (c, a)):
            edges2[(min(e), max(e))].append(t)
    adj2 = [(ts[0], ts[1]) for ts in edges2.values() if len(ts) == 2]
    print(f"   unique positions {len(key)}; interior edges by position {len(adj2)}")
    cls = defaultdict(int)
    cand = []
    for a, b in adj2:
        eqn = bool(np.all(n[a] == n[b]))
        opp = bool(np.all(n[a] == -n[b]))
        dotn = float(np.dot(n[a].astype(np.float64), n[b].astype(np.float64)))
        # max distance of b's vertices from a's plane (float64)
        dist = np.abs((tv[b].astype(np.float64) - tv[a][0].astype(np.float64)) @ n[a].astype(np.float64)).max()
        samev0 = bool(np.all(tv[a][0] == tv[b][0]))
        if eqn: cls["normals =="] += 1
        if eqn and samev0: cls["normals == and v0 =="] += 1
        if opp: cls["normals opposite"] += 1
        if dotn > 1 - 1e-6 and dist < 1e-3: cls["coplanar 1mm same orientation"] += 1
        if dotn > 1 - 1e-6 and dist < 1e-3 and not eqn: cls["coplanar but normals !="] += 1
        if eqn: cand.append((a, b, samev0))
    print("   ", dict(cls))
    # greedy maximal matching among == normal adjacent pairs whose union is a convex quad
    def quad_of(a, b):
        sa, sb = list(Tk[a]), list(Tk[b])
        shared = [v for v in sa if v in sb]
        if len(shared) != 2: return None
        # rotate a so that the shared edge is (a[i], a[i+1]) in a's orientation
        for i in range(3):
            if sa[i] in shared and sa[(i + 1) % 3] in shared:
                p, q, r = sa[i], sa[(i + 1) % 3], sa[(i + 2) % 3]
                break
        o = [v for v in sb if v not in shared][0]
        # b must traverse the shared edge in the opposite direction (q -> p) for consistent orientation
        ok = any(sb[j] == q and sb[(j + 1) % 3] == p for j in range(3))
        if not ok: return None
        return (r, p, o, q)  # quad r -> p -> o -> q, counter-clockwise w.r.t. a's normal
    pos = {v: k for k, v in key.items()}
    def convex(quad, nrm):
        P = [np.array(pos[v], np.float64) for v in quad]
        for k in range(4):
            e0 = P[(k + 1) % 4] - P[k]; e1 = P[(k + 2) % 4] - P[(k + 1) % 4]
            turn = np.dot(np.cross(e0, e1), nrm)
            if not turn > 1e-3 * np.sqrt(np.dot(e0, e0) * np.dot(e1, e1)): return False
        return True
    used = np.zeros(len(T), bool)
    npairs = npairs_v0 = 0
    ok_pairs = []
    for a, b, samev0 in cand:
        q = quad_of(a, b)
        if q is None or not convex(q, n[a].astype(np.float64)): continue
        ok_pairs.append((a, b, samev0))
    # prefer same-v0 pairs first
    for a, b, samev0 in sorted(ok_pairs, key=lambda x: not x[2]):
        if used[a] or used[b]: continue
        used[a] = used[b] = True
        npairs += 1; npairs_v0 += samev0
    print(f"    convex == normal pairs available {len(ok_pairs)}; greedy matching: {npairs} pairs ({npairs_v0} with equal first vertex) -> {len(T) - npairs} primitives")
    cons = sum(1 for a, b, _ in ok_pairs if abs(a - b) == 1 and min(a, b) % 2 == 0)
    print(f"    of them consecutive (2i, 2i+1): {cons}")
