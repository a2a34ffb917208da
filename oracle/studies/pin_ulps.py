"""Exploration: which float32 contraction pattern reproduces the reference's golden path vertices
(test_scene.py:116-160, exact float32 literals) bit for bit?  Exact arithmetic via Fractions."""
import itertools, json, sys
from fractions import Fraction as Fr
import numpy as np

f32 = np.float32

def rn(fr):
    """round an exact Fraction to nearest float32, ties to even"""
    d = float(fr)
    r = f32(d)
    if not np.isfinite(r):
        return r
    best = None
    for c in (np.nextafter(r, f32(-np.inf)), r, np.nextafter(r, f32(np.inf))):
        err = abs(Fr(float(c)) - fr)
        even = (int(np.array(c, f32).view(np.uint32)) & 1) == 0
        key = (err, 0 if even else 1)
        if best is None or key < best[0]:
            best = (key, c)
    return f32(best[1])

F = lambda x: Fr(float(x))
mul = lambda a, b: rn(F(a) * F(b))
add = lambda a, b: rn(F(a) + F(b))
sub = lambda a, b: rn(F(a) - F(b))
fma = lambda a, b, c: rn(F(a) * F(b) + F(c))
div = lambda a, b: rn(F(a) / F(b))
def sqrt32(a):
    # correctly rounded sqrt via float64 sqrt (53 bits >> 2*24+2: safe)
    return f32(np.sqrt(np.float64(a)))

def dot(x, y, v):
    p = [mul(x[i], y[i]) for i in range(3)]
    if v == 0: return add(add(p[0], p[1]), p[2])
    if v == 1: return fma(x[2], y[2], fma(x[1], y[1], p[0]))
    if v == 2: return fma(x[2], y[2], fma(x[0], y[0], p[1]))
    if v == 3: return fma(x[2], y[2], add(p[0], p[1]))
    if v == 4: return add(fma(x[0], y[0], p[1]), p[2])
    if v == 5: return fma(x[0], y[0], fma(x[1], y[1], p[2]))
    if v == 6: return add(p[0], add(p[1], p[2]))
    if v == 7: return add(fma(x[1], y[1], p[0]), p[2])
    if v == 8: return fma(x[1], y[1], fma(x[2], y[2], p[0]))
    if v == 9: return add(add(p[0], p[2]), p[1])
NDOT = 10

def cross(a, b, v):
    idx = [(1, 2), (2, 0), (0, 1)]
    out = []
    for (i, j) in idx:
        if v == 0: out.append(sub(mul(a[i], b[j]), mul(a[j], b[i])))
        if v == 1: out.append(fma(a[i], b[j], -mul(a[j], b[i])))
        if v == 2: out.append(fma(-a[j], b[i], mul(a[i], b[j])))
    return out

def normal(v0, v1, v2, cfg):
    a = [sub(v1[i], v0[i]) for i in range(3)]
    b = [sub(v2[i], v1[i]) for i in range(3)] if cfg.get("nform", 0) == 0 else [sub(v2[i], v0[i]) for i in range(3)]
    c = cross(a, b, cfg["cross"])
    l = sqrt32(dot(c, c, cfg["ndot"]))
    if l == 0: l = f32(1)
    if cfg["ndiv"] == 0: return [div(ci, l) for ci in c]
    r = div(f32(1), l)
    return [mul(ci, r) for ci in c]

def image(x, p, n, cfg):
    inc = [sub(x[i], p[i]) for i in range(3)]
    c = mul(f32(2), dot(inc, n, cfg["dot"]))
    if cfg["img"] == 0: return [sub(x[i], mul(c, n[i])) for i in range(3)]
    return [fma(-c, n[i], x[i]) for i in range(3)]

def rayplane(o, d, p, n, cfg):
    v = [sub(p[i], o[i]) for i in range(3)]
    un = dot(d, n, cfg["dot"]); vn = dot(v, n, cfg["dot"])
    t = div(vn, un) if cfg.get("rpdiv", 0) == 0 else mul(vn, div(f32(1), un))
    if cfg["rp"] == 0: return [add(o[i], mul(d[i], t)) for i in range(3)]
    return [fma(d[i], t, o[i]) for i in range(3)]

def chain(tx, rx, tris, V, Tr, cfg, ncache):
    ps, ns = [], []
    for t in tris:
        v0, v1, v2 = (V[Tr[t][k]] for k in range(3))
        ps.append((v0, v1, v2)[cfg.get("pv", 0)])
        key = (t, cfg["cross"], cfg["ndot"], cfg["ndiv"], cfg.get("nform", 0))
        if key not in ncache: ncache[key] = normal(v0, v1, v2, cfg)
        ns.append(ncache[key])
    imgs, prev = [], tx
    for p, n in zip(ps, ns):
        prev = image(prev, p, n, cfg); imgs.append(prev)
    out, cur = [None] * len(tris), rx
    for j in reversed(range(len(tris))):
        d = [sub(imgs[j][i], cur[i]) for i in range(3)]
        cur = rayplane(cur, d, ps[j], ns[j], cfg); out[j] = cur
    return out

g = json.load(open('tests/golden/reference_goldens.json'))["advanced_path_tracing_example"]
tb = json.load(open('tests/golden/two_buildings.json'))
V = np.asarray(tb["vertices"], f32); Tr = np.asarray(tb["triangles"], np.int32)
tx = np.asarray(g["tx"], f32); rx = np.asarray(g["rx"], f32)
cases = []
for order in (1, 2, 3, 4):
    e = g["orders"][str(order)]
    cases.append((e["objects"][0][1:-1], np.asarray(e["path_vertices"], f32).reshape(order, 3)))
results = []
ncache = {}
for dotv, ndot, crs, ndiv, img, rp, nform, pv, rpdiv in itertools.product(range(NDOT), (0, 2), range(3), range(2), range(2), range(2), range(2), range(3), range(2)):
    cfg = dict(dot=dotv, ndot=ndot, cross=crs, ndiv=ndiv, img=img, rp=rp, nform=nform, pv=pv, rpdiv=rpdiv)
    score, tot = 0, 0
    for tris, exp in cases:
        got = np.asarray(chain(tx, rx, tris, V, Tr, cfg, ncache), f32)
        score += int((got.view(np.uint32) == exp.view(np.uint32)).sum()); tot += exp.size
    results.append((score, cfg))
results.sort(key=lambda r: -r[0])
for s, c in results[:25]:
    print(s, "/", tot, c)
print("baseline no-FMA:", [r for r in results if all(v == 0 for v in r[1].values())][0])

