"""Per-path EM channel kernel (drt_paths_channel) vs oracle/em_ref.py over random scenes, materials,
slab thicknesses, frequencies and polarisations.  Lengths / delays must be bit-identical; the complex
coefficient within max(2e-5, 8x the float32 restatement's own error vs float64) of the path's amplitude
scale |a| + 1e-2 lambda/(4 pi s); angles within 1e-3 deg.
python scratch/em_stress.py [seconds]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402
import oracle as orc  # noqa: E402
from differt_amd.plugins import deepmimo  # noqa: E402
from oracle import em_ref as emo  # noqa: E402
import synthetic_scenes as S  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(5)
st = {"cases": 0, "paths": 0, "valid_paths": 0, "length_bits_mismatch": 0, "coeff_out_of_tol": 0,
      "coeff_max_scaled_err": 0.0, "angle_out_of_tol": 0, "power_out_of_tol": 0, "nonfinite_mismatch": 0}
t0 = time.time()
while time.time() - t0 < budget:
    if rng.random() < 0.5:
        boxes = int(rng.integers(1, 6))
        pitch = float(rng.uniform(20, 45))
        V, Tr, c, h = S.manhattan(boxes, pitch=pitch, seed=int(rng.integers(1 << 30)))
        tx = (rng.uniform(-1, 1, (int(rng.integers(1, 3)), 3)) * pitch).astype(np.float32)
        rx = (rng.uniform(-1, 1, (int(rng.integers(1, 4)), 3)) * pitch).astype(np.float32)
        tx[:, 2] = np.abs(tx[:, 2]) + 1
        rx[:, 2] = np.abs(rx[:, 2]) + 1
    else:
        V, Tr = orc.box_mesh(*rng.uniform(2, 30, 3), with_top=True)
        tx = (rng.uniform(-0.9, 0.9, (int(rng.integers(1, 3)), 3))).astype(np.float32)
        rx = (rng.uniform(-0.9, 0.9, (int(rng.integers(1, 4)), 3))).astype(np.float32)
    T = Tr.shape[0]
    order = int(rng.choice([0, 1, 2, 2, 3]))
    full = orc.generate_all_path_candidates(T, order)
    if full.shape[0] > 2000:
        full = full[np.sort(rng.choice(full.shape[0], 2000, replace=False))]
    M = int(rng.integers(1, 5))
    fm = rng.integers(0, M, T).astype(np.int32)
    f = float(10 ** rng.uniform(8, 10.5))
    eta = rng.uniform(1.0, 9.0, M).astype(np.float32)
    sig = (10 ** rng.uniform(-4, 0.5, M)).astype(np.float32) * (rng.random(M) > 0.2)
    n_c = emo.complex_refractive_index(eta, sig, f)
    th = np.where(rng.random(M) < 0.5, -1.0, 10 ** rng.uniform(-3, 0, M)).astype(np.float32)

    def pol():
        k = rng.random()
        if k < 0.3:
            return "V"
        if k < 0.6:
            return "H"
        return tuple(float(x) for x in emo.normalize(rng.normal(size=3))[0])

    polarization = (pol(), pol()) if rng.random() < 0.5 else pol()
    scene = G.Scene(tx, rx, G.Mesh(V, Tr).set_face_materials(fm))
    paths = scene.trace_paths(path_candidates=full.astype(np.int32))
    got = deepmimo.paths_channel(paths, scene.mesh, np.stack((n_c.real, n_c.imag), -1), th, f, polarization)
    v, o = paths.vertices.cpu().numpy(), paths.objects.cpu().numpy()
    nr = orc.mesh_normals(orc.triangle_vertices(V, Tr))
    with np.errstate(all="ignore"):
        exp = emo.channel(v, o, nr, fm, n_c, th, f, polarization)
        emo.F, emo.C64 = np.float64, np.complex128  # the same restatement evaluated in float64
        try:
            n64 = emo.complex_refractive_index(eta.astype(np.float64), sig.astype(np.float64), f)
            e64 = emo.channel(v.astype(np.float64), o, nr.astype(np.float64), fm, n64, th.astype(np.float64), f,
                              polarization)
        finally:
            emo.F, emo.C64 = np.float32, np.complex64
    valid = paths.mask.cpu().numpy()
    st["cases"] += 1
    st["paths"] += valid.size
    st["valid_paths"] += int(valid.sum())
    if not valid.any():
        continue
    g = {k: t.cpu().numpy() for k, t in got.items()}
    st["length_bits_mismatch"] += int((g["length"].view(np.uint32) != exp["length"].view(np.uint32))[valid].sum())
    st["length_bits_mismatch"] += int((g["delay"].view(np.uint32) != exp["delay"].view(np.uint32))[valid].sum())
    fin_g, fin_e = np.isfinite(g["a"]), np.isfinite(exp["a"])
    st["nonfinite_mismatch"] += int((fin_g != fin_e)[valid].sum())
    ok = valid & fin_g & fin_e
    friis = (emo.c / f) / (4 * np.pi * np.maximum(exp["length"], 1e-9))
    scale = np.abs(exp["a"]) + 1e-2 * friis
    err = np.abs(g["a"] - exp["a"]) / scale
    st["coeff_within_2e-5_of_f32_oracle"] = st.get("coeff_within_2e-5_of_f32_oracle", 0) + int((err[ok] <= 2e-5).sum())
    st["coeff_max_scaled_err"] = max(st["coeff_max_scaled_err"], float(err[ok].max()) if ok.any() else 0.0)
    # slab resonances and grazing incidence are ill-conditioned: the yardstick is how far the float32
    # restatement itself is from its float64 evaluation on the same path
    ok &= np.isfinite(e64["a"])
    err_gpu = np.abs(g["a"] - e64["a"]) / scale
    err_ref = np.abs(exp["a"] - e64["a"]) / scale
    st["coeff_out_of_tol"] += int((err_gpu[ok] > np.maximum(2e-5, 8 * err_ref[ok])).sum())
    st["coeff_max_err_gpu_vs_f64"] = max(st.get("coeff_max_err_gpu_vs_f64", 0.0), float(err_gpu[ok].max()) if ok.any() else 0.0)
    st["coeff_max_err_f32_oracle_vs_f64"] = max(st.get("coeff_max_err_f32_oracle_vs_f64", 0.0), float(err_ref[ok].max()) if ok.any() else 0.0)
    co = ok & (np.abs(exp["a"]) > 1e-2 * friis) & (err_ref <= 2e-6)  # well-conditioned, co-polar sized
    st["power_out_of_tol"] += int((np.abs(g["power"] - exp["power"])[co] > 1e-5 * np.abs(exp["power"][co]) + 2e-4).sum())
    for k in ("aoa_az", "aoa_el", "aod_az", "aod_el"):
        d = (g[k] - exp[k] + 180.0) % 360.0 - 180.0
        st["angle_out_of_tol"] += int((np.abs(d)[valid] > 1e-3).sum())
st["seconds"] = time.time() - t0
print(json.dumps(st))
