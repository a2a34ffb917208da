"""Smoothed tracer (forward + VJP) vs the torch restatement over random scenes.
Forward: error vs the float64 restatement <= max(1e-5*|ref| + 1e-6, 8x the worst error of the float32
restatement in the same case); vertices/objects identical to the hard mode.  Gradient (every 4th
case): float64 autograd; error relative to the largest gradient entry <= max(1e-5, 16x the error of
plain float32 autograd).  python scratch/smooth_stress.py [seconds]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402
import oracle as orc  # noqa: E402
from oracle import torch_ref as tr  # noqa: E402
import synthetic_scenes as S  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(77)
EPS, TOL, MINLEN = orc.DEFAULT_EPSILON, orc.DEFAULT_HIT_TOL, orc.DEFAULT_MIN_LEN
st = {"cases": 0, "confidences": 0, "informative": 0, "nan": 0, "nan_mismatch": 0, "conf_out_of_tol": 0,
      "vertex_bits_vs_hard_mismatch": 0, "object_mismatch": 0, "grad_cases": 0,
      "grad_max_rel_err_gpu": 0.0, "grad_max_rel_err_f32_autograd": 0.0, "grad_worse_than_16x_f32": 0,
      "grad_nonfinite": 0}
t0 = time.time()
while time.time() - t0 < budget:
    if rng.random() < 0.5:
        boxes = int(rng.integers(1, 5))
        pitch = float(rng.uniform(20, 45))
        V, Tr, c, h = S.manhattan(boxes, pitch=pitch, seed=int(rng.integers(1 << 30)))
        scale = float(10 ** rng.uniform(-2, 0))  # shrink: confidences saturate in 40 m scenes
        V = (V * scale).astype(np.float32)
        tx = (rng.uniform(-1, 1, (int(rng.integers(1, 3)), 3)) * pitch * scale).astype(np.float32)
        rx = (rng.uniform(-1, 1, (int(rng.integers(1, 4)), 3)) * pitch * scale).astype(np.float32)
    else:
        V, Tr = orc.box_mesh(*rng.uniform(1, 5, 3), with_top=True)
        V = (V + rng.normal(size=V.shape) * 0.03).astype(np.float32)
        tx = rng.uniform(-1, 1, (int(rng.integers(1, 3)), 3)).astype(np.float32)
        rx = rng.uniform(-1, 1, (int(rng.integers(1, 4)), 3)).astype(np.float32)
    quads = bool(rng.random() < 0.3)
    mask = (rng.random(Tr.shape[0]) > 0.15) if rng.random() < 0.4 else None
    if mask is not None and quads:
        mask[1::2] = mask[0::2]
    order = int(rng.choice([0, 1, 2, 2, 3]))
    n = Tr.shape[0] // 2 if quads else Tr.shape[0]
    full = orc.generate_all_path_candidates(n, order)
    if full.shape[0] > 300:
        full = full[np.sort(rng.choice(full.shape[0], 300, replace=False))]
    cand = (full * (2 if quads else 1)).astype(np.int32)
    if rng.random() < 0.1 and order:
        cand[rng.integers(0, len(cand))] = -1
    sf = float(10 ** rng.uniform(-1, 3.5))
    bs = [None, 512, 7, 1][int(rng.integers(0, 4))]
    tracer = G.ExhaustivePathTracer(smoothing_factor=sf, batch_size=bs)
    do_grad = st["cases"] % 4 == 0
    vg = torch.tensor(V, device="cuda", requires_grad=do_grad)
    txg = torch.tensor(tx, device="cuda", requires_grad=do_grad)
    rxg = torch.tensor(rx, device="cuda", requires_grad=do_grad)
    scene = G.Scene(txg, rxg, G.Mesh(vg, Tr, mask=mask, assume_quads=quads))
    got = scene.trace_paths(path_candidates=cand, solver=tracer)
    hard = scene.trace_paths(path_candidates=cand)
    Trl = torch.tensor(Tr, dtype=torch.long)
    candl = torch.tensor(cand.astype(np.int64))
    maskt = None if mask is None else torch.tensor(mask)

    def ref(dtype, grad):
        ins = [torch.tensor(a, dtype=dtype, requires_grad=grad) for a in (V, tx, rx)]
        fullp, m = tr.trace_smooth(ins[0], Trl, ins[1], ins[2], candl, mask=maskt, assume_quads=quads, epsilon=EPS,
                                   hit_tol=TOL, min_len=MINLEN, smoothing_factor=sf, batch_size=bs)
        return ins, fullp, m

    _, _, m32 = ref(torch.float32, False)
    _, _, m64 = ref(torch.float64, False)
    e = m32.numpy()
    e64 = m64.numpy()
    g = got.mask.detach().cpu().numpy()
    st["cases"] += 1
    st["confidences"] += e.size
    nan = np.isnan(e)
    st["nan"] += int(nan.sum())
    st["nan_mismatch"] += int((np.isnan(g) != nan).sum())
    ok = ~nan & ~np.isnan(g) & ~np.isnan(e64)
    err = np.abs(g[ok] - e[ok])
    tol = 1e-5 * np.abs(e64[ok]) + 1e-6
    st["conf_within_1e-5_of_f32_restatement"] = st.get("conf_within_1e-5_of_f32_restatement", 0) + int((err <= tol).sum())
    err_gpu = np.abs(g[ok] - e64[ok])
    err_ref = np.abs(e[ok] - e64[ok])
    # large slopes amplify the last ulp of the path vertices (alpha * delta in the exponent): the yardstick
    # is how far the float32 restatement itself lands from float64 in the same case
    yard = 8.0 * (err_ref.max() if err_ref.size else 0.0)
    st["conf_out_of_tol"] += int((err_gpu > np.maximum(tol, yard)).sum())
    st["conf_max_err_gpu_vs_f64"] = max(st.get("conf_max_err_gpu_vs_f64", 0.0), float(err_gpu.max()) if err_gpu.size else 0.0)
    st["conf_max_err_f32_vs_f64"] = max(st.get("conf_max_err_f32_vs_f64", 0.0), float(err_ref.max()) if err_ref.size else 0.0)
    st["conf_max_abs_err_vs_f32_restatement"] = max(st.get("conf_max_abs_err_vs_f32_restatement", 0.0), float(err.max()) if err.size else 0.0)
    st["informative"] += int(((e[ok] > 1e-3) & (e[ok] < 1 - 1e-3)).sum())
    st["vertex_bits_vs_hard_mismatch"] += int(
        (got.vertices.detach().cpu().numpy().view(np.uint32) != hard.vertices.detach().cpu().numpy().view(np.uint32)).sum())
    st["object_mismatch"] += int((got.objects.cpu().numpy() != hard.objects.cpu().numpy()).sum())
    if do_grad:
        w = rng.normal(size=e.shape)
        w[nan] = 0.0
        wv = rng.normal(size=(*e.shape, order + 2, 3)) * 1e-2

        def loss(fullp, m, dtype, dev):
            mm = torch.where(torch.isnan(m), torch.zeros_like(m), m)
            return (mm * torch.tensor(w, dtype=dtype, device=dev)).sum() + (fullp * torch.tensor(wv, dtype=dtype, device=dev)).sum()

        grads = {}
        for name, dtype in (("f64", torch.float64), ("f32", torch.float32)):
            ins, fullp, m = ref(dtype, True)
            loss(fullp, m, dtype, "cpu").backward()
            grads[name] = [t.grad.numpy().astype(np.float64) for t in ins]
        loss(got.vertices, got.mask, torch.float32, "cuda").backward()
        gg = [t.grad.cpu().numpy().astype(np.float64) for t in (vg, txg, rxg)]
        if not all(np.isfinite(x).all() for x in grads["f64"]):
            continue  # the restatement itself went non-finite (NaN confidences leak through torch.where)
        st["grad_cases"] += 1
        for a, b, c in zip(gg, grads["f32"], grads["f64"]):
            scale_ = np.abs(c).max() + 1e-30
            eg, er = np.abs(a - c).max() / scale_, np.abs(b - c).max() / scale_
            st["grad_nonfinite"] += int(not np.isfinite(a).all())
            st["grad_max_rel_err_gpu"] = max(st["grad_max_rel_err_gpu"], float(eg))
            st["grad_max_rel_err_f32_autograd"] = max(st["grad_max_rel_err_f32_autograd"], float(er))
            if scale_ < 1e-20:
                continue
            st["grad_comparisons"] = st.get("grad_comparisons", 0) + 1
            st["grad_gpu_within_1e-5"] = st.get("grad_gpu_within_1e-5", 0) + int(eg <= 1e-5)
            st["grad_f32_autograd_within_1e-5"] = st.get("grad_f32_autograd_within_1e-5", 0) + int(er <= 1e-5)
            st["grad_worse_than_16x_f32"] += int(eg > max(1e-5, 16 * er))
            if eg > max(1e-5, 16 * er) and st.setdefault("gdiag", 0) < 15:
                st["gdiag"] += 1
                i = int(np.argmax(np.abs(a - c)))
                print("GDIAG sf=%.3g order=%d quads=%d bs=%s mask=%d shape=%s eg=%.3g er=%.3g  gpu=%.6g f32=%.6g f64=%.6g scale=%.3g" % (
                    sf, order, quads, bs, mask is not None, a.shape, eg, er, a.reshape(-1)[i], b.reshape(-1)[i], c.reshape(-1)[i], scale_), flush=True)
st["seconds"] = time.time() - t0
print(json.dumps(st))
