"""Numeric core of ``differt.plugins.deepmimo.export`` (plugins/deepmimo.py:407-728) on the GPU.

For every path of one or several :class:`TracedPaths` (all candidates, valid or not -- the mask is
exported next to them, as in the reference) one HIP kernel (``drt_paths_channel``, csrc/em.hip)
computes the complex channel coefficient and the DeepMIMO quantities: power [dBW], phase [deg],
delay [s], angles of arrival / departure [deg].  Same assumptions as the reference: far field,
isotropic antennas, every interaction a specular reflection.  Differentiable in the path vertices
(``drt_paths_channel_vjp``: forward-mode duals per path), hence -- through the tracer's VJP -- in the
transmitters, receivers and mesh vertices; mesh normals and material constants enter as constants.
"""

from __future__ import annotations

import ctypes as C
from collections.abc import Iterable, Mapping
from dataclasses import asdict, dataclass
from typing import Any

import numpy as np
import torch

from .. import _lib
from .._tensors import as_f32, device, ptr, stream
from ..em._material import Material, materials
from ..geometry._paths import TracedPaths

__all__ = ["DeepMIMO", "export"]

_NO_INTERACTION = -1  # plugins/deepmimo.py:494


@dataclass
class DeepMIMO:
    """Field set of the reference's ``DeepMIMO`` container (plugins/deepmimo.py:63-331); arrays are
    ``[num_tx, num_rx, num_paths, ...]``."""

    power: Any
    phase: Any
    delay: Any
    aoa_az: Any
    aoa_el: Any
    aod_az: Any
    aod_el: Any
    inter: Any
    inter_pos: Any
    rx_pos: Any
    tx_pos: Any
    mask: Any
    primitives: Any = None

    @property
    def num_tx(self) -> int:
        return self.tx_pos.shape[0]

    @property
    def num_rx(self) -> int:
        return self.rx_pos.shape[0]

    @property
    def num_paths(self) -> int:
        return self.power.shape[-1]

    def asdict(self) -> dict:
        return asdict(self)

    def numpy(self) -> "DeepMIMO":
        return DeepMIMO(**{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v)
                           for k, v in asdict(self).items()})


def _pad_cat(left: torch.Tensor, right: torch.Tensor, fill) -> torch.Tensor:
    """plugins/deepmimo.py:50-60: pad the interaction axis (3) to the longer one, concatenate paths (2)."""
    n = max(left.shape[3], right.shape[3])

    def pad(x):
        if x.shape[3] == n:
            return x
        shape = list(x.shape)
        shape[3] = n - x.shape[3]
        return torch.cat((x, torch.full(shape, fill, dtype=x.dtype, device=x.device)), dim=3)

    return torch.cat((pad(left), pad(right)), dim=2)


def _polarization(p) -> tuple[int, tuple[float, float, float]]:
    if isinstance(p, str):
        if p not in ("V", "H"):
            raise ValueError(f"Unknown polarization {p!r}: expected 'V', 'H' or a 3D vector")
        return (0 if p == "V" else 1), (0.0, 0.0, 0.0)
    v = np.asarray(p.detach().cpu() if isinstance(p, torch.Tensor) else p, dtype=np.float32).reshape(3)
    return 2, (float(v[0]), float(v[1]), float(v[2]))


def material_tables(material_names, radio_materials: Mapping[str, Material], frequency: float):
    """Host side of plugins/deepmimo.py:461-482: per material complex refractive index
    ``sqrt(eta_r - j sigma / (omega eps0))`` (C ABI, host) and thickness (``-1`` = half space)."""
    eta = np.array([radio_materials[m].relative_permittivity(frequency) for m in material_names], np.float32)
    sig = np.array([radio_materials[m].conductivity(frequency) for m in material_names], np.float32)
    th = np.array([radio_materials[m].thickness if radio_materials[m].thickness is not None else -1.0
                   for m in material_names], np.float32)
    n = np.zeros((len(material_names), 2), np.float32)
    if len(material_names):
        _lib.call("drt_complex_refractive_index", eta.ctypes.data_as(C.c_void_p), sig.ctypes.data_as(C.c_void_p),
                  len(material_names), float(frequency), n.ctypes.data_as(C.c_void_p))
    return n, th


_CHANNEL_OUTPUTS = ("a_re", "a_im", "power", "phase", "length", "delay", "aoa_az", "aoa_el", "aod_az", "aod_el")


class _PathsChannelFn(torch.autograd.Function):
    """``drt_paths_channel`` / ``drt_paths_channel_vjp``: the ten per-path outputs as one ``[N, 10]`` tensor,
    differentiable in the path vertices (which carry the dependence on transmitters, receivers and mesh
    vertices through the tracer's own VJP); normals and material constants are constants."""

    @staticmethod
    def forward(ctx, v, o, tabs, pr, order):
        normals, fm, nc, th, T, M = tabs
        N = v.shape[0]
        out = torch.empty((10, N), dtype=torch.float32, device=v.device)
        a = torch.empty((N, 2), dtype=torch.float32, device=v.device)
        if N:
            _lib.call("drt_paths_channel", ptr(v), ptr(o), N, order, ptr(normals), ptr(fm), T, ptr(nc), ptr(th), M,
                      C.byref(pr), ptr(a), *(ptr(out[k]) for k in range(2, 10)), stream())
        res = torch.cat((a, out[2:].t()), dim=1)  # [N, 10] in _CHANNEL_OUTPUTS order
        ctx.save_for_backward(v, o)
        ctx.cfg = (tabs, pr, order)
        return res

    @staticmethod
    def backward(ctx, g):
        v, o = ctx.saved_tensors
        (normals, fm, nc, th, T, M), pr, order = ctx.cfg
        gv = torch.zeros_like(v)
        if v.shape[0]:
            _lib.call("drt_paths_channel_vjp", ptr(v), ptr(o), v.shape[0], order, ptr(normals), ptr(fm), T, ptr(nc),
                      ptr(th), M, C.byref(pr), ptr(g.contiguous()), ptr(gv), stream())
        return gv, None, None, None, None


def paths_channel(paths: TracedPaths, mesh, n_complex, thickness, frequency: float, polarization="V") -> dict:
    """Per-path channel quantities (``drt_paths_channel``) for ONE :class:`TracedPaths` of any batch
    shape: dict of ``a`` (complex64), ``power``, ``phase``, ``length``, ``delay``, ``aoa_az``,
    ``aoa_el``, ``aod_az``, ``aod_el``, each ``[*batch]``; differentiable in ``paths.vertices``."""
    dev = device()
    tx_pol, rx_pol = polarization if isinstance(polarization, tuple) and len(polarization) == 2 else (polarization,) * 2
    pr = _lib.EmParams()
    pr.frequency = float(frequency)
    pr.tx_polarization, txv = _polarization(tx_pol)
    pr.rx_polarization, rxv = _polarization(rx_pol)
    pr.tx_vector[:] = txv
    pr.rx_vector[:] = rxv
    batch, order = tuple(paths.objects.shape[:-1]), paths.order
    N = int(np.prod(batch, dtype=np.int64))
    v = as_f32(paths.vertices, dev).reshape(N, order + 2, 3).contiguous()
    o = paths.objects.to(device=dev, dtype=torch.int32).reshape(N, order + 2).contiguous()
    normals = fm = nc = th = None
    T = M = 0
    if order > 0:
        normals = mesh.normals.contiguous()
        fm = mesh.face_materials.to(device=dev, dtype=torch.int32).contiguous()
        nc = torch.as_tensor(np.ascontiguousarray(n_complex, dtype=np.float32), device=dev)
        th = torch.as_tensor(np.ascontiguousarray(thickness, dtype=np.float32), device=dev)
        T, M = mesh.num_triangles, nc.shape[0]
        # Mesh.append() writes -1 for sub-meshes without materials and set_face_materials() does no bounds
        # check: the kernel would alias such faces to material 0, so refuse them here (one reduction)
        if fm.numel() and (fm.shape[0] != T or int(fm.min()) < 0 or int(fm.max()) >= M):
            raise ValueError(f"face_materials must hold one index in [0, {M}) per triangle "
                             f"(got {fm.shape[0]} entries in [{int(fm.min())}, {int(fm.max())}] for {T} triangles)")
    res = _PathsChannelFn.apply(v, o, (normals, fm, nc, th, T, M), pr, order)
    out = {k: res[:, i].reshape(batch) for i, k in enumerate(_CHANNEL_OUTPUTS) if i >= 2}
    out["a"] = torch.view_as_complex(res[:, :2].contiguous()).reshape(batch)
    return out


def export(*, paths: TracedPaths | Iterable[TracedPaths], scene, radio_materials: Mapping[str, Material] | None = None,
           frequency: float, include_primitives: bool = False, polarization="V") -> DeepMIMO:
    """``deepmimo.export`` (plugins/deepmimo.py:407-728): ``paths`` is one :class:`TracedPaths` or an
    iterable of them (different orders, or the chunks of ``Scene.trace_paths(chunk_size=...)``);
    ``polarization`` is ``"V"``, ``"H"``, a 3-vector, or a ``(tx, rx)`` pair."""
    mesh = scene.mesh
    if mesh.face_materials is None:  # :451-453
        raise ValueError("Scene must contain information about face materials.")
    if radio_materials is None:
        radio_materials = materials
    n_complex, thickness = material_tables(mesh.material_names, radio_materials, frequency)
    dev = device()
    tx_pos = scene.transmitters.detach().reshape(-1, 3)
    rx_pos = scene.receivers.detach().reshape(-1, 3)
    ntx, nrx = tx_pos.shape[0], rx_pos.shape[0]

    def empty(*tail, dtype=torch.float32):
        return torch.zeros((ntx, nrx, 0, *tail), dtype=dtype, device=dev)

    cols = {k: empty() for k in ("power", "phase", "delay", "aoa_az", "aoa_el", "aod_az", "aod_el")}
    inter, inter_pos = empty(0, dtype=torch.int32), empty(0, 3)
    prims = empty(0, dtype=torch.int32) if include_primitives else None
    masks = []
    for p in ([paths] if isinstance(paths, TracedPaths) else paths):
        p = p.reshape(ntx, nrx, -1)  # :520
        ch = paths_channel(p, mesh, n_complex, thickness, frequency, polarization)
        for k in cols:
            cols[k] = torch.cat((cols[k], ch[k]), dim=-1)
        ids = p.objects[..., 1:-1].to(torch.int32)
        if prims is not None:
            prims = _pad_cat(prims, ids, _NO_INTERACTION)  # :527-532
        types = p.interaction_types if p.interaction_types is not None else torch.zeros_like(ids)
        inter = _pad_cat(inter, types.to(torch.int32), _NO_INTERACTION)  # :534-546
        inter_pos = _pad_cat(inter_pos, p.vertices[..., 1:-1, :], 0.0)  # :548-552
        masks.append(p.mask)  # :679-689 (float confidences are exported as they are)
    soft = any(m.dtype != torch.bool for m in masks)
    mask = torch.cat([empty(dtype=torch.float32 if soft else torch.bool),
                      *(m.to(torch.float32) if soft else m for m in masks)], dim=-1)
    return DeepMIMO(inter=inter, inter_pos=inter_pos, rx_pos=rx_pos, tx_pos=tx_pos, mask=mask, primitives=prims,
                    **cols)
