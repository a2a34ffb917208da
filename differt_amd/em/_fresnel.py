"""``differt.em._fresnel`` on the GPU (complex64)."""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .._tensors import as_f32, device, ptr, stream

__all__ = ["fresnel_coefficients", "reflection_coefficients", "refraction_coefficients", "refractive_index"]


def _as_c64(x, dev):
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=torch.complex64)
    return torch.as_tensor(np.asarray(x).astype(np.complex64), device=dev)


def refractive_index(epsilon_r, mu_r=None):
    """em/_fresnel.py:10-44: ``sqrt(epsilon_r * mu_r)``; real for real non-negative inputs, complex64
    otherwise (``mu_r`` multiplies on the host side of the call: it is a scalar material constant)."""
    dev = device()
    e = _as_c64(epsilon_r, dev)
    real_in = not (torch.is_complex(epsilon_r) if isinstance(epsilon_r, torch.Tensor) else np.iscomplexobj(epsilon_r))
    if mu_r is not None:
        e = e * _as_c64(mu_r, dev)
        real_in = real_in and not (torch.is_complex(mu_r) if isinstance(mu_r, torch.Tensor) else np.iscomplexobj(mu_r))
    z = torch.view_as_real(e.contiguous()).contiguous()
    out = torch.empty_like(z)
    if e.numel():
        _lib.call("drt_refractive_index", ptr(z), e.numel(), ptr(out), stream())
    n = torch.view_as_complex(out).reshape(e.shape)
    return n.real.contiguous() if real_in and bool((e.real >= 0).all()) else n


def fresnel_coefficients(n_r, cos_theta_i):
    """em/_fresnel.py:47-214: ``((r_s, r_p), (t_s, t_p))`` for the relative refractive index ``n_r``
    (complex) and ``cos(theta_i)``; ``safe_divide`` semantics (0 where the denominator vanishes)."""
    dev = device()
    n = _as_c64(n_r, dev)
    ct = as_f32(cos_theta_i, dev)
    batch = torch.broadcast_shapes(n.shape, ct.shape)
    B = int(np.prod(batch, dtype=np.int64))
    nf = torch.view_as_real(n.expand(batch).contiguous()).contiguous()
    cf = ct.detach().expand(batch).contiguous()
    outs = [torch.empty((*batch, 2), dtype=torch.float32, device=dev) for _ in range(4)]
    if B:
        _lib.call("drt_fresnel_coefficients", ptr(nf), ptr(cf), B, *(ptr(o) for o in outs), stream())
    r_s, r_p, t_s, t_p = (torch.view_as_complex(o) for o in outs)
    return (r_s, r_p), (t_s, t_p)


def reflection_coefficients(n_r, cos_theta_i):
    """em/_fresnel.py:217-488."""
    return fresnel_coefficients(n_r, cos_theta_i)[0]


def refraction_coefficients(n_r, cos_theta_i):
    """em/_fresnel.py:491-516."""
    return fresnel_coefficients(n_r, cos_theta_i)[1]
