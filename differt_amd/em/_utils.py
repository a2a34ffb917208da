"""``differt.em._utils`` on the GPU: delays, s/p bases, basis rotations, free-space path loss."""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .._tensors import as_f32, device, ptr, stream
from ._constants import c

__all__ = ["fspl", "length_to_delay", "path_delay", "sp_directions", "sp_rotation_matrix"]


def _binary(name: str, a, b, *extra):
    dev = device()
    a, b = as_f32(a, dev).detach(), as_f32(b, dev).detach()
    batch = torch.broadcast_shapes(a.shape, b.shape)
    a, b = a.expand(batch).contiguous(), b.expand(batch).contiguous()
    out = torch.empty(batch, dtype=torch.float32, device=dev)
    if out.numel():
        _lib.call(name, ptr(a), ptr(b), out.numel(), *extra, ptr(out), stream())
    return out


def length_to_delay(length, speed=c):
    """em/_utils.py:14-44: ``length / speed``."""
    return _binary("drt_length_to_delay", length, speed)


def path_length(path):
    """geometry/_utils.py:150-181: sum of the segment lengths of ``path [*batch, L, 3]``."""
    dev = device()
    p = as_f32(path, dev).detach().contiguous()
    batch, L = p.shape[:-2], p.shape[-2]
    B = int(np.prod(batch, dtype=np.int64))
    out = torch.zeros(batch, dtype=torch.float32, device=dev)
    if B and L:
        _lib.call("drt_path_length", ptr(p), B, L, ptr(out), stream())
    return out


def path_delay(path, **kwargs):
    """em/_utils.py:47-81."""
    return length_to_delay(path_length(path), **kwargs)


def _bcast3(*xs):
    dev = device()
    ts = [as_f32(x, dev).detach() for x in xs]
    batch = torch.broadcast_shapes(*(t.shape[:-1] for t in ts))
    return batch, [t.expand(*batch, 3).contiguous() for t in ts]


def sp_directions(k_i, k_r, normals):
    """em/_utils.py:84-265: ``((e_i_s, e_i_p), (e_r_s, e_r_p))`` for incident / reflected unit
    directions and surface normals; at normal incidence ``e_i_s`` is any unit vector perpendicular to
    ``k_i`` (``perpendicular_vector``, geometry/_utils.py:76-109)."""
    batch, (ki, kr, n) = _bcast3(k_i, k_r, normals)
    B = int(np.prod(batch, dtype=np.int64))
    outs = [torch.empty((*batch, 3), dtype=torch.float32, device=ki.device) for _ in range(4)]
    if B:
        _lib.call("drt_sp_directions", ptr(ki), ptr(kr), ptr(n), B, *(ptr(o) for o in outs), stream())
    return (outs[0], outs[1]), (outs[2], outs[3])


def sp_rotation_matrix(e_a_s, e_a_p, e_b_s, e_b_p):
    """em/_utils.py:268-303: ``[*batch, 2, 2]`` matrix from the (a_s, a_p) to the (b_s, b_p) basis."""
    batch, (a_s, a_p, b_s, b_p) = _bcast3(e_a_s, e_a_p, e_b_s, e_b_p)
    B = int(np.prod(batch, dtype=np.int64))
    out = torch.empty((*batch, 2, 2), dtype=torch.float32, device=a_s.device)
    if B:
        _lib.call("drt_sp_rotation_matrix", ptr(a_s), ptr(a_p), ptr(b_s), ptr(b_p), B, ptr(out), stream())
    return out


def fspl(d, f, *, dB: bool = False):  # noqa: N803
    """em/_utils.py:345-367."""
    return _binary("drt_fspl", d, f, int(bool(dB)))
