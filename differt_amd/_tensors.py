"""Device-buffer plumbing: torch is used ONLY to own HBM allocations, streams and autograd glue.

Every compute call goes through the C ABI (``differt_amd._lib``) with raw device pointers.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

F32_EPS = float(np.finfo(np.float32).eps)


def device() -> torch.device:
    """The GPU the hot path runs on.  Raises (never falls back) when there is none."""
    if not torch.cuda.is_available():
        raise RuntimeError(
            "differt_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False and "
            "there is no CPU fallback"
        )
    _lib.require_device()
    return torch.device("cuda", torch.cuda.current_device())


def as_f32(x, dev: torch.device | None = None) -> torch.Tensor:
    dev = dev or device()
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=torch.float32)
    return torch.as_tensor(np.asarray(x, dtype=np.float32), device=dev)


def as_i32(x, dev: torch.device | None = None) -> torch.Tensor:
    dev = dev or device()
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=torch.int32)
    return torch.as_tensor(np.asarray(x).astype(np.int32), device=dev)


def as_u8(x, dev: torch.device | None = None) -> torch.Tensor:
    dev = dev or device()
    if isinstance(x, torch.Tensor):
        return x.to(device=dev).to(torch.uint8)
    return torch.as_tensor(np.asarray(x).astype(np.uint8), device=dev)


def ptr(t: torch.Tensor | None):
    """Raw device pointer (or NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "C ABI takes densely packed buffers"
    return C.c_void_p(t.data_ptr())


def stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def shape_of(x) -> tuple[int, ...]:
    return tuple(x.shape) if hasattr(x, "shape") else tuple(np.asarray(x).shape)
