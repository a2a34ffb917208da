"""drt_comm_* / drt_allreduce_* / drt_allgather_bytes (csrc/comm.hip): the path's collectives on RCCL without
torch.distributed (SURVEY.md section 8b export list, 8e).  CPU: symbols, error behaviour without a GPU.  GPU box
(one GPU): a world-size-1 communicator runs every collective for real through librccl."""

from __future__ import annotations

import ctypes as C

import pytest
import torch

from differt_amd import _lib

NAMES = ["drt_comm_unique_id", "drt_comm_init", "drt_comm_destroy", "drt_comm_rank", "drt_comm_world",
         "drt_allreduce_min_u64", "drt_allreduce_max_u8", "drt_allreduce_sum_f32", "drt_allgather_bytes"]


def test_symbols_exported_and_declared():
    L = _lib.load()
    declared = _lib.declared_symbols()
    for n in NAMES:
        assert hasattr(L, n) and n in declared and n in _lib._SIGNATURES


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_init_without_a_gpu_is_a_status_code():
    h = C.c_void_p()
    ident = (C.c_uint8 * 128)()
    with pytest.raises(_lib.DrtError):
        _lib.call("drt_comm_init", ident, 0, 1, C.byref(h))
    assert not h.value
    with pytest.raises(ValueError):
        _lib.call("drt_comm_init", ident, 3, 2, C.byref(h))  # rank outside the world: invalid argument first
    assert _lib.load().drt_comm_destroy(None) == 0


@pytest.mark.gpu
def test_world_size_one_collectives_on_rccl():
    from differt_amd.distributed import NativeComm

    torch.cuda.set_device(0)
    comm = NativeComm(NativeComm.unique_id(), 0, 1)
    try:
        assert _lib.load().drt_comm_rank(comm.h) == 0 and _lib.load().drt_comm_world(comm.h) == 1
        g = torch.Generator(device="cuda").manual_seed(3)
        keys = torch.randint(-(1 << 62), 1 << 62, (1 << 17,), dtype=torch.int64, device="cuda", generator=g)
        ref = keys.clone()
        assert torch.equal(comm.allreduce_min_u64(keys), ref)
        flags = (torch.rand(4097, device="cuda", generator=g) < 0.3).to(torch.uint8)
        assert torch.equal(comm.allreduce_max_u8(flags.clone()), flags)
        x = torch.randn(3 * 16 + 3 * 64, device="cuda", generator=g)
        assert torch.equal(comm.allreduce_sum_f32(x.clone()), x)
        rec = torch.randint(0, 255, (1000, 40), dtype=torch.uint8, device="cuda", generator=g)
        out = comm.allgather_bytes(rec)
        assert out.shape == (1, 1000, 40) and torch.equal(out[0], rec)
        torch.cuda.synchronize()
    finally:
        comm.close()
