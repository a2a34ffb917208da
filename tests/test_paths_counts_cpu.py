"""TracedPaths.num_valid_paths from the dense tracer's device-side counters (differt_amd/geometry/_paths.py): the cached
pair is trusted only while `mask` is the very tensor the kernel wrote -- same storage, same version, same element count.
Runs without a GPU (the logic is host-side; tests/test_trace_gpu.py checks the counters themselves)."""
import torch

from differt_amd.geometry._paths import TracedPaths


def _paths():
    mask = torch.tensor([[True, False, True], [False, False, True]])
    verts = torch.zeros((2, 3, 4, 3))
    objs = torch.zeros((2, 3, 4), dtype=torch.int32)
    return TracedPaths(verts, objs, mask)


def test_counters_are_used_while_the_mask_is_untouched():
    p = _paths()
    assert int(p.num_valid_paths) == 3
    # counters that DISAGREE with the mask on purpose: the fast path returns survivors - cleared
    p._attach_valid_count(torch.tensor([7, 2]))
    assert int(p.num_valid_paths) == 5
    assert int(p.reshape(-1).num_valid_paths) == 5 and int(p.reshape(3, 2).num_valid_paths) == 5
    assert int(p.reshape(1, 2, 3).squeeze(0).num_valid_paths) == 5


def test_any_edit_slice_or_replacement_falls_back_to_the_reduction():
    p = _paths()
    p._attach_valid_count(torch.tensor([7, 2]))
    q = p.reshape(-1)
    p.mask[0, 0] = False  # in-place edit: the version moves for every view
    assert int(p.num_valid_paths) == 2 and int(q.num_valid_paths) == 2
    p = _paths()
    p._attach_valid_count(torch.tensor([7, 2]))
    sl = TracedPaths(p.vertices[:1], p.objects[:1], p.mask[:1])  # same storage start, fewer elements
    sl.__dict__["_valid_count"] = p.__dict__["_valid_count"]
    assert int(sl.num_valid_paths) == 2
    p.mask = p.mask.clone()  # replaced tensor
    assert int(p.num_valid_paths) == 3
