"""Source rules that a GPU-less box can check: what may NOT sit inside a capturable entry point.

On this ROCm a HIP graph that holds memset nodes replays correctly once and then fills garbage (csrc/core.hip), and
rocPRIM's radix sort turns into such nodes above 2^20 items (csrc/sort_safe.hpp): the second replay of a captured
drt_trace_paths_beam_async on a 200 000-triangle mesh died with a memory aperture violation (round 4).  Every
hipMemsetAsync and every default-configuration radix sort of the kernel sources is therefore listed here with the
(synchronous, never captured) function it belongs to; a new one fails this test and sends its author to fill_bytes_async /
CaptureSafeSort."""
import re
from pathlib import Path

CSRC = Path(__file__).resolve().parents[1] / "differt_amd" / "csrc"

# file -> number of allowed occurrences, all inside entry points that allocate or synchronise (never capturable)
MEMSET_ALLOWED = {
    "bvh.hip": 4,    # drt_mesh_build_bvh (allocates, synchronises)
    "mesh.hip": 1,   # drt_mesh_create
    "trace.hip": 1,  # drt_trace_paths_compact (reads its counters back)
}
DEFAULT_SORT_ALLOWED = {
    "beam.hip": 7,   # three temp-size queries + the two sorts of the SYNCHRONOUS drt_trace_paths_beam + query and sort of
                     # the pairing pass (pair_triangles: allocates, synchronises, once per mesh)
    "bvh.hip": 2,    # drt_mesh_build_bvh
    "trace.hip": 2,  # temp-size query + the sort of the SYNCHRONOUS drt_trace_paths_compact
}


def _count(pattern: str) -> dict:
    out = {}
    for f in sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.hpp")):
        text = re.sub(r"//[^\n]*", "", f.read_text())  # comments may talk about them
        n = len(re.findall(pattern, text))
        if n:
            out[f.name] = n
    return out


def test_memset_nodes_only_in_non_capturable_entry_points():
    assert _count(r"hipMemsetAsync\s*\(") == MEMSET_ALLOWED


def test_default_radix_sorts_only_in_synchronous_entry_points():
    assert _count(r"rocprim::radix_sort_(?:keys|pairs)\s*\(") == DEFAULT_SORT_ALLOWED


def test_capturable_entry_points_use_the_safe_configuration():
    beam = (CSRC / "beam.hip").read_text()
    i = beam.index("int32_t drt_trace_paths_beam_async(")
    body = beam[i:]
    assert body.count("capture_safe_sort(") >= 2
    assert not re.search(r"rocprim::radix_sort_(?:keys|pairs)\s*\(", re.sub(r"//[^\n]*", "", body))
    trace = (CSRC / "trace.hip").read_text()
    j = trace.index("int32_t drt_trace_paths_compact_async(")
    k = trace.index("\n}\n", j)
    assert "capture_safe_sort(" in trace[j:k]
    # the dispatcher itself: the library's own radix sort (kernels only) or rocPRIM's merge-sort configuration
    safe = re.sub(r"//[^\n]*", "", (CSRC / "sort_safe.hpp").read_text())
    assert "radix_sort_u64(" in safe and "radix_sort_keys<CaptureSafeSort>" in safe and "radix_sort_pairs<CaptureSafeSort>" in safe
    own = re.sub(r"//[^\n]*", "", (CSRC / "radix_sort.hip").read_text())
    assert "hipMemset" not in own and "rocprim::" not in own


def test_cluster_boxes_are_read_through_the_constant_address_space():
    """The cluster box tables are wave-uniform and read-only, but next to the LDS-DMA builtin of the same loop the compiler
    makes a plain `C.boxes[...]` read a VECTOR load with a full vmcnt(0) wait (three memory latencies in series per cluster
    trip of every clustered expansion kernel until round 5, profiles/r05/beam.md section 7): every device-side read goes
    through ro() / ro4() (constant address space -> s_load)."""
    beam = re.sub(r"//[^\n]*", "", (CSRC / "beam.hip").read_text())
    reads = re.findall(r"[^\n]*\bC\.(?:sub)?boxes\b[^\n]*", beam)
    device_reads = [ln for ln in reads if "=" not in ln.split("C.")[0] or "ro" in ln]
    assert len(device_reads) >= 6
    for ln in reads:
        if re.search(r"C\.(?:sub)?boxes\s*=", ln):  # the host side filling the struct
            continue
        assert re.search(r"\bro4?\(C\.(?:sub)?boxes", ln), ln.strip()
