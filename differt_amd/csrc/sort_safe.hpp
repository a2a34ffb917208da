// sort_safe.hpp -- the radix-sort configuration for every sort that can sit inside a CAPTURED call.
//
// rocPRIM's radix sort switches to its one-sweep algorithm above 2^20 items, and that algorithm resets its histogram and
// look-back buffers with hipMemsetAsync (rocprim/device/device_radix_sort.hpp:122, :251).  Under stream capture these become
// memset NODES, and on ROCm 7.x a graph that holds memset nodes replays correctly once and then fills garbage (core.hip,
// fill_bytes_kernel): the look-back states of the second replay are wild, the scatter offsets with them -- a memory
// aperture violation on the second replay of drt_trace_paths_beam_async on a 200 000-triangle mesh (found by the round-4
// bench leg that captures configs[4]; the 10 000-triangle graphs happened to survive their garbage).  With an unbounded
// merge-sort limit rocPRIM takes its merge sort at every size: kernels only, nothing to reset from the host side.
// The synchronous entry points keep the default configuration (one-sweep is ~3x faster at 2^24 keys).
#pragma once

#include <rocprim/rocprim.hpp>

namespace drt {
using CaptureSafeSort = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config,
                                                   ~static_cast<size_t>(0)>;
}
