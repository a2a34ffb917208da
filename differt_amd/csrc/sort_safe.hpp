// sort_safe.hpp -- the radix-sort configuration for every sort that can sit inside a CAPTURED call.
//
// rocPRIM's radix sort switches to its one-sweep algorithm above 2^20 items, and that algorithm resets its histogram and
// look-back buffers with hipMemsetAsync (rocprim/device/device_radix_sort.hpp:122, :251).  Under stream capture these become
// memset NODES, and on ROCm 7.x a graph that holds memset nodes replays correctly once and then fills garbage (core.hip,
// fill_bytes_kernel): the look-back states of the second replay are wild, the scatter offsets with them -- a memory
// aperture violation on the second replay of drt_trace_paths_beam_async on a 200 000-triangle mesh (found by the round-4
// bench leg that captures configs[4]; the 10 000-triangle graphs happened to survive their garbage).  With an unbounded
// merge-sort limit rocPRIM takes its merge sort at every size: kernels only, nothing to reset from the host side.
// Round 5: from 2^15 keys on the capturable entry points use the library's OWN radix sort (csrc/radix_sort.hip: kernels
// only, every counter written by a kernel before it is read); rocPRIM's merge sort stays for the small sorts (a few
// thousand valid paths), where it is one or two launches.
#pragma once

#include <cstring>
#include <string.h>  // (rocPRIM's texture iterator calls the global memset)

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

namespace drt {
using CaptureSafeSort = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config,
                                                   ~static_cast<size_t>(0)>;

// csrc/radix_sort.hip: stable LSD radix sort of u64 keys (+ optional u32 payload), capture-safe
size_t radix_sort_u64_temp_bytes(int64_t n, bool values);
hipError_t radix_sort_u64(void *tmp, size_t tmp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out,
                          const uint32_t *vals_in, uint32_t *vals_out, int64_t n, int begin_bit, int end_bit, hipStream_t s);

#if defined(DRT_LAB) && defined(DRT_LAB_OWN_SORT_MIN)  // lab: A/B against rocPRIM's merge sort (a huge value disables the own sort)
constexpr int64_t kOwnSortMin = DRT_LAB_OWN_SORT_MIN;
#else
constexpr int64_t kOwnSortMin = 1 << 15;
#endif

inline size_t capture_safe_sort_temp_bytes(int64_t n, bool values) {
    if (n <= 0) return 0;
    size_t safe = 0;
    if (values)
        (void)rocprim::radix_sort_pairs<CaptureSafeSort>(nullptr, safe, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                                         (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0, 64, nullptr);
    else
        (void)rocprim::radix_sort_keys<CaptureSafeSort>(nullptr, safe, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                                        (size_t)n, 0, 64, nullptr);
    const size_t own = radix_sort_u64_temp_bytes(n, values);
    return safe > own ? safe : own;
}
// keys_in is left intact; `tmp_bytes` >= capture_safe_sort_temp_bytes(n, vals_in != nullptr)
inline hipError_t capture_safe_sort(void *tmp, size_t tmp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out,
                                    const uint32_t *vals_in, uint32_t *vals_out, int64_t n, int begin_bit, int end_bit,
                                    hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n >= kOwnSortMin) return radix_sort_u64(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, s);
    if (vals_in)
        return rocprim::radix_sort_pairs<CaptureSafeSort>(tmp, tmp_bytes, const_cast<unsigned long long *>(keys_in), keys_out,
                                                          const_cast<uint32_t *>(vals_in), vals_out, (size_t)n, begin_bit, end_bit, s);
    return rocprim::radix_sort_keys<CaptureSafeSort>(tmp, tmp_bytes, const_cast<unsigned long long *>(keys_in), keys_out, (size_t)n,
                                                     begin_bit, end_bit, s);
}
}
