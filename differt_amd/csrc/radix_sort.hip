// radix_sort.hip -- a capture-safe, stable LSD radix sort of u64 keys (optionally with a u32 payload) for the entry
// points that may sit inside a captured HIP graph (drt_trace_paths_beam_async, drt_trace_paths_compact_async, the
// deterministic VJP).
//
// Why our own: rocPRIM's radix sort switches to its one-sweep algorithm above 2^20 items and resets that algorithm's
// histogram / look-back state with hipMemsetAsync; under stream capture these become memset NODES, and on ROCm 7.x a
// graph that holds memset nodes replays correctly once and then fills garbage (core.hip, fill_bytes_kernel; found in
// round 4 by capturing configs[4]).  Round 4 fell back to rocPRIM's merge sort at every size (sort_safe.hpp), "several
// milliseconds at 2^25 rows" (DESIGN.md).  This sort is kernels only -- every counter it uses is WRITTEN by a kernel before
// it is read, nothing to reset -- 8 bits per pass, three small steps per pass:
//   rs_hist     one workgroup per tile of 2048 keys: 256-bin histogram in LDS -> counts[digit][tile]
//   rs_scan*    exclusive scan of the digit-major counter array (chunks of 2048, then the chunk sums by one workgroup)
//   rs_scatter  the tile again: a STABLE rank per key -- lanes hold consecutive keys, same-digit peers by eight wave
//               ballots, v_mbcnt for the rank among them, a wave-private LDS counter per digit carried from round to
//               round, the four waves' counts prefix-summed per digit -- and the scatter to base[digit][tile] + rank
// Traffic per pass: the keys twice in, once out (HBM-bound: 24 B per key per pass, + 8 B per payload); 8 passes for 63 bits.
#include "common.hpp"
#include "sort_safe.hpp"

namespace drt {

constexpr int kRsThreads = 256;
constexpr int kRsKeysPerThread = 8;
constexpr int kRsTile = kRsThreads * kRsKeysPerThread;  // 2048 keys per workgroup
constexpr int kRsChunk = 2048;                          // counters per workgroup of the scan

__global__ __launch_bounds__(kRsThreads) void rs_hist_kernel(const unsigned long long *__restrict__ keys, int64_t n, int shift,
                                                             uint32_t mask, uint32_t *__restrict__ counts, int64_t ntiles) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kRsTile;
#pragma unroll
    for (int i = 0; i < kRsKeysPerThread; ++i) {
        const int64_t idx = base + i * kRsThreads + threadIdx.x;
        if (idx < n) atomicAdd(&h[(uint32_t)(keys[idx] >> shift) & mask], 1u);
    }
    __syncthreads();
    counts[(int64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of 256 values held one per thread (value in, exclusive prefix out; *total = the sum, on every thread)
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t *lds_wave_sums /*[4]*/, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) lds_wave_sums[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t s = lds_wave_sums[w];
        before += (w < wave) ? s : 0u;
        all += s;
    }
    __syncthreads();
    *total = all;
    return before + incl - v;
}

// chunks of 2048 counters (8 consecutive per thread): exclusive scan in place, the chunk's sum -> sums[chunk]
__global__ __launch_bounds__(kRsThreads) void rs_scan_chunks_kernel(uint32_t *__restrict__ counts, int64_t m, uint32_t *__restrict__ sums) {
    __shared__ uint32_t ws[4];
    const int64_t base = (int64_t)blockIdx.x * kRsChunk + (int64_t)threadIdx.x * 8;
    uint32_t v[8], s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = (base + i < m) ? counts[base + i] : 0u;
        s += v[i];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan_256(s, ws, &total);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (base + i < m) counts[base + i] = run;
        run += v[i];
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// one workgroup: exclusive scan of the chunk sums, in place
__global__ __launch_bounds__(kRsThreads) void rs_scan_sums_kernel(uint32_t *__restrict__ sums, int64_t nchunks) {
    __shared__ uint32_t ws[4];
    uint32_t carry = 0;
    for (int64_t base = 0; base < nchunks; base += kRsThreads) {
        const int64_t i = base + threadIdx.x;
        const uint32_t v = (i < nchunks) ? sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_256(v, ws, &total);
        if (i < nchunks) sums[i] = carry + ex;
        carry += total;
    }
}

template <bool VALUES>
__global__ __launch_bounds__(kRsThreads) void rs_scatter_kernel(const unsigned long long *__restrict__ keys_in,
                                                                const uint32_t *__restrict__ vals_in,
                                                                unsigned long long *__restrict__ keys_out,
                                                                uint32_t *__restrict__ vals_out, int64_t n, int shift,
                                                                uint32_t mask, const uint32_t *__restrict__ counts,
                                                                const uint32_t *__restrict__ sums, int64_t ntiles) {
    __shared__ uint32_t cnt[4][256];   // per wave and digit: keys seen so far (then: the wave's base)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#pragma unroll
    for (int w = 0; w < 4; ++w) cnt[w][threadIdx.x] = 0u;
    __syncthreads();
    const int64_t wbase = (int64_t)blockIdx.x * kRsTile + (int64_t)wave * (kRsTile / 4);
    unsigned long long key[kRsKeysPerThread];
    uint32_t val[kRsKeysPerThread], off[kRsKeysPerThread];
#pragma unroll
    for (int r = 0; r < kRsKeysPerThread; ++r) {  // lanes hold consecutive keys of round r
        const int64_t idx = wbase + r * 64 + lane;
        key[r] = (idx < n) ? keys_in[idx] : 0ull;
        if (VALUES) val[r] = (idx < n) ? vals_in[idx] : 0u;
    }
#pragma unroll
    for (int r = 0; r < kRsKeysPerThread; ++r) {
        const int64_t idx = wbase + r * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bb = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bb : ~bb;
        }
        const uint32_t below = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        const uint32_t prev = cnt[wave][d];  // the same value for all peers; read before the leader's write (in-order LDS queue)
        off[r] = prev + below;
        if (valid && (peers >> lane) == 1ull) cnt[wave][d] = prev + (uint32_t)__popcll(peers);  // highest peer
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {  // thread = digit: where this tile's keys of that digit start, per wave
        const int64_t ci = (int64_t)threadIdx.x * ntiles + blockIdx.x;
        uint32_t run = counts[ci] + sums[ci / kRsChunk];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t c = cnt[w][threadIdx.x];
            cnt[w][threadIdx.x] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRsKeysPerThread; ++r) {
        const int64_t idx = wbase + r * 64 + lane;
        if (idx < n) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
            const uint32_t pos = cnt[wave][d] + off[r];
            // (pos < n by construction; the compare costs nothing next to the store and turns a caller's misuse -- two
            // calls in flight on ONE workspace overwrite each other's counters -- into wrong output instead of a wild write)
            if ((int64_t)pos < n) {
                keys_out[pos] = key[r];
                if (VALUES) vals_out[pos] = val[r];
            }
        }
    }
}

__global__ __launch_bounds__(256) void rs_copy_kernel(const unsigned long long *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                      unsigned long long *__restrict__ kout, uint32_t *__restrict__ vout, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    kout[i] = kin[i];
    if (vin) vout[i] = vin[i];
}

static size_t rs_align(size_t x) { return (x + 255) / 256 * 256; }

size_t radix_sort_u64_temp_bytes(int64_t n, bool values) {
    if (n <= 0) return 256;
    const int64_t ntiles = ceil_div(n, (int64_t)kRsTile), m = ntiles * 256, nchunks = ceil_div(m, (int64_t)kRsChunk);
    return rs_align((size_t)n * 8) + (values ? rs_align((size_t)n * 4) : 0) + rs_align((size_t)m * 4) + rs_align((size_t)nchunks * 4);
}

hipError_t radix_sort_u64(void *tmp, size_t tmp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out,
                          const uint32_t *vals_in, uint32_t *vals_out, int64_t n, int begin_bit, int end_bit, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const bool values = vals_in != nullptr;
    if (n >= (1ll << 31) || tmp_bytes < radix_sort_u64_temp_bytes(n, values) || begin_bit < 0 || end_bit > 64 || end_bit < begin_bit)
        return hipErrorInvalidValue;
    const int64_t ntiles = ceil_div(n, (int64_t)kRsTile), m = ntiles * 256, nchunks = ceil_div(m, (int64_t)kRsChunk);
    char *p = reinterpret_cast<char *>(tmp);
    auto *kt = reinterpret_cast<unsigned long long *>(p);
    p += rs_align((size_t)n * 8);
    uint32_t *vt = nullptr;
    if (values) {
        vt = reinterpret_cast<uint32_t *>(p);
        p += rs_align((size_t)n * 4);
    }
    auto *counts = reinterpret_cast<uint32_t *>(p);
    p += rs_align((size_t)m * 4);
    auto *sums = reinterpret_cast<uint32_t *>(p);
    const int passes = (end_bit - begin_bit + 7) / 8;
    if (passes == 0) {  // nothing to order by: a stable sort is a copy
        hipLaunchKernelGGL(rs_copy_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, keys_in, vals_in, keys_out, vals_out, n);
        return hipGetLastError();
    }
    // ping-pong between the caller's output and the temporary so that the LAST pass writes the output
    const unsigned long long *kin = keys_in;
    const uint32_t *vin = vals_in;
    for (int pass = 0; pass < passes; ++pass) {
        const bool to_out = ((passes - 1 - pass) % 2) == 0;
        unsigned long long *kout = to_out ? keys_out : kt;
        uint32_t *vout = to_out ? vals_out : vt;
        const int shift = begin_bit + 8 * pass;
        const int bits = (end_bit - shift < 8) ? end_bit - shift : 8;
        const uint32_t mask = (1u << bits) - 1u;
        hipLaunchKernelGGL(rs_hist_kernel, dim3((unsigned)ntiles), dim3(kRsThreads), 0, s, kin, n, shift, mask, counts, ntiles);
        hipLaunchKernelGGL(rs_scan_chunks_kernel, dim3((unsigned)nchunks), dim3(kRsThreads), 0, s, counts, m, sums);
        hipLaunchKernelGGL(rs_scan_sums_kernel, dim3(1), dim3(kRsThreads), 0, s, sums, nchunks);
        if (values)
            hipLaunchKernelGGL(rs_scatter_kernel<true>, dim3((unsigned)ntiles), dim3(kRsThreads), 0, s, kin, vin, kout, vout, n, shift, mask,
                               counts, sums, ntiles);
        else
            hipLaunchKernelGGL(rs_scatter_kernel<false>, dim3((unsigned)ntiles), dim3(kRsThreads), 0, s, kin, vin, kout, vout, n, shift, mask,
                               counts, sums, ntiles);
        kin = kout;
        vin = vout;
    }
    return hipGetLastError();
}

}  // namespace drt

extern "C" {

size_t drt_sort_u64_workspace_size(int64_t n, int32_t with_values) { return drt::radix_sort_u64_temp_bytes(n, with_values != 0); }

int32_t drt_sort_u64(const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *values_in, uint32_t *values_out, int64_t n,
                     int32_t begin_bit, int32_t end_bit, void *workspace, size_t workspace_bytes, void *stream) {
    using namespace drt;
    DRT_REQUIRE(n >= 0 && n < (1ll << 31), "n must be in [0, 2^31)");
    if (n == 0) return DRT_OK;
    DRT_REQUIRE(keys_in && keys_out && keys_in != keys_out, "keys_in / keys_out must be distinct device arrays");
    DRT_REQUIRE((values_in == nullptr) == (values_out == nullptr), "values_in and values_out go together");
    DRT_REQUIRE(begin_bit >= 0 && end_bit <= 64 && begin_bit <= end_bit, "bad bit range");
    const size_t need = radix_sort_u64_temp_bytes(n, values_in != nullptr);
    if (!workspace || workspace_bytes < need) return fail(DRT_E_CAPACITY, "workspace too small: need %zu bytes", need);
    DRT_HIP(radix_sort_u64(workspace, workspace_bytes, reinterpret_cast<const unsigned long long *>(keys_in),
                           reinterpret_cast<unsigned long long *>(keys_out), values_in, values_out, n, begin_bit, end_bit,
                           as_stream(stream)));
    return DRT_OK;
}

}  // extern "C"
