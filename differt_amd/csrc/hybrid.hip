// hybrid.hip -- the per-pair visibility-pruned tracer behind ONE C entry point.
//
// Reference: HybridPathTracer (geometry/_solvers.py:960-1176) estimates which primitives are visible from the
// transmitters / receivers, restricts the first (last) interaction of a candidate to them -- merged over all
// transmitters (receivers), :969-973 -- and traces the pruned DiGraph exhaustively.  The MI355X extension keeps the
// sets PER end point: pair (i, j) traces F_i x N^(order-2) x L_j, all pairs in one launch over the concatenated
// (ragged) spaces of the compact tracer.  Round 3 built the CSR sets and the pair offsets in torch (nonzero / cumsum
// and three .item() synchronisations); here they are kernels, so a host without torch gets the whole tracer:
//
//   drt_trace_paths_hybrid_pairs(mesh, params, tx, rx, order, vis_tx u8[Ntx,T], vis_rx u8[Nrx,T], ...)
//     vis_prim   : triangle visibility -> primitive visibility (quads: either triangle, _solvers.py:1024-1031),
//                  AND the primitive's mask (:1038-1042)
//     row counts : one wavefront per end point, popcount of ballots
//     offsets    : exclusive scan of the counts (a single-block scan: end points are few)
//     compaction : one wavefront per end point writes its visible primitive ids, ascending
//     pair sizes : |F_i| n^(order-2) |L_j| -> exclusive scan -> pair_offsets (rocPRIM)
//   then drt_trace_paths_compact on the ragged description, and the keys are rewritten as PACKED keys
//   ((tx nrx + rx) n^order + sum_j m_j n^(order-1-j), like drt_trace_paths_beam) so that the VJP needs no set.
#include <algorithm>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "common.hpp"
#include "mesh.hpp"

namespace drt {

// vis u8[V, T] (non-zero = seen) -> u8[V, P]: primitive p = triangles [p scale, (p+1) scale)
__global__ __launch_bounds__(256) void hyb_prim_vis_kernel(const uint8_t *__restrict__ vis, int64_t V, int64_t T, int64_t P,
                                                           int32_t scale, const uint8_t *__restrict__ mask,
                                                           uint8_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= V * P) return;
    const int64_t v = i / P, p = i - v * P;
    bool seen = false, active = true;
    for (int t = 0; t < scale; ++t) {
        const int64_t f = p * scale + t;
        seen = seen || vis[v * T + f] != 0;
        if (mask) active = active && mask[f] != 0;
    }
    out[i] = (uint8_t)(seen && active);
}

// rows of flags u8[V, P]: counts[v] = number of set flags (one wavefront per row)
__global__ __launch_bounds__(64) void hyb_row_count_kernel(const uint8_t *__restrict__ flags, int64_t P,
                                                           long long *__restrict__ counts) {
    const int64_t v = blockIdx.x;
    const int lane = threadIdx.x;
    long long c = 0;
    for (int64_t p0 = 0; p0 < P; p0 += 64) {
        const int64_t p = p0 + lane;
        c += __popcll(__ballot(p < P && flags[v * P + p] != 0));
    }
    if (lane == 0) counts[v] = c;
}

// offsets[0..V] = exclusive scan of counts (V is a number of end points: one thread is enough)
__global__ void hyb_scan_small_kernel(const long long *__restrict__ counts, int64_t V, long long *__restrict__ offsets) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long s = 0;
    for (int64_t v = 0; v < V; ++v) {
        offsets[v] = s;
        s += counts[v];
    }
    offsets[V] = s;
}

// ids of the set flags of every row, ascending, at offsets[v]
__global__ __launch_bounds__(64) void hyb_compact_kernel(const uint8_t *__restrict__ flags, int64_t P,
                                                         const long long *__restrict__ offsets, int32_t *__restrict__ ids) {
    const int64_t v = blockIdx.x;
    const int lane = threadIdx.x;
    long long at = offsets[v];
    for (int64_t p0 = 0; p0 < P; p0 += 64) {
        const int64_t p = p0 + lane;
        const bool on = p < P && flags[v * P + p] != 0;
        const unsigned long long vote = __ballot(on);
        if (on) ids[at + __popcll(vote & ((1ull << lane) - 1ull))] = (int32_t)p;
        at += __popcll(vote);
    }
}

// active flags of the primitives (mask given): u8[P]
__global__ __launch_bounds__(256) void hyb_active_kernel(const uint8_t *__restrict__ mask, int64_t P, int32_t scale,
                                                         uint8_t *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    bool a = true;
    for (int t = 0; t < scale; ++t) a = a && mask[p * scale + t] != 0;
    out[p] = (uint8_t)a;
}

// sizes[i nrx + j] = nf[i] * mid * nl[j]  (mid = n^(order-2); the host checked that the total fits 62 bits)
__global__ __launch_bounds__(256) void hyb_pair_sizes_kernel(const long long *__restrict__ nf, const long long *__restrict__ nl,
                                                             int64_t ntx, int64_t nrx, long long mid,
                                                             long long *__restrict__ sizes) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p > ntx * nrx) return;
    if (p == ntx * nrx) {
        sizes[p] = 0;  // the scan's last slot: pair_offsets[npairs] = total
        return;
    }
    const int64_t i = p / nrx, j = p - i * nrx;
    sizes[p] = nf[i] * mid * nl[j];
}

// objects [n, order+2] (tx, triangle ids ..., rx) -> packed keys (tx nrx + rx) N^order + sum_j (id_j / scale) N^(order-1-j)
__global__ __launch_bounds__(256) void hyb_pack_keys_kernel(const int32_t *__restrict__ objects, int64_t n, int32_t order,
                                                            int64_t nrx, long long N, int32_t scale,
                                                            long long *__restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t *o = objects + i * (order + 2);
    long long k = (long long)o[0] * nrx + (long long)o[order + 1];
    for (int j = 0; j < order; ++j) k = k * N + (long long)(o[1 + j] / scale);
    keys[i] = k;
}

static size_t hyb_align(size_t x) { return (x + 255) / 256 * 256; }

struct HybLayout {
    size_t vis_tx, vis_rx, active, cnt, off_f, off_l, ids_f, ids_l, middle, mid_off, sizes, pair_off, scan_tmp, trace_ws,
        trace_ws_bytes, total;
};

static size_t hyb_scan_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (long long *)nullptr, (long long *)nullptr, 0ll, (size_t)(n > 0 ? n : 1),
                                  rocprim::plus<long long>(), nullptr);
    return bytes;
}

static HybLayout hyb_layout(int64_t ntx, int64_t nrx, int64_t P, int64_t max_survivors, int64_t max_paths) {
    HybLayout L{};
    size_t off = 0;
    auto take = [&](size_t b) {
        const size_t at = off;
        off += hyb_align(b > 0 ? b : 1);
        return at;
    };
    const int64_t p1 = P > 0 ? P : 1;
    L.vis_tx = take((size_t)ntx * p1);
    L.vis_rx = take((size_t)nrx * p1);
    L.active = take((size_t)p1);
    L.cnt = take((size_t)(ntx + nrx + 2) * 8);
    L.off_f = take((size_t)(ntx + 1) * 8);
    L.off_l = take((size_t)(nrx + 1) * 8);
    L.ids_f = take((size_t)ntx * p1 * 4);
    L.ids_l = take((size_t)nrx * p1 * 4);
    L.middle = take((size_t)p1 * 4);
    L.mid_off = take(16);
    L.sizes = take((size_t)(ntx * nrx + 1) * 8);
    L.pair_off = take((size_t)(ntx * nrx + 1) * 8);
    L.scan_tmp = take(hyb_scan_temp_bytes(ntx * nrx + 1));
    L.trace_ws_bytes = drt_trace_compact_workspace_size(max_survivors, max_paths);
    L.trace_ws = take(L.trace_ws_bytes);
    L.total = off;
    return L;
}

}  // namespace drt

using namespace drt;

extern "C" {

size_t drt_trace_hybrid_pairs_workspace_size(int64_t num_tx, int64_t num_rx, int64_t num_primitives,
                                             int64_t max_survivors, int64_t max_paths) {
    if (num_tx < 0) num_tx = 0;
    if (num_rx < 0) num_rx = 0;
    if (num_primitives < 0) num_primitives = 0;
    if (max_survivors < 0) max_survivors = 0;
    if (max_paths < 0) max_paths = 0;
    return hyb_layout(num_tx, num_rx, num_primitives, max_survivors, max_paths).total;
}

int32_t drt_trace_paths_hybrid_pairs(drt_mesh_t mesh, const drt_trace_params *pr, const float *tx, int64_t ntx,
                                     const float *rx, int64_t nrx, int32_t order, const uint8_t *vis_tx,
                                     const uint8_t *vis_rx, int32_t flags, int64_t max_survivors, int64_t max_paths,
                                     int64_t *keys, float *vertices, int32_t *objects, int64_t *num_valid_host,
                                     int64_t *num_evaluated_host, void *ws, size_t ws_bytes, void *stream) {
    DRT_REQUIRE(mesh && pr && num_valid_host, "null argument");
    *num_valid_host = 0;
    if (num_evaluated_host) *num_evaluated_host = 0;
    DRT_REQUIRE(ntx >= 0 && nrx >= 0 && max_survivors >= 0 && max_paths >= 0, "negative size");
    DRT_REQUIRE(order >= 2 && order <= DRT_MAX_ORDER, "per-pair pruning restricts the FIRST and the LAST interaction: order >= 2");
    DRT_REQUIRE(nrx < (1ll << 31), "too many receivers for one launch");
    const int32_t scale = mesh->assume_quads ? 2 : 1;
    const int64_t T = mesh->num_triangles, P = T / scale;
    if (ntx == 0 || nrx == 0 || P == 0) return DRT_OK;
    DRT_REQUIRE(tx && rx && vis_tx && vis_rx, "null pointer");
    const HybLayout L = hyb_layout(ntx, nrx, P, max_survivors, max_paths);
    if (!ws || ws_bytes < L.total) return fail(DRT_E_CAPACITY, "workspace too small: need %zu bytes", L.total);
    hipStream_t s = as_stream(stream);
    char *base = reinterpret_cast<char *>(ws);
    auto *pv_tx = reinterpret_cast<uint8_t *>(base + L.vis_tx);
    auto *pv_rx = reinterpret_cast<uint8_t *>(base + L.vis_rx);
    auto *active = reinterpret_cast<uint8_t *>(base + L.active);
    auto *cnt = reinterpret_cast<long long *>(base + L.cnt);  // nf[ntx], nl[nrx], n_active
    auto *off_f = reinterpret_cast<long long *>(base + L.off_f);
    auto *off_l = reinterpret_cast<long long *>(base + L.off_l);
    auto *ids_f = reinterpret_cast<int32_t *>(base + L.ids_f);
    auto *ids_l = reinterpret_cast<int32_t *>(base + L.ids_l);
    auto *middle = reinterpret_cast<int32_t *>(base + L.middle);
    auto *mid_off = reinterpret_cast<long long *>(base + L.mid_off);
    auto *sizes = reinterpret_cast<long long *>(base + L.sizes);
    auto *pair_off = reinterpret_cast<long long *>(base + L.pair_off);
    const uint8_t *mask = mesh->has_mask ? mesh->mask : nullptr;

    hipLaunchKernelGGL(hyb_prim_vis_kernel, dim3((unsigned)ceil_div(ntx * P, 256)), dim3(256), 0, s, vis_tx, ntx, T, P, scale,
                       mask, pv_tx);
    hipLaunchKernelGGL(hyb_prim_vis_kernel, dim3((unsigned)ceil_div(nrx * P, 256)), dim3(256), 0, s, vis_rx, nrx, T, P, scale,
                       mask, pv_rx);
    hipLaunchKernelGGL(hyb_row_count_kernel, dim3((unsigned)ntx), dim3(64), 0, s, pv_tx, P, cnt);
    hipLaunchKernelGGL(hyb_row_count_kernel, dim3((unsigned)nrx), dim3(64), 0, s, pv_rx, P, cnt + ntx);
    hipLaunchKernelGGL(hyb_scan_small_kernel, dim3(1), dim3(64), 0, s, cnt, ntx, off_f);
    hipLaunchKernelGGL(hyb_scan_small_kernel, dim3(1), dim3(64), 0, s, cnt + ntx, nrx, off_l);
    hipLaunchKernelGGL(hyb_compact_kernel, dim3((unsigned)ntx), dim3(64), 0, s, pv_tx, P, off_f, ids_f);
    hipLaunchKernelGGL(hyb_compact_kernel, dim3((unsigned)nrx), dim3(64), 0, s, pv_rx, P, off_l, ids_l);
    if (mask) {  // middle interactions: the active primitives (_solvers.py:1038-1042)
        hipLaunchKernelGGL(hyb_active_kernel, dim3((unsigned)ceil_div(P, 256)), dim3(256), 0, s, mask, P, scale, active);
        hipLaunchKernelGGL(hyb_row_count_kernel, dim3(1), dim3(64), 0, s, active, P, cnt + ntx + nrx);
        DRT_HIP(fill_bytes_async(mid_off, 0, 16, s));
        hipLaunchKernelGGL(hyb_compact_kernel, dim3(1), dim3(64), 0, s, active, P, mid_off, middle);
    }
    DRT_LAUNCH_CHECK();
    // the counts are a few thousand numbers: exact totals, the overflow check and the launch parameters on the host
    std::vector<long long> h((size_t)(ntx + nrx + 1), 0);
    DRT_HIP(hipMemcpyAsync(h.data(), cnt, (size_t)(ntx + nrx + (mask ? 1 : 0)) * 8, hipMemcpyDeviceToHost, s));
    DRT_HIP(hipStreamSynchronize(s));
    const int64_t n = mask ? (int64_t)h[(size_t)(ntx + nrx)] : P;
    unsigned __int128 sum_f = 0, sum_l = 0, mid = 1;
    long long max_f = 0, max_l = 0;
    for (int64_t i = 0; i < ntx; ++i) {
        sum_f += (unsigned __int128)h[(size_t)i];
        max_f = std::max(max_f, h[(size_t)i]);
    }
    for (int64_t j = 0; j < nrx; ++j) {
        sum_l += (unsigned __int128)h[(size_t)(ntx + j)];
        max_l = std::max(max_l, h[(size_t)(ntx + j)]);
    }
    for (int j = 0; j < order - 2; ++j) {
        mid *= (unsigned __int128)(n > 0 ? n : 0);
        if (mid >= ((unsigned __int128)1 << 62)) return fail(DRT_E_OVERFLOW, "per-pair candidate spaces do not fit 62 bits");
    }
    const unsigned __int128 total128 = sum_f * mid * sum_l;
    if (total128 >= ((unsigned __int128)1 << 62))
        return fail(DRT_E_OVERFLOW, "per-pair candidate spaces hold >= 2^62 rows in all: trace fewer pairs per call or lower the order");
    const int64_t total = (int64_t)total128;
    if (num_evaluated_host) *num_evaluated_host = total;
    if (total == 0) return DRT_OK;
    hipLaunchKernelGGL(hyb_pair_sizes_kernel, dim3((unsigned)ceil_div(ntx * nrx + 1, 256)), dim3(256), 0, s, cnt, cnt + ntx, ntx,
                       nrx, (long long)mid, sizes);
    size_t tb = hyb_scan_temp_bytes(ntx * nrx + 1);
    DRT_HIP(rocprim::exclusive_scan(base + L.scan_tmp, tb, sizes, pair_off, 0ll, (size_t)(ntx * nrx + 1),
                                    rocprim::plus<long long>(), s));
    const unsigned __int128 max_pair = (unsigned __int128)max_f * mid * (unsigned __int128)max_l;
    const bool small = max_pair < ((unsigned __int128)1 << 32);
    // huge pair spaces at order >= 3: the prefix kernel amortises unranking / gathers / forward images over the
    // receivers and their last interactions (measured at configs[3]: 14.2 s plain ragged, 10.6 s per-pair launches)
    bool prefix = false;
    if (order >= 3) {
        if (flags & DRT_HYBRID_PREFIX) prefix = true;
        else if (!(flags & DRT_HYBRID_RAGGED)) prefix = (double)total / (double)(ntx * nrx) > 2e7;
    }
    drt_candidates c{};
    c.table = nullptr;
    c.num_candidates = total;
    c.rank_lo = 0;
    c.num_nodes = n;
    c.node_map = mask ? middle : nullptr;
    c.order = order;
    c.first_map = ids_f;
    c.last_map = ids_l;
    c.num_first = max_f;  // grid sizing of the prefix kernel
    c.pair_offsets = reinterpret_cast<const int64_t *>(pair_off);
    c.first_offsets = reinterpret_cast<const int64_t *>(off_f);
    c.last_offsets = reinterpret_cast<const int64_t *>(off_l);
    c.reserved = (small ? 1 : 0) | (prefix ? 2 : 0);
    int32_t rc = drt_trace_paths_compact(mesh, pr, tx, ntx, rx, nrx, &c, max_survivors, max_paths, keys, vertices, objects,
                                         num_valid_host, base + L.trace_ws, L.trace_ws_bytes, stream);
    if (rc != DRT_OK) return rc;
    const int64_t nv = *num_valid_host;
    if (nv > 0) {  // ragged row indices -> self-describing packed keys (what drt_trace_paths_vjp takes with DRT_CAND_PACKED_KEYS)
        unsigned __int128 span = (unsigned __int128)ntx * (unsigned __int128)nrx;
        for (int j = 0; j < order; ++j) span *= (unsigned __int128)P;
        if (span >= ((unsigned __int128)1 << 62)) return fail(DRT_E_OVERFLOW, "tx * rx * primitives^order does not fit a 62-bit key");
        hipLaunchKernelGGL(hyb_pack_keys_kernel, dim3((unsigned)ceil_div(nv, 256)), dim3(256), 0, s, objects, nv, order, nrx,
                           (long long)P, scale, reinterpret_cast<long long *>(keys));
        DRT_LAUNCH_CHECK();
    }
    return DRT_OK;
}

}  // extern "C"
