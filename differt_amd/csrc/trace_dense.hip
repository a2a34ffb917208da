// trace_dense.hip -- the fused image-method tracer in the REFERENCE'S DENSE LAYOUT: what sits under the unchanged
// signatures `Scene.trace_paths(order, chunk_size=...)` -> `solver.trace_path_candidates(scene, candidates, types)`
// (reference geometry/_scene.py:735-764, _solvers.py:499-770, 936-957): for every (tx, rx, candidate)
//     vertices [Ntx,Nrx,C,K+2,3] f32, objects [Ntx,Nrx,C,K+2] i32, mask [Ntx,Nrx,C] u8, interaction_types [Ntx,Nrx,C,K] i32.
//
// This operator is HBM-WRITE bound (12(K+2) + 4(K+2) + 1 + 4K bytes written and 4K read per
// candidate = 81 B at K = 2, 105 B at K = 3, against ~170 VALU instructions), so the kernel is built around its
// stores:
//   * lane = candidate row, a wave = 64 consecutive rows, loops over every (tx, rx): the rows of a wave are ONE
//     contiguous segment of every output array (3072 B of vertices, 1024 B of objects, 512 B of types, 64 B of mask
//     at K = 2);
//   * each wave stages its rows in a wave-private LDS region and flushes them with 16-B nontemporal stores whose
//     chunks start on 128-B lines of the output (stores.hpp).  Measured against the direct form of round 3 (one lane
//     = one row, K+2 `st3` at a 12(K+2)-byte lane stride, 24+ lines per store instruction) on the same box,
//     profiles/r04/dense.md: 5.74 TB/s of stores against 5.25 TB/s -- the L2 already merged most of the direct
//     form's partial lines, the staged form is worth 9 % and writes the interaction types too;
//   * the mask row comes straight from the wave ballot that also drives the survivor queue (four lanes store 16 B);
//   * no block barrier anywhere: waves run free, 8 blocks per CU at K = 2 (18 KiB of LDS per block).
// The arithmetic (image chain with the reference's where-guards, inside / same-side / length / finite / active
// checks) is the compact tracer's, bit for bit; candidates that pass go to the same stage B (occlusion), which
// clears their mask byte when blocked.
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>

#include "common.hpp"
#include "geom.hpp"
#include "image_chain.hpp"
#include "mesh.hpp"
#include "stores.hpp"
#include "trace_common.hpp"
#include "trace_stages.hpp"

#pragma clang fp contract(off)

namespace drt {

enum : uint32_t { kVecVertices = 1, kVecObjects = 2, kVecTypes = 4, kVecMask = 8,
                  kRowBlocksInterleaved = 0x100 };  // lab (DRT_DENSE_LAB_MODE=1): round 4's row-block order, for the A/B

#if defined(DRT_LAB) && defined(DRT_DENSE_LAB_OCC)  // lab: occupancy the register allocator targets
#define DRT_DENSE_ATTR __attribute__((amdgpu_waves_per_eu(DRT_DENSE_LAB_OCC, DRT_DENSE_LAB_OCC)))
#else
#define DRT_DENSE_ATTR
#endif
template <int K, bool QUADS>
__global__ __launch_bounds__(256) DRT_DENSE_ATTR void trace_dense_kernel(
    TraceArgs a, const float *__restrict__ txp, const float *__restrict__ rxp, CandSrc cs,
    unsigned long long *__restrict__ q_count, long long *__restrict__ queue, int64_t q_cap, int64_t tx_per_block,
    float *__restrict__ d_vertices, int32_t *__restrict__ d_objects, uint8_t *__restrict__ d_mask,
    const int32_t *__restrict__ types_in, int32_t *__restrict__ d_types, uint32_t vec) {
    constexpr int VDW = 3 * (K + 2), ODW = K + 2, TDW = KA<K>::n;
    __shared__ __attribute__((aligned(16))) uint32_t lds_v[4 * 64 * VDW];
    __shared__ __attribute__((aligned(16))) uint32_t lds_o[4 * 64 * ODW];
    __shared__ __attribute__((aligned(16))) uint32_t lds_t[4 * 64 * TDW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *wv = lds_v + wave * 64 * VDW;
    uint32_t *wo = lds_o + wave * 64 * ODW;
    uint32_t *wt = lds_t + wave * 64 * TDW;
    const int64_t it0 = (int64_t)blockIdx.y * tx_per_block;
    const int64_t it1 = (it0 + tx_per_block < a.ntx) ? it0 + tx_per_block : a.ntx;
    // Row blocks of 256, ONE CONTIGUOUS RANGE PER XCD: workgroups go round-robin to the 8 XCDs (workgroup b -> XCD b % 8;
    // the launcher keeps gridDim.x a multiple of 8 whenever it exceeds 8), so walking index rb owns row block
    // (rb % 8) * ceil(nrb / 8) + rb / 8 -- every XCD's L2 and fabric port stream one eighth of each output array instead
    // of every eighth 12-KiB piece of all of it.  Same rows, same bytes; measured 1.05 -> 0.96 ms with the outputs packed
    // in one region and 0.79 -> 0.755 ms with the vertices 32 GiB away (profiles/r04/dense.md, scratch/dense_modes.py).
    const int64_t nrb = (cs.count + 255) / 256;
    const int64_t per_xcd = (nrb + 7) / 8;
    for (int64_t rb = (int64_t)blockIdx.x; rb < 8 * per_xcd; rb += (int64_t)gridDim.x) {
        const int64_t rbm = (vec & kRowBlocksInterleaved) ? rb : (rb % 8) * per_xcd + rb / 8;
        if (rbm >= nrb) continue;  // (wave-uniform: the padding of the last XCD's range)
        const int64_t row0 = rbm * 256;
        const int64_t row_w = row0 + wave * 64;  // wave-uniform: the wave's first row
        const int64_t left = cs.count - row_w;
        const uint32_t nrows = left >= 64 ? 64u : (left > 0 ? (uint32_t)left : 0u);
        const int64_t row = row_w + lane;
        const bool in_range = (uint32_t)lane < nrows;
        int32_t id[KA<K>::n];
        Mirrors<K, QUADS> m;
        load_candidate<K>(cs, in_range ? row : 0, id);
        load_mirrors<K, QUADS>(a, id, m);
        const bool cand_ok = in_range && m.ok;
        const uint64_t live_mask = __builtin_amdgcn_ballot_w64(cand_ok && m.active);
        if (K > 0 && d_types) {  // the interaction types of a row are the same for every (tx, rx): staged once
            wave_lds_fence();    // the previous row block's flushes have read the region
#pragma unroll
            for (int j = 0; j < K; ++j) wt[lane * TDW + j] = (types_in && in_range) ? (uint32_t)types_in[row * K + j] : 0u;
        }

        for (int64_t it = it0; it < it1; ++it) {
            const V3 tx = ld3(txp + 3 * it);  // wave-uniform scalar loads
            V3 img[KA<K>::n];
            {
                V3 prev = tx;  // forward scan, IM:191-195
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    img[j] = image_of_vertex(prev, m.p[j], m.n[j]);
                    prev = img[j];
                }
            }
            const float *prx = rxp;
            V3 rx_next = ld3(prx);
            const int nrx = (int)a.nrx;
            for (int ir = 0; ir < nrx; ++ir) {
                const V3 rx = rx_next;
                prx += (ir + 1 < nrx) ? 3 : 0;
                rx_next = ld3(prx);
                V3 full[K + 2];
                full[0] = tx;
                full[K + 1] = rx;
                {
                    // reverse scan, IM:196-201: the plain quotient when no lane of the wave needs the reference's
                    // where-guards, the guarded form (same bits for the untroubled lanes) otherwise -- see
                    // trace_filter_kernel (trace.hip) for the argument
                    V3 cur = rx;
#pragma unroll
                    for (int j = K - 1; j >= 0; --j) {
                        const V3 dir = img[j] - cur;
                        const V3 v = m.p[j] - cur;
                        const float un = dot(dir, m.n[j]);
                        const float vn = dot(v, m.n[j]);
                        const float t = vn / un;
                        if (__builtin_expect(__builtin_amdgcn_fcmpf(__builtin_fabsf(t), kInf, 4) !=
                                                 __builtin_amdgcn_read_exec(), 0)) {
                            cur = backward_step(cur, img[j], m.p[j], m.n[j]);
                        } else {
                            cur = V3{cur.x + dir.x * t, cur.y + dir.y * t, cur.z + dir.z * t};
                        }
                        full[j + 1] = cur;
                    }
                }
                const bool fin = path_finite<K>(full);
                const bool keep = fin && cand_ok;  // SV:696-699: non-finite paths are zeroed

                // ---- stage this wave's rows (vertices, objects) ----
                {
                    float *lv = reinterpret_cast<float *>(wv) + lane * VDW;
#pragma unroll
                    for (int j = 0; j < K + 2; ++j) st3(lv + 3 * j, keep ? full[j] : V3{0, 0, 0});
                    uint32_t *lo = wo + lane * ODW;
                    lo[0] = (uint32_t)it;
#pragma unroll
                    for (int j = 0; j < K; ++j) lo[1 + j] = (uint32_t)id[j];
                    lo[K + 1] = (uint32_t)ir;
                }

                // ---- validity (the mask is a pure AND, SV:715-717: the order of the checks is free) ----
                uint64_t alive_mask = live_mask;
#pragma unroll
                for (int j = K - 1; j >= 0; --j)  // inside tests as wave masks, most selective (last mirror) first
                    if (alive_mask != 0) alive_mask = inside_one_wave<K, QUADS>(m, full, j, a.eps, alive_mask);
                bool alive = false;
                if (alive_mask != 0) {
                    alive = ((alive_mask >> lane) & 1ull) && fin;
#pragma unroll
                    for (int j = 0; j < K; ++j)  // IM:443-454
                        alive = alive && same_sign(dot(full[j] - m.p[j], m.n[j]), dot(full[j + 2] - m.p[j], m.n[j]));
#pragma unroll
                    for (int s = 0; s <= K; ++s) {  // SV:684-693 (squared length)
                        const V3 d = full[s + 1] - full[s];
                        alive = alive && !(dot(d, d) < a.min_len);
                    }
                }
                const unsigned long long vote = (alive_mask != 0) ? __ballot(alive) : 0ull;

                // ---- flush ----
                wave_lds_fence();
                const int64_t rowg = ((int64_t)it * a.nrx + (int64_t)ir) * cs.count + row_w;  // wave-uniform
                if (nrows) {
                    if (vec & kVecVertices)
                        flush_region_b128<64 * VDW * 4>(wv, reinterpret_cast<char *>(d_vertices + rowg * VDW), nrows * VDW * 4, lane);
                    else
                        flush_region_b32(wv, reinterpret_cast<uint32_t *>(d_vertices + rowg * VDW), nrows * VDW, lane);
#if !(defined(DRT_LAB) && defined(DRT_DENSE_LAB_ONLYV))  // lab: vertices only (not a valid build)
                    if (vec & kVecObjects)
                        flush_region_b128<64 * ODW * 4>(wo, reinterpret_cast<char *>(d_objects + rowg * ODW), nrows * ODW * 4, lane);
                    else
                        flush_region_b32(wo, reinterpret_cast<uint32_t *>(d_objects + rowg * ODW), nrows * ODW, lane);
                    if (K > 0 && d_types) {
                        if (vec & kVecTypes)
                            flush_region_b128<64 * TDW * 4>(wt, reinterpret_cast<char *>(d_types + rowg * K), nrows * K * 4, lane);
                        else
                            flush_region_b32(wt, reinterpret_cast<uint32_t *>(d_types + rowg * K), nrows * K, lane);
                    }
                    if (vec & kVecMask) {  // C % 16 == 0: nrows is a multiple of 16, lanes 0 .. nrows/16-1 store 16 rows each
                        if ((uint32_t)lane * 16u < nrows) {
                            const uint32_t bits = (uint32_t)(vote >> (16 * lane)) & 0xffffu;
                            st_u32x4 mv;
                            mv.x = nibble_to_bytes(bits & 0xfu);
                            mv.y = nibble_to_bytes((bits >> 4) & 0xfu);
                            mv.z = nibble_to_bytes((bits >> 8) & 0xfu);
                            mv.w = nibble_to_bytes((bits >> 12) & 0xfu);
                            __builtin_nontemporal_store(mv, reinterpret_cast<st_u32x4 *>(d_mask + rowg) + lane);
                        }
                    } else if (in_range) {
                        d_mask[rowg + lane] = (uint8_t)alive;  // stage B clears it when the path is blocked
                    }
#endif
                }
                wave_lds_fence();  // flush reads precede the next iteration's staging writes

                // survivors -> stage B queue: one ballot + one atomic per wave
                if (vote) {
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(q_count, (unsigned long long)__popcll(vote));
                    base = __shfl(base, 0, 64);
                    if (alive) {
                        const unsigned long long below = vote & ((1ull << lane) - 1ull);
                        const unsigned long long slot = base + (unsigned long long)__popcll(below);
                        if ((int64_t)slot < q_cap) {
                            queue[slot] = rowg + lane;
                        } else {
                            // the queue is full (drt_trace_paths_dense_capped): this row will never be occlusion-tested, so it
                            // is not reported as a valid path either -- its mask byte, set by the flush above, is cleared
                            // (after that store has completed: both come from this wave) and the status word names the
                            // overflow.  A caller that forgets to read the status gets fewer paths, never unverified ones.
                            __builtin_amdgcn_s_waitcnt(/*vmcnt 0; expcnt, lgkmcnt: no wait*/ 0x0f70);
                            d_mask[rowg + lane] = 0;
                        }
                    }
                }
            }
        }
    }
}

static void dense_grid(const Launch &L, dim3 *grid, int64_t *tx_per_block) {
    int64_t bx = ceil_div(L.cs.count, 256);
    if (bx > 256 * 8) bx = 256 * 8;
    if (bx < 1) bx = 1;
    if (bx > 8) bx = (bx + 7) / 8 * 8;  // walking index % 8 == workgroup % 8 == XCD on every trip of the grid-stride loop
    int64_t by = 1;  // few candidates but many transmitters: split the tx loop over blockIdx.y
    if (bx < 1024 && L.a.ntx > 1) {
        by = ceil_div(2048, bx);
        if (by > L.a.ntx) by = L.a.ntx;
        if (by > 65535) by = 65535;
    }
    *tx_per_block = ceil_div(L.a.ntx, by);
    by = ceil_div(L.a.ntx, *tx_per_block);
    *grid = dim3((unsigned)bx, (unsigned)by);
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// counter word [2] = status: the survivor queue overflowed (drt_trace_paths_dense_capped)
__global__ void dense_status_kernel(unsigned long long *__restrict__ qc, int64_t qcap) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && (int64_t)qc[0] > qcap) qc[2] |= (unsigned long long)DRT_TRACE_OVERFLOW_SURVIVORS;
}

}  // namespace drt

using namespace drt;

extern "C" {

size_t drt_trace_dense_workspace_size(int64_t ntx, int64_t nrx, int64_t C) {
    if (ntx <= 0 || nrx <= 0 || C <= 0) return 64;
    return 64 + (size_t)ntx * (size_t)nrx * (size_t)C * 8;
}
size_t drt_trace_dense_capped_workspace_size(int64_t max_survivors) {
    return 64 + (size_t)(max_survivors > 0 ? max_survivors : 0) * 8;
}

int32_t drt_trace_paths_dense_ex(drt_mesh_t mesh, const drt_trace_params *pr, const float *tx, int64_t ntx,
                                 const float *rx, int64_t nrx, const drt_candidates *cands,
                                 const int32_t *types_in, float *vertices, int32_t *objects, uint8_t *mask,
                                 int32_t *types_out, void *ws, size_t ws_bytes, void *stream) {
    // the worst case: every row survives the geometric checks
    return drt_trace_paths_dense_capped(mesh, pr, tx, ntx, rx, nrx, cands, types_in, vertices, objects, mask, types_out, -1, ws,
                                        ws_bytes, stream);
}

int32_t drt_trace_paths_dense_capped(drt_mesh_t mesh, const drt_trace_params *pr, const float *tx, int64_t ntx,
                                     const float *rx, int64_t nrx, const drt_candidates *cands,
                                     const int32_t *types_in, float *vertices, int32_t *objects, uint8_t *mask,
                                     int32_t *types_out, int64_t max_survivors, void *ws, size_t ws_bytes, void *stream) {
    DRT_REQUIRE(mesh && pr && cands, "null argument");
    DRT_REQUIRE(ntx >= 0 && nrx >= 0, "negative size");
    DRT_REQUIRE(nrx < (1ll << 31), "too many receivers for one launch");
    Launch L;
    L.s = as_stream(stream);
    L.quads = mesh->assume_quads != 0;
    int32_t rc = make_cand_src(cands, L.quads ? 2 : 1, &L.cs);
    if (rc != DRT_OK) return rc;
    DRT_REQUIRE(!L.cs.ragged, "ragged pair spaces have no dense layout: use drt_trace_paths_compact");
    DRT_REQUIRE(!L.cs.packed, "packed keys address traced paths (drt_trace_paths_vjp), they are not a candidate source");
    L.a = make_args(mesh, pr, tx, ntx, rx, nrx);
    if ((pr->flags & DRT_TRACE_USE_BVH) && mesh->num_triangles > 0) {
        rc = drt_mesh_build_bvh(mesh, stream);
        if (rc != DRT_OK) return rc;
        L.bvh = reinterpret_cast<const BvhNode *>(mesh->bvh_nodes);
        L.bvh_leaf_ids = mesh->bvh_leaf_ids;
    }
    const int64_t C = L.cs.count;
    const int64_t total = ntx * nrx * C;
    if (total == 0) return DRT_OK;  // SV:566-573
    DRT_REQUIRE(tx && rx && vertices && objects && mask, "null pointer");
    // survivor queue: `max_survivors` entries (< 0: one per row, the worst case); a queue that overflows sets bit
    // DRT_TRACE_OVERFLOW_SURVIVORS of the workspace's third counter word -- never silent
    const int64_t qcap = (max_survivors < 0 || max_survivors > total) ? total : max_survivors;
    const size_t need = drt_trace_dense_capped_workspace_size(qcap);
    if (!ws || ws_bytes < need) return fail(DRT_E_CAPACITY, "workspace too small: need %zu bytes", need);
    auto *qc = reinterpret_cast<unsigned long long *>(ws);
    auto *q = reinterpret_cast<long long *>(reinterpret_cast<char *>(ws) + 64);
    DRT_HIP(fill_bytes_async(qc, 0, 64, L.s));
    const int k = cands->order;
    // the 16-byte store path of an array needs every wave segment to start on a 16-byte boundary: the array does
    // and C rows are a multiple of 16 bytes (a wave's first row is a multiple of 64)
    uint32_t vec = 0;
#ifdef DRT_LAB  // lab builds only: DRT_DENSE_LAB_MODE=1 walks the row blocks in round 4's interleaved order
    if (const char *lab = getenv("DRT_DENSE_LAB_MODE")) vec |= (atoi(lab) & 1) ? kRowBlocksInterleaved : 0u;
#endif
    if (aligned16(vertices) && (C * 12 * (k + 2)) % 16 == 0) vec |= kVecVertices;
    if (aligned16(objects) && (C * 4 * (k + 2)) % 16 == 0) vec |= kVecObjects;
    if (types_out && aligned16(types_out) && (C * 4 * k) % 16 == 0) vec |= kVecTypes;
    if (aligned16(mask) && C % 16 == 0) vec |= kVecMask;
    dim3 grid;
    int64_t tpb;
    dense_grid(L, &grid, &tpb);
    drt_trace_stats *st = pr->stats;  // optional HIP-event times of the two kernels (costs a stream synchronisation)
    StageTimer timer(st != nullptr, L.s);
    timer.mark(0);
#define CALL(K)                                                                                              \
    do {                                                                                                     \
        if (L.quads)                                                                                         \
            hipLaunchKernelGGL((trace_dense_kernel<K, true>), grid, dim3(256), 0, L.s, L.a, L.a.tx, L.a.rx,  \
                               L.cs, qc, q, qcap, tpb, vertices, objects, mask, types_in, types_out, vec);   \
        else                                                                                                 \
            hipLaunchKernelGGL((trace_dense_kernel<K, false>), grid, dim3(256), 0, L.s, L.a, L.a.tx, L.a.rx, \
                               L.cs, qc, q, qcap, tpb, vertices, objects, mask, types_in, types_out, vec);   \
        timer.mark(1);                                                                                       \
        launch_occlusion<K, true>(L, qc, q, qcap, qc + 1, nullptr, 0, mask);                                 \
        timer.mark(2);                                                                                       \
    } while (0)
    DRT_ORDER_SWITCH(k, CALL)
#undef CALL
    if (qcap < total)
        hipLaunchKernelGGL(dense_status_kernel, dim3(1), dim3(64), 0, L.s, qc, qcap);
    DRT_LAUNCH_CHECK();
    if (st) {
        unsigned long long sv[2] = {0, 0};
        DRT_HIP(hipMemcpyAsync(sv, qc, 16, hipMemcpyDeviceToHost, L.s));
        DRT_HIP(hipStreamSynchronize(L.s));
        const unsigned long long survivors = sv[0], blocked = sv[1];
        st->candidates = total;
        st->survivors = (int64_t)survivors;
        // (rows beyond a full queue are neither tested nor reported: their mask bytes are cleared)
        st->valid = (int64_t)((survivors < (unsigned long long)qcap ? survivors : (unsigned long long)qcap) - blocked);
        st->filter_ms = timer.elapsed(0, 1);
        st->occlusion_ms = timer.elapsed(1, 2);
        st->sort_emit_ms = 0.0f;
    }
    return DRT_OK;
}

int32_t drt_trace_paths_dense(drt_mesh_t mesh, const drt_trace_params *pr, const float *tx, int64_t ntx,
                              const float *rx, int64_t nrx, const drt_candidates *cands, float *vertices,
                              int32_t *objects, uint8_t *mask, void *ws, size_t ws_bytes, void *stream) {
    return drt_trace_paths_dense_ex(mesh, pr, tx, ntx, rx, nrx, cands, nullptr, vertices, objects, mask, nullptr,
                                    ws, ws_bytes, stream);
}

}  // extern "C"
