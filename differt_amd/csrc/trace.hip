// trace.hip -- the fused image-method tracer (reference geometry/_solvers.py:499-770,
// `_trace_path_candidates`, hard-mask mode) and its VJP, for gfx950.
//
// Pipeline (all device-resident, candidates are never materialised when given as a rank window):
//   stage A  trace_filter_kernel     one LANE per candidate row, looping over every (tx, rx) pair:
//            GPU unranking of the candidate (replaces differt-core's host generator), mirror
//            gather, image chain in registers, then the four cheap validity checks of the
//            reference (inside-triangle SV:637-642, same-side SV:655-659, too-small SV:684-693,
//            finite SV:696-699, active SV:544-549), most selective first with wave-uniform
//            early-outs.  Survivors are compacted into a queue with one ballot + one atomic per
//            wave.
//   stage B  trace_occlusion_kernel  one WAVEFRONT per surviving candidate: its order+1 segments
//            are tested against the whole mesh, lanes striding over LDS-staged triangle tiles
//            shared by the block's waves; ballot any-hit with early exit ("blocked", SV:676-680,
//            predicate of _utils.py:1469).
//   sort     radix sort of the (few) valid flat indices -> the stable order of
//            TracedPaths.masked_vertices (geometry/_paths.py:274-297), independent of atomics.
//   emit     trace_emit_kernel       vertices / objects of the valid paths.
//   vjp      trace_vjp_kernel        hand-derived reverse of the image chain per valid path.
// The dense mode writes the reference's full [Ntx,Nrx,C,...] layout (parity tests, small sizes).
#include <cstring>
#include <string.h>

#include <hip/hip_runtime.h>
#ifdef DRT_FILTER_DEBUG  // scratch instrumentation: how often a wave leaves the fast paths (never in the product build)
namespace drt {
__device__ unsigned long long drt_dbg_counts[8];
}
#define DRT_DBG(i) do { if ((threadIdx.x & 63) == 0) atomicAdd(&drt::drt_dbg_counts[i], 1ull); } while (0)
#define DRT_MT_DBG() DRT_DBG(3)
#else
#define DRT_DBG(i) do { } while (0)
#endif

#include <rocprim/rocprim.hpp>

#include "common.hpp"
#include "geom.hpp"
#include "image_chain.hpp"
#include "mesh.hpp"
#include "tri_tile.hpp"
#include "bvh.hpp"
#include "trace_common.hpp"
#include "sort_safe.hpp"
#include "trace_stages.hpp"

#pragma clang fp contract(off)

namespace drt {

// ------------------------------------------------------------------------------------------
// stage A
// ------------------------------------------------------------------------------------------
#ifndef DRT_FILTER_WAVES  // occupancy experiment hook (profiles/r02): waves per SIMD the register allocator targets
#define DRT_FILTER_ATTR
#else
#define DRT_FILTER_ATTR __attribute__((amdgpu_waves_per_eu(DRT_FILTER_WAVES, DRT_FILTER_WAVES)))
#endif
template <int K, bool QUADS>
__global__ __launch_bounds__(256) DRT_FILTER_ATTR void trace_filter_kernel(
    TraceArgs a, const float *__restrict__ txp, const float *__restrict__ rxp, CandSrc cs,
    unsigned long long *__restrict__ q_count,
    long long *__restrict__ queue, int64_t q_cap, int64_t tx_per_block) {
    const int lane = threadIdx.x & 63;
    const int64_t it0 = (int64_t)blockIdx.y * tx_per_block;
    const int64_t it1 = (it0 + tx_per_block < a.ntx) ? it0 + tx_per_block : a.ntx;
    for (int64_t row0 = (int64_t)blockIdx.x * 256; row0 < cs.count; row0 += (int64_t)gridDim.x * 256) {
        const int64_t row = row0 + threadIdx.x;
        const bool in_range = row < cs.count;
        int32_t id[KA<K>::n];
        Mirrors<K, QUADS> m;
        load_candidate<K>(cs, in_range ? row : 0, id);
        load_mirrors<K, QUADS>(a, id, m);
        const bool cand_ok = in_range && m.ok;
        // lanes that can survive at all, as a wave mask: the per-receiver tests below stay on the scalar unit
        const uint64_t live_mask = __builtin_amdgcn_ballot_w64(cand_ok && m.active);

        for (int64_t it = it0; it < it1; ++it) {
            // txp / rxp are separate `const __restrict__` kernel arguments so that these wave-uniform
            // reads become scalar loads (through the by-value TraceArgs they were per-lane global loads
            // with a vmcnt(0) stall in every iteration)
            const V3 tx = ld3(txp + 3 * it);
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
            const int nrx = (int)a.nrx;  // < 2^31 (checked by the launcher); 32-bit scalar loop control
            for (int ir = 0; ir < nrx; ++ir) {
                // the next receiver's scalar loads are in flight during this one's arithmetic (the last
                // one is re-read)
                const V3 rx = rx_next;
                prx += (ir + 1 < nrx) ? 3 : 0;
                rx_next = ld3(prx);
                V3 full[K + 2];
                full[0] = tx;
                full[K + 1] = rx;
                {
                    // reverse scan, IM:196-201.  Fast path per mirror when NO lane of the wave needs
                    // the reference's where-guards (parallel ray IM:123-135, infinite previous point
                    // IM:165-181): the plain t = vn / un; x = o + d * t, which is exactly what the
                    // guarded form computes in that case.  Otherwise the whole wave runs the guarded
                    // form (bit-identical for the untroubled lanes).
                    V3 cur = rx;
#pragma unroll
                    for (int j = K - 1; j >= 0; --j) {
                        const V3 dir = img[j] - cur;
                        const V3 v = m.p[j] - cur;
                        const float un = dot(dir, m.n[j]);
                        const float vn = dot(v, m.n[j]);
                        // Both guard cases leave a non-finite quotient: un == 0 gives +-inf or NaN; a
                        // non-finite previous point makes BOTH differences to it, hence un and vn,
                        // non-finite (a term +-inf * n_i is +-inf or NaN and nothing finite cancels it), and
                        // inf / inf = NaN.  One compare of the speculative quotient covers them; a finite
                        // overflow only sends the wave to the guarded form (same bits).
                        const float t = vn / un;
                        // (llvm.amdgcn.fcmp, predicate 4 = ordered less-than: a plain v_cmp whose lane mask
                        // stays on the scalar unit; `fabs(t) < inf` as an expression becomes a class test
                        // plus a v_cndmask / v_cmp pair to rebuild the mask)
                        if (__builtin_expect(__builtin_amdgcn_fcmpf(__builtin_fabsf(t), kInf, 4) !=
                                                 __builtin_amdgcn_read_exec(), 0)) {
                            DRT_DBG(1 + j);
                            cur = backward_step(cur, img[j], m.p[j], m.n[j]);
                        } else {
                            cur = V3{cur.x + dir.x * t, cur.y + dir.y * t, cur.z + dir.z * t};
                        }
                        full[j + 1] = cur;
                    }
                }
                // most selective test first: the last reflection point lies in its triangle
                uint64_t alive_mask = live_mask;
                if (K > 0) alive_mask = inside_one_wave<K, QUADS>(m, full, K - 1, a.eps, live_mask);
                bool alive = false, fin = true;
                DRT_DBG(0);
                // Some lane passes the last mirror's test in 9.4 % of the wave-iterations of configs[2]
                // (scratch/filter_debug.py) -- not rare enough to run all remaining checks for: the other
                // mirrors' inside tests come next, still as wave masks, and the wave leaves as soon as the mask
                // is empty (the checks are independent, so their order cannot change the result)
#pragma unroll
                for (int j = K - 2; j >= 0; --j)
                    if (alive_mask != 0) alive_mask = inside_one_wave<K, QUADS>(m, full, j, a.eps, alive_mask);
                if (alive_mask != 0) {
                    DRT_DBG(4);
                    alive = (alive_mask >> lane) & 1ull;
                    fin = path_finite<K>(full);
                    alive = alive && fin;
#pragma unroll
                    for (int j = 0; j < K; ++j)  // IM:443-454
                        alive = alive && same_sign(dot(full[j] - m.p[j], m.n[j]),
                                                   dot(full[j + 2] - m.p[j], m.n[j]));
#pragma unroll
                    for (int s = 0; s <= K; ++s) {  // SV:684-693 (squared length)
                        const V3 d = full[s + 1] - full[s];
                        alive = alive && !(dot(d, d) < a.min_len);
                    }
                }
                const int64_t flat = (it * a.nrx + (int64_t)ir) * cs.count + row;
                // wave-level compaction of the survivors: one ballot + one atomic per wave
                const unsigned long long vote = (alive_mask != 0) ? __ballot(alive) : 0ull;
                if (vote) {
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(q_count, (unsigned long long)__popcll(vote));
                    base = __shfl(base, 0, 64);
                    if (alive) {
                        const unsigned long long below = vote & ((1ull << lane) - 1ull);
                        const unsigned long long slot = base + (unsigned long long)__popcll(below);
                        if ((int64_t)slot < q_cap) queue[slot] = flat;
                    }
                }
            }
        }
    }
}

// Stage A for RAGGED per-pair candidate spaces (CandSrc::ragged): one lane per (pair, candidate) row of
// the concatenated spaces -- no (tx, rx) loops, the pair comes out of the row index.  Same checks, same
// arithmetic (image_chain is the guarded form the fast path above is identical to).
template <int K, bool QUADS>
__global__ __launch_bounds__(256) void trace_filter_ragged_kernel(TraceArgs a, CandSrc cs,
                                                                  unsigned long long *__restrict__ q_count,
                                                                  long long *__restrict__ queue, int64_t q_cap) {
    const int lane = threadIdx.x & 63;
    for (int64_t g0 = (int64_t)blockIdx.x * 256; g0 < cs.count; g0 += (int64_t)gridDim.x * 256) {
        const int64_t g = g0 + threadIdx.x;
        const bool in_range = g < cs.count;
        int64_t it = 0, ir = 0;
        int32_t id[KA<K>::n];
        ragged_decode<K>(cs, a.nrx, in_range ? g : 0, it, ir, id);
        Mirrors<K, QUADS> m;
        load_mirrors<K, QUADS>(a, id, m);
        V3 full[K + 2];
        full[0] = ld3(a.tx + 3 * it);
        full[K + 1] = ld3(a.rx + 3 * ir);
        if constexpr (K > 0) {
            V3 path[KA<K>::n];
            image_chain<KA<K>::n>(full[0], full[K + 1], m.p, m.n, path);
#pragma unroll
            for (int j = 0; j < K; ++j) full[j + 1] = path[j];
        }
        bool alive = in_range && m.ok && m.active;
        if (K > 0) alive = alive && inside_one<K, QUADS>(m, full, K - 1, a.eps);
        if (__any(alive)) {
            alive = alive && path_finite<K>(full);
#pragma unroll
            for (int j = K - 2; j >= 0; --j) alive = alive && inside_one<K, QUADS>(m, full, j, a.eps);
#pragma unroll
            for (int j = 0; j < K; ++j)
                alive = alive && same_sign(dot(full[j] - m.p[j], m.n[j]), dot(full[j + 2] - m.p[j], m.n[j]));
#pragma unroll
            for (int sgm = 0; sgm <= K; ++sgm) {
                const V3 d = full[sgm + 1] - full[sgm];
                alive = alive && !(dot(d, d) < a.min_len);
            }
        }
        const unsigned long long vote = __ballot(alive);
        if (vote) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(q_count, (unsigned long long)__popcll(vote));
            base = __shfl(base, 0, 64);
            if (alive) {
                const unsigned long long below = vote & ((1ull << lane) - 1ull);
                const unsigned long long slot = base + (unsigned long long)__popcll(below);
                if ((int64_t)slot < q_cap) queue[slot] = g;
            }
        }
    }
}

// Stage A for a per-pair table of COPLANAR-PAIR BLOCKS (DRT_CAND_PAIR_BLOCKS, include/differt_amd.h): lane = block
// of 2^K rows that name the triangle choices (first or second triangle, bit_j(c)) of one sequence of primitives, each a
// pair of triangles that are the same mirror (equal first vertex, equal unit normal up to the sign of zero components)
// or a single triangle.  All rows of the block have the same images and reflection points as VALUES: the image method
// is sums, products and guarded quotients of the mirror's point and normal, none of which turns the sign of a zero into
// a different number, and every later decision (Moller-Trumbore's ranges with its |a| > eps guard, jnp.sign of the
// same-side test, the squared lengths, isfinite) is a comparison, which does not see that sign either.  The chain is
// evaluated ONCE with the first triangle's mirror, Moller-Trumbore runs against both triangles of every pair, and row c
// survives iff its triangle passes at every mirror -- the decisions trace_filter_ragged_kernel takes row by row (8 chains
// at order 3), the same queue of global table rows.  Stage B and the emit stage read each surviving row's own triangles
// from the table, as before.  The block's FIRST row names the first triangles, its LAST row the second ones; an id of a
// padding row is stored as -2 - id (beam.hip, rows_expand_pairs_kernel), -1 = no such triangle / a whole padding block.
__device__ __forceinline__ int32_t pairblock_id(int32_t x) { return (x >= 0) ? x : ((x == -1) ? -1 : -2 - x); }
template <int K>
__global__ __launch_bounds__(256) void trace_filter_pairblocks_kernel(TraceArgs a, CandSrc cs,
                                                                      unsigned long long *__restrict__ q_count,
                                                                      long long *__restrict__ queue, int64_t q_cap) {
    constexpr int COMBOS = 1 << K;
    const int lane = threadIdx.x & 63;
    const int64_t nblocks = cs.count >> K;
    for (int64_t b0 = (int64_t)blockIdx.x * 256; b0 < nblocks; b0 += (int64_t)gridDim.x * 256) {
        const int64_t b = b0 + threadIdx.x;
        const bool in_range = b < nblocks;
        const int64_t g0 = (in_range ? b : 0) << K;
        int64_t lo = 0, hi = cs.npairs;  // pair of the block: largest lo with pair_offsets[lo] <= g0
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (cs.pair_offsets[mid] <= g0) lo = mid; else hi = mid;
        }
        const int64_t it = lo / a.nrx, ir = lo - it * a.nrx;
        int32_t t0[K], t1[K];  // the two triangles of every primitive (t1 = -1: a single triangle)
        bool ok = in_range;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            t0[j] = pairblock_id(cs.table[g0 * K + j]);
            t1[j] = pairblock_id(cs.table[(g0 + COMBOS - 1) * K + j]);
            ok = ok && t0[j] >= 0 && (int64_t)t0[j] < a.T && (int64_t)t1[j] < a.T;
            if (!ok) t0[j] = 0, t1[j] = -1;
        }
        V3 p[K], n[K];
        bool active = true;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int64_t s = (int64_t)t0[j];
            p[j] = ld3(a.tri_verts + 9 * s);
            n[j] = ld3(a.normals + 3 * s);
            if (a.mask) active = active && (a.mask[s] != 0) && (t1[j] < 0 || a.mask[t1[j]] != 0);
        }
        V3 full[K + 2];
        full[0] = ld3(a.tx + 3 * it);
        full[K + 1] = ld3(a.rx + 3 * ir);
        {
            V3 path[K];
            image_chain<K>(full[0], full[K + 1], p, n, path);
#pragma unroll
            for (int j = 0; j < K; ++j) full[j + 1] = path[j];
        }
        bool alive = ok && active;
        // inside tests, last mirror first (the most selective); hit[j] bit t = triangle 2 q_j + t contains X_{j+1}
        uint32_t hit[K];
#pragma unroll
        for (int j = 0; j < K; ++j) hit[j] = 0u;
#pragma unroll
        for (int j = K - 1; j >= 0; --j) {
            if (!__any(alive)) break;
            const V3 o = full[j];
            const V3 d = full[j + 1] - full[j];
            float t;
            const bool h0 = moller_trumbore(o, d, load_tri(a.tri_verts + 9 * (int64_t)t0[j]), a.eps, t);
            // (a single triangle: the test runs on the first triangle again and is masked -- no divergent branch)
            const bool h1 = moller_trumbore(o, d, load_tri(a.tri_verts + 9 * (int64_t)(t1[j] >= 0 ? t1[j] : t0[j])), a.eps, t) &&
                            t1[j] >= 0;
            hit[j] = (h0 ? 1u : 0u) | (h1 ? 2u : 0u);
            alive = alive && hit[j] != 0u;
        }
        if (__any(alive)) {
            alive = alive && path_finite<K>(full);
#pragma unroll
            for (int j = 0; j < K; ++j)
                alive = alive && same_sign(dot(full[j] - p[j], n[j]), dot(full[j + 2] - p[j], n[j]));
#pragma unroll
            for (int sgm = 0; sgm <= K; ++sgm) {
                const V3 d = full[sgm + 1] - full[sgm];
                alive = alive && !(dot(d, d) < a.min_len);
            }
        }
        if (!__any(alive)) continue;
        // the rows of the block that survive: every mirror's chosen triangle is hit, no triangle named twice in a row
        uint32_t rows = 0u;
#pragma unroll
        for (int c = 0; c < COMBOS; ++c) {
            bool v = alive;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const uint32_t bit = (uint32_t)(c >> (K - 1 - j)) & 1u;
                v = v && ((hit[j] >> bit) & 1u);
                if (j > 0) {  // no triangle named twice in a row
                    const uint32_t pbit = (uint32_t)(c >> (K - j)) & 1u;
                    v = v && ((bit ? t1[j] : t0[j]) != (pbit ? t1[j - 1] : t0[j - 1]));
                }
            }
            rows |= v ? (1u << c) : 0u;
        }
        // one atomic per wave; lanes take consecutive slots for their rows
        const uint32_t mine = (uint32_t)__popc(rows);
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        const uint32_t total = __shfl(incl, 63, 64);
        if (total == 0u) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(q_count, (unsigned long long)total);
        base = __shfl(base, 0, 64);
        unsigned long long slot = base + (unsigned long long)(incl - mine);
        uint32_t r = rows;
        while (r) {
            const int c = __builtin_ctz(r);
            r &= r - 1u;
            if ((int64_t)slot < q_cap) queue[slot] = g0 + c;
            ++slot;
        }
    }
}

// Stage A for ragged per-pair spaces of order >= 3 with LARGE pair spaces: lane = PREFIX (the first K-1
// interactions of transmitter blockIdx.y: F_tx x N^(K-2) rows), inner loops over the receivers and over
// each receiver's visible last interactions.  The unranking, the K-1 mirror gathers and the forward images
// are paid once per prefix instead of once per candidate, the last mirror is wave-uniform (scalar loads).
// Emits the same keys as trace_filter_ragged_kernel (global ragged rows), same arithmetic.
template <int K, bool QUADS>
__global__ __launch_bounds__(256) void trace_filter_prefix_kernel(TraceArgs a, CandSrc cs,
                                                                  unsigned long long *__restrict__ q_count,
                                                                  long long *__restrict__ queue, int64_t q_cap) {
    static_assert(K >= 2, "prefix kernel needs at least two interactions");
    const int lane = threadIdx.x & 63;
    const int64_t it = blockIdx.y;
    const int32_t *F = cs.first_map + cs.first_off[it];
    const int64_t nF = cs.first_off[it + 1] - cs.first_off[it];
    const int64_t nN = cs.num_nodes;
    const int64_t prefixes = nF * cs.mid_pw;
    const V3 tx = ld3(a.tx + 3 * it);
    for (int64_t row0 = (int64_t)blockIdx.x * 256; row0 < prefixes; row0 += (int64_t)gridDim.x * 256) {
        const int64_t row = row0 + threadIdx.x;
        const bool in_range = row < prefixes;
        // prefix digits, most significant first: f, then K-2 middle digits base N
        int32_t id[KA<K>::n];
        bool bad = false;
        {
            uint64_t r = (uint64_t)(in_range ? row : 0);
            int32_t dig[KA<K>::n];
#pragma unroll
            for (int j = K - 2; j >= 1; --j) {
                const uint64_t q = r / (uint64_t)nN;
                const uint64_t d = r - q * (uint64_t)nN;
                dig[j] = cs.node_map ? cs.node_map[d] : (int32_t)d;
                r = q;
            }
            dig[0] = F[nF > 0 ? r : 0];
#pragma unroll
            for (int j = 1; j <= K - 2; ++j) bad = bad || (dig[j] == dig[j - 1]);
#pragma unroll
            for (int j = 0; j <= K - 2; ++j) id[j] = dig[j] * cs.id_scale;
        }
        Mirrors<K, QUADS> m;
        bool prefix_ok = in_range && !bad, prefix_active = true;
#pragma unroll
        for (int j = 0; j <= K - 2; ++j) prefix_ok = load_one_mirror<K, QUADS>(a, id[j], j, m, prefix_active) && prefix_ok;
        V3 img[KA<K>::n];
        {
            V3 prev = tx;
#pragma unroll
            for (int j = 0; j <= K - 2; ++j) {
                img[j] = image_of_vertex(prev, m.p[j], m.n[j]);
                prev = img[j];
            }
        }
        for (int64_t ir = 0; ir < a.nrx; ++ir) {
            const V3 rx = ld3(a.rx + 3 * ir);
            const int32_t *L = cs.last_map + cs.last_off[ir];
            const int64_t nL = cs.last_off[ir + 1] - cs.last_off[ir];
            const int64_t key0 = cs.pair_offsets[it * a.nrx + ir] + row * nL;
            for (int64_t l = 0; l < nL; ++l) {
                const int32_t last = L[l];  // wave-uniform
                id[K - 1] = last * cs.id_scale;
                bool active = prefix_active;
                const bool last_ok = load_one_mirror<K, QUADS>(a, id[K - 1], K - 1, m, active);
                img[K - 1] = image_of_vertex(img[K - 2], m.p[K - 1], m.n[K - 1]);
                V3 full[K + 2];
                full[0] = tx;
                full[K + 1] = rx;
                V3 cur = rx;
#pragma unroll
                for (int j = K - 1; j >= 0; --j) {
                    cur = backward_step(cur, img[j], m.p[j], m.n[j]);
                    full[j + 1] = cur;
                }
                bool alive = prefix_ok && last_ok && active && (id[K - 1] != id[K - 2]);
                alive = alive && inside_one<K, QUADS>(m, full, K - 1, a.eps);
                if (__any(alive)) {
                    alive = alive && path_finite<K>(full);
#pragma unroll
                    for (int j = K - 2; j >= 0; --j) alive = alive && inside_one<K, QUADS>(m, full, j, a.eps);
#pragma unroll
                    for (int j = 0; j < K; ++j)
                        alive = alive &&
                                same_sign(dot(full[j] - m.p[j], m.n[j]), dot(full[j + 2] - m.p[j], m.n[j]));
#pragma unroll
                    for (int sgm = 0; sgm <= K; ++sgm) {
                        const V3 d = full[sgm + 1] - full[sgm];
                        alive = alive && !(dot(d, d) < a.min_len);
                    }
                }
                const unsigned long long vote = __ballot(alive);
                if (vote) {
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(q_count, (unsigned long long)__popcll(vote));
                    base = __shfl(base, 0, 64);
                    if (alive) {
                        const unsigned long long below = vote & ((1ull << lane) - 1ull);
                        const unsigned long long slot = base + (unsigned long long)__popcll(below);
                        if ((int64_t)slot < q_cap) queue[slot] = key0 + l;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// emit: vertices / objects of the sorted valid keys
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void trace_emit_kernel(TraceArgs a, CandSrc cs,
                                                         const long long *__restrict__ keys,
                                                         int64_t num, float *__restrict__ vertices,
                                                         int32_t *__restrict__ objects) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= num) return;
    int64_t it, ir;
    int32_t id[KA<K>::n];
    V3 p[KA<K>::n], n[KA<K>::n], full[K + 2];
    key_to_path<K>(a, cs, keys[i], it, ir, id, p, n, full);
#pragma unroll
    for (int j = 0; j < K + 2; ++j) st3(vertices + (i * (K + 2) + j) * 3, full[j]);
    int32_t *ob = objects + i * (K + 2);
    ob[0] = (int32_t)it;
#pragma unroll
    for (int j = 0; j < K; ++j) ob[1 + j] = id[j];
    ob[K + 1] = (int32_t)ir;
}

// Static-shape variant for the no-sync entry point: `cap` rows are always written.  Rows at or beyond
// the device-side valid count (their sorted key is a sentinel >= 2^62) become padding: key -1,
// vertices 0, objects -1.  Also folds the two overflow conditions into the status word.
constexpr long long kKeySentinelMin = 1ll << 62;

template <int K>
__global__ __launch_bounds__(256) void trace_emit_padded_kernel(TraceArgs a, CandSrc cs,
                                                                long long *__restrict__ keys, int64_t cap,
                                                                int64_t q_cap,
                                                                const unsigned long long *__restrict__ counters,
                                                                long long *__restrict__ counts_out,
                                                                float *__restrict__ vertices,
                                                                int32_t *__restrict__ objects) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && counts_out) {
        const long long surv = (long long)counters[0], valid = (long long)counters[1];
        counts_out[0] = surv;
        counts_out[1] = valid;
        counts_out[2] = (surv > q_cap ? DRT_TRACE_OVERFLOW_SURVIVORS : 0) | (valid > cap ? DRT_TRACE_OVERFLOW_PATHS : 0);
        counts_out[3] = 0;
    }
    if (i >= cap) return;
    const long long key = keys[i];
    float *v = vertices + i * (K + 2) * 3;
    int32_t *ob = objects + i * (K + 2);
    if (key >= kKeySentinelMin || key < 0) {
        keys[i] = -1;
#pragma unroll
        for (int j = 0; j < (K + 2) * 3; ++j) v[j] = 0.0f;
#pragma unroll
        for (int j = 0; j < K + 2; ++j) ob[j] = -1;
        return;
    }
    int64_t it, ir;
    int32_t id[KA<K>::n];
    V3 p[KA<K>::n], n[KA<K>::n], full[K + 2];
    key_to_path<K>(a, cs, key, it, ir, id, p, n, full);
#pragma unroll
    for (int j = 0; j < K + 2; ++j) st3(v + j * 3, full[j]);
    ob[0] = (int32_t)it;
#pragma unroll
    for (int j = 0; j < K; ++j) ob[1 + j] = id[j];
    ob[K + 1] = (int32_t)ir;
}

// ------------------------------------------------------------------------------------------
// VJP: cotangent of the (K+2) path vertices -> tx, rx and mesh-vertex gradients
// ------------------------------------------------------------------------------------------

template <int K>
__global__ __launch_bounds__(256) void trace_vjp_kernel(
    TraceArgs a, CandSrc cs, const float *__restrict__ mesh_vertices,
    const int32_t *__restrict__ mesh_triangles, const long long *__restrict__ keys,
    const float *__restrict__ cot, int64_t num, float *__restrict__ g_tx, float *__restrict__ g_rx,
    float *__restrict__ g_vertices) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= num) return;
    int64_t it, ir;
    int32_t id[KA<K>::n];
    V3 p[KA<K>::n], n[KA<K>::n], full[K + 2];
    // padding rows and non-finite paths have constant (zeroed) vertices: no gradient (SV:696-699);
    // key -1 = padding row of the static-shape (async) compact output
    const long long key = keys[i];
    if (key < 0) return;
    if (!key_to_path<K>(a, cs, key, it, ir, id, p, n, full) || !path_finite<K>(full)) return;
    const float *g = cot + i * (K + 2) * 3;
    V3 tx_bar = ld3(g);                  // vertices[0] = tx
    V3 rx_bar = ld3(g + 3 * (K + 1));    // vertices[K+1] = rx
    if constexpr (K > 0) {
        V3 gp[KA<K>::n], pb[KA<K>::n], nb[KA<K>::n], fb, tb;
#pragma unroll
        for (int j = 0; j < K; ++j) gp[j] = ld3(g + 3 * (j + 1));
        image_chain_vjp<KA<K>::n>(full[0], full[K + 1], p, n, gp, fb, tb, pb, nb);
        tx_bar = tx_bar + fb;
        rx_bar = rx_bar + tb;
        if (g_vertices) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                mirror_vjp_to_mesh(mesh_vertices, mesh_triangles, id[j], pb[j], nb[j], g_vertices);
            }
        }
    }
    if (g_tx) atomic_add3(g_tx + 3 * it, tx_bar);
    if (g_rx) atomic_add3(g_rx + 3 * ir, rx_bar);
}

// Deterministic variant (DRT_TRACE_DETERMINISTIC_GRAD, SURVEY.md section 7 hard part 6): float atomics add in
// arrival order, so two runs differ in the last bits.  Here every path WRITES its contributions -- slot 0 its
// transmitter's, slot 1 its receiver's, slots 2.. the three vertices of each mirror -- next to a 64-bit
// destination key (space << 40 | index; all ones = nothing); a STABLE radix sort groups them by destination with the
// path order kept inside a group, and one lane per group adds them up in that fixed order.
template <int K>
__global__ __launch_bounds__(256) void trace_vjp_contrib_kernel(
    TraceArgs a, CandSrc cs, const float *__restrict__ mesh_vertices, const int32_t *__restrict__ mesh_triangles,
    const long long *__restrict__ keys, const float *__restrict__ cot, int64_t num, int want_tx, int want_rx,
    int want_vertices, unsigned long long *__restrict__ dest, uint32_t *__restrict__ slot_ids,
    float *__restrict__ vecs) {
    constexpr int S = 2 + 3 * K;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= num) return;
    unsigned long long dk[S];
    V3 dv[S];
#pragma unroll
    for (int q = 0; q < S; ++q) {
        dk[q] = ~0ull;
        dv[q] = V3{0, 0, 0};
    }
    int64_t it, ir;
    int32_t id[KA<K>::n];
    V3 p[KA<K>::n], n[KA<K>::n], full[K + 2];
    const long long key = keys[i];
    if (key >= 0 && key_to_path<K>(a, cs, key, it, ir, id, p, n, full) && path_finite<K>(full)) {
        const float *g = cot + i * (K + 2) * 3;
        V3 tx_bar = ld3(g), rx_bar = ld3(g + 3 * (K + 1));
        if constexpr (K > 0) {
            V3 gp[KA<K>::n], pb[KA<K>::n], nb[KA<K>::n], fb, tb;
#pragma unroll
            for (int j = 0; j < K; ++j) gp[j] = ld3(g + 3 * (j + 1));
            image_chain_vjp<KA<K>::n>(full[0], full[K + 1], p, n, gp, fb, tb, pb, nb);
            tx_bar = tx_bar + fb;
            rx_bar = rx_bar + tb;
            if (want_vertices) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    int32_t vi[3];
                    V3 vv[3];
                    mirror_vjp_contrib(mesh_vertices, mesh_triangles, id[j], pb[j], nb[j], vi, vv);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        dk[2 + 3 * j + k] = (2ull << 40) | (unsigned long long)(uint32_t)vi[k];
                        dv[2 + 3 * j + k] = vv[k];
                    }
                }
            }
        }
        if (want_tx) {
            dk[0] = (unsigned long long)it;
            dv[0] = tx_bar;
        }
        if (want_rx) {
            dk[1] = (1ull << 40) | (unsigned long long)ir;
            dv[1] = rx_bar;
        }
    }
#pragma unroll
    for (int q = 0; q < S; ++q) {
        const int64_t slot = i * S + q;
        dest[slot] = dk[q];
        slot_ids[slot] = (uint32_t)slot;
        st3(vecs + 3 * slot, dv[q]);
    }
}

// Ordered sum of every destination group in TWO levels, so that a group of g contributions costs one lane g / 256 + 256
// dependent adds instead of g (one transmitter with 1e6 paths: 4 000 instead of 1e6): the sorted list is
// cut into fixed chunks of 256; level 1 sums, in order, the part of a group that CONTINUES from the previous chunk
// (at most one such run per chunk: the leading one) into carry[chunk]; level 2, one lane per group head, sums the
// group's elements inside the head's chunk in order and then the carries of the following chunks in order.  The
// association is fixed by the positions alone: bit-identical from run to run.
constexpr int kVjpChunk = 256;
__global__ __launch_bounds__(256) void trace_vjp_carry_kernel(const unsigned long long *__restrict__ dest_sorted,
                                                              const uint32_t *__restrict__ slots_sorted,
                                                              const float *__restrict__ vecs, int64_t n,
                                                              float *__restrict__ carry) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;  // chunk
    const int64_t i0 = c * kVjpChunk;
    if (c == 0 || i0 >= n) return;
    const unsigned long long d = dest_sorted[i0];
    V3 acc{0, 0, 0};
    if (d != ~0ull && dest_sorted[i0 - 1] == d) {
        const int64_t end = (i0 + kVjpChunk < n) ? i0 + kVjpChunk : n;
        acc = ld3(vecs + 3 * (int64_t)slots_sorted[i0]);
        for (int64_t j = i0 + 1; j < end && dest_sorted[j] == d; ++j) acc = acc + ld3(vecs + 3 * (int64_t)slots_sorted[j]);
    }
    st3(carry + 3 * c, acc);
}

__global__ __launch_bounds__(256) void trace_vjp_reduce_kernel(const unsigned long long *__restrict__ dest_sorted,
                                                               const uint32_t *__restrict__ slots_sorted,
                                                               const float *__restrict__ vecs, int64_t n,
                                                               const float *__restrict__ carry,
                                                               float *__restrict__ g_tx, float *__restrict__ g_rx,
                                                               float *__restrict__ g_vertices) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long d = dest_sorted[i];
    if (d == ~0ull || (i > 0 && dest_sorted[i - 1] == d)) return;  // not the head of a group
    const int64_t chunk_end = (i / kVjpChunk + 1) * kVjpChunk;
    V3 acc = ld3(vecs + 3 * (int64_t)slots_sorted[i]);
    int64_t j = i + 1;
    for (; j < n && j < chunk_end && dest_sorted[j] == d; ++j) acc = acc + ld3(vecs + 3 * (int64_t)slots_sorted[j]);
    // the group runs on into the next chunks: their leading runs were summed by trace_vjp_carry_kernel
    for (int64_t c = chunk_end / kVjpChunk; j == c * kVjpChunk && j < n && dest_sorted[j] == d; ++c) {
        acc = acc + ld3(carry + 3 * c);
        const int64_t last = ((c + 1) * kVjpChunk < n ? (c + 1) * kVjpChunk : n) - 1;
        j = (dest_sorted[last] == d) ? last + 1 : j + 1;  // (j + 1: any value off the next chunk boundary ends the loop)
    }
    const int space = (int)(d >> 40);
    float *out = ((space == 0) ? g_tx : ((space == 1) ? g_rx : g_vertices)) + 3 * (int64_t)(d & 0xffffffffffull);
    // gradients are ACCUMULATED into the caller's buffers: the group's only writer adds its total
    out[0] += acc.x;
    out[1] += acc.y;
    out[2] += acc.z;
}

// (a12) GPU-resident candidate table
template <int K>
__global__ __launch_bounds__(256) void candidates_fill_kernel(CandSrc cs, int32_t *__restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= cs.count) return;
    int32_t id[KA<K>::n];
    load_candidate<K>(cs, row, id);
#pragma unroll
    for (int j = 0; j < K; ++j) out[row * K + j] = id[j];
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static void filter_grid(const Launch &L, dim3 *grid, int64_t *tx_per_block) {
    int64_t bx = ceil_div(L.cs.count, 256);
    if (bx > 256 * 8) bx = 256 * 8;
    if (bx < 1) bx = 1;
    // few candidates but many transmitters: split the tx loop over blockIdx.y
    int64_t by = 1;
    if (bx < 1024 && L.a.ntx > 1) {
        by = ceil_div(2048, bx);
        if (by > L.a.ntx) by = L.a.ntx;
        if (by > 65535) by = 65535;
    }
    *tx_per_block = ceil_div(L.a.ntx, by);
    by = ceil_div(L.a.ntx, *tx_per_block);
    *grid = dim3((unsigned)bx, (unsigned)by);
}

template <int K, bool QUADS>
static void launch_filter(const Launch &L, unsigned long long *qc, long long *q, int64_t qcap) {
    if (L.cs.ragged) {
        if constexpr (K >= 3) {
            if (L.cs.prefix_kernel) {
                // rows per transmitter = F_tx * N^(K-2) <= num_first_max * mid_pw: size the grid for the largest
                int64_t bx = ceil_div(L.cs.max_prefixes, 256);
                if (bx > 256 * 16) bx = 256 * 16;
                hipLaunchKernelGGL((trace_filter_prefix_kernel<K, QUADS>), dim3((unsigned)(bx < 1 ? 1 : bx), (unsigned)L.a.ntx),
                                   dim3(256), 0, L.s, L.a, L.cs, qc, q, qcap);
                return;
            }
        }
        if constexpr (K >= 1 && K <= 4 && !QUADS) {
            if (L.cs.pair_blocks) {  // coplanar-pair blocks: one lane per block of 2^K rows
                int64_t bx = ceil_div(L.cs.count >> K, 256);
                if (bx > 256 * 32) bx = 256 * 32;
                hipLaunchKernelGGL((trace_filter_pairblocks_kernel<K>), dim3((unsigned)(bx < 1 ? 1 : bx)), dim3(256), 0, L.s,
                                   L.a, L.cs, qc, q, qcap);
                return;
            }
        }
        if constexpr (K >= 1) {  // K == 1 only occurs with a per-pair TABLE (beam-pruned rows)
            int64_t bx = ceil_div(L.cs.count, 256);
            if (bx > 256 * 32) bx = 256 * 32;
            hipLaunchKernelGGL((trace_filter_ragged_kernel<K, QUADS>), dim3((unsigned)(bx < 1 ? 1 : bx)), dim3(256), 0,
                               L.s, L.a, L.cs, qc, q, qcap);
        }
        return;
    }
    dim3 grid;
    int64_t tpb;
    filter_grid(L, &grid, &tpb);
    hipLaunchKernelGGL((trace_filter_kernel<K, QUADS>), grid, dim3(256), 0, L.s, L.a, L.a.tx,
                       L.a.rx, L.cs, qc, q, qcap, tpb);
}

static size_t sort_temp_bytes(int64_t n) {  // enough for the default and for the capture-safe sorts (sort_safe.hpp)
    if (n <= 0) return 0;
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (unsigned long long *)nullptr,
                                   (unsigned long long *)nullptr, (size_t)n, 0, 64, nullptr);
    const size_t safe = capture_safe_sort_temp_bytes(n, false);
    return bytes > safe ? bytes : safe;
}

}  // namespace drt

using namespace drt;
#ifdef DRT_FILTER_DEBUG
extern "C" void drt_debug_counts(unsigned long long *out, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(drt::drt_dbg_counts), sizeof(unsigned long long) * 8);
    if (reset) {
        unsigned long long z[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(drt::drt_dbg_counts), z, sizeof(z));
    }
}
#endif

extern "C" {

int32_t drt_candidates_fill(int64_t num_nodes, int32_t order, int64_t rank_lo, int64_t rank_hi,
                            const int32_t *node_map, int32_t id_scale, int32_t *out, void *stream) {
    DRT_REQUIRE(rank_hi >= rank_lo, "rank_hi < rank_lo");
    if (rank_hi == rank_lo || order == 0) return DRT_OK;
    DRT_REQUIRE(out, "out is null");
    drt_candidates c{};
    c.table = nullptr;
    c.num_candidates = rank_hi - rank_lo;
    c.rank_lo = rank_lo;
    c.num_nodes = num_nodes;
    c.node_map = node_map;
    c.order = order;
    CandSrc cs;
    int32_t rc = make_cand_src(&c, id_scale <= 0 ? 1 : id_scale, &cs);
    if (rc != DRT_OK) return rc;
    const dim3 grid((unsigned)ceil_div(cs.count, 256));
#define CALL(K) hipLaunchKernelGGL(candidates_fill_kernel<K>, grid, dim3(256), 0, as_stream(stream), cs, out)
    DRT_ORDER_SWITCH(order, CALL)
#undef CALL
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

size_t drt_trace_compact_workspace_size(int64_t max_survivors, int64_t max_paths) {
    if (max_survivors < 0) max_survivors = 0;
    if (max_paths < 0) max_paths = 0;
    return 64 + align_up((size_t)max_survivors * 8, 256) + align_up((size_t)max_paths * 8, 256) +
           align_up(sort_temp_bytes(max_paths), 256) + 256;
}

int32_t drt_trace_paths_compact(drt_mesh_t mesh, const drt_trace_params *pr, const float *tx,
                                int64_t ntx, const float *rx, int64_t nrx, const drt_candidates *cands,
                                int64_t max_survivors, int64_t max_paths, int64_t *keys,
                                float *vertices, int32_t *objects, int64_t *num_valid_host, void *ws,
                                size_t ws_bytes, void *stream) {
    DRT_REQUIRE(mesh && pr && cands && num_valid_host, "null argument");
    DRT_REQUIRE(ntx >= 0 && nrx >= 0 && max_survivors >= 0 && max_paths >= 0, "negative size");
    DRT_REQUIRE(nrx < (1ll << 31), "too many receivers for one launch");
    *num_valid_host = 0;
    Launch L;
    L.s = as_stream(stream);
    L.quads = mesh->assume_quads != 0;
    int32_t rc = make_cand_src(cands, L.quads ? 2 : 1, &L.cs);
    if (rc != DRT_OK) return rc;
    L.a = make_args(mesh, pr, tx, ntx, rx, nrx);
    if ((pr->flags & DRT_TRACE_USE_BVH) && mesh->num_triangles > 0) {
        rc = drt_mesh_build_bvh(mesh, stream);
        if (rc != DRT_OK) return rc;
        L.bvh = reinterpret_cast<const BvhNode *>(mesh->bvh_nodes);
        L.bvh_leaf_ids = mesh->bvh_leaf_ids;
    }
    DRT_REQUIRE(!L.cs.packed, "packed keys address traced paths (drt_trace_paths_vjp), they are not a candidate source");
    L.cs.npairs = ntx * nrx;
    const unsigned __int128 total = L.cs.ragged ? (unsigned __int128)L.cs.count
                                                : (unsigned __int128)ntx * (unsigned __int128)nrx *
                                                      (unsigned __int128)L.cs.count;
    if (total == 0) return DRT_OK;
    DRT_REQUIRE(total < ((unsigned __int128)1 << 62), "tx*rx*candidates does not fit a 62-bit key");
    DRT_REQUIRE(tx && rx, "null pointer");
    const size_t need = drt_trace_compact_workspace_size(max_survivors, max_paths);
    if (!ws || ws_bytes < need) return fail(DRT_E_CAPACITY, "workspace too small: need %zu bytes", need);
    char *base = reinterpret_cast<char *>(ws);
    auto *counters = reinterpret_cast<unsigned long long *>(base);  // [0] survivors, [1] valid
    auto *q1 = reinterpret_cast<long long *>(base + 64);
    auto *q2 = reinterpret_cast<long long *>(base + 64 + align_up((size_t)max_survivors * 8, 256));
    char *sort_tmp = reinterpret_cast<char *>(q2) + align_up((size_t)max_paths * 8, 256);
    DRT_HIP(hipMemsetAsync(counters, 0, 64, L.s));
    const int k = cands->order;
    // optional per-stage timers (SURVEY.md section 5 "metrics"): HIP events on the caller's stream
    drt_trace_stats *st = pr->stats;
    StageTimer timer(st != nullptr, L.s);
    timer.mark(0);
#define CALL(K)                                                                                   \
    do {                                                                                          \
        if (L.quads)                                                                              \
            launch_filter<K, true>(L, counters, q1, max_survivors); \
        else                                                                                      \
            launch_filter<K, false>(L, counters, q1, max_survivors); \
        timer.mark(1);                                                                            \
        launch_occlusion<K, false>(L, counters, q1, max_survivors, counters + 1, q2, max_paths,  \
                                   nullptr);                                                      \
        timer.mark(2);                                                                            \
    } while (0)
    DRT_ORDER_SWITCH(k, CALL)
#undef CALL
    DRT_LAUNCH_CHECK();
    unsigned long long host_counts[2] = {0, 0};
    DRT_HIP(hipMemcpyAsync(host_counts, counters, 16, hipMemcpyDeviceToHost, L.s));
    DRT_HIP(hipStreamSynchronize(L.s));
    if (st) {
        st->candidates = (int64_t)total;
        st->survivors = (int64_t)host_counts[0];
        st->valid = (int64_t)host_counts[1];
        st->filter_ms = timer.elapsed(0, 1);
        st->occlusion_ms = timer.elapsed(1, 2);
        st->sort_emit_ms = 0.0f;
    }
    if ((int64_t)host_counts[0] > max_survivors) {
        *num_valid_host = (int64_t)host_counts[0];
        return fail(DRT_E_CAPACITY,
                    "survivor queue overflow: %llu candidates passed the geometric checks, capacity %lld",
                    host_counts[0], (long long)max_survivors);
    }
    const int64_t nv = (int64_t)host_counts[1];
    *num_valid_host = nv;
    if (nv > max_paths)
        return fail(DRT_E_CAPACITY, "%lld valid paths, output capacity %lld", (long long)nv,
                    (long long)max_paths);
    if (nv == 0) return DRT_OK;
    DRT_REQUIRE(keys && vertices && objects, "null output");
    size_t tmp_bytes = sort_temp_bytes(nv);
    DRT_HIP(rocprim::radix_sort_keys(sort_tmp, tmp_bytes, reinterpret_cast<unsigned long long *>(q2),
                                     reinterpret_cast<unsigned long long *>(keys), (size_t)nv, 0, 64,
                                     L.s));
#define CALL(K)                                                                                 \
    hipLaunchKernelGGL(trace_emit_kernel<K>, dim3((unsigned)ceil_div(nv, 256)), dim3(256), 0, L.s, L.a, \
                       L.cs, reinterpret_cast<const long long *>(keys), nv, vertices, objects)
    DRT_ORDER_SWITCH(k, CALL)
#undef CALL
    DRT_LAUNCH_CHECK();
    if (st) {
        timer.mark(3);
        DRT_HIP(hipStreamSynchronize(L.s));
        st->sort_emit_ms = timer.elapsed(2, 3);
    }
    return DRT_OK;
}

int32_t drt_trace_paths_compact_async(drt_mesh_t mesh, const drt_trace_params *pr, const float *tx,
                                      int64_t ntx, const float *rx, int64_t nrx, const drt_candidates *cands,
                                      int64_t max_survivors, int64_t max_paths, int64_t *keys,
                                      float *vertices, int32_t *objects, int64_t *counts_dev, void *ws,
                                      size_t ws_bytes, void *stream) {
    DRT_REQUIRE(mesh && pr && cands, "null argument");
    DRT_REQUIRE(ntx >= 0 && nrx >= 0 && max_survivors >= 0 && max_paths >= 0, "negative size");
    DRT_REQUIRE(nrx < (1ll << 31), "too many receivers for one launch");
    DRT_REQUIRE(max_paths == 0 || (keys && vertices && objects), "null output");
    Launch L;
    L.s = as_stream(stream);
    L.quads = mesh->assume_quads != 0;
    int32_t rc = make_cand_src(cands, L.quads ? 2 : 1, &L.cs);
    if (rc != DRT_OK) return rc;
    L.a = make_args(mesh, pr, tx, ntx, rx, nrx);
    if ((pr->flags & DRT_TRACE_USE_BVH) && mesh->num_triangles > 0) {
        // building the LBVH allocates and synchronises: it must exist before a capture / async call
        DRT_REQUIRE(mesh->bvh_nodes != nullptr,
                    "DRT_TRACE_USE_BVH in the async entry point needs drt_mesh_build_bvh() beforehand");
        L.bvh = reinterpret_cast<const BvhNode *>(mesh->bvh_nodes);
        L.bvh_leaf_ids = mesh->bvh_leaf_ids;
    }
    DRT_REQUIRE(!L.cs.packed, "packed keys address traced paths (drt_trace_paths_vjp), they are not a candidate source");
    L.cs.npairs = ntx * nrx;
    const unsigned __int128 total = L.cs.ragged ? (unsigned __int128)L.cs.count
                                                : (unsigned __int128)ntx * (unsigned __int128)nrx *
                                                      (unsigned __int128)L.cs.count;
    DRT_REQUIRE(total < ((unsigned __int128)1 << 62), "tx*rx*candidates does not fit a 62-bit key");
    DRT_REQUIRE(total == 0 || (tx && rx), "null pointer");
    const size_t need = drt_trace_compact_workspace_size(max_survivors, max_paths);
    if (!ws || ws_bytes < need) return fail(DRT_E_CAPACITY, "workspace too small: need %zu bytes", need);
    char *base = reinterpret_cast<char *>(ws);
    auto *counters = reinterpret_cast<unsigned long long *>(base);  // [0] survivors, [1] valid
    auto *q1 = reinterpret_cast<long long *>(base + 64);
    auto *q2 = reinterpret_cast<long long *>(base + 64 + align_up((size_t)max_survivors * 8, 256));
    char *sort_tmp = reinterpret_cast<char *>(q2) + align_up((size_t)max_paths * 8, 256);
    DRT_HIP(fill_bytes_async(counters, 0, 64, L.s));
    // every slot of the valid-key queue starts as a sentinel (0x7f7f... >= 2^62 > any key): the sort
    // then runs over the FIXED capacity and pushes the unused slots behind the valid keys
    if (max_paths > 0) DRT_HIP(fill_bytes_async(q2, 0x7f, (size_t)max_paths * 8, L.s));
    const int k = cands->order;
    if (total > 0) {
#define CALL(K)                                                                                   \
    do {                                                                                          \
        if (L.quads)                                                                              \
            launch_filter<K, true>(L, counters, q1, max_survivors); \
        else                                                                                      \
            launch_filter<K, false>(L, counters, q1, max_survivors); \
        launch_occlusion<K, false>(L, counters, q1, max_survivors, counters + 1, q2, max_paths,  \
                                   nullptr);                                                      \
    } while (0)
        DRT_ORDER_SWITCH(k, CALL)
#undef CALL
        DRT_LAUNCH_CHECK();
    }
    if (max_paths > 0) {
        size_t tmp_bytes = sort_temp_bytes(max_paths);
        DRT_HIP(capture_safe_sort(sort_tmp, tmp_bytes, reinterpret_cast<unsigned long long *>(q2),
                                  reinterpret_cast<unsigned long long *>(keys), nullptr, nullptr, max_paths, 0, 64, L.s));
    }
    const int64_t rows = max_paths > 0 ? max_paths : 1;  // one thread at least: it writes the counts
#define CALL(K)                                                                                        \
    hipLaunchKernelGGL(trace_emit_padded_kernel<K>, dim3((unsigned)ceil_div(rows, 256)), dim3(256), 0, L.s, \
                       L.a, L.cs, reinterpret_cast<long long *>(keys), max_paths, max_survivors, counters,   \
                       reinterpret_cast<long long *>(counts_dev), vertices, objects)
    DRT_ORDER_SWITCH(k, CALL)
#undef CALL
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_trace_paths_vjp(drt_mesh_t mesh, const float *tx, int64_t ntx, const float *rx,
                            int64_t nrx, const drt_candidates *cands, const int64_t *keys,
                            const float *cot, int64_t num, float *g_tx, float *g_rx,
                            float *g_vertices, void *stream) {
    DRT_REQUIRE(mesh && cands, "null argument");
    DRT_REQUIRE(num >= 0, "negative size");
    if (num == 0) return DRT_OK;
    DRT_REQUIRE(tx && rx && keys && cot, "null pointer");
    Launch L;
    L.s = as_stream(stream);
    L.quads = mesh->assume_quads != 0;
    int32_t rc = make_cand_src(cands, L.quads ? 2 : 1, &L.cs);
    if (rc != DRT_OK) return rc;
    L.cs.npairs = ntx * nrx;
    L.a = make_args(mesh, nullptr, tx, ntx, rx, nrx);
    const int k = cands->order;
#define CALL(K)                                                                                  \
    hipLaunchKernelGGL(trace_vjp_kernel<K>, dim3((unsigned)ceil_div(num, 256)), dim3(256), 0, L.s, L.a, \
                       L.cs, mesh->vertices, mesh->triangles, reinterpret_cast<const long long *>(keys), \
                       cot, num, g_tx, g_rx, g_vertices)
    DRT_ORDER_SWITCH(k, CALL)
#undef CALL
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

static size_t vjp_sort_temp_bytes(int64_t n) {
    if (n <= 0) return 0;
    return capture_safe_sort_temp_bytes(n, true);
}

size_t drt_trace_vjp_workspace_size(int64_t num_paths, int32_t order) {
    if (num_paths < 0) num_paths = 0;
    if (order < 0) order = 0;
    const int64_t n = num_paths * (2 + 3 * (int64_t)order);
    return 256 + 2 * align_up((size_t)n * 8, 256) + 2 * align_up((size_t)n * 4, 256) + align_up((size_t)n * 12, 256) +
           align_up(vjp_sort_temp_bytes(n), 256);
}

int32_t drt_trace_paths_vjp_ex(drt_mesh_t mesh, const drt_trace_params *pr, const float *tx, int64_t ntx,
                               const float *rx, int64_t nrx, const drt_candidates *cands, const int64_t *keys,
                               const float *cot, int64_t num, float *g_tx, float *g_rx, float *g_vertices, void *ws,
                               size_t ws_bytes, void *stream) {
    if (!pr || !(pr->flags & DRT_TRACE_DETERMINISTIC_GRAD))
        return drt_trace_paths_vjp(mesh, tx, ntx, rx, nrx, cands, keys, cot, num, g_tx, g_rx, g_vertices, stream);
    DRT_REQUIRE(mesh && cands, "null argument");
    DRT_REQUIRE(num >= 0, "negative size");
    if (num == 0) return DRT_OK;
    DRT_REQUIRE(tx && rx && keys && cot, "null pointer");
    Launch L;
    L.s = as_stream(stream);
    L.quads = mesh->assume_quads != 0;
    int32_t rc = make_cand_src(cands, L.quads ? 2 : 1, &L.cs);
    if (rc != DRT_OK) return rc;
    L.cs.npairs = ntx * nrx;
    L.a = make_args(mesh, nullptr, tx, ntx, rx, nrx);
    const int k = cands->order;
    const int64_t n = num * (2 + 3 * (int64_t)k);
    DRT_REQUIRE(n < (1ll << 32), "too many gradient contributions for one call");
    const size_t need = drt_trace_vjp_workspace_size(num, k);
    if (!ws || ws_bytes < need) return fail(DRT_E_CAPACITY, "workspace too small: need %zu bytes", need);
    char *base = reinterpret_cast<char *>(ws);
    const size_t a8 = align_up((size_t)n * 8, 256), a4 = align_up((size_t)n * 4, 256);
    auto *dest = reinterpret_cast<unsigned long long *>(base);
    auto *dest_sorted = reinterpret_cast<unsigned long long *>(base + a8);
    auto *slots = reinterpret_cast<uint32_t *>(base + 2 * a8);
    auto *slots_sorted = reinterpret_cast<uint32_t *>(base + 2 * a8 + a4);
    auto *vecs = reinterpret_cast<float *>(base + 2 * a8 + 2 * a4);
    char *sort_tmp = base + 2 * a8 + 2 * a4 + align_up((size_t)n * 12, 256);
#define CALL(K)                                                                                                       \
    hipLaunchKernelGGL(trace_vjp_contrib_kernel<K>, dim3((unsigned)ceil_div(num, 256)), dim3(256), 0, L.s, L.a, L.cs,  \
                       mesh->vertices, mesh->triangles, reinterpret_cast<const long long *>(keys), cot, num,         \
                       g_tx ? 1 : 0, g_rx ? 1 : 0, g_vertices ? 1 : 0, dest, slots, vecs)
    DRT_ORDER_SWITCH(k, CALL)
#undef CALL
    DRT_LAUNCH_CHECK();
    size_t tb = vjp_sort_temp_bytes(n);
    // radix sort is stable: inside a destination the slots stay in path order
    // (capture-safe configuration, sort_safe.hpp: this entry point may sit in a HIP graph next to the asynchronous tracers)
    DRT_HIP(capture_safe_sort(sort_tmp, tb, dest, dest_sorted, slots, slots_sorted, n, 0, 42, L.s));
    // (the unsorted destination keys are dead after the sort: their buffer holds the chunk carries, 12 B per 256 slots)
    float *carry = reinterpret_cast<float *>(dest);
    const int64_t nchunks = ceil_div(n, kVjpChunk);
    hipLaunchKernelGGL(trace_vjp_carry_kernel, dim3((unsigned)ceil_div(nchunks, 256)), dim3(256), 0, L.s, dest_sorted,
                       slots_sorted, vecs, n, carry);
    hipLaunchKernelGGL(trace_vjp_reduce_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, L.s, dest_sorted,
                       slots_sorted, vecs, n, carry, g_tx, g_rx, g_vertices);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

}  // extern "C"
