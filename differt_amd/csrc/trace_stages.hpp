// trace_stages.hpp -- pieces of the hard-mask tracer shared by trace.hip (compact forms) and
// trace_dense.hip (the reference's dense layout): the inside-triangle tests of stage A, stage B (occlusion of
// the survivors, brute force and LBVH) and the launch descriptor.
// Reference: geometry/_solvers.py:598-642 (inside), :676-680 (blocked), _utils.py:1469 (predicate).
#pragma once

#include <hip/hip_runtime.h>

#include "common.hpp"
#include "geom.hpp"
#include "image_chain.hpp"
#include "mesh.hpp"
#include "tri_tile.hpp"
#include "bvh.hpp"
#include "trace_common.hpp"

#ifndef DRT_DBG
#define DRT_DBG(i) do { } while (0)
#endif

#pragma clang fp contract(off)

namespace drt {

template <int K, bool QUADS>
__device__ __forceinline__ bool inside_one(const Mirrors<K, QUADS> &m, const V3 (&full)[K + 2], int j,
                                           float eps) {
    const V3 o = full[j];
    const V3 d = full[j + 1] - full[j];
    float t;
    bool h = moller_trumbore(o, d, m.tri[j], eps, t);
    if (QUADS) h = h || moller_trumbore(o, d, m.tri2[j], eps, t);  // SV:615-627 any over the pair
    return h;
}

// wave-mask form (bit l = lane l of `want` passes), see moller_trumbore_wave
template <int K, bool QUADS>
__device__ __forceinline__ uint64_t inside_one_wave(const Mirrors<K, QUADS> &m, const V3 (&full)[K + 2], int j,
                                                    float eps, uint64_t want) {
    const V3 o = full[j];
    const V3 d = full[j + 1] - full[j];
    uint64_t h = moller_trumbore_wave(o, d, m.tri[j], eps, want);
    if (QUADS) h |= moller_trumbore_wave(o, d, m.tri2[j], eps, want & ~h);  // SV:615-627 any over the pair
    return h;
}

// ------------------------------------------------------------------------------------------
// stage B: one wavefront per surviving candidate
// ------------------------------------------------------------------------------------------
template <int K, bool DENSE>
__global__ __launch_bounds__(256) void trace_occlusion_kernel(
    TraceArgs a, CandSrc cs, const unsigned long long *__restrict__ q_count,
    const long long *__restrict__ queue, int64_t q_cap, unsigned long long *__restrict__ v_count,
    long long *__restrict__ valid, int64_t v_cap, uint8_t *__restrict__ d_mask) {
    __shared__ TriRec lds[kTile];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int64_t count = (int64_t)*q_count;
    if (count > q_cap) count = q_cap;
    for (int64_t e0 = (int64_t)blockIdx.x * 4; e0 < count; e0 += (int64_t)gridDim.x * 4) {
        const int64_t e = e0 + wave;
        const bool have = e < count;  // wave-uniform
        V3 full[K + 2];
        int64_t flat = 0;
        if (have) {
            flat = queue[e];
            int64_t it, ir;
            int32_t id[KA<K>::n];
            V3 p[KA<K>::n], n[KA<K>::n];
            key_to_path<K>(a, cs, flat, it, ir, id, p, n, full);
        } else {
#pragma unroll
            for (int j = 0; j < K + 2; ++j) full[j] = V3{0, 0, 0};
        }
        V3 dir[K + 1];
#pragma unroll
        for (int s = 0; s <= K; ++s) dir[s] = full[s + 1] - full[s];
        bool blocked = !have;  // idle waves count as done
        for (int64_t base = 0; base < a.T_occ; base += kTile) {
            // block-wide early exit (also the barrier that protects the previous tile's readers)
            if (__syncthreads_and(blocked ? 1 : 0)) break;
            stage_tile(lds, a.tri_verts, a.mask, base, a.T_occ);
            __syncthreads();
            if (!blocked) {
                const int n = (int)((a.T_occ - base < kTile) ? a.T_occ - base : kTile);
                bool hit = false;
                for (int j = lane; j < n; j += 64) {
                    const TriRec rec = lds[j];
                    const TriE tr = rec_tri(rec);
#pragma unroll
                    for (int s = 0; s <= K; ++s) {
                        float t;
                        const bool h = moller_trumbore(full[s], dir[s], tr, a.eps, t);
                        hit = hit || (h && (t < a.thr) && rec.active);  // _utils.py:1469
                    }
                }
                blocked = __any(hit);
            }
        }
        __syncthreads();
        if (have && lane == 0) {
            if (DENSE) {
                if (blocked) {
                    d_mask[flat] = 0;
                    if (v_count) atomicAdd(v_count, 1ull);  // dense layout: the number of survivors the stage cleared
                }
            } else if (!blocked) {
                const unsigned long long slot = atomicAdd(v_count, 1ull);
                if ((int64_t)slot < v_cap) valid[slot] = flat;
            }
        }
    }
}

// stage B, BVH flavour (opt-in, drt_trace_params.flags & DRT_TRACE_USE_BVH): one LANE per surviving
// candidate walks the mesh LBVH for each of its order+1 segments.  O(log T) per segment instead of
// O(T): the choice for very large meshes (configs[4], 200k triangles).
template <int K, bool DENSE>
__global__ __launch_bounds__(256) void trace_occlusion_bvh_kernel(
    TraceArgs a, CandSrc cs, const BvhNode *__restrict__ nodes, const uint32_t *__restrict__ leaf_ids,
    const unsigned long long *__restrict__ q_count, const long long *__restrict__ queue, int64_t q_cap,
    unsigned long long *__restrict__ v_count, long long *__restrict__ valid, int64_t v_cap,
    uint8_t *__restrict__ d_mask) {
    DRT_BVH_LDS_STACK(lds_stack, 256);
    int32_t *col = &lds_stack[0][threadIdx.x];
    int64_t count = (int64_t)*q_count;
    if (count > q_cap) count = q_cap;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (int64_t)gridDim.x * 256) {
        const int64_t flat = queue[e];
        int64_t it, ir;
        int32_t id[KA<K>::n];
        V3 p[KA<K>::n], n[KA<K>::n], full[K + 2];
        key_to_path<K>(a, cs, flat, it, ir, id, p, n, full);
        bool blocked = false;
        // unrolled over the K + 1 segments: `full[s]` with a runtime s puts the path in scratch memory
#pragma unroll
        for (int s = 0; s <= K; ++s) {
            if (!blocked)
                blocked = bvh_any_hit<256>(nodes, leaf_ids, a.T, a.tri_verts, a.mask, full[s], full[s + 1] - full[s],
                                           a.eps, a.thr, col);
        }
        if (DENSE) {
            if (blocked) {
                d_mask[flat] = 0;
                if (v_count) atomicAdd(v_count, 1ull);
            }
        } else if (!blocked) {
            const unsigned long long slot = atomicAdd(v_count, 1ull);
            if ((int64_t)slot < v_cap) valid[slot] = flat;
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct Launch {
    hipStream_t s;
    TraceArgs a;
    CandSrc cs;
    bool quads;
    const BvhNode *bvh = nullptr;  // non-null: stage B walks the LBVH
    const uint32_t *bvh_leaf_ids = nullptr;
};

template <int K, bool DENSE>
static void launch_occlusion(const Launch &L, const unsigned long long *qc, const long long *q,
                             int64_t qcap, unsigned long long *vc, long long *v, int64_t vcap,
                             uint8_t *dm) {
    // persistent-style grid: the survivor count lives on the device
    if (L.bvh && L.a.T_occ > 0)
        hipLaunchKernelGGL((trace_occlusion_bvh_kernel<K, DENSE>), dim3(256 * 8), dim3(256), 0, L.s, L.a,
                           L.cs, L.bvh, L.bvh_leaf_ids, qc, q, qcap, vc, v, vcap, dm);
    else
        hipLaunchKernelGGL((trace_occlusion_kernel<K, DENSE>), dim3(256 * 4), dim3(256), 0, L.s, L.a,
                           L.cs, qc, q, qcap, vc, v, vcap, dm);
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// HIP-event timer of the stages of one call (only when the caller asked for drt_trace_stats)
struct StageTimer {
    bool on;
    hipStream_t s;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    StageTimer(bool enable, hipStream_t stream) : on(enable), s(stream) {
        if (on)
            for (auto &e : ev)
                if (hipEventCreate(&e) != hipSuccess) on = false;
    }
    ~StageTimer() {
        for (auto &e : ev)
            if (e) (void)hipEventDestroy(e);
    }
    void mark(int i) {
        if (on) (void)hipEventRecord(ev[i], s);
    }
    float elapsed(int i, int j) {
        float ms = 0.0f;
        if (on && hipEventElapsedTime(&ms, ev[i], ev[j]) != hipSuccess) ms = 0.0f;
        return ms;
    }
};



}  // namespace drt
