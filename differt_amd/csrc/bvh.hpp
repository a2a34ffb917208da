// bvh.hpp -- node layout and traversal primitives of the LBVH (built in bvh.hip), shared with the
// occlusion stage of the tracer (trace.hip).
#pragma once

#include "geom.hpp"

#pragma clang fp contract(off)

namespace drt {

struct __attribute__((aligned(16))) BvhNode {
    float llo[3];
    int32_t left;   // >= 0: internal node index;  < 0: leaf, triangle id = ~left
    float lhi[3];
    int32_t right;
    float rlo[3];
    uint32_t first;  // leaves of this node's subtree = positions [first, last] of the Morton-sorted leaf list
    float rhi[3];
    uint32_t last;
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 B");

// ---- traversal ---------------------------------------------------------------------------------
struct RayPrep {
    V3 o, d, inv;
};

__device__ __forceinline__ RayPrep prep_ray(V3 o, V3 d) {
    return RayPrep{o, d, V3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z}};
}

// entry / exit parameters of the ray through a box, widened by a few ulps; NaNs (0 * inf) are
// ignored by fminf/fmaxf
__device__ __forceinline__ void slab(const RayPrep &r, const float *lo, const float *hi, float &t0,
                                     float &t1) {
    const float ax = (lo[0] - r.o.x) * r.inv.x, bx = (hi[0] - r.o.x) * r.inv.x;
    const float ay = (lo[1] - r.o.y) * r.inv.y, by = (hi[1] - r.o.y) * r.inv.y;
    const float az = (lo[2] - r.o.z) * r.inv.z, bz = (hi[2] - r.o.z) * r.inv.z;
    t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    t0 = t0 - fabsf(t0) * 0x1p-20f - 1e-30f;
    t1 = t1 + fabsf(t1) * 0x1p-20f + 1e-30f;
}


// Traversal stack.  A depth-first walk holds at most one pending node per tree level, and a Karras radix tree
// over DISTINCT 62-bit keys (30-bit Morton code << 32 | triangle index, bvh.hip pad_and_morton_kernel) can be 62
// levels deep whatever the geometry.  A private 64-entry array lives in scratch memory (272-384 B per lane in
// round 2: every push / pop a trip through the vector memory path, in the dependent chain pop -> node load).
// Here the stack is a COLUMN OF LDS per lane -- lds[level][thread], so lanes at different depths still hit
// different banks -- of kBvhLdsStack entries; the rare walk that needs more does not push: it tests the far
// child's whole subtree at once through the node's leaf range (first / last in the Morton-sorted leaf list).
// Any order of visiting gives the same answer (any-hit is an OR, first-hit a minimum of packed keys), so results
// are unchanged, and a deep pending node has few leaves below it.  No scratch, no overflow, no depth limit.
// 20 entries = 20 KiB per 256-thread block = 8 blocks per CU: measured 1.41e9 rays/s on the 200k-triangle first hit
// against 1.13e9 with 32 entries (5 blocks per CU) and 1.0e9 with the scratch stack of round 2.
#ifndef DRT_BVH_LDS_STACK_N  // test hook: a 2-entry column sends almost every walk through the overflow path
#define DRT_BVH_LDS_STACK_N 20
#endif
constexpr int kBvhLdsStack = DRT_BVH_LDS_STACK_N;
#define DRT_BVH_LDS_STACK(name, block) __shared__ int32_t name[::drt::kBvhLdsStack][block]

// The walk shared by every query.  `leaf(j)` tests triangle j and returns true to stop the walk; `limit` is read
// at every node (boxes entered after it cannot matter): a first-hit functor lowers it as it finds hits.
// ORDERED: nearer child first.  BLOCK = threads per block (stride of the LDS column `col` = &lds[0][threadIdx.x]).
template <int BLOCK, bool ORDERED, class Leaf>
__device__ __forceinline__ void bvh_walk(const BvhNode *__restrict__ nodes, const uint32_t *__restrict__ leaf_ids,
                                         int64_t T, const RayPrep &ray, const float &limit, int32_t *col, Leaf &&leaf) {
    int sp = 0;
    int32_t node = (T == 1) ? ~0 : 0;  // T == 1: the single triangle is tested directly
    for (;;) {
        if (node < 0) {
            if (leaf((int64_t)~node)) return;
        } else {
            const BvhNode nd = nodes[node];
            float l0, l1, r0, r1;
            slab(ray, nd.llo, nd.lhi, l0, l1);
            slab(ray, nd.rlo, nd.rhi, r0, r1);
            const bool hl = (l0 <= l1) && (l1 >= 0.0f) && (l0 <= limit);
            const bool hr = (r0 <= r1) && (r1 >= 0.0f) && (r0 <= limit);
            if (hl && hr) {
                const bool left_first = !ORDERED || (l0 <= r0);
                const int32_t nearc = left_first ? nd.left : nd.right;
                const int32_t farc = left_first ? nd.right : nd.left;
                if (sp < kBvhLdsStack) {
                    col[sp * BLOCK] = farc;
                    ++sp;
                } else if (farc < 0) {  // column full: the far child is dealt with here and now
                    if (leaf((int64_t)~farc)) return;
                } else {
                    const uint32_t p0 = nodes[farc].first, p1 = nodes[farc].last;
                    for (uint32_t q = p0; q <= p1; ++q)
                        if (leaf((int64_t)leaf_ids[q])) return;
                }
                node = nearc;
                continue;
            }
            if (hl) { node = nd.left; continue; }
            if (hr) { node = nd.right; continue; }
        }
        if (sp == 0) return;
        --sp;
        node = col[sp * BLOCK];
    }
}

// packed first-hit key, see ray_ops.hip: smallest t, then the LATEST batch_size-tile, then the lowest
// index inside the tile (reference geometry/_utils.py:1865-1867, 1886)
struct TileTieB {
    int64_t bs, nb, ntiles;
};
__device__ __forceinline__ uint64_t first_hit_key_b(float t, int64_t j, const TileTieB &tt) {
    const int64_t tile = (j < tt.nb * tt.bs) ? j / tt.bs : tt.nb;
    const int64_t in_tile = j - tile * tt.bs;
    return ((uint64_t)float_to_ordered(t) << 32) | (uint64_t)((tt.ntiles - 1 - tile) * tt.bs + in_tile);
}


inline TileTieB make_tie_b(int64_t T, int64_t batch_size) {
    int64_t bs = batch_size <= 0 ? T : batch_size;
    if (bs > T) bs = T;
    if (bs < 1) bs = 1;
    TileTieB tt;
    tt.bs = bs;
    tt.nb = T / bs;
    tt.ntiles = tt.nb + ((T % bs) ? 1 : 0);
    return tt;
}


// closest hit through the BVH as a packed key (~0 = miss); boxes entered at t <= best t are still
// visited so that ties resolve exactly like the brute-force kernels
template <int BLOCK>
__device__ __forceinline__ uint64_t bvh_first_hit(const BvhNode *__restrict__ nodes, const uint32_t *__restrict__ leaf_ids,
                                                  int64_t T, const float *__restrict__ tv,
                                                  const uint8_t *__restrict__ mask, V3 o, V3 d,
                                                  float eps, const TileTieB &tt, int32_t *col) {
    const RayPrep ray = prep_ray(o, d);
    uint64_t best = ~0ull;
    float best_t = kInf;
    bvh_walk<BLOCK, true>(nodes, leaf_ids, T, ray, best_t, col, [&](int64_t j) {
        float t;
        const bool h = moller_trumbore(ray.o, ray.d, load_tri(tv + 9 * j), eps, t) && (!mask || mask[j]);
        if (h && is_finite(t)) {
            const uint64_t k = first_hit_key_b(t, j, tt);
            if (k < best) { best = k; best_t = t; }
        }
        return false;
    });
    return best;
}

__device__ __forceinline__ void decode_first_hit(uint64_t key, const TileTieB &tt, int32_t &idx, float &t) {
    if (key == ~0ull) {
        idx = -1;
        t = kInf;
        return;
    }
    const uint64_t tie = key & 0xffffffffull;
    const int64_t tile = tt.ntiles - 1 - (int64_t)(tie / (uint64_t)tt.bs);
    idx = (int32_t)(tile * tt.bs + (int64_t)(tie % (uint64_t)tt.bs));
    t = ordered_to_float((uint32_t)(key >> 32));
}

// any-hit with the predicate of reference geometry/_utils.py:1469: exists an active triangle with
// hit && t < thr.  Leaf test = the shared Moller-Trumbore.
template <int BLOCK>
__device__ __forceinline__ bool bvh_any_hit(const BvhNode *__restrict__ nodes, const uint32_t *__restrict__ leaf_ids,
                                            int64_t T, const float *__restrict__ tv,
                                            const uint8_t *__restrict__ mask, V3 o, V3 d, float eps,
                                            float thr, int32_t *col) {
    const RayPrep ray = prep_ray(o, d);
    bool any = false;
    bvh_walk<BLOCK, false>(nodes, leaf_ids, T, ray, thr, col, [&](int64_t j) {
        float t;
        const bool h = moller_trumbore(ray.o, ray.d, load_tri(tv + 9 * j), eps, t) && (!mask || mask[j]);
        any = h && (t < thr);
        return any;
    });
    return any;
}

}  // namespace drt
