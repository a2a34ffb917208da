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
    uint32_t pad0;
    float rhi[3];
    uint32_t pad1;
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


// Traversal pushes at most one node per tree level, and a Karras radix tree over DISTINCT 62-bit keys
// (30-bit Morton code << 32 | triangle index, bvh.hip pad_and_morton_kernel) splits on a strictly
// increasing bit position along any root-to-leaf path: depth <= 62 whatever the geometry (coincident
// centroids only deepen the tree down the index bits).  A 64-entry stack therefore cannot overflow;
// the `sp < kBvhStack` guards below are unreachable belt-and-braces.
constexpr int kBvhKeyBits = 62;
constexpr int kBvhStack = 64;
static_assert(kBvhStack >= kBvhKeyBits, "traversal stack must cover the maximum radix-tree depth");

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
__device__ __forceinline__ uint64_t bvh_first_hit(const BvhNode *__restrict__ nodes, int64_t T,
                                                  const float *__restrict__ tv,
                                                  const uint8_t *__restrict__ mask, V3 o, V3 d,
                                                  float eps, const TileTieB &tt) {
    const RayPrep ray = prep_ray(o, d);
    uint64_t best = ~0ull;
    float best_t = kInf;
    int32_t stack[kBvhStack];
    int sp = 0;
    int32_t node = (T == 1) ? ~0 : 0;
    for (;;) {
        if (node < 0) {
            const int64_t j = ~node;
            float t;
            const bool h = moller_trumbore(ray.o, ray.d, load_tri(tv + 9 * j), eps, t) &&
                           (!mask || mask[j]);
            if (h && is_finite(t)) {
                const uint64_t k = first_hit_key_b(t, j, tt);
                if (k < best) { best = k; best_t = t; }
            }
        } else {
            const BvhNode nd = nodes[node];
            float l0, l1, r0, r1;
            slab(ray, nd.llo, nd.lhi, l0, l1);
            slab(ray, nd.rlo, nd.rhi, r0, r1);
            const bool hl = (l0 <= l1) && (l1 >= 0.0f) && (l0 <= best_t);
            const bool hr = (r0 <= r1) && (r1 >= 0.0f) && (r0 <= best_t);
            if (hl && hr) {
                const bool left_first = l0 <= r0;
                if (sp < kBvhStack) stack[sp++] = left_first ? nd.right : nd.left;
                node = left_first ? nd.left : nd.right;
                continue;
            }
            if (hl) { node = nd.left; continue; }
            if (hr) { node = nd.right; continue; }
        }
        if (sp == 0) return best;
        node = stack[--sp];
    }
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
__device__ __forceinline__ bool bvh_any_hit(const BvhNode *__restrict__ nodes, int64_t T,
                                            const float *__restrict__ tv,
                                            const uint8_t *__restrict__ mask, V3 o, V3 d, float eps,
                                            float thr) {
    const RayPrep ray = prep_ray(o, d);
    int32_t stack[kBvhStack];
    int sp = 0;
    int32_t node = (T == 1) ? ~0 : 0;
    for (;;) {
        if (node < 0) {
            const int64_t j = ~node;
            float t;
            const bool h = moller_trumbore(ray.o, ray.d, load_tri(tv + 9 * j), eps, t) &&
                           (!mask || mask[j]);
            if (h && (t < thr)) return true;
        } else {
            const BvhNode nd = nodes[node];
            float l0, l1, r0, r1;
            slab(ray, nd.llo, nd.lhi, l0, l1);
            slab(ray, nd.rlo, nd.rhi, r0, r1);
            const bool hl = (l0 <= l1) && (l1 >= 0.0f) && (l0 <= thr);
            const bool hr = (r0 <= r1) && (r1 >= 0.0f) && (r0 <= thr);
            if (hl && hr) {
                if (sp < kBvhStack) stack[sp++] = nd.right;
                node = nd.left;
                continue;
            }
            if (hl) { node = nd.left; continue; }
            if (hr) { node = nd.right; continue; }
        }
        if (sp == 0) return false;
        node = stack[--sp];
    }
}

}  // namespace drt
