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

// The traversal structure: the binary radix tree collapsed to a 4-ary one (a node's children are its
// grandchildren in the binary tree; a leaf child stays where it is).  Half the dependent node fetches per walk:
// with incoherent rays on a 200k-triangle mesh the walk is a chain of L2 / MALL latencies, not arithmetic.
// Wide node i describes binary node i (every binary node gets one; only those reachable from the root by
// two-level steps are ever visited).  They live in the same allocation, behind the binary nodes (which stay:
// leaf ranges, refit, and the walk of a mesh whose boxes are not all finite).  child: >= 0 binary / wide node
// index, < 0 leaf (triangle id = ~child); an empty slot has child = kBvhNoChild.
struct __attribute__((aligned(16))) Bvh4Node {
    uint16_t qlo[4][3];  // child boxes on a 16-bit grid over the padded scene bounds (Bvh4Grid), rounded OUTWARD:
    uint16_t qhi[4][3];  // decode(qlo) <= lo and decode(qhi) >= hi, checked with the decode expression itself at build
    int32_t child[4];
};
static_assert(sizeof(Bvh4Node) == 64, "Bvh4Node must be 64 B");  // four 16-byte loads per lane and node: with incoherent
                                                                   // rays the walk is bound by L1 look-ups per lane
// header of the wide region: the grid, and whether the wide walk may be used at all (every triangle box finite)
struct __attribute__((aligned(16))) Bvh4Grid {
    float lo[3];
    float cell[3];
    uint32_t ok;
    uint32_t pad[9];
};
static_assert(sizeof(Bvh4Grid) == 64, "Bvh4Grid must be 64 B");
constexpr int32_t kBvhNoChild = 0x7fffffff;
__host__ __device__ inline int64_t bvh_wide_offset(int64_t T) {  // in BvhNode units, 128-B aligned
    const int64_t nn = T > 1 ? T - 1 : 1;
    return (nn + 1) & ~(int64_t)1;
}
__host__ __device__ inline size_t bvh_wide_bytes(int64_t T) {  // header + one wide node per binary node
    const int64_t nn = T > 1 ? T - 1 : 1;
    return sizeof(Bvh4Grid) + (size_t)nn * sizeof(Bvh4Node);
}
__device__ __forceinline__ const Bvh4Grid *bvh_grid(const BvhNode *nodes, int64_t T) {
    return reinterpret_cast<const Bvh4Grid *>(nodes + bvh_wide_offset(T));
}
__device__ __forceinline__ const Bvh4Node *bvh_wide(const BvhNode *nodes, int64_t T) {
    return reinterpret_cast<const Bvh4Node *>(bvh_grid(nodes, T) + 1);
}
// THE decode expression (build-time check and walk use this one function)
__device__ __forceinline__ float bvh_q_decode(uint32_t q, float cell, float lo) {
    return __builtin_fmaf((float)q, cell, lo);
}

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
__device__ __forceinline__ void bvh_walk4(const BvhNode *__restrict__ nodes, const uint32_t *__restrict__ leaf_ids,
                                          int64_t T, const RayPrep &ray, const float &limit, int32_t *col, Leaf &&leaf) {
    const Bvh4Node *__restrict__ wide = bvh_wide(nodes, T);
    const Bvh4Grid *__restrict__ grid = bvh_grid(nodes, T);  // wave-uniform: scalar loads
    const float glo[3] = {grid->lo[0], grid->lo[1], grid->lo[2]}, cell[3] = {grid->cell[0], grid->cell[1], grid->cell[2]};
    int sp = 0;
    int32_t node = (T == 1) ? ~0 : 0;  // T == 1: the single triangle is tested directly
    // a child that cannot be pushed (column full) is dealt with here and now: its whole subtree through the
    // binary node's leaf range
    auto flush = [&](int32_t c) -> bool {
        if (c < 0) return leaf((int64_t)~c);
        const uint32_t p0 = nodes[c].first, p1 = nodes[c].last;
        for (uint32_t q = p0; q <= p1; ++q)
            if (leaf((int64_t)leaf_ids[q])) return true;
        return false;
    };
    for (;;) {
        if (node < 0) {
            if (leaf((int64_t)~node)) return;
        } else {
            const Bvh4Node nd = wide[node];
            float t0[4];
            int32_t ch[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a0, a1;
                const float lo[3] = {bvh_q_decode(nd.qlo[i][0], cell[0], glo[0]), bvh_q_decode(nd.qlo[i][1], cell[1], glo[1]),
                                     bvh_q_decode(nd.qlo[i][2], cell[2], glo[2])};
                const float hi[3] = {bvh_q_decode(nd.qhi[i][0], cell[0], glo[0]), bvh_q_decode(nd.qhi[i][1], cell[1], glo[1]),
                                     bvh_q_decode(nd.qhi[i][2], cell[2], glo[2])};
                slab(ray, lo, hi, a0, a1);
                // (an empty slot has no box: test the id)
                const bool hit = (a0 <= a1) && (a1 >= 0.0f) && (a0 <= limit) && (nd.child[i] != kBvhNoChild);
                t0[i] = hit ? a0 : kInf;
                ch[i] = hit ? nd.child[i] : kBvhNoChild;
            }
            if (ORDERED) {
                // nearest first: a 4-element sorting network on (entry distance, child); dropping the fifth
                // compare-exchange (middle two unordered) measured 10-15 % slower on first hits
#define DRT_CSWAP(a, b)                                  \
    do {                                                 \
        const bool sw = t0[b] < t0[a];                   \
        const float ta = sw ? t0[b] : t0[a], tb = sw ? t0[a] : t0[b]; \
        const int32_t ca = sw ? ch[b] : ch[a], cb = sw ? ch[a] : ch[b]; \
        t0[a] = ta; t0[b] = tb; ch[a] = ca; ch[b] = cb;  \
    } while (0)
                DRT_CSWAP(0, 1);
                DRT_CSWAP(2, 3);
                DRT_CSWAP(0, 2);
                DRT_CSWAP(1, 3);
                DRT_CSWAP(1, 2);
#undef DRT_CSWAP
            } else {  // hit children to the front, any order
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int k = i + 1; k < 4; ++k)
                        if (ch[i] == kBvhNoChild && ch[k] != kBvhNoChild) {
                            ch[i] = ch[k];
                            ch[k] = kBvhNoChild;
                        }
            }
            if (ch[0] != kBvhNoChild) {
                // the others wait in the column, farthest pushed first
#pragma unroll
                for (int i = 3; i >= 1; --i) {
                    if (ch[i] != kBvhNoChild) {
                        if (sp < kBvhLdsStack) {
                            col[sp * BLOCK] = ch[i];
                            ++sp;
                        } else if (flush(ch[i])) {
                            return;
                        }
                    }
                }
                node = ch[0];
                continue;
            }
        }
        if (sp == 0) return;
        --sp;
        node = col[sp * BLOCK];
    }
}

// the binary walk (nearer child first): two boxes per 64-B node
template <int BLOCK, bool ORDERED, class Leaf>
__device__ __forceinline__ void bvh_walk2(const BvhNode *__restrict__ nodes, const uint32_t *__restrict__ leaf_ids,
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

// Which tree a walk uses (measured, 1e6 rays, profiles/r03/bvh.md).  Incoherent rays -- every lane of a load another
// line -- are bound by L1 look-ups per lane: the 64-B wide node serves four children with the four 16-byte loads the
// binary node spends on two, and wins at every size, ordered or not (10k triangles: first hit 5.3e9 vs 4.5e9, any hit
// 6.9e9 vs 5.5e9 rays/s).  COHERENT rays (the lattices of launch_paths and of the visibility estimate) share their
// lines; there the ordered wide walk pays for decoding and sorting four children and loses below ~65k triangles
// (launch_paths order 3, 10k triangles: 1.23e9 vs 1.14e9 rays/s) -- a wave-uniform switch on T for those callers.
#ifndef DRT_BVH_WIDE_ORDERED_MIN_T
#define DRT_BVH_WIDE_ORDERED_MIN_T 65536
#endif
constexpr int64_t kBvhWideOrderedMinT = DRT_BVH_WIDE_ORDERED_MIN_T;
template <int BLOCK, bool ORDERED, bool COHERENT = false, class Leaf>
__device__ __forceinline__ void bvh_walk(const BvhNode *__restrict__ nodes, const uint32_t *__restrict__ leaf_ids,
                                         int64_t T, const RayPrep &ray, const float &limit, int32_t *col, Leaf &&leaf) {
    // (a mesh with a non-finite triangle box has no grid: binary walk on the float boxes)
    const bool wide_ok = __builtin_amdgcn_readfirstlane((int)bvh_grid(nodes, T)->ok) != 0;
    if (wide_ok && (!(ORDERED && COHERENT) || T >= kBvhWideOrderedMinT))
        bvh_walk4<BLOCK, ORDERED>(nodes, leaf_ids, T, ray, limit, col, leaf);
    else
        bvh_walk2<BLOCK, ORDERED>(nodes, leaf_ids, T, ray, limit, col, leaf);
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
template <int BLOCK, bool COHERENT = false>
__device__ __forceinline__ uint64_t bvh_first_hit(const BvhNode *__restrict__ nodes, const uint32_t *__restrict__ leaf_ids,
                                                  int64_t T, const float *__restrict__ tv,
                                                  const uint8_t *__restrict__ mask, V3 o, V3 d,
                                                  float eps, const TileTieB &tt, int32_t *col) {
    const RayPrep ray = prep_ray(o, d);
    uint64_t best = ~0ull;
    float best_t = kInf;
    bvh_walk<BLOCK, true, COHERENT>(nodes, leaf_ids, T, ray, best_t, col, [&](int64_t j) {
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
