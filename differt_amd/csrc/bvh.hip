// bvh.hip -- own LBVH over a drt_mesh_t and ray queries that use it ("next" row f1 of SURVEY.md 8f).
//
// The reference's mesh-bound queries (geometry/_mesh.py:142-223, 3018-3162) run on NVIDIA Warp's BVH
// (warp-lang, not under /root/reference, no ROCm support).  This is an independent implementation:
//   build   : Morton codes of triangle centroids -> radix sort (rocPRIM) -> Karras' parallel radix
//             tree (2012) -> bottom-up AABB refit with atomic arrival flags.  64-byte nodes hold BOTH
//             children's boxes so that one node visit is one 64-B load and two slab tests.
//   any-hit : lane = ray, depth-first traversal with a private stack, leaf test = the SAME
//             Moller-Trumbore as the brute-force operators (geom.hpp), predicate of _utils.py:1469.
//   first-hit: same traversal ordered near child first, packed (t, tie) key as in ray_ops.hip, boxes
//             with entry distance <= best t are still visited so that ties resolve exactly like the
//             brute-force kernel (lowest index in a tile, later tile wins).
// Boxes are padded and the slab test is widened by a few ulps: a triangle test is skipped only if the
// ray misses the padded box.  For non-degenerate rays the result equals the brute-force kernels bit
// for bit (tests/test_bvh_gpu.py); for rays grazing a triangle's plane within ~1e-7 rad the
// brute-force outcome itself is rounding noise and the two may differ -- the brute-force operators
// remain the normative ones.
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "common.hpp"
#include "geom.hpp"
#include "mesh.hpp"
#include "bvh.hpp"
#include "lattice.hpp"

#pragma clang fp contract(off)

namespace drt {

struct Box {
    float lo[3], hi[3];
};

__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// per-triangle padded boxes + scene bounds (atomic min/max on ordered uints)
__global__ __launch_bounds__(256) void tri_boxes_kernel(const float *__restrict__ tv, int64_t T,
                                                        Box *__restrict__ boxes,
                                                        uint32_t *__restrict__ scene /*[6] + [6]: non-finite flag*/) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const V3 a = ld3(tv + 9 * t), b = ld3(tv + 9 * t + 3), c = ld3(tv + 9 * t + 6);
    Box bx;
    bx.lo[0] = fminf(a.x, fminf(b.x, c.x)); bx.hi[0] = fmaxf(a.x, fmaxf(b.x, c.x));
    bx.lo[1] = fminf(a.y, fminf(b.y, c.y)); bx.hi[1] = fmaxf(a.y, fmaxf(b.y, c.y));
    bx.lo[2] = fminf(a.z, fminf(b.z, c.z)); bx.hi[2] = fmaxf(a.z, fmaxf(b.z, c.z));
    boxes[t] = bx;
    bool all_finite = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (is_finite(bx.lo[k])) atomicMin(scene + k, float_to_ordered(bx.lo[k]));
        if (is_finite(bx.hi[k])) atomicMax(scene + 3 + k, float_to_ordered(bx.hi[k]));
        all_finite = all_finite && is_finite(bx.lo[k]) && is_finite(bx.hi[k]);
    }
    if (!all_finite) atomicOr(scene + 6, 1u);
}

__global__ __launch_bounds__(256) void pad_and_morton_kernel(Box *__restrict__ boxes, int64_t T,
                                                             const uint32_t *__restrict__ scene,
                                                             uint64_t *__restrict__ keys,
                                                             uint32_t *__restrict__ ids) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    float slo[3], shi[3], ext = 0.0f, mag = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        slo[k] = ordered_to_float(scene[k]);
        shi[k] = ordered_to_float(scene[3 + k]);
        ext = fmaxf(ext, shi[k] - slo[k]);
        mag = fmaxf(mag, fmaxf(fabsf(slo[k]), fabsf(shi[k])));
    }
    // padding: far above the rounding of any coordinate in the scene (2^-14 of its magnitude)
    const float pad = fmaxf(mag, ext) * 0x1p-14f + 1e-30f;
    Box bx = boxes[t];
    uint32_t code = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float cen = 0.5f * (bx.lo[k] + bx.hi[k]);
        float u = (ext > 0.0f) ? (cen - slo[k]) / ext : 0.0f;
        u = fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f);
        code |= expand_bits((uint32_t)u) << (2 - k);
        bx.lo[k] -= pad;
        bx.hi[k] += pad;
    }
    boxes[t] = bx;
    keys[t] = ((uint64_t)code << 32) | (uint32_t)t;  // index in the low bits: all keys distinct
    ids[t] = (uint32_t)t;
}

__device__ __forceinline__ int delta(const uint64_t *keys, int64_t n, int64_t i, int64_t j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));  // keys are distinct -> xor != 0
}

// Karras 2012, one thread per internal node
__global__ __launch_bounds__(256) void radix_tree_kernel(const uint64_t *__restrict__ keys, int64_t n,
                                                         BvhNode *__restrict__ nodes,
                                                         int32_t *__restrict__ parent_internal,
                                                         int32_t *__restrict__ parent_leaf) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int64_t lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int64_t l = 0;
    for (int64_t t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int64_t j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int64_t s = 0;
    for (int64_t t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int64_t gamma = i + s * d + (d < 0 ? -1 : 0);
    const int64_t lo = i < j ? i : j, hi = i < j ? j : i;
    const bool left_leaf = (lo == gamma), right_leaf = (hi == gamma + 1);
    nodes[i].first = (uint32_t)lo;  // this node's leaves: sorted positions [lo, hi]
    nodes[i].last = (uint32_t)hi;
    nodes[i].left = left_leaf ? ~(int32_t)gamma : (int32_t)gamma;
    nodes[i].right = right_leaf ? ~(int32_t)(gamma + 1) : (int32_t)(gamma + 1);
    if (left_leaf) parent_leaf[gamma] = (int32_t)i; else parent_internal[gamma] = (int32_t)i;
    if (right_leaf) parent_leaf[gamma + 1] = (int32_t)i; else parent_internal[gamma + 1] = (int32_t)i;
    if (i == 0) parent_internal[0] = -1;
}

__device__ __forceinline__ void box_union(const Box &a, const Box &b, Box &o) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o.lo[k] = fminf(a.lo[k], b.lo[k]);
        o.hi[k] = fmaxf(a.hi[k], b.hi[k]);
    }
}

// one thread per leaf walks up; the second arrival at a node owns it
__global__ __launch_bounds__(256) void refit_kernel(int64_t n, const uint32_t *__restrict__ sorted_ids,
                                                    const Box *__restrict__ tri_boxes,
                                                    BvhNode *__restrict__ nodes,
                                                    const int32_t *__restrict__ parent_internal,
                                                    const int32_t *__restrict__ parent_leaf,
                                                    Box *__restrict__ node_boxes,
                                                    uint32_t *__restrict__ flags) {
    const int64_t leaf = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (leaf >= n) return;
    int32_t p = parent_leaf[leaf];
    while (p >= 0) {
        __threadfence();
        if (atomicAdd(flags + p, 1u) == 0u) return;  // first arrival: the sibling will finish
        __threadfence();
        BvhNode nd = nodes[p];
        Box lb, rb;
        if (nd.left < 0) { lb = tri_boxes[sorted_ids[~nd.left]]; } else { lb = node_boxes[nd.left]; }
        if (nd.right < 0) { rb = tri_boxes[sorted_ids[~nd.right]]; } else { rb = node_boxes[nd.right]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            nd.llo[k] = lb.lo[k]; nd.lhi[k] = lb.hi[k];
            nd.rlo[k] = rb.lo[k]; nd.rhi[k] = rb.hi[k];
        }
        // leaves are stored as triangle ids from now on
        if (nd.left < 0) nd.left = ~(int32_t)sorted_ids[~nd.left];
        if (nd.right < 0) nd.right = ~(int32_t)sorted_ids[~nd.right];
        nodes[p] = nd;
        Box u;
        box_union(lb, rb, u);
        node_boxes[p] = u;
        p = parent_internal[p];
    }
}

// binary -> 4-ary collapse, one thread per binary node (after the refit)
struct WideEntry {
    float lo[3], hi[3];
    int32_t child;
};
__device__ __forceinline__ WideEntry wide_entry(const float *lo, const float *hi, int32_t child) {
    return WideEntry{{lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}, child};
}
// the 16-bit grid of the wide nodes: the scene bounds widened by twice the boxes' padding (every finite padded box
// lies inside), 65535 cells per axis; ok = 0 (binary walk) when some triangle box is not finite or the bounds are
__global__ void wide_grid_kernel(const uint32_t *__restrict__ scene, Bvh4Grid *__restrict__ grid) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float slo[3], shi[3], ext = 0.0f, mag = 0.0f;
    bool ok = scene[6] == 0u;
    for (int k = 0; k < 3; ++k) {
        slo[k] = ordered_to_float(scene[k]);
        shi[k] = ordered_to_float(scene[3 + k]);
        ok = ok && is_finite(slo[k]) && is_finite(shi[k]) && slo[k] <= shi[k];
        ext = fmaxf(ext, shi[k] - slo[k]);
        mag = fmaxf(mag, fmaxf(fabsf(slo[k]), fabsf(shi[k])));
    }
    const float pad = fmaxf(mag, ext) * 0x1p-14f + 1e-30f;  // pad_and_morton_kernel's
    Bvh4Grid g{};
    for (int k = 0; k < 3; ++k) {
        const float lo = slo[k] - 2.0f * pad, hi = shi[k] + 2.0f * pad;
        float cell = (hi - lo) * (1.0f / 65535.0f);
        // the last grid line must not fall short of hi (one or two steps of the float above, if at all)
        for (int it = 0; it < 64 && ok && !(bvh_q_decode(65535u, cell, lo) >= hi); ++it)
            cell = __uint_as_float(__float_as_uint(cell) + 1u);
        ok = ok && is_finite(lo) && is_finite(hi) && is_finite(cell) && cell > 0.0f && bvh_q_decode(65535u, cell, lo) >= hi;
        g.lo[k] = lo;
        g.cell[k] = cell;
    }
    g.ok = ok ? 1u : 0u;
    *grid = g;
}

// largest grid line <= v / smallest grid line >= v, by the decode expression itself (v finite and inside the grid
// whenever the grid is in use; anything else lands on an end line and the wide walk is off)
__device__ __forceinline__ uint16_t quant_down(float v, float cell, float lo) {
    const float x = (v - lo) / cell;
    uint32_t q = (x >= 1.0f) ? ((x < 65535.0f) ? (uint32_t)x : 65535u) : 0u;  // NaN -> 0
    while (q > 0u && bvh_q_decode(q, cell, lo) > v) --q;
    return (uint16_t)q;
}
__device__ __forceinline__ uint16_t quant_up(float v, float cell, float lo) {
    const float x = (v - lo) / cell;
    uint32_t q = (x <= 65534.0f) ? ((x > 0.0f) ? (uint32_t)x + 1u : 0u) : 65535u;  // NaN -> 65535
    while (q < 65535u && bvh_q_decode(q, cell, lo) < v) ++q;
    return (uint16_t)q;
}

__global__ __launch_bounds__(256) void collapse_kernel(const BvhNode *__restrict__ nodes, int64_t n_internal,
                                                       const Bvh4Grid *__restrict__ grid, Bvh4Node *__restrict__ wide) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= n_internal) return;
    const BvhNode nd = nodes[b];
    const float inf3[3] = {kInf, kInf, kInf}, ninf3[3] = {-kInf, -kInf, -kInf};
    const WideEntry none = wide_entry(inf3, ninf3, kBvhNoChild);
    // each binary child contributes itself (a leaf) or its two children
    WideEntry l0 = wide_entry(nd.llo, nd.lhi, nd.left), l1 = none, r0 = wide_entry(nd.rlo, nd.rhi, nd.right), r1 = none;
    if (nd.left >= 0) {
        const BvhNode g = nodes[nd.left];
        l0 = wide_entry(g.llo, g.lhi, g.left);
        l1 = wide_entry(g.rlo, g.rhi, g.right);
    }
    if (nd.right >= 0) {
        const BvhNode g = nodes[nd.right];
        r0 = wide_entry(g.llo, g.lhi, g.left);
        r1 = wide_entry(g.rlo, g.rhi, g.right);
    }
    // slots: l0, then l1 if there is one, then r0, r1; empty slots last (kBvhNoChild)
    const bool two_l = nd.left >= 0;
    const WideEntry e[4] = {l0, two_l ? l1 : r0, two_l ? r0 : r1, two_l ? r1 : none};
    const float glo[3] = {grid->lo[0], grid->lo[1], grid->lo[2]}, cell[3] = {grid->cell[0], grid->cell[1], grid->cell[2]};
    Bvh4Node w;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            w.qlo[k][c] = quant_down(e[k].lo[c], cell[c], glo[c]);
            w.qhi[k][c] = quant_up(e[k].hi[c], cell[c], glo[c]);
        }
        w.child[k] = e[k].child;
    }
    wide[b] = w;
}

template <bool FIRST>
__global__ __launch_bounds__(256) void bvh_query_kernel(
    const BvhNode *__restrict__ nodes, const uint32_t *__restrict__ leaf_ids, int64_t T,
    const float *__restrict__ tv, const uint8_t *__restrict__ mask, const float *__restrict__ ro,
    const float *__restrict__ rd, int64_t R, float eps, float thr, TileTieB tt, uint8_t *__restrict__ any_out,
    int32_t *__restrict__ idx_out, float *__restrict__ t_out) {
    DRT_BVH_LDS_STACK(lds_stack, 256);
    int32_t *col = &lds_stack[0][threadIdx.x];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const V3 o = ld3(ro + 3 * r), d = ld3(rd + 3 * r);
    if (FIRST) {
        int32_t idx;
        float t;
        decode_first_hit(bvh_first_hit<256>(nodes, leaf_ids, T, tv, mask, o, d, eps, tt, col), tt, idx, t);
        idx_out[r] = idx;
        t_out[r] = t;
    } else {
        any_out[r] = (uint8_t)bvh_any_hit<256>(nodes, leaf_ids, T, tv, mask, o, d, eps, thr, col);
    }
}

// visibility (reference geometry/_mesh.py:3164-3253): lane = lattice ray, closest hit through the BVH
// with the tie rule of first_triangle_hit_by_ray(batch_size=None): lowest index among equal t.
__global__ __launch_bounds__(256) void bvh_visibility_kernel(
    const BvhNode *__restrict__ nodes, const uint32_t *__restrict__ leaf_ids, int64_t T,
    const float *__restrict__ tv, const uint8_t *__restrict__ mask, const float *__restrict__ view,
    const float *__restrict__ frusta, int64_t num_rays, float eps, uint8_t *__restrict__ visible) {
    DRT_BVH_LDS_STACK(lds_stack, 256);
    int32_t *col = &lds_stack[0][threadIdx.x];
    const int64_t b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= num_rays) return;
    const RayPrep ray = prep_ray(ld3(view + 3 * b), lattice_direction(i, num_rays, frusta + 6 * b));
    float best_t = kInf;
    int64_t best_j = -1;
    bvh_walk<256, true, true>(nodes, leaf_ids, T, ray, best_t, col, [&](int64_t j) {  // lattice rays: coherent
        float t;
        const bool h = moller_trumbore(ray.o, ray.d, load_tri(tv + 9 * j), eps, t) && (!mask || mask[j]);
        if (h && is_finite(t) && (t < best_t || (t == best_t && j < best_j))) {
            best_t = t;
            best_j = j;
        }
        return false;
    });
    if (best_j >= 0) visible[b * T + best_j] = 1;
}

__global__ __launch_bounds__(256) void fill_miss_kernel(int64_t R, int32_t *__restrict__ idx,
                                                        float *__restrict__ t) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    idx[r] = -1;
    t[r] = kInf;
}

// Triangle-driven complement of the lattice visibility (an extension: the reference samples DIRECTIONS only,
// so a small far-away face can fall between the rays): lane = (triangle, interior sample point); the face is
// visible when the segment viewpoint -> sample is not blocked by any OTHER active triangle before the
// sample (t < 1 - 1e-4).  ORs into `visible`.
__global__ __launch_bounds__(256) void bvh_visibility_samples_kernel(
    const BvhNode *__restrict__ nodes, const uint32_t *__restrict__ leaf_ids, int64_t T, const float *__restrict__ tv,
    const uint8_t *__restrict__ mask, const float *__restrict__ view, float eps, uint8_t *__restrict__ visible) {
    constexpr int kSamples = 7;
    DRT_BVH_LDS_STACK(lds_stack, 256);
    int32_t *col = &lds_stack[0][threadIdx.x];
    const int64_t b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= T * kSamples) return;
    const int64_t tri = i / kSamples;
    const int sidx = (int)(i - tri * kSamples);
    if (mask && !mask[tri]) return;
    if (visible[b * T + tri]) return;  // the lattice pass (or another sample) already saw it
    // barycentric weights: centroid, three points near the vertices, three near the edge mid-points
    const float w[kSamples][3] = {{1.f / 3, 1.f / 3, 1.f / 3}, {0.8f, 0.1f, 0.1f}, {0.1f, 0.8f, 0.1f}, {0.1f, 0.1f, 0.8f},
                                  {0.45f, 0.45f, 0.1f}, {0.1f, 0.45f, 0.45f}, {0.45f, 0.1f, 0.45f}};
    const float *t9 = tv + 9 * tri;
    const V3 v0 = ld3(t9), v1 = ld3(t9 + 3), v2 = ld3(t9 + 6);
    const V3 p = v0 * w[sidx][0] + v1 * w[sidx][1] + v2 * w[sidx][2];
    const V3 o = ld3(view + 3 * b);
    const RayPrep ray = prep_ray(o, p - o);
    const float thr = 1.0f - 1e-4f;
    bool blocked = false;
    bvh_walk<256, false>(nodes, leaf_ids, T, ray, thr, col, [&](int64_t j) {
        if (j != tri && (!mask || mask[j])) {
            float t;
            if (moller_trumbore(ray.o, ray.d, load_tri(tv + 9 * j), eps, t) && t < thr) blocked = true;
        }
        return blocked;
    });
    if (!blocked) visible[b * T + tri] = 1;
}

}  // namespace drt

using namespace drt;

extern "C" {

int32_t drt_mesh_build_bvh(drt_mesh_t m, void *stream) {
    DRT_REQUIRE(m, "mesh is null");
    if (m->bvh_nodes) return DRT_OK;
    const int64_t T = m->num_triangles;
    if (T == 0) return DRT_OK;
    hipStream_t s = as_stream(stream);
    const size_t nn = (size_t)(T > 1 ? T - 1 : 1);
    BvhNode *nodes = nullptr;
    Box *tri_boxes = nullptr, *node_boxes = nullptr;
    uint64_t *keys = nullptr, *keys_sorted = nullptr;
    uint32_t *ids = nullptr, *ids_sorted = nullptr, *scene = nullptr, *flags = nullptr;
    int32_t *par_int = nullptr, *par_leaf = nullptr;
    void *tmp = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(tri_boxes); (void)hipFree(node_boxes); (void)hipFree(keys);
        (void)hipFree(keys_sorted); (void)hipFree(ids); (void)hipFree(scene);
        (void)hipFree(flags); (void)hipFree(par_int); (void)hipFree(par_leaf); (void)hipFree(tmp);
    };
#define TRY_HIP(expr)                                                                    \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            cleanup();                                                                   \
            (void)hipFree(nodes);                                                        \
            (void)hipFree(ids_sorted);                                                   \
            return fail(DRT_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));       \
        }                                                                                \
    } while (0)
    // binary nodes, then (128-B aligned) the grid header and one 4-ary node per binary node
    const size_t wide_off = (size_t)bvh_wide_offset(T);
    TRY_HIP(hipMalloc(&nodes, wide_off * sizeof(BvhNode) + bvh_wide_bytes(T)));
    TRY_HIP(hipMalloc(&tri_boxes, (size_t)T * sizeof(Box)));
    TRY_HIP(hipMalloc(&node_boxes, nn * sizeof(Box)));
    TRY_HIP(hipMalloc(&keys, (size_t)T * 8));
    TRY_HIP(hipMalloc(&keys_sorted, (size_t)T * 8));
    TRY_HIP(hipMalloc(&ids, (size_t)T * 4));
    TRY_HIP(hipMalloc(&ids_sorted, (size_t)T * 4));
    TRY_HIP(hipMalloc(&scene, 32));
    TRY_HIP(hipMalloc(&flags, nn * 4));
    TRY_HIP(hipMalloc(&par_int, nn * 4));
    TRY_HIP(hipMalloc(&par_leaf, (size_t)T * 4));
    // scene bounds as ordered uints: min slots start at +max, max slots at 0
    TRY_HIP(hipMemsetAsync(scene, 0xff, 12, s));
    TRY_HIP(hipMemsetAsync(scene + 3, 0, 20, s));  // max slots and the non-finite flag
    TRY_HIP(hipMemsetAsync(flags, 0, nn * 4, s));
    TRY_HIP(hipMemsetAsync(nodes, 0, wide_off * sizeof(BvhNode) + sizeof(Bvh4Grid), s));  // (grid.ok = 0 until built)
    const dim3 gt((unsigned)ceil_div(T, 256));
    hipLaunchKernelGGL(tri_boxes_kernel, gt, dim3(256), 0, s, m->tri_verts, T, tri_boxes, scene);
    hipLaunchKernelGGL(pad_and_morton_kernel, gt, dim3(256), 0, s, tri_boxes, T, scene, keys, ids);
    TRY_HIP(hipGetLastError());
    if (T > 1) {
        size_t tmp_bytes = 0;
        TRY_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys_sorted, ids, ids_sorted,
                                          (size_t)T, 0, 64, s));
        TRY_HIP(hipMalloc(&tmp, tmp_bytes));
        TRY_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_sorted, ids, ids_sorted, (size_t)T,
                                          0, 64, s));
        hipLaunchKernelGGL(radix_tree_kernel, dim3((unsigned)ceil_div(T - 1, 256)), dim3(256), 0, s,
                           keys_sorted, T, nodes, par_int, par_leaf);
        hipLaunchKernelGGL(refit_kernel, gt, dim3(256), 0, s, T, ids_sorted, tri_boxes, nodes, par_int,
                           par_leaf, node_boxes, flags);
        Bvh4Grid *grid = reinterpret_cast<Bvh4Grid *>(nodes + wide_off);
        hipLaunchKernelGGL(wide_grid_kernel, dim3(1), dim3(64), 0, s, scene, grid);
        hipLaunchKernelGGL(collapse_kernel, dim3((unsigned)ceil_div(T - 1, 256)), dim3(256), 0, s, nodes, T - 1, grid,
                           reinterpret_cast<Bvh4Node *>(grid + 1));
        TRY_HIP(hipGetLastError());
    }
    TRY_HIP(hipStreamSynchronize(s));
#undef TRY_HIP
    cleanup();
    if (T == 1) {  // no sort ran: the single leaf is position 0
        const uint32_t zero = 0;
        (void)hipMemcpy(ids_sorted, &zero, 4, hipMemcpyHostToDevice);
    }
    m->bvh_leaf_ids = ids_sorted;
    m->bvh_nodes = nodes;
    return DRT_OK;
}

int32_t drt_mesh_has_bvh(drt_mesh_t m) { return (m && m->bvh_nodes) ? 1 : 0; }

int32_t drt_mesh_ray_intersect_any_triangle(drt_mesh_t m, const float *ro, const float *rd, int64_t R,
                                            float epsilon, float hit_tol, uint8_t *out, void *stream) {
    DRT_REQUIRE(m, "mesh is null");
    DRT_REQUIRE(R >= 0, "negative size");
    if (R == 0) return DRT_OK;
    DRT_REQUIRE(out, "null output");
    hipStream_t s = as_stream(stream);
    if (m->num_triangles == 0) {  // _mesh.py:3053-3057
        DRT_HIP(fill_bytes_async(out, 0, (size_t)R, s));
        return DRT_OK;
    }
    DRT_REQUIRE(ro && rd, "null pointer");
    int32_t rc = drt_mesh_build_bvh(m, stream);
    if (rc != DRT_OK) return rc;
    const TileTieB tt = make_tie_b(m->num_triangles, 0);
    hipLaunchKernelGGL(bvh_query_kernel<false>, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, s,
                       reinterpret_cast<const BvhNode *>(m->bvh_nodes), m->bvh_leaf_ids, m->num_triangles, m->tri_verts,
                       m->has_mask ? m->mask : nullptr, ro, rd, R, epsilon, 1.0f - hit_tol, tt, out,
                       (int32_t *)nullptr, (float *)nullptr);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_mesh_triangles_visible_samples(drt_mesh_t m, const float *vertices, int64_t B, float epsilon,
                                           uint8_t *visible_inout, void *stream) {
    DRT_REQUIRE(m, "mesh is null");
    DRT_REQUIRE(B >= 0, "negative size");
    const int64_t T = m->num_triangles;
    if (B == 0 || T == 0) return DRT_OK;
    DRT_REQUIRE(vertices && visible_inout, "null pointer");
    DRT_REQUIRE(B <= 65535, "at most 65535 viewing vertices per call");
    int32_t rc = drt_mesh_build_bvh(m, stream);
    if (rc != DRT_OK) return rc;
    hipLaunchKernelGGL(bvh_visibility_samples_kernel, dim3((unsigned)ceil_div(T * 7, 256), (unsigned)B), dim3(256), 0,
                       as_stream(stream), reinterpret_cast<const BvhNode *>(m->bvh_nodes), m->bvh_leaf_ids, T, m->tri_verts,
                       m->has_mask ? m->mask : nullptr, vertices, epsilon, visible_inout);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_mesh_triangles_visible_from_vertex(drt_mesh_t m, const float *vertices, int64_t B,
                                               int64_t num_rays, float epsilon, uint8_t *visible_out,
                                               float *frustum_workspace, void *stream) {
    DRT_REQUIRE(m, "mesh is null");
    DRT_REQUIRE(B >= 0, "negative size");
    DRT_REQUIRE(num_rays > 0, "num_rays must be strictly positive");
    const int64_t T = m->num_triangles;
    if (B == 0 || T == 0) return DRT_OK;
    DRT_REQUIRE(vertices && visible_out && frustum_workspace, "null pointer");
    DRT_REQUIRE(B <= 65535, "at most 65535 viewing vertices per call");
    int32_t rc = drt_mesh_build_bvh(m, stream);
    if (rc != DRT_OK) return rc;
    hipStream_t s = as_stream(stream);
    const uint8_t *mask = m->has_mask ? m->mask : nullptr;
    DRT_HIP(fill_bytes_async(visible_out, 0, (size_t)B * (size_t)T, s));  // (a kernel: memset nodes do not survive graph replay, core.hip)
    launch_frustum_kernel(vertices, B, m->tri_verts, T, mask, frustum_workspace, s);
    DRT_LAUNCH_CHECK();
    hipLaunchKernelGGL(bvh_visibility_kernel, dim3((unsigned)ceil_div(num_rays, 256), (unsigned)B),
                       dim3(256), 0, s, reinterpret_cast<const BvhNode *>(m->bvh_nodes), m->bvh_leaf_ids, T, m->tri_verts,
                       mask, vertices, frustum_workspace, num_rays, epsilon, visible_out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_mesh_first_triangle_hit_by_ray(drt_mesh_t m, const float *ro, const float *rd, int64_t R,
                                           float epsilon, int64_t batch_size, int32_t *idx, float *t,
                                           void *stream) {
    DRT_REQUIRE(m, "mesh is null");
    DRT_REQUIRE(R >= 0, "negative size");
    if (R == 0) return DRT_OK;
    DRT_REQUIRE(idx && t, "null output");
    hipStream_t s = as_stream(stream);
    if (m->num_triangles == 0) {  // _mesh.py:3129-3136
        hipLaunchKernelGGL(fill_miss_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, s, R, idx, t);
        DRT_LAUNCH_CHECK();
        return DRT_OK;
    }
    DRT_REQUIRE(ro && rd, "null pointer");
    int32_t rc = drt_mesh_build_bvh(m, stream);
    if (rc != DRT_OK) return rc;
    const TileTieB tt = make_tie_b(m->num_triangles, batch_size);
    hipLaunchKernelGGL(bvh_query_kernel<true>, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, s,
                       reinterpret_cast<const BvhNode *>(m->bvh_nodes), m->bvh_leaf_ids, m->num_triangles, m->tri_verts,
                       m->has_mask ? m->mask : nullptr, ro, rd, R, epsilon, 0.0f, tt,
                       (uint8_t *)nullptr, idx, t);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

}  // extern "C"
