// image_method.hip -- stand-alone image-method operators over flat batches.
//   drt_image_method              reference geometry/_solver_image_method.py:206-363
//   drt_image_method_vjp          reverse of the two lax.scans (:185-203), hand-derived
//   drt_consecutive_vertices_same_side   reference :386-454
// One lane per batch element, rows staged per wave through LDS (stores.hpp); the chain is fully unrolled for the
// (small) number of mirrors.
#include "common.hpp"
#include "image_chain.hpp"
#include "stores.hpp"

#pragma clang fp contract(off)

namespace drt {

// Rows of one batch element: from / to 3 dwords, mirrors / path / cotangents 3K dwords.  All of them are array-of-
// struct rows at a 12- or 12K-byte lane stride, so every array goes through the wave's LDS region (stores.hpp):
// 16-B coalesced global accesses, the per-lane row on the LDS side.  The operator moves 24 + 36K bytes per element
// for ~40K VALU instructions: HBM bound.  A stride of 0 marks an input that is the same for every element (the
// reference broadcasts `from` / `to` of shape [3] against [B, K, 3] mirrors, tests/benchmarks/fixtures.py:19-40): it is
// read once per lane from one address instead of being materialised B times by the caller.
struct ImArgs {
    const float *from, *to, *mv, *mn;
    int64_t fs, ts, mvs, mns;  // floats between consecutive elements: 0 (shared) or the dense row size
    int64_t B;
    uint32_t vec;  // bit i: array i (0 from, 1 to, 2 mv, 3 mn, 4 out / g, 5 gfrom, 6 gto, 7 gmv, 8 gmn) is 16-B aligned
};

// this wave's rows [b0, b0 + nrows) of a dense array of ROWDW-dword rows -> row[] of every lane
template <int ROWDW>
__device__ __forceinline__ void wave_rows_in(uint32_t *lds_w, const float *base, int64_t stride, int64_t b0,
                                             uint32_t nrows, bool vec_ok, int lane, float (&row)[ROWDW]) {
    if (stride == 0) {  // shared row: every lane reads the same ROWDW dwords
#pragma unroll
        for (int j = 0; j < ROWDW; ++j) row[j] = base[j];
        return;
    }
    const float *g = base + b0 * ROWDW;
    const uint32_t nbytes = nrows * ROWDW * 4u;
    wave_lds_fence();  // earlier readers of the region are done
    if (vec_ok && (nbytes & 15u) == 0) {
        RegionRegs<64 * ROWDW * 4> r;
        region_load_b128<64 * ROWDW * 4>(reinterpret_cast<const char *>(g), nbytes, lane, r);
        region_to_lds<64 * ROWDW * 4>(lds_w, lane, r);
    } else {
        region_load_b32(lds_w, reinterpret_cast<const uint32_t *>(g), nrows * ROWDW, lane);
    }
    wave_lds_fence();
    const float *l = reinterpret_cast<const float *>(lds_w) + lane * ROWDW;
#pragma unroll
    for (int j = 0; j < ROWDW; ++j) row[j] = l[j];
}

template <int ROWDW>
__device__ __forceinline__ void wave_rows_out(uint32_t *lds_w, float *base, int64_t b0, uint32_t nrows, bool vec_ok,
                                              int lane, const float (&row)[ROWDW]) {
    float *g = base + b0 * ROWDW;
    const uint32_t nbytes = nrows * ROWDW * 4u;
    wave_lds_fence();
    float *l = reinterpret_cast<float *>(lds_w) + lane * ROWDW;
#pragma unroll
    for (int j = 0; j < ROWDW; ++j) l[j] = row[j];
    wave_lds_fence();
    if (vec_ok && (nbytes & 15u) == 0)
        flush_region_b128<64 * ROWDW * 4>(lds_w, reinterpret_cast<char *>(g), nbytes, lane);
    else
        flush_region_b32(lds_w, reinterpret_cast<uint32_t *>(g), nrows * ROWDW, lane);
}

template <int K>
__device__ __forceinline__ void rows_to_v3(const float (&row)[3 * K], V3 (&v)[K]) {
#pragma unroll
    for (int j = 0; j < K; ++j) v[j] = V3{row[3 * j], row[3 * j + 1], row[3 * j + 2]};
}
template <int K>
__device__ __forceinline__ void v3_to_rows(const V3 (&v)[K], float (&row)[3 * K]) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
        row[3 * j] = v[j].x;
        row[3 * j + 1] = v[j].y;
        row[3 * j + 2] = v[j].z;
    }
}

template <int K>
__global__ __launch_bounds__(256) void image_method_kernel(ImArgs a, float *__restrict__ out) {
    constexpr int RDW = 3 * K;
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 64 * RDW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *lw = lds + wave * 64 * RDW;
    const int64_t b0 = ((int64_t)blockIdx.x * 4 + wave) * 64;  // wave-uniform
    if (b0 >= a.B) return;                                     // no block barrier below: a wave may leave
    const uint32_t nrows = (a.B - b0 >= 64) ? 64u : (uint32_t)(a.B - b0);
    float fr[3], tr[3], pr[RDW], nr[RDW];
    wave_rows_in<3>(lw, a.from, a.fs, b0, nrows, a.vec & 1u, lane, fr);
    wave_rows_in<3>(lw, a.to, a.ts, b0, nrows, a.vec & 2u, lane, tr);
    wave_rows_in<RDW>(lw, a.mv, a.mvs, b0, nrows, a.vec & 4u, lane, pr);
    wave_rows_in<RDW>(lw, a.mn, a.mns, b0, nrows, a.vec & 8u, lane, nr);
    V3 p[K], n[K], path[K];
    rows_to_v3<K>(pr, p);
    rows_to_v3<K>(nr, n);
    image_chain<K>(V3{fr[0], fr[1], fr[2]}, V3{tr[0], tr[1], tr[2]}, p, n, path);
    float orow[RDW];
    v3_to_rows<K>(path, orow);
    wave_rows_out<RDW>(lw, out, b0, nrows, a.vec & 16u, lane, orow);
}

template <int K>
__global__ __launch_bounds__(256) void image_method_vjp_kernel(ImArgs a, const float *__restrict__ gpath,
                                                               float *__restrict__ gfrom, float *__restrict__ gto,
                                                               float *__restrict__ gmv, float *__restrict__ gmn) {
    constexpr int RDW = 3 * K;
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 64 * RDW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *lw = lds + wave * 64 * RDW;
    const int64_t b0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    if (b0 >= a.B) return;
    const uint32_t nrows = (a.B - b0 >= 64) ? 64u : (uint32_t)(a.B - b0);
    float fr[3], tr[3], pr[RDW], nr[RDW], gr[RDW];
    wave_rows_in<3>(lw, a.from, a.fs, b0, nrows, a.vec & 1u, lane, fr);
    wave_rows_in<3>(lw, a.to, a.ts, b0, nrows, a.vec & 2u, lane, tr);
    wave_rows_in<RDW>(lw, a.mv, a.mvs, b0, nrows, a.vec & 4u, lane, pr);
    wave_rows_in<RDW>(lw, a.mn, a.mns, b0, nrows, a.vec & 8u, lane, nr);
    wave_rows_in<RDW>(lw, gpath, RDW, b0, nrows, a.vec & 16u, lane, gr);
    V3 p[K], n[K], g[K], pb[K], nb[K];
    rows_to_v3<K>(pr, p);
    rows_to_v3<K>(nr, n);
    rows_to_v3<K>(gr, g);
    V3 fb, tb;
    image_chain_vjp<K>(V3{fr[0], fr[1], fr[2]}, V3{tr[0], tr[1], tr[2]}, p, n, g, fb, tb, pb, nb);
    if (gfrom) {
        const float r3[3] = {fb.x, fb.y, fb.z};
        wave_rows_out<3>(lw, gfrom, b0, nrows, a.vec & 32u, lane, r3);
    }
    if (gto) {
        const float r3[3] = {tb.x, tb.y, tb.z};
        wave_rows_out<3>(lw, gto, b0, nrows, a.vec & 64u, lane, r3);
    }
    float orow[RDW];
    if (gmv) {
        v3_to_rows<K>(pb, orow);
        wave_rows_out<RDW>(lw, gmv, b0, nrows, a.vec & 128u, lane, orow);
    }
    if (gmn) {
        v3_to_rows<K>(nb, orow);
        wave_rows_out<RDW>(lw, gmn, b0, nrows, a.vec & 256u, lane, orow);
    }
}

// geometry/_solver_image_method.py:443-454: one lane per (batch element, mirror)
__global__ __launch_bounds__(256) void same_side_kernel(const float *__restrict__ vertices,
                                                        const float *__restrict__ mv,
                                                        const float *__restrict__ mn, int64_t B,
                                                        int K, uint8_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * K) return;
    const int64_t b = i / K;
    const int j = (int)(i % K);
    const V3 p = ld3(mv + 3 * i), n = ld3(mn + 3 * i);
    const V3 vp = ld3(vertices + 3 * ((K + 2) * b + j));
    const V3 vn = ld3(vertices + 3 * ((K + 2) * b + j + 2));
    out[i] = (uint8_t)same_sign(dot(vp - p, n), dot(vn - p, n));
}

// _solver_image_method.py:68-79 and :110-135, one lane per element
__global__ __launch_bounds__(256) void image_of_vertex_kernel(const float *__restrict__ x,
                                                              const float *__restrict__ p,
                                                              const float *__restrict__ n, int64_t B,
                                                              float *__restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    st3(out + 3 * b, image_of_vertex(ld3(x + 3 * b), ld3(p + 3 * b), ld3(n + 3 * b)));
}

__global__ __launch_bounds__(256) void ray_plane_kernel(const float *__restrict__ o,
                                                        const float *__restrict__ d,
                                                        const float *__restrict__ p,
                                                        const float *__restrict__ n, int64_t B,
                                                        float *__restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    st3(out + 3 * b, ray_plane(ld3(o + 3 * b), ld3(d + 3 * b), ld3(p + 3 * b), ld3(n + 3 * b)));
}

// geometry/_utils.py:66-72 normalize: v / where(|v| == 0, 1, |v|), lengths returned too
__global__ __launch_bounds__(256) void normalize_kernel(const float *__restrict__ v, int64_t B,
                                                        float *__restrict__ out,
                                                        float *__restrict__ len_out) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const V3 x = ld3(v + 3 * b);
    const float len = __builtin_sqrtf(dot(x, x));
    const float den = (len == 0.0f) ? 1.0f : len;
    st3(out + 3 * b, V3{x.x / den, x.y / den, x.z / den});
    if (len_out) len_out[b] = len;
}

template <int K>
static void launch_fwd(const ImArgs &a, float *out, hipStream_t s) {
    hipLaunchKernelGGL(image_method_kernel<K>, dim3((unsigned)ceil_div(a.B, 256)), dim3(256), 0, s, a, out);
}
template <int K>
static void launch_vjp(const ImArgs &a, const float *g, float *gf, float *gt, float *gmv, float *gmn, hipStream_t s) {
    hipLaunchKernelGGL(image_method_vjp_kernel<K>, dim3((unsigned)ceil_div(a.B, 256)), dim3(256), 0, s, a, g, gf, gt,
                       gmv, gmn);
}

static uint32_t al16(const void *p, int bit) { return ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) ? (1u << bit) : 0u; }

static int32_t check_stride(int64_t s, int64_t dense, const char *what) {
    if (s != 0 && s != dense) return fail(DRT_E_INVALID, "%s stride must be 0 (shared) or %lld (dense), got %lld", what,
                                          (long long)dense, (long long)s);
    return DRT_OK;
}

}  // namespace drt

using namespace drt;

#define DRT_DISPATCH_ORDER(k, CALL)                                                        \
    switch (k) {                                                                           \
        case 1: CALL(1); break;                                                            \
        case 2: CALL(2); break;                                                            \
        case 3: CALL(3); break;                                                            \
        case 4: CALL(4); break;                                                            \
        case 5: CALL(5); break;                                                            \
        case 6: CALL(6); break;                                                            \
        case 7: CALL(7); break;                                                            \
        case 8: CALL(8); break;                                                            \
        default:                                                                           \
            return fail(DRT_E_UNSUPPORTED, "order %d not supported (max %d)", (int)(k),    \
                        DRT_MAX_ORDER);                                                    \
    }

extern "C" {

int32_t drt_normalize(const float *vectors, int64_t B, float *out, float *lengths_out, void *stream) {
    DRT_REQUIRE(B >= 0, "negative size");
    if (B == 0) return DRT_OK;
    DRT_REQUIRE(vectors && out, "null pointer");
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0,
                       as_stream(stream), vectors, B, out, lengths_out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_image_of_vertex(const float *x, const float *p, const float *n, int64_t B, float *out,
                            void *stream) {
    DRT_REQUIRE(B >= 0, "negative size");
    if (B == 0) return DRT_OK;
    DRT_REQUIRE(x && p && n && out, "null pointer");
    hipLaunchKernelGGL(image_of_vertex_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0,
                       as_stream(stream), x, p, n, B, out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_intersection_of_ray_with_plane(const float *o, const float *d, const float *p,
                                           const float *n, int64_t B, float *out, void *stream) {
    DRT_REQUIRE(B >= 0, "negative size");
    if (B == 0) return DRT_OK;
    DRT_REQUIRE(o && d && p && n && out, "null pointer");
    hipLaunchKernelGGL(ray_plane_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0,
                       as_stream(stream), o, d, p, n, B, out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_image_method_strided(const float *from, int64_t from_stride, const float *to, int64_t to_stride,
                                 const float *mv, int64_t mv_stride, const float *mn, int64_t mn_stride, int64_t B,
                                 int32_t k, float *out, void *stream) {
    DRT_REQUIRE(B >= 0 && k >= 0, "negative size");
    if (B == 0 || k == 0) return DRT_OK;  // _solver_image_method.py:349-358
    DRT_REQUIRE(from && to && mv && mn && out, "null pointer");
    int32_t rc;
    if ((rc = check_stride(from_stride, 3, "from")) || (rc = check_stride(to_stride, 3, "to")) ||
        (rc = check_stride(mv_stride, 3 * (int64_t)k, "mirror_vertices")) ||
        (rc = check_stride(mn_stride, 3 * (int64_t)k, "mirror_normals")))
        return rc;
    hipStream_t s = as_stream(stream);
    ImArgs a{from, to, mv, mn, from_stride, to_stride, mv_stride, mn_stride, B,
             al16(from, 0) | al16(to, 1) | al16(mv, 2) | al16(mn, 3) | al16(out, 4)};
#define CALL(K) launch_fwd<K>(a, out, s)
    DRT_DISPATCH_ORDER(k, CALL)
#undef CALL
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_image_method(const float *from, const float *to, const float *mv, const float *mn,
                         int64_t B, int32_t k, float *out, void *stream) {
    return drt_image_method_strided(from, 3, to, 3, mv, 3 * (int64_t)k, mn, 3 * (int64_t)k, B, k, out, stream);
}

int32_t drt_image_method_vjp_strided(const float *from, int64_t from_stride, const float *to, int64_t to_stride,
                                     const float *mv, int64_t mv_stride, const float *mn, int64_t mn_stride,
                                     const float *g, int64_t B, int32_t k, float *gf, float *gt, float *gmv,
                                     float *gmn, void *stream) {
    DRT_REQUIRE(B >= 0 && k >= 0, "negative size");
    hipStream_t s = as_stream(stream);
    if (B == 0) return DRT_OK;
    if (k == 0) {  // no mirrors: the (empty) output does not depend on anything
        if (gf) DRT_HIP(fill_bytes_async(gf, 0, (size_t)B * 12, s));
        if (gt) DRT_HIP(fill_bytes_async(gt, 0, (size_t)B * 12, s));
        return DRT_OK;
    }
    DRT_REQUIRE(from && to && mv && mn && g, "null pointer");
    int32_t rc;
    if ((rc = check_stride(from_stride, 3, "from")) || (rc = check_stride(to_stride, 3, "to")) ||
        (rc = check_stride(mv_stride, 3 * (int64_t)k, "mirror_vertices")) ||
        (rc = check_stride(mn_stride, 3 * (int64_t)k, "mirror_normals")))
        return rc;
    ImArgs a{from, to, mv, mn, from_stride, to_stride, mv_stride, mn_stride, B,
             al16(from, 0) | al16(to, 1) | al16(mv, 2) | al16(mn, 3) | al16(g, 4) | al16(gf, 5) | al16(gt, 6) |
                 al16(gmv, 7) | al16(gmn, 8)};
#define CALL(K) launch_vjp<K>(a, g, gf, gt, gmv, gmn, s)
    DRT_DISPATCH_ORDER(k, CALL)
#undef CALL
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_image_method_vjp(const float *from, const float *to, const float *mv, const float *mn,
                             const float *g, int64_t B, int32_t k, float *gf, float *gt, float *gmv,
                             float *gmn, void *stream) {
    return drt_image_method_vjp_strided(from, 3, to, 3, mv, 3 * (int64_t)k, mn, 3 * (int64_t)k, g, B, k, gf, gt, gmv,
                                        gmn, stream);
}

int32_t drt_consecutive_vertices_same_side(const float *vertices, const float *mv, const float *mn,
                                           int64_t B, int32_t k, uint8_t *out, void *stream) {
    DRT_REQUIRE(B >= 0 && k >= 0, "negative size");
    if (B == 0 || k == 0) return DRT_OK;
    DRT_REQUIRE(vertices && mv && mn && out, "null pointer");
    hipLaunchKernelGGL(same_side_kernel, dim3((unsigned)ceil_div(B * k, 256)), dim3(256), 0,
                       as_stream(stream), vertices, mv, mn, B, (int)k, out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

}  // extern "C"
