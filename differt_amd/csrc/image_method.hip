// image_method.hip -- stand-alone image-method operators over flat batches.
//   drt_image_method              reference geometry/_solver_image_method.py:206-363
//   drt_image_method_vjp          reverse of the two lax.scans (:185-203), hand-derived
//   drt_consecutive_vertices_same_side   reference :386-454
// One lane per batch element; the chain is fully unrolled for the (small) number of mirrors.
#include "common.hpp"
#include "image_chain.hpp"

#pragma clang fp contract(off)

namespace drt {

template <int K>
__global__ __launch_bounds__(256) void image_method_kernel(const float *__restrict__ from,
                                                           const float *__restrict__ to,
                                                           const float *__restrict__ mv,
                                                           const float *__restrict__ mn, int64_t B,
                                                           float *__restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    V3 p[K], n[K], path[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        p[j] = ld3(mv + 3 * (K * b + j));
        n[j] = ld3(mn + 3 * (K * b + j));
    }
    image_chain<K>(ld3(from + 3 * b), ld3(to + 3 * b), p, n, path);
#pragma unroll
    for (int j = 0; j < K; ++j) st3(out + 3 * (K * b + j), path[j]);
}

template <int K>
__global__ __launch_bounds__(256) void image_method_vjp_kernel(
    const float *__restrict__ from, const float *__restrict__ to, const float *__restrict__ mv,
    const float *__restrict__ mn, const float *__restrict__ gpath, int64_t B,
    float *__restrict__ gfrom, float *__restrict__ gto, float *__restrict__ gmv,
    float *__restrict__ gmn) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    V3 p[K], n[K], g[K], pb[K], nb[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        p[j] = ld3(mv + 3 * (K * b + j));
        n[j] = ld3(mn + 3 * (K * b + j));
        g[j] = ld3(gpath + 3 * (K * b + j));
    }
    V3 fb, tb;
    image_chain_vjp<K>(ld3(from + 3 * b), ld3(to + 3 * b), p, n, g, fb, tb, pb, nb);
    if (gfrom) st3(gfrom + 3 * b, fb);
    if (gto) st3(gto + 3 * b, tb);
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (gmv) st3(gmv + 3 * (K * b + j), pb[j]);
        if (gmn) st3(gmn + 3 * (K * b + j), nb[j]);
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
static void launch_fwd(const float *from, const float *to, const float *mv, const float *mn,
                       int64_t B, float *out, hipStream_t s) {
    hipLaunchKernelGGL(image_method_kernel<K>, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, s,
                       from, to, mv, mn, B, out);
}
template <int K>
static void launch_vjp(const float *from, const float *to, const float *mv, const float *mn,
                       const float *g, int64_t B, float *gf, float *gt, float *gmv, float *gmn,
                       hipStream_t s) {
    hipLaunchKernelGGL(image_method_vjp_kernel<K>, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, s,
                       from, to, mv, mn, g, B, gf, gt, gmv, gmn);
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

int32_t drt_image_method(const float *from, const float *to, const float *mv, const float *mn,
                         int64_t B, int32_t k, float *out, void *stream) {
    DRT_REQUIRE(B >= 0 && k >= 0, "negative size");
    if (B == 0 || k == 0) return DRT_OK;  // _solver_image_method.py:349-358
    DRT_REQUIRE(from && to && mv && mn && out, "null pointer");
    hipStream_t s = as_stream(stream);
#define CALL(K) launch_fwd<K>(from, to, mv, mn, B, out, s)
    DRT_DISPATCH_ORDER(k, CALL)
#undef CALL
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_image_method_vjp(const float *from, const float *to, const float *mv, const float *mn,
                             const float *g, int64_t B, int32_t k, float *gf, float *gt, float *gmv,
                             float *gmn, void *stream) {
    DRT_REQUIRE(B >= 0 && k >= 0, "negative size");
    hipStream_t s = as_stream(stream);
    if (B == 0) return DRT_OK;
    if (k == 0) {  // no mirrors: the (empty) output does not depend on anything
        if (gf) DRT_HIP(hipMemsetAsync(gf, 0, (size_t)B * 12, s));
        if (gt) DRT_HIP(hipMemsetAsync(gt, 0, (size_t)B * 12, s));
        return DRT_OK;
    }
    DRT_REQUIRE(from && to && mv && mn && g, "null pointer");
#define CALL(K) launch_vjp<K>(from, to, mv, mn, g, B, gf, gt, gmv, gmn, s)
    DRT_DISPATCH_ORDER(k, CALL)
#undef CALL
    DRT_LAUNCH_CHECK();
    return DRT_OK;
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
