// visibility.hip -- triangles visible from a vertex by ray launching ("next" row f2 of SURVEY.md 8f):
//   viewing_frustum              reference geometry/_utils.py:639-927
//   fibonacci_lattice            reference geometry/_utils.py:369-490 (split-modulus frac, :426-462)
//   triangles_visible_from_vertex reference geometry/_utils.py:1540-1772, Mesh method _mesh.py:3164-3253
// The lattice is never materialised: each lane derives its ray from (i, frustum), finds the first
// triangle it hits over LDS-staged tiles (same arithmetic and tie rule as first_triangle_hit_by_ray
// with batch_size=None: lowest index among equal t) and marks it visible (benign race, like the
// reference's Warp kernel, _mesh.py:364-366).
#include "common.hpp"
#include "geom.hpp"
#include "lattice.hpp"
#include "tri_tile.hpp"

#pragma clang fp contract(off)

namespace drt {

struct Frustum {
    float r_min, p_min, a_min, r_max, p_max, a_max;
};

// one block per viewing vertex; world vertices = triangle vertices + triangle centres
// (geometry/_utils.py:1669-1673), active per triangle.
__global__ __launch_bounds__(256) void frustum_kernel(const float *__restrict__ view, int64_t B,
                                                      const float *__restrict__ tv, int64_t T,
                                                      const uint8_t *__restrict__ active,
                                                      float *__restrict__ out) {
    const int64_t b = blockIdx.x;
    const V3 vv = ld3(view + 3 * b);
    float r_min = kInf, r_max = 0.0f, p_min = kPi, p_max = 0.0f;
    float a_min = kPi, a_max = -kPi, a0_min = kTwoPi, a0_max = 0.0f;
    for (int64_t t = threadIdx.x; t < T; t += 256) {
        if (active && !active[t]) continue;
        const V3 v0 = ld3(tv + 9 * t), v1 = ld3(tv + 9 * t + 3), v2 = ld3(tv + 9 * t + 6);
        const V3 s = (v0 + v1) + v2;
        const V3 w[4] = {v0, v1, v2, V3{s.x / 3.0f, s.y / 3.0f, s.z / 3.0f}};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const V3 x = w[k] - vv;
            float r = __builtin_sqrtf(dot(x, x));  // cartesian_to_spherical, :952-956
            r = (r == 0.0f) ? 1.0f : r;
            const float p = acosf(x.z / r);
            const float a = atan2f(x.y, x.x);
            const float a0 = fmodf(a + kTwoPi, kTwoPi);
            r_min = fminf(r_min, r); r_max = fmaxf(r_max, r);
            p_min = fminf(p_min, p); p_max = fmaxf(p_max, p);
            a_min = fminf(a_min, a); a_max = fmaxf(a_max, a);
            a0_min = fminf(a0_min, a0); a0_max = fmaxf(a0_max, a0);
        }
    }
    __shared__ float red[8][256];
    float vals[8] = {r_min, -r_max, p_min, -p_max, a_min, -a_max, a0_min, -a0_max};  // all as minima
#pragma unroll
    for (int k = 0; k < 8; ++k) red[k][threadIdx.x] = vals[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                red[k][threadIdx.x] = fminf(red[k][threadIdx.x], red[k][threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        r_min = red[0][0]; r_max = -red[1][0]; p_min = red[2][0]; p_max = -red[3][0];
        a_min = red[4][0]; a_max = -red[5][0]; a0_min = red[6][0]; a0_max = -red[7][0];
        // azimuth: keep the narrower of the [-pi,pi) and [0,2pi) domains (:850-877), full circle
        // when both exceed 270 degrees (:879-890)
        const float a_width = a_max - a_min, a0_width = a0_max - a0_min;
        if (a_width > a0_width) { a_min = a0_min; a_max = a0_max; }
        if (fminf(a_width, a0_width) > 1.5f * kPi) { a_min = -kPi; a_max = kPi; }
        // degenerate polar band (:892-915)
        float p0_min = p_min, p0_max = p_max;
        if (p_min == p_max) { p_min = 0.0f; p0_max = kPi; }
        if ((p_max - p_min) > (p0_max - p0_min)) { p_min = p0_min; p_max = p0_max; }
        float *o = out + 6 * b;
        o[0] = r_min; o[1] = p_min; o[2] = a_min; o[3] = r_max; o[4] = p_max; o[5] = a_max;
    }
}

__global__ __launch_bounds__(256) void lattice_kernel(int64_t n, const float *__restrict__ fr,
                                                      float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    st3(out + 3 * i, lattice_direction(i, n, fr));
}

// grid (ceil(num_rays / 256), B); lane = lattice ray
__global__ __launch_bounds__(256) void visibility_kernel(const float *__restrict__ view,
                                                         const float *__restrict__ frusta,
                                                         int64_t num_rays,
                                                         const float *__restrict__ tv, int64_t T,
                                                         const uint8_t *__restrict__ active, float eps,
                                                         uint8_t *__restrict__ visible) {
    __shared__ TriRec lds[kTile];
    const int64_t b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < num_rays;
    const V3 o = ld3(view + 3 * b);
    const V3 d = lattice_direction(valid ? i : 0, num_rays, frusta + 6 * b);
    float best_t = kInf;
    int64_t best_j = -1;
    for (int64_t base = 0; base < T; base += kTile) {
        __syncthreads();
        stage_tile(lds, tv, active, base, T);
        __syncthreads();
        const int n = (int)((T - base < kTile) ? T - base : kTile);
        for (int j = 0; j < n; ++j) {
            const TriRec rec = lds[j];
            const TriE tr = rec_tri(rec);
            float t;
            const bool h = moller_trumbore(o, d, tr, eps, t);
            if (h && rec.active && t < best_t) {  // strict <: the lowest index wins ties (argmin)
                best_t = t;
                best_j = base + j;
            }
        }
    }
    if (valid && best_j >= 0 && is_finite(best_t)) visible[b * T + best_j] = 1;
}

// frustum over an arbitrary list of world points (SBR: triangle vertices + receivers,
// geometry/_solvers.py:1213-1219)
__global__ __launch_bounds__(256) void frustum_points_kernel(const float *__restrict__ view,
                                                             const float *__restrict__ pts, int64_t N,
                                                             float *__restrict__ out) {
    const int64_t b = blockIdx.x;
    const V3 vv = ld3(view + 3 * b);
    float r_min = kInf, r_max = 0.0f, p_min = kPi, p_max = 0.0f;
    float a_min = kPi, a_max = -kPi, a0_min = kTwoPi, a0_max = 0.0f;
    for (int64_t k = threadIdx.x; k < N; k += 256) {
        const V3 x = ld3(pts + 3 * k) - vv;
        float r = __builtin_sqrtf(dot(x, x));
        r = (r == 0.0f) ? 1.0f : r;
        const float p = acosf(x.z / r);
        const float a = atan2f(x.y, x.x);
        const float a0 = fmodf(a + kTwoPi, kTwoPi);
        r_min = fminf(r_min, r); r_max = fmaxf(r_max, r);
        p_min = fminf(p_min, p); p_max = fmaxf(p_max, p);
        a_min = fminf(a_min, a); a_max = fmaxf(a_max, a);
        a0_min = fminf(a0_min, a0); a0_max = fmaxf(a0_max, a0);
    }
    __shared__ float red[8][256];
    float vals[8] = {r_min, -r_max, p_min, -p_max, a_min, -a_max, a0_min, -a0_max};
#pragma unroll
    for (int k = 0; k < 8; ++k) red[k][threadIdx.x] = vals[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                red[k][threadIdx.x] = fminf(red[k][threadIdx.x], red[k][threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        r_min = red[0][0]; r_max = -red[1][0]; p_min = red[2][0]; p_max = -red[3][0];
        a_min = red[4][0]; a_max = -red[5][0]; a0_min = red[6][0]; a0_max = -red[7][0];
        const float a_width = a_max - a_min, a0_width = a0_max - a0_min;
        if (a_width > a0_width) { a_min = a0_min; a_max = a0_max; }
        if (fminf(a_width, a0_width) > 1.5f * kPi) { a_min = -kPi; a_max = kPi; }
        float p0_min = p_min, p0_max = p_max;
        if (p_min == p_max) { p_min = 0.0f; p0_max = kPi; }
        if ((p_max - p_min) > (p0_max - p0_min)) { p_min = p0_min; p_max = p0_max; }
        float *o = out + 6 * b;
        o[0] = r_min; o[1] = p_min; o[2] = a_min; o[3] = r_max; o[4] = p_max; o[5] = a_max;
    }
}
// General form of viewing_frustum (_utils.py:639-927): per-viewer point sets (stride 3 N or 0 = shared),
// per-point mask (stride N or 0), and `reduce` (one frustum over every batch entry: the min / max run over
// viewers AND points before the azimuth / polar selection logic, :838-846 with axis = None).
__device__ __forceinline__ void frustum_finish(const float (&m)[8], float *o) {
    float r_min = m[0], r_max = -m[1], p_min = m[2], p_max = -m[3];
    float a_min = m[4], a_max = -m[5], a0_min = m[6], a0_max = -m[7];
    const float a_width = a_max - a_min, a0_width = a0_max - a0_min;
    if (a_width > a0_width) { a_min = a0_min; a_max = a0_max; }
    if (fminf(a_width, a0_width) > 1.5f * kPi) { a_min = -kPi; a_max = kPi; }
    float p0_min = p_min, p0_max = p_max;
    if (p_min == p_max) { p_min = 0.0f; p0_max = kPi; }
    if ((p_max - p_min) > (p0_max - p0_min)) { p_min = p0_min; p_max = p0_max; }
    o[0] = r_min; o[1] = p_min; o[2] = a_min; o[3] = r_max; o[4] = p_max; o[5] = a_max;
}

__global__ __launch_bounds__(256) void frustum_general_kernel(const float *__restrict__ view,
                                                              const float *__restrict__ pts, int64_t N,
                                                              int64_t pts_stride, const uint8_t *__restrict__ act,
                                                              int64_t act_stride, float *__restrict__ raw,
                                                              float *__restrict__ out) {
    const int64_t b = blockIdx.x;
    const V3 vv = ld3(view + 3 * b);
    const float *P = pts + b * pts_stride;
    const uint8_t *A = act ? act + b * act_stride : nullptr;
    float r_min = kInf, r_max = 0.0f, p_min = kPi, p_max = 0.0f;
    float a_min = kPi, a_max = -kPi, a0_min = kTwoPi, a0_max = 0.0f;
    for (int64_t k = threadIdx.x; k < N; k += 256) {
        if (A && !A[k]) continue;
        const V3 x = ld3(P + 3 * k) - vv;
        float r = __builtin_sqrtf(dot(x, x));
        r = (r == 0.0f) ? 1.0f : r;
        const float p = acosf(x.z / r);
        const float a = atan2f(x.y, x.x);
        const float a0 = fmodf(a + kTwoPi, kTwoPi);
        r_min = fminf(r_min, r); r_max = fmaxf(r_max, r);
        p_min = fminf(p_min, p); p_max = fmaxf(p_max, p);
        a_min = fminf(a_min, a); a_max = fmaxf(a_max, a);
        a0_min = fminf(a0_min, a0); a0_max = fmaxf(a0_max, a0);
    }
    __shared__ float red[8][256];
    float vals[8] = {r_min, -r_max, p_min, -p_max, a_min, -a_max, a0_min, -a0_max};
#pragma unroll
    for (int k = 0; k < 8; ++k) red[k][threadIdx.x] = vals[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                red[k][threadIdx.x] = fminf(red[k][threadIdx.x], red[k][threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = red[k][0];
        if (raw) {
#pragma unroll
            for (int k = 0; k < 8; ++k) raw[8 * b + k] = m[k];
        } else {
            frustum_finish(m, out + 6 * b);
        }
    }
}

__global__ __launch_bounds__(256) void frustum_reduce_kernel(const float *__restrict__ raw, int64_t B,
                                                             float *__restrict__ out) {
    __shared__ float red[8][256];
    float m[8] = {kInf, 0.0f, kPi, 0.0f, kPi, kPi, kTwoPi, 0.0f};  // minima of (x, -x_max): identities
    for (int64_t b = threadIdx.x; b < B; b += 256)
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = fminf(m[k], raw[8 * b + k]);
#pragma unroll
    for (int k = 0; k < 8; ++k) red[k][threadIdx.x] = m[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                red[k][threadIdx.x] = fminf(red[k][threadIdx.x], red[k][threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = red[k][0];
        frustum_finish(m, out);
    }
}


void launch_frustum_kernel(const float *view, int64_t B, const float *tv, int64_t T,
                           const uint8_t *active, float *out, hipStream_t s) {
    hipLaunchKernelGGL(frustum_kernel, dim3((unsigned)B), dim3(256), 0, s, view, B, tv, T, active, out);
}

// geometry/_utils.py:930-993: (r, polar, azimuth) <-> (x, y, z); r == 0 -> polar = acos(z / 1)
__global__ __launch_bounds__(256) void cart_to_sph_kernel(const float *__restrict__ xyz, int64_t B,
                                                          float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const V3 v = ld3(xyz + 3 * i);
    float r = __builtin_sqrtf(dot(v, v));
    r = (r == 0.0f) ? 1.0f : r;
    st3(out + 3 * i, V3{r, acosf(v.z / r), atan2f(v.y, v.x)});
}

__global__ __launch_bounds__(256) void sph_to_cart_kernel(const float *__restrict__ rpa, int64_t B, int width,
                                                          float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const float *q = rpa + (int64_t)width * i;
    const float p = q[width - 2], a = q[width - 1];
    const float sp = sinf(p), cp = cosf(p);
    V3 v{sp * cosf(a), sp * sinf(a), cp};
    if (width == 3) v = v * q[0];
    st3(out + 3 * i, v);
}

}  // namespace drt

using namespace drt;

extern "C" {

int32_t drt_cartesian_to_spherical(const float *xyz, int64_t batch, float *rpa_out, void *stream) {
    DRT_REQUIRE(batch >= 0, "negative size");
    if (batch == 0) return DRT_OK;
    DRT_REQUIRE(xyz && rpa_out, "null pointer");
    hipLaunchKernelGGL(cart_to_sph_kernel, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, as_stream(stream),
                       xyz, batch, rpa_out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_spherical_to_cartesian(const float *rpa, int64_t batch, int32_t width, float *xyz_out, void *stream) {
    DRT_REQUIRE(batch >= 0, "negative size");
    DRT_REQUIRE(width == 2 || width == 3, "rpa must hold (polar, azimuth) or (r, polar, azimuth)");
    if (batch == 0) return DRT_OK;
    DRT_REQUIRE(rpa && xyz_out, "null pointer");
    hipLaunchKernelGGL(sph_to_cart_kernel, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, as_stream(stream),
                       rpa, batch, (int)width, xyz_out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_viewing_frustum(const float *viewing_vertices, int64_t B, const float *tv, int64_t T,
                            const uint8_t *active, float *frustum_out, void *stream) {
    DRT_REQUIRE(B >= 0 && T >= 0, "negative size");
    if (B == 0) return DRT_OK;
    DRT_REQUIRE(viewing_vertices && frustum_out && (T == 0 || tv), "null pointer");
    hipLaunchKernelGGL(frustum_kernel, dim3((unsigned)B), dim3(256), 0, as_stream(stream),
                       viewing_vertices, B, tv, T, active, frustum_out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_viewing_frustum_points(const float *viewing_vertices, int64_t B, const float *points,
                                   int64_t N, float *frustum_out, void *stream) {
    DRT_REQUIRE(B >= 0 && N >= 0, "negative size");
    if (B == 0) return DRT_OK;
    DRT_REQUIRE(viewing_vertices && frustum_out && (N == 0 || points), "null pointer");
    hipLaunchKernelGGL(frustum_points_kernel, dim3((unsigned)B), dim3(256), 0, as_stream(stream),
                       viewing_vertices, points, N, frustum_out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_viewing_frustum_general(const float *viewing_vertices, int64_t B, const float *points, int64_t N,
                                    int64_t points_viewer_stride, const uint8_t *active,
                                    int64_t active_viewer_stride, int32_t reduce, float *workspace,
                                    float *frustum_out, void *stream) {
    DRT_REQUIRE(B >= 0 && N >= 0, "negative size");
    if (B == 0) return DRT_OK;
    DRT_REQUIRE(viewing_vertices && frustum_out && (N == 0 || points), "null pointer");
    DRT_REQUIRE(points_viewer_stride == 0 || points_viewer_stride == 3 * N, "points_viewer_stride must be 0 or 3*N");
    DRT_REQUIRE(active_viewer_stride == 0 || active_viewer_stride == N, "active_viewer_stride must be 0 or N");
    DRT_REQUIRE(!reduce || workspace, "reduce needs a workspace of 8*B floats");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(frustum_general_kernel, dim3((unsigned)B), dim3(256), 0, s, viewing_vertices, points, N,
                       points_viewer_stride, active, active_viewer_stride, reduce ? workspace : (float *)nullptr,
                       frustum_out);
    if (reduce)
        hipLaunchKernelGGL(frustum_reduce_kernel, dim3(1), dim3(256), 0, s, workspace, B, frustum_out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_fibonacci_lattice(int64_t n, const float *frustum, float *out, void *stream) {
    DRT_REQUIRE(n > 0, "Invalid size %lld, must be strictly positive.", (long long)n);
    DRT_REQUIRE(out, "null output");
    hipLaunchKernelGGL(lattice_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       as_stream(stream), n, frustum, out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_triangles_visible_from_vertex(const float *vertices, int64_t B, const float *tv, int64_t T,
                                          const uint8_t *active, int64_t num_rays, float epsilon,
                                          uint8_t *visible_out, float *frustum_workspace,
                                          void *stream) {
    DRT_REQUIRE(B >= 0 && T >= 0, "negative size");
    DRT_REQUIRE(num_rays > 0, "num_rays must be strictly positive");
    if (B == 0 || T == 0) return DRT_OK;
    DRT_REQUIRE(vertices && tv && visible_out && frustum_workspace, "null pointer");
    DRT_REQUIRE(B <= 65535, "at most 65535 viewing vertices per call");
    hipStream_t s = as_stream(stream);
    DRT_HIP(fill_bytes_async(visible_out, 0, (size_t)B * (size_t)T, s));  // (a kernel: memset nodes do not survive graph replay, core.hip)
    hipLaunchKernelGGL(frustum_kernel, dim3((unsigned)B), dim3(256), 0, s, vertices, B, tv, T, active,
                       frustum_workspace);
    DRT_LAUNCH_CHECK();
    hipLaunchKernelGGL(visibility_kernel, dim3((unsigned)ceil_div(num_rays, 256), (unsigned)B),
                       dim3(256), 0, s, vertices, frustum_workspace, num_rays, tv, T, active, epsilon,
                       visible_out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

}  // extern "C"
