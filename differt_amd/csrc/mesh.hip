// mesh.hip -- drt_mesh_t: an explicit, caller-owned handle holding what the hot path needs from the
// reference's `Mesh` (geometry/_mesh.py:624-688): gathered triangle vertices (:899-905) and unit
// normals (:950-956 = normalize(cross(v1 - v0, v2 - v1))), plus the optional triangle mask.
// It replaces the reference's hidden, id()-keyed `_WARP_MESHES_CACHE` (_mesh.py:55, 170-174).
#include "mesh.hpp"

#include "beam_margins.hpp"

#include "common.hpp"
#include "geom.hpp"

#pragma clang fp contract(off)

namespace drt {

__global__ __launch_bounds__(256) void mesh_prepare_kernel(const float *__restrict__ vertices,
                                                           int64_t num_vertices,
                                                           const int32_t *__restrict__ triangles,
                                                           int64_t T, float *__restrict__ tv,
                                                           float *__restrict__ normals,
                                                           float *__restrict__ shape,
                                                           int32_t *__restrict__ bad_index) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    V3 v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int64_t i = triangles[3 * t + c];
        if (i < 0 || i >= num_vertices) {
            *bad_index = 1;
            i = 0;
        }
        v[c] = ld3(vertices + 3 * i);
        st3(tv + 9 * t + 3 * c, v[c]);
    }
    // jnp.diff(tv, axis=1) -> (v1 - v0, v2 - v1); cross; normalize (_utils.py:66-72)
    const V3 c = cross(v[1] - v[0], v[2] - v[1]);
    const float len = __builtin_sqrtf(dot(c, c));
    const float den = (len == 0.0f) ? 1.0f : len;
    st3(normals + 3 * t, V3{c.x / den, c.y / den, c.z / den});
    // shape factor sigma >= 1 for the error bounds of the beam pruning (csrc/beam.hip): the largest
    // 1 / sin(corner angle) = |a| |b| / |a x b|; a Moller-Trumbore inside test loses that factor in
    // position accuracy on a sliver.  Degenerate triangle -> +inf (its tests switch themselves off).
    const V3 ea = v[1] - v[0], eb = v[2] - v[1], ec = v[0] - v[2];
    const float la = __builtin_sqrtf(dot(ea, ea)), lb = __builtin_sqrtf(dot(eb, eb)), lc = __builtin_sqrtf(dot(ec, ec));
    const float pm = fmaxf(la * lb, fmaxf(lb * lc, lc * la));
    const float sg = (len > 0.0f) ? pm / len : kInf;
    shape[t] = (sg >= 1.0f && sg < kInf) ? sg * margins::kSigmaRoundUp : ((sg < 1.0f) ? 1.0f : kInf);
}

}  // namespace drt

using namespace drt;

// Ray preparation of the reference's Warp-backed mesh queries (opt-in `semantics="warp"`):
//   mode 0, any-hit  (_mesh.py:3065-3070): direction normalised (len = |d|, zero vector kept: UT:29-72),
//           origin moved by hit_tol * len along it, segment shortened to len * (1 - 2 hit_tol);
//           written back as a SEGMENT (o', d' = dir * max_t) so that the any-hit operator with
//           hit_tol = 0 evaluates `0 < t <= max_t` on it;
//   mode 1, first-hit (_mesh.py:195-199): origin nudged by 1e-5 * d, direction unchanged.
__global__ __launch_bounds__(256) void warp_ray_prep_kernel(const float *__restrict__ ro,
                                                            const float *__restrict__ rd, int64_t n, int mode,
                                                            float param, float *__restrict__ o_out,
                                                            float *__restrict__ d_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const V3 o = ld3(ro + 3 * i), d = ld3(rd + 3 * i);
    if (mode == 0) {
        const float len = sqrtf(dot(d, d));
        const float den = (len == 0.0f) ? 1.0f : len;
        const V3 dir = V3{d.x / den, d.y / den, d.z / den};
        const float off = param * len;
        const float max_t = len * (1.0f - 2.0f * param);
        st3(o_out + 3 * i, V3{o.x + dir.x * off, o.y + dir.y * off, o.z + dir.z * off});
        st3(d_out + 3 * i, dir * max_t);
    } else {
        st3(o_out + 3 * i, V3{o.x + d.x * param, o.y + d.y * param, o.z + d.z * param});
        st3(d_out + 3 * i, d);
    }
}

// t += nudge on hits (res.t + epsilon, _mesh.py:199); misses keep (-1, inf)
__global__ __launch_bounds__(256) void warp_first_hit_finish_kernel(const int32_t *__restrict__ idx,
                                                                    float *__restrict__ t, int64_t n, float nudge) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && idx[i] >= 0) t[i] = t[i] + nudge;
}

extern "C" {

int32_t drt_warp_ray_prep(const float *ro, const float *rd, int64_t n, int32_t mode, float param, float *o_out,
                          float *d_out, void *stream) {
    DRT_REQUIRE(n >= 0 && (mode == 0 || mode == 1), "bad argument");
    if (n == 0) return DRT_OK;
    DRT_REQUIRE(ro && rd && o_out && d_out, "null pointer");
    hipLaunchKernelGGL(warp_ray_prep_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), ro, rd,
                       n, (int)mode, param, o_out, d_out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_warp_first_hit_finish(const int32_t *idx, float *t, int64_t n, float nudge, void *stream) {
    DRT_REQUIRE(n >= 0, "negative size");
    if (n == 0) return DRT_OK;
    DRT_REQUIRE(idx && t, "null pointer");
    hipLaunchKernelGGL(warp_first_hit_finish_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       as_stream(stream), idx, t, n, nudge);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_mesh_destroy(drt_mesh_t m) {
    if (!m) return DRT_OK;
    (void)hipFree(m->vertices);
    (void)hipFree(m->triangles);
    (void)hipFree(m->tri_verts);
    (void)hipFree(m->normals);
    (void)hipFree(m->shape);
    (void)hipFree(m->mask);
    (void)hipFree(m->bvh_nodes);
    (void)hipFree(m->bvh_leaf_ids);
    (void)hipFree(m->beam[0].blob);
    (void)hipFree(m->beam[1].blob);
    (void)hipFree(m->pair_blob);
    delete m;
    return DRT_OK;
}

int32_t drt_mesh_create(const float *vertices, int64_t num_vertices, const int32_t *triangles,
                        int64_t T, const uint8_t *mask, int32_t assume_quads, void *stream,
                        drt_mesh_t *mesh_out) {
    DRT_REQUIRE(mesh_out, "mesh_out is null");
    *mesh_out = nullptr;
    DRT_REQUIRE(num_vertices >= 0 && T >= 0, "negative size");
    DRT_REQUIRE(T == 0 || (vertices && triangles), "null vertices/triangles");
    DRT_REQUIRE(!assume_quads || T % 2 == 0, "assume_quads needs an even number of triangles");
    DRT_REQUIRE(T < (1ll << 31), "too many triangles");
    hipStream_t s = as_stream(stream);
    drt_mesh *m = new drt_mesh();
    m->num_vertices = num_vertices;
    m->num_triangles = T;
    m->assume_quads = assume_quads ? 1 : 0;
    m->has_mask = mask ? 1 : 0;
    int32_t *bad = nullptr;
    int32_t rc = DRT_OK;
    auto bail = [&](int32_t code) {
        (void)hipFree(bad);
        drt_mesh_destroy(m);
        return code;
    };
#define TRY_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return bail(fail(DRT_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)));      \
    } while (0)
    // zero-size allocations are avoided: keep 1 element minimum
    const size_t nv = (size_t)(num_vertices > 0 ? num_vertices : 1), nt = (size_t)(T > 0 ? T : 1);
    TRY_HIP(hipMalloc(&m->vertices, nv * 12));
    TRY_HIP(hipMalloc(&m->triangles, nt * 12));
    TRY_HIP(hipMalloc(&m->tri_verts, nt * 36));
    TRY_HIP(hipMalloc(&m->normals, nt * 12));
    TRY_HIP(hipMalloc(&m->shape, nt * 4));
    TRY_HIP(hipMalloc(&bad, 4));
    TRY_HIP(hipMemsetAsync(bad, 0, 4, s));
    if (num_vertices > 0)
        TRY_HIP(hipMemcpyAsync(m->vertices, vertices, (size_t)num_vertices * 12,
                               hipMemcpyDeviceToDevice, s));
    if (T > 0) {
        TRY_HIP(hipMemcpyAsync(m->triangles, triangles, (size_t)T * 12, hipMemcpyDeviceToDevice, s));
        if (mask) {
            TRY_HIP(hipMalloc(&m->mask, nt));
            TRY_HIP(hipMemcpyAsync(m->mask, mask, (size_t)T, hipMemcpyDeviceToDevice, s));
        }
        hipLaunchKernelGGL(mesh_prepare_kernel, dim3((unsigned)ceil_div(T, 256)), dim3(256), 0, s,
                           m->vertices, num_vertices, m->triangles, T, m->tri_verts, m->normals,
                           m->shape, bad);
        TRY_HIP(hipGetLastError());
    }
    int32_t bad_host = 0;
    TRY_HIP(hipMemcpyAsync(&bad_host, bad, 4, hipMemcpyDeviceToHost, s));
    TRY_HIP(hipStreamSynchronize(s));
    if (bad_host) return bail(fail(DRT_E_INVALID, "triangle index out of range [0, %lld)",
                                   (long long)num_vertices));
#undef TRY_HIP
    (void)hipFree(bad);
    (void)rc;
    *mesh_out = m;
    return DRT_OK;
}

int32_t drt_mesh_copy(drt_mesh_t m, float *tv_out, float *normals_out, void *stream) {
    DRT_REQUIRE(m, "mesh is null");
    const size_t T = (size_t)m->num_triangles;
    if (T == 0) return DRT_OK;
    if (tv_out)
        DRT_HIP(hipMemcpyAsync(tv_out, m->tri_verts, T * 36, hipMemcpyDeviceToDevice, as_stream(stream)));
    if (normals_out)
        DRT_HIP(hipMemcpyAsync(normals_out, m->normals, T * 12, hipMemcpyDeviceToDevice,
                               as_stream(stream)));
    return DRT_OK;
}

int64_t drt_mesh_num_triangles(drt_mesh_t m) { return m ? m->num_triangles : 0; }
const float *drt_mesh_triangle_vertices(drt_mesh_t m) { return m ? m->tri_verts : nullptr; }
const float *drt_mesh_normals(drt_mesh_t m) { return m ? m->normals : nullptr; }

}  // extern "C"
