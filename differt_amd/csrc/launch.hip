// launch.hip -- shooting-and-bouncing rays ("next" row f3 of SURVEY.md 8f): the launch_paths loop of
// the reference (geometry/_solvers.py:358-491) fused into one kernel.  lane = (transmitter, ray);
// per bounce: first hit over the mesh (LBVH, bvh.hpp -- same Moller-Trumbore, epsilon and tie-break
// as first_triangle_hit_by_ray), receiver-vicinity filter (filter_rays, :320-356), specular bounce
// (bounce_rays, :279-318).  Rays come from the caller (`launch_rays` is the reference's extension
// point, :262-277); SBRPathLauncher feeds a frustum-bounded Fibonacci lattice (:1202-1226).
#include "bvh.hpp"
#include "common.hpp"
#include "geom.hpp"
#include "mesh.hpp"

#pragma clang fp contract(off)

namespace drt {

__global__ __launch_bounds__(256) void launch_paths_kernel(
    const BvhNode *__restrict__ nodes, const uint32_t *__restrict__ leaf_ids, int64_t T,
    const float *__restrict__ tv, const float *__restrict__ normals, const uint8_t *__restrict__ mask,
    const float *__restrict__ ro, const float *__restrict__ rd, int64_t ntx, int64_t num_rays,
    const float *__restrict__ rx, int64_t nrx, int order, float eps, TileTieB tt, float max_dist,
    int32_t *__restrict__ tri_out, float *__restrict__ vert_out, uint8_t *__restrict__ masks_out) {
    DRT_BVH_LDS_STACK(lds_stack, 256);
    int32_t *col = &lds_stack[0][threadIdx.x];
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= ntx * num_rays) return;
    const int64_t it = g / num_rays, i = g - it * num_rays;
    V3 o = ld3(ro + 3 * g), d = ld3(rd + 3 * g);
    bool valid = true;
    for (int b = 0; b <= order; ++b) {
        int32_t tri = -1;
        float t_hit = kInf;
        if (T > 0) decode_first_hit(bvh_first_hit<256, true>(nodes, leaf_ids, T, tv, mask, o, d, eps, tt, col), tt, tri, t_hit);
        // filter_rays (_solvers.py:340-356): squared distance between the receiver and the ray, only
        // for receivers ahead of the origin and before the hit
        for (int64_t ir = 0; ir < nrx; ++ir) {
            const V3 v = ld3(rx + 3 * ir) - o;
            const V3 c = cross(d, v);
            const float dist2 = (c.x * c.x + c.y * c.y) + c.z * c.z;
            const float t_rx = dot(d, v);
            const bool near = (t_rx > 0.0f) && (t_rx < t_hit) && valid && (dist2 < max_dist);
            masks_out[((it * nrx + ir) * num_rays + i) * (order + 1) + b] = (uint8_t)near;
        }
        // bounce_rays (_solvers.py:300-318)
        const bool inside = is_finite(t_hit);
        valid = valid && inside;
        const float t = inside ? t_hit : 0.0f;
        o = V3{o.x + t * d.x, o.y + t * d.y, o.z + t * d.z};
        const V3 n = (T > 0) ? ld3(normals + 3 * (int64_t)((tri >= 0) ? tri : T - 1)) : V3{0, 0, 0};
        const float c2 = 2.0f * dot(d, n);
        d = V3{d.x - c2 * n.x, d.y - c2 * n.y, d.z - c2 * n.z};
        if (b < order) {
            tri_out[g * order + b] = tri;
            st3(vert_out + (g * order + b) * 3, o);
        }
    }
}

}  // namespace drt

using namespace drt;

// Reverse of the bounce chain (launch_paths is differentiable in the reference: every step of the scan,
// _solvers.py:385-444, is plain JAX code around Mesh.first_triangle_hit_by_ray, whose t carries the
// custom VJP of _mesh.py:258-344).  lane = (transmitter, ray): the forward chain is recomputed from the
// stored hit triangles (t = Moller-Trumbore on the hit face = the differentiable distance), then walked
// backwards:  o' = o + t d;  n = normal(tri);  d' = d - 2 <d, n> n.
// The receiver masks are booleans (no gradient).  grad_origins / grad_directions are written per ray,
// the mesh-vertex gradient is accumulated with atomics.
constexpr int kMaxBounces = DRT_MAX_ORDER;

__device__ __forceinline__ void normal_vjp_to_mesh(const float *__restrict__ mesh_vertices,
                                                   const int32_t *__restrict__ mesh_triangles, int64_t tri,
                                                   V3 n_bar, float *__restrict__ g_vertices) {
    const int32_t i0 = mesh_triangles[3 * tri], i1 = mesh_triangles[3 * tri + 1], i2 = mesh_triangles[3 * tri + 2];
    const V3 v0 = ld3(mesh_vertices + 3 * (int64_t)i0), v1 = ld3(mesh_vertices + 3 * (int64_t)i1),
             v2 = ld3(mesh_vertices + 3 * (int64_t)i2);
    const V3 ea = v1 - v0, eb = v2 - v1;  // _mesh.py:950-956: normalize(cross(v1 - v0, v2 - v1))
    const V3 c = cross(ea, eb);
    const float len = __builtin_sqrtf(dot(c, c));
    V3 cbar;
    if (len == 0.0f) {
        cbar = n_bar;  // normalize divides by 1 for zero-length vectors
    } else {
        const float inv = 1.0f / len;
        const float proj = dot(n_bar, c) * inv * inv * inv;
        cbar = n_bar * inv - c * proj;
    }
    const V3 ea_bar = cross(eb, cbar), eb_bar = cross(cbar, ea);
    const V3 g0 = V3{0, 0, 0} - ea_bar, g1 = ea_bar - eb_bar;
    atomicAdd(g_vertices + 3 * (int64_t)i0 + 0, g0.x);
    atomicAdd(g_vertices + 3 * (int64_t)i0 + 1, g0.y);
    atomicAdd(g_vertices + 3 * (int64_t)i0 + 2, g0.z);
    atomicAdd(g_vertices + 3 * (int64_t)i1 + 0, g1.x);
    atomicAdd(g_vertices + 3 * (int64_t)i1 + 1, g1.y);
    atomicAdd(g_vertices + 3 * (int64_t)i1 + 2, g1.z);
    atomicAdd(g_vertices + 3 * (int64_t)i2 + 0, eb_bar.x);
    atomicAdd(g_vertices + 3 * (int64_t)i2 + 1, eb_bar.y);
    atomicAdd(g_vertices + 3 * (int64_t)i2 + 2, eb_bar.z);
}

__global__ __launch_bounds__(256) void launch_paths_vjp_kernel(
    int64_t T, const float *__restrict__ mesh_vertices, const int32_t *__restrict__ mesh_triangles,
    const float *__restrict__ tv, const float *__restrict__ normals, const float *__restrict__ ro,
    const float *__restrict__ rd, int64_t n_rays_total, int order, float eps, const int32_t *__restrict__ tri_in,
    const float *__restrict__ vert_cot, float *__restrict__ g_ro, float *__restrict__ g_rd,
    float *__restrict__ g_vertices) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_rays_total) return;
    V3 os[kMaxBounces], ds[kMaxBounces];
    float ts[kMaxBounces];
    bool ins[kMaxBounces];
    V3 o = ld3(ro + 3 * g), d = ld3(rd + 3 * g);
#pragma unroll
    for (int b = 0; b < kMaxBounces; ++b) {
        if (b < order) {
            const int32_t tri = tri_in[g * order + b];
            float t = kInf;
            if (tri >= 0) (void)moller_trumbore(o, d, load_tri(tv + 9 * (int64_t)tri), eps, t);
            const bool inside = (tri >= 0) && is_finite(t);
            os[b] = o;
            ds[b] = d;
            ts[b] = inside ? t : 0.0f;
            ins[b] = inside;
            o = V3{o.x + ts[b] * d.x, o.y + ts[b] * d.y, o.z + ts[b] * d.z};
            const V3 n = ld3(normals + 3 * (int64_t)((tri >= 0) ? tri : T - 1));
            const float c2 = 2.0f * dot(d, n);
            d = V3{d.x - c2 * n.x, d.y - c2 * n.y, d.z - c2 * n.z};
        }
    }
    V3 obar{0, 0, 0}, dbar{0, 0, 0};
#pragma unroll
    for (int b = kMaxBounces - 1; b >= 0; --b) {
        if (b < order) {
            const int32_t tri = tri_in[g * order + b];
            const int64_t tn = (tri >= 0) ? tri : T - 1;
            const V3 n = ld3(normals + 3 * tn);
            obar = obar + ld3(vert_cot + (g * order + b) * 3);  // cotangent of the bounce point o_{b+1}
            // d' = d - 2 <d, n> n
            const float s = dot(ds[b], n), dn = dot(dbar, n);
            V3 dprev = V3{dbar.x - 2.0f * dn * n.x, dbar.y - 2.0f * dn * n.y, dbar.z - 2.0f * dn * n.z};
            const V3 nbar = V3{-2.0f * (ds[b].x * dn + s * dbar.x), -2.0f * (ds[b].y * dn + s * dbar.y),
                               -2.0f * (ds[b].z * dn + s * dbar.z)};
            if (g_vertices) normal_vjp_to_mesh(mesh_vertices, mesh_triangles, tn, nbar, g_vertices);
            // o' = o + t d
            V3 oprev = obar;
            dprev = dprev + obar * ts[b];
            if (ins[b]) {
                const float tbar = dot(obar, ds[b]);
                const int32_t i0 = mesh_triangles[3 * (int64_t)tri], i1 = mesh_triangles[3 * (int64_t)tri + 1],
                              i2 = mesh_triangles[3 * (int64_t)tri + 2];
                const MtHardBar gb = mt_t_vjp(os[b], ds[b], ld3(mesh_vertices + 3 * (int64_t)i0),
                                              ld3(mesh_vertices + 3 * (int64_t)i1),
                                              ld3(mesh_vertices + 3 * (int64_t)i2), tbar);
                oprev = oprev + gb.o;
                dprev = dprev + gb.d;
                if (g_vertices) {
                    atomicAdd(g_vertices + 3 * (int64_t)i0 + 0, gb.v0.x);
                    atomicAdd(g_vertices + 3 * (int64_t)i0 + 1, gb.v0.y);
                    atomicAdd(g_vertices + 3 * (int64_t)i0 + 2, gb.v0.z);
                    atomicAdd(g_vertices + 3 * (int64_t)i1 + 0, gb.v1.x);
                    atomicAdd(g_vertices + 3 * (int64_t)i1 + 1, gb.v1.y);
                    atomicAdd(g_vertices + 3 * (int64_t)i1 + 2, gb.v1.z);
                    atomicAdd(g_vertices + 3 * (int64_t)i2 + 0, gb.v2.x);
                    atomicAdd(g_vertices + 3 * (int64_t)i2 + 1, gb.v2.y);
                    atomicAdd(g_vertices + 3 * (int64_t)i2 + 2, gb.v2.z);
                }
            }
            obar = oprev;
            dbar = dprev;
        }
    }
    if (g_ro) st3(g_ro + 3 * g, obar);
    if (g_rd) st3(g_rd + 3 * g, dbar);
}

extern "C" {

int32_t drt_launch_paths_vjp(drt_mesh_t m, const float *ro, const float *rd, int64_t ntx, int64_t num_rays,
                             int32_t order, float epsilon, const int32_t *triangles_in,
                             const float *vertices_cotangent, float *grad_origins, float *grad_directions,
                             float *grad_vertices, void *stream) {
    DRT_REQUIRE(m, "mesh is null");
    DRT_REQUIRE(ntx >= 0 && num_rays >= 0 && order >= 0 && order <= DRT_MAX_ORDER, "bad size");
    const int64_t n = ntx * num_rays;
    if (n == 0) return DRT_OK;
    hipStream_t s = as_stream(stream);
    if (order == 0 || m->num_triangles == 0) {  // no bounce point depends on anything
        if (grad_origins) DRT_HIP(fill_bytes_async(grad_origins, 0, (size_t)n * 12, s));
        if (grad_directions) DRT_HIP(fill_bytes_async(grad_directions, 0, (size_t)n * 12, s));
        return DRT_OK;
    }
    DRT_REQUIRE(ro && rd && triangles_in && vertices_cotangent, "null pointer");
    hipLaunchKernelGGL(launch_paths_vjp_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, m->num_triangles,
                       m->vertices, m->triangles, m->tri_verts, m->normals, ro, rd, n, (int)order, epsilon,
                       triangles_in, vertices_cotangent, grad_origins, grad_directions, grad_vertices);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_launch_paths(drt_mesh_t m, const float *ro, const float *rd, int64_t ntx, int64_t num_rays,
                         const float *rx, int64_t nrx, int32_t order, float epsilon, int64_t batch_size,
                         float max_dist, int32_t *triangles_out, float *vertices_out,
                         uint8_t *masks_out, void *stream) {
    DRT_REQUIRE(m, "mesh is null");
    DRT_REQUIRE(ntx >= 0 && num_rays >= 0 && nrx >= 0 && order >= 0, "negative size");
    if (ntx * num_rays == 0) return DRT_OK;
    DRT_REQUIRE(ro && rd && (nrx == 0 || (rx && masks_out)), "null pointer");
    DRT_REQUIRE(order == 0 || (triangles_out && vertices_out), "null output");
    const int64_t T = m->num_triangles;
    if (T > 0) {
        int32_t rc = drt_mesh_build_bvh(m, stream);
        if (rc != DRT_OK) return rc;
    }
    const TileTieB tt = make_tie_b(T > 0 ? T : 1, batch_size);
    hipLaunchKernelGGL(launch_paths_kernel, dim3((unsigned)ceil_div(ntx * num_rays, 256)), dim3(256), 0,
                       as_stream(stream), reinterpret_cast<const BvhNode *>(m->bvh_nodes), m->bvh_leaf_ids, T, m->tri_verts,
                       m->normals, m->has_mask ? m->mask : nullptr, ro, rd, ntx, num_rays, rx, nrx,
                       (int)order, epsilon, tt, max_dist, triangles_out, vertices_out, masks_out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

}  // extern "C"
