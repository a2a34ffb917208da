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
    const BvhNode *__restrict__ nodes, int64_t T, const float *__restrict__ tv,
    const float *__restrict__ normals, const uint8_t *__restrict__ mask,
    const float *__restrict__ ro, const float *__restrict__ rd, int64_t ntx, int64_t num_rays,
    const float *__restrict__ rx, int64_t nrx, int order, float eps, TileTieB tt, float max_dist,
    int32_t *__restrict__ tri_out, float *__restrict__ vert_out, uint8_t *__restrict__ masks_out) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= ntx * num_rays) return;
    const int64_t it = g / num_rays, i = g - it * num_rays;
    V3 o = ld3(ro + 3 * g), d = ld3(rd + 3 * g);
    bool valid = true;
    for (int b = 0; b <= order; ++b) {
        int32_t tri = -1;
        float t_hit = kInf;
        if (T > 0) decode_first_hit(bvh_first_hit(nodes, T, tv, mask, o, d, eps, tt), tt, tri, t_hit);
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

extern "C" {

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
                       as_stream(stream), reinterpret_cast<const BvhNode *>(m->bvh_nodes), T, m->tri_verts,
                       m->normals, m->has_mask ? m->mask : nullptr, ro, rd, ntx, num_rays, rx, nrx,
                       (int)order, epsilon, tt, max_dist, triangles_out, vertices_out, masks_out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

}  // extern "C"
