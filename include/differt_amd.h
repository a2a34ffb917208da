/*
 * differt_amd.h -- C ABI of the MI355X-native DiffeRT ray-tracing core (libdiffert_amd.so).
 *
 * This is the drop-in boundary for the reference's hot path.  In the reference the same
 * boundary is `wp.jax_callable(func, output_dims=...)` (flat contiguous device buffers in,
 * flat buffers out, static shapes, no gradients: differt/src/differt/geometry/_mesh.py:266-276,
 * 3082-3092, 3239-3250) for the ray queries, XLA-jitted JAX for the image method
 * (geometry/_solver_image_method.py:206-363, geometry/_solvers.py:499-770) and a PyO3 module for
 * the candidate generator (differt-core/src/geometry/graph.rs:1194-1205,
 * differt-core/python/differt_core/_differt_core/geometry/graph.pyi:1-103).
 *
 * Conventions
 *   - every function returns an int32 status: 0 = ok, <0 = error (see DRT_E_*); the message is
 *     available from drt_last_error() (thread-local).  Nothing throws, nothing aborts.
 *   - all array pointers are DEVICE pointers (HBM) unless the name ends in `_host`.
 *   - float = IEEE binary32, row-major, densely packed.  Indices are int32 (reference:
 *     `dtype=int` under x64-off JAX), candidate ranks / flat path keys are int64.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous
 *     on that stream unless documented otherwise; the caller owns every buffer.
 *   - the library keeps no hidden global state: device state lives in explicit handles.
 */
#ifndef DIFFERT_AMD_H
#define DIFFERT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRT_ABI_VERSION 7

enum {
    DRT_OK = 0,
    DRT_E_INVALID = -1,    /* invalid argument */
    DRT_E_HIP = -2,        /* HIP runtime error */
    DRT_E_NO_DEVICE = -3,  /* no usable GPU */
    DRT_E_CAPACITY = -4,   /* an output/workspace capacity given by the caller is too small */
    DRT_E_OVERFLOW = -5,   /* a count does not fit in 64 bits */
    DRT_E_UNSUPPORTED = -6
};

#define DRT_MAX_ORDER 8

int32_t drt_abi_version(void);
const char *drt_last_error(void);
/* 0 if a gfx950 device is visible and usable, DRT_E_NO_DEVICE otherwise (never aborts). */
int32_t drt_device_check(void);

/* ---------------------------------------------------------------------------------------------
 * (a1) ray_intersect_triangle -- reference: geometry/_utils.py:1157-1322 (hard mode).
 * dense : rays [R,3] x triangles [T,3,3] -> t [R,T] f32, hit [R,T] u8   (the o[...,None,:] form)
 * paired: element i of each input -> t[i], hit[i]                        (fully broadcast form)
 * ------------------------------------------------------------------------------------------- */
int32_t drt_ray_intersect_triangle_dense(const float *ray_origins, const float *ray_directions,
                                         int64_t num_rays, const float *triangle_vertices,
                                         int64_t num_triangles, float epsilon, float *t_out,
                                         uint8_t *hit_out, void *stream);
int32_t drt_ray_intersect_triangle_paired(const float *ray_origins, const float *ray_directions,
                                          const float *triangle_vertices, int64_t n, float epsilon,
                                          float *t_out, uint8_t *hit_out, void *stream);
/* batched: `num_problems` independent dense problems of num_rays x num_triangles each in ONE launch --
 * the outer form under leading batch axes (`vmap` of the call above; broadcasting rules of
 * docs/source/batch_axes.md:67-80).  Problem b reads rays at ray_origins + b * ray_batch_stride
 * (stride in floats: 3 * num_rays, or 0 = every problem shares the rays) and triangles at
 * triangle_vertices + b * tv_batch_stride (9 * num_triangles, or 0 = shared); outputs
 * t [B, R, T] f32 and hit [B, R, T] u8.  Bit-identical to B separate dense calls; a single 256-ray
 * problem (BASELINE configs[1]) is latency-bound, a batch of them runs at the dense kernel's bandwidth. */
int32_t drt_ray_intersect_triangle_dense_batched(const float *ray_origins, const float *ray_directions,
                                                 int64_t ray_batch_stride, int64_t num_rays,
                                                 const float *triangle_vertices, int64_t tv_batch_stride,
                                                 int64_t num_triangles, int64_t num_problems, float epsilon,
                                                 float *t_out, uint8_t *hit_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * (a2) ray_intersect_any_triangle -- reference: geometry/_utils.py:1353-1537 (hard mode), and the
 * mesh-bound form geometry/_mesh.py:3018-3094.
 *   out[r] = any_j [ (t_rj < 1 - hit_tol) & hit_rj & active_j ]
 * tv_ray_stride: 0 = one triangle set [T,3,3] shared by all rays; 9*T = one set per ray.
 * active: NULL or u8; active_ray_stride: 0 = shared [T], T = per ray [R,T].
 * ------------------------------------------------------------------------------------------- */
int32_t drt_ray_intersect_any_triangle(const float *ray_origins, const float *ray_directions,
                                       int64_t num_rays, const float *triangle_vertices,
                                       int64_t num_triangles, int64_t tv_ray_stride,
                                       const uint8_t *active, int64_t active_ray_stride,
                                       float epsilon, float hit_tol, uint8_t *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * (a4) first_triangle_hit_by_ray -- reference: geometry/_utils.py:1775-1960, incl. its tile
 * tie-break (lowest index inside a `batch_size` tile, the LATER tile wins ties, remainder tile
 * last: _utils.py:1865-1867, 1886, 1939-1955).  batch_size <= 0 means "None" (one tile).
 * Miss: index -1, t = +inf.  workspace: num_rays * 8 bytes (device), see the _workspace_size query.
 * ------------------------------------------------------------------------------------------- */
size_t drt_first_triangle_hit_by_ray_workspace_size(int64_t num_rays);
int32_t drt_first_triangle_hit_by_ray(const float *ray_origins, const float *ray_directions,
                                      int64_t num_rays, const float *triangle_vertices,
                                      int64_t num_triangles, int64_t tv_ray_stride,
                                      const uint8_t *active, int64_t active_ray_stride,
                                      float epsilon, int64_t batch_size, int32_t *index_out,
                                      float *t_out, void *workspace, size_t workspace_bytes,
                                      void *stream);

/* Triangle-block sharding (BASELINE configs[4]): every GPU holds a block
 * [index_offset, index_offset + T_block) of the mesh and computes, for the SAME rays, packed 64-bit
 * keys (ordered(t) << 32) | tie with GLOBAL tile ids -- so the MIN over blocks (one RCCL all-reduce of
 * 8 B per ray) is the single-GPU first hit including the reference tie-break, whatever the
 * partition.  keys are in/out (init != 0 resets them to "miss"); finalize decodes them. */
int32_t drt_first_hit_keys(const float *ray_origins, const float *ray_directions, int64_t num_rays,
                           const float *triangle_vertices_block, int64_t block_triangles,
                           int64_t index_offset, int64_t total_triangles,
                           const uint8_t *active_block, float epsilon, int64_t batch_size,
                           uint64_t *keys, int32_t init, void *stream);
int32_t drt_first_hit_finalize(const uint64_t *keys, int64_t num_rays, int64_t total_triangles,
                               int64_t batch_size, int32_t *index_out, float *t_out, void *stream);

/* Reverse mode of the hard-mode `t` of ray_intersect_triangle (reference: t = f * <q, e2> is plain
 * JAX code, geometry/_utils.py:1263-1316, differentiated by jax; `hit` carries no gradient).
 *   dense != 0: rays [R,3] x triangles [T,3,3], t_cotangent [R,T]; grad_origins / grad_directions
 *               [R,3] and grad_triangle_vertices [T,3,3] are ACCUMULATED with atomic adds
 *               (zero-initialise them);
 *   dense == 0: paired, R == T elements, t_cotangent [R]; every gradient is WRITTEN.
 * Each gradient pointer may be NULL.  A zero cotangent contributes exactly zero (no 0 * inf NaN). */
int32_t drt_ray_intersect_triangle_vjp(const float *ray_origins, const float *ray_directions,
                                       int64_t num_rays, const float *triangle_vertices,
                                       int64_t num_triangles, int32_t dense, const float *t_cotangent,
                                       float *grad_origins, float *grad_directions,
                                       float *grad_triangle_vertices, void *stream);

/* (a5) backward of the first-hit distance -- reference: geometry/_mesh.py:226-344: the cotangent of
 * t flows through Moller-Trumbore on the hit face only.  Gradients are ACCUMULATED (atomic adds)
 * into grad_vertices [Nv,3] (may be NULL); grad_origins / grad_directions [R,3] are written. */
int32_t drt_first_hit_vjp(const float *vertices, const int32_t *triangles,
                          const float *ray_origins, const float *ray_directions,
                          const int32_t *hit_index, const float *t_cotangent, int64_t num_rays,
                          float *grad_vertices, float *grad_origins, float *grad_directions,
                          void *stream);

/* ---------------------------------------------------------------------------------------------
 * (a6-a10) image method -- reference: geometry/_solver_image_method.py:11-454.
 * Flat batches: from/to [B,3], mirrors [B,k,3] -> paths [B,k,3] (end points excluded).
 * ------------------------------------------------------------------------------------------- */
/* normalize (geometry/_utils.py:29-72): out = v / where(|v| == 0, 1, |v|); lengths_out [B] or NULL */
int32_t drt_normalize(const float *vectors, int64_t batch, float *out, float *lengths_out,
                      void *stream);
/* element-wise helpers (_solver_image_method.py:11-79 and :82-135), flat batches [B,3] */
int32_t drt_image_of_vertex(const float *vertices, const float *mirror_vertices,
                            const float *mirror_normals, int64_t batch, float *out, void *stream);
int32_t drt_intersection_of_ray_with_plane(const float *ray_origins, const float *ray_directions,
                                           const float *plane_vertices, const float *plane_normals,
                                           int64_t batch, float *out, void *stream);
int32_t drt_image_method(const float *from_vertices, const float *to_vertices,
                         const float *mirror_vertices, const float *mirror_normals, int64_t batch,
                         int32_t num_mirrors, float *paths_out, void *stream);
/* VJP of drt_image_method w.r.t. all four inputs (any grad pointer may be NULL). */
int32_t drt_image_method_vjp(const float *from_vertices, const float *to_vertices,
                             const float *mirror_vertices, const float *mirror_normals,
                             const float *paths_cotangent, int64_t batch, int32_t num_mirrors,
                             float *grad_from, float *grad_to, float *grad_mirror_vertices,
                             float *grad_mirror_normals, void *stream);
/* The same operators with BROADCAST inputs (reference signature `(3),(3),(n,3),(n,3)->(n,3)` over `*#batch`,
 * _solver_image_method.py:360-363; its benchmark passes from / to of shape [3] against [10000, 8, 3] mirrors,
 * tests/benchmarks/fixtures.py:19-40): every *_stride is the number of floats between consecutive batch elements,
 * either 0 (one row shared by the whole batch, read in place) or the dense row size (3, resp. 3 * num_mirrors).
 * Outputs and the cotangent are dense; the VJP writes PER-ELEMENT gradients (the caller sums those of shared rows). */
int32_t drt_image_method_strided(const float *from_vertices, int64_t from_stride, const float *to_vertices,
                                 int64_t to_stride, const float *mirror_vertices, int64_t mirror_vertices_stride,
                                 const float *mirror_normals, int64_t mirror_normals_stride, int64_t batch,
                                 int32_t num_mirrors, float *paths_out, void *stream);
int32_t drt_image_method_vjp_strided(const float *from_vertices, int64_t from_stride, const float *to_vertices,
                                     int64_t to_stride, const float *mirror_vertices,
                                     int64_t mirror_vertices_stride, const float *mirror_normals,
                                     int64_t mirror_normals_stride, const float *paths_cotangent, int64_t batch,
                                     int32_t num_mirrors, float *grad_from, float *grad_to,
                                     float *grad_mirror_vertices, float *grad_mirror_normals, void *stream);
int32_t drt_consecutive_vertices_same_side(const float *vertices, const float *mirror_vertices,
                                           const float *mirror_normals, int64_t batch,
                                           int32_t num_mirrors, uint8_t *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * (a16) mesh handle: device-resident copy of what the path needs from `Mesh`
 * (geometry/_mesh.py:624-688 fields; triangle_vertices :899-905; normals :950-956).
 * Replaces the reference's hidden `_WARP_MESHES_CACHE` keyed on Python id() (_mesh.py:55, 170-174).
 * ------------------------------------------------------------------------------------------- */
typedef struct drt_mesh *drt_mesh_t;
int32_t drt_mesh_create(const float *vertices, int64_t num_vertices, const int32_t *triangles,
                        int64_t num_triangles, const uint8_t *mask /* NULL = all active */,
                        int32_t assume_quads, void *stream, drt_mesh_t *mesh_out);
int32_t drt_mesh_destroy(drt_mesh_t mesh);
int64_t drt_mesh_num_triangles(drt_mesh_t mesh);
/* borrowed device pointers, valid until drt_mesh_destroy */
const float *drt_mesh_triangle_vertices(drt_mesh_t mesh); /* [T,3,3] */
const float *drt_mesh_normals(drt_mesh_t mesh);           /* [T,3]   */
/* copies into caller-owned device buffers ([T,3,3] and [T,3]; either may be NULL) */
int32_t drt_mesh_copy(drt_mesh_t mesh, float *triangle_vertices_out, float *normals_out,
                      void *stream);

/* ---------------------------------------------------------------------------------------------
 * (f1, "next" row) BVH-accelerated mesh queries -- the counterpart of the reference's Warp-backed
 * Mesh.ray_intersect_any_triangle / Mesh.first_triangle_hit_by_ray (geometry/_mesh.py:142-223,
 * 3018-3162).  Own LBVH (Morton sort + Karras radix tree), built lazily and cached in the handle.
 * Same predicate, epsilon, mask and tie-break as the brute-force entry points above.
 * ------------------------------------------------------------------------------------------- */
int32_t drt_mesh_build_bvh(drt_mesh_t mesh, void *stream);
int32_t drt_mesh_has_bvh(drt_mesh_t mesh);
int32_t drt_mesh_ray_intersect_any_triangle(drt_mesh_t mesh, const float *ray_origins,
                                            const float *ray_directions, int64_t num_rays,
                                            float epsilon, float hit_tol, uint8_t *out, void *stream);
/* BVH flavour of drt_triangles_visible_from_vertex (geometry/_mesh.py:3164-3253) */
int32_t drt_mesh_triangles_visible_from_vertex(drt_mesh_t mesh, const float *vertices,
                                               int64_t num_vertices, int64_t num_rays, float epsilon,
                                               uint8_t *visible_out, float *frustum_workspace,
                                               void *stream);
/* Extension: triangle-driven complement of the lattice estimate -- a face is also visible when the segment
 * from the viewing vertex to one of 7 interior sample points of the face is not blocked by another active
 * triangle.  ORs into visible_inout [B,T] (run it after drt_mesh_triangles_visible_from_vertex). */
int32_t drt_mesh_triangles_visible_samples(drt_mesh_t mesh, const float *viewing_vertices,
                                           int64_t num_vertices, float epsilon, uint8_t *visible_inout,
                                           void *stream);
int32_t drt_mesh_first_triangle_hit_by_ray(drt_mesh_t mesh, const float *ray_origins,
                                           const float *ray_directions, int64_t num_rays,
                                           float epsilon, int64_t batch_size, int32_t *index_out,
                                           float *t_out, void *stream);

/* Opt-in Warp query semantics of the reference's mesh-bound operators (third-party warp-lang BVH
 * queries behind geometry/_mesh.py:142-223; the normative predicate of this library stays the
 * pure-JAX operator, which the reference asserts equal: tests/geometry/test_mesh.py:1984-2002).
 *   mode 0 (any-hit, _mesh.py:3065-3070): normalise the direction, move the origin hit_tol * |d| along
 *     it, shorten the segment to |d| * (1 - 2 hit_tol); param = hit_tol.  Outputs a segment
 *     (origin', direction' = unit direction * max_t): query it with hit_tol = 0.
 *   mode 1 (first-hit, _mesh.py:195-199): origin' = origin + param * direction (param = 1e-5).
 * drt_warp_first_hit_finish adds the nudge back to t on hits (res.t + epsilon). */
int32_t drt_warp_ray_prep(const float *ray_origins, const float *ray_directions, int64_t num_rays,
                          int32_t mode, float param, float *origins_out, float *directions_out,
                          void *stream);
int32_t drt_warp_first_hit_finish(const int32_t *hit_index, float *t_inout, int64_t num_rays, float nudge,
                                  void *stream);

/* ---------------------------------------------------------------------------------------------
 * (f2, "next" row) visibility by ray launching -- reference: geometry/_utils.py:369-490
 * (fibonacci_lattice), :639-927 (viewing_frustum), :1540-1772 (triangles_visible_from_vertex) and the
 * Mesh method geometry/_mesh.py:3164-3253.  World vertices of the frustum are the triangle vertices
 * plus the triangle centres (_utils.py:1669-1673).  frustum layout: [B,2,3] =
 * [[r_min, polar_min, azim_min], [r_max, polar_max, azim_max]].
 * ------------------------------------------------------------------------------------------- */
int32_t drt_viewing_frustum(const float *viewing_vertices, int64_t num_vertices,
                            const float *triangle_vertices, int64_t num_triangles,
                            const uint8_t *active_triangles, float *frustum_out, void *stream);
/* same over an explicit list of world points [N,3] (no centres, no mask) */
int32_t drt_viewing_frustum_points(const float *viewing_vertices, int64_t num_vertices,
                                   const float *points, int64_t num_points, float *frustum_out,
                                   void *stream);
/* General form (geometry/_utils.py:639-927): per-viewer point sets (points_viewer_stride = 3*N) or one
 * shared set (0); optional per-point mask with its own stride (N or 0); reduce != 0 -> ONE frustum
 * [2,3] over all viewers and points (min / max taken before the azimuth / polar selection, axis=None
 * in the reference), workspace = 8*B floats; else frustum_out [B,2,3]. */
int32_t drt_viewing_frustum_general(const float *viewing_vertices, int64_t num_viewers, const float *points,
                                    int64_t num_points, int64_t points_viewer_stride, const uint8_t *active,
                                    int64_t active_viewer_stride, int32_t reduce, float *workspace,
                                    float *frustum_out, void *stream);
/* geometry/_utils.py:930-993: xyz [batch,3] -> (r, polar, azimuth); rpa [batch,width] (width 2 = unit
 * radius, 3 = with radius) -> xyz. */
int32_t drt_cartesian_to_spherical(const float *xyz, int64_t batch, float *rpa_out, void *stream);
int32_t drt_spherical_to_cartesian(const float *rpa, int64_t batch, int32_t width, float *xyz_out,
                                   void *stream);
/* n lattice directions [n,3]; frustum = device [2,3] or NULL (full sphere) */
int32_t drt_fibonacci_lattice(int64_t n, const float *frustum, float *out, void *stream);
/* visible_out u8 [B,T]; frustum_workspace: device float [B,6] */
int32_t drt_triangles_visible_from_vertex(const float *vertices, int64_t num_vertices,
                                          const float *triangle_vertices, int64_t num_triangles,
                                          const uint8_t *active_triangles, int64_t num_rays,
                                          float epsilon, uint8_t *visible_out,
                                          float *frustum_workspace, void *stream);

/* ---------------------------------------------------------------------------------------------
 * (f3, "next" row) shooting-and-bouncing rays -- reference: AbstractPathLauncher.launch_paths
 * geometry/_solvers.py:358-491 (first hit + filter_rays :320-356 + bounce_rays :279-318 per bounce).
 * rays [Ntx, num_rays, 3] are given (launch_rays is the reference's extension point).  Outputs, NOT
 * broadcast over receivers: triangles [Ntx, num_rays, order] i32 (-1 = left the scene),
 * vertices [Ntx, num_rays, order, 3] (bounce points), masks [Ntx, Nrx, num_rays, order+1] u8.
 * ------------------------------------------------------------------------------------------- */
int32_t drt_launch_paths(drt_mesh_t mesh, const float *ray_origins, const float *ray_directions,
                         int64_t num_tx, int64_t num_rays, const float *rx, int64_t num_rx,
                         int32_t order, float epsilon, int64_t batch_size, float max_dist,
                         int32_t *triangles_out, float *vertices_out, uint8_t *masks_out, void *stream);

/* Reverse mode of the bounce points of drt_launch_paths w.r.t. ray origins, ray directions and mesh
 * vertices (the reference's launch_paths is differentiable: geometry/_solvers.py:385-444 around the
 * custom VJP of Mesh.first_triangle_hit_by_ray, _mesh.py:258-344).  triangles_in = triangles_out of
 * the forward call; vertices_cotangent [Ntx, num_rays, order, 3].  grad_origins / grad_directions
 * [Ntx, num_rays, 3] are WRITTEN, grad_vertices [Nv,3] is ACCUMULATED (atomic adds); each may be NULL.
 * The receiver masks are booleans and carry no gradient. */
int32_t drt_launch_paths_vjp(drt_mesh_t mesh, const float *ray_origins, const float *ray_directions,
                             int64_t num_tx, int64_t num_rays, int32_t order, float epsilon,
                             const int32_t *triangles_in, const float *vertices_cotangent,
                             float *grad_origins, float *grad_directions, float *grad_vertices,
                             void *stream);

/* ---------------------------------------------------------------------------------------------
 * (a12-a14) path-candidate enumeration -- reference: differt-core/src/geometry/graph.rs
 * (CompleteGraph :127-277, iterator :286-491, count :314-377) and geometry/_utils.py:1047-1132.
 * Host-side (no GPU needed): exact count and lexicographic unranking of "no two equal neighbours"
 * tuples; rows are produced for ranks [rank_lo, rank_hi).
 * ------------------------------------------------------------------------------------------- */
/* count of paths of `depth` nodes from `from_` to `to` in the complete graph on num_nodes nodes
 * (from_/to may be outside the graph, i.e. >= num_nodes).  *overflow_out = 1 and *count_out =
 * UINT64_MAX when the count does not fit (graph.rs:368-375). */
int32_t drt_complete_graph_count(uint64_t num_nodes, uint64_t from_, uint64_t to, uint64_t depth,
                                 uint64_t *count_out, int32_t *overflow_out);
/* the true number of paths (the closed form above reports an overflow in a few degenerate cases
 * where the real count is small, e.g. num_nodes = 1); saturates at UINT64_MAX with *exceeds_out = 1. */
int32_t drt_complete_graph_count_exact(uint64_t num_nodes, uint64_t from_, uint64_t to,
                                       uint64_t depth, uint64_t *count_out, int32_t *exceeds_out);
/* rows [rank_lo, rank_hi) into out_host (uint64 [rank_hi-rank_lo, depth or depth-2]). */
int32_t drt_complete_graph_fill_host(uint64_t num_nodes, uint64_t from_, uint64_t to,
                                     uint64_t depth, int32_t include_from_and_to,
                                     uint64_t rank_lo, uint64_t rank_hi, uint64_t *out_host);
/* GPU-resident table for the tracer: candidates of `order` interactions among num_nodes primitives
 * (from/to outside the graph), ranks [rank_lo, rank_hi), optional node_map [num_nodes] (device)
 * from compact node id to primitive id, ids multiplied by id_scale (2 for assume_quads).
 * out: device int32 [rank_hi-rank_lo, order]. */
int32_t drt_candidates_fill(int64_t num_nodes, int32_t order, int64_t rank_lo, int64_t rank_hi,
                            const int32_t *node_map, int32_t id_scale, int32_t *out, void *stream);

/* DiGraph (graph.rs:594-1120): adjacency lists + DFS path iterator, host side. */
typedef struct drt_digraph *drt_digraph_t;
int32_t drt_digraph_from_complete_graph(uint64_t num_nodes, drt_digraph_t *out);
int32_t drt_digraph_from_adjacency_matrix(const uint8_t *matrix_host, uint64_t num_nodes,
                                          drt_digraph_t *out);
int32_t drt_digraph_destroy(drt_digraph_t g);
uint64_t drt_digraph_num_nodes(drt_digraph_t g);
int32_t drt_digraph_insert_from_and_to_nodes(drt_digraph_t g, int32_t direct_path,
                                             const uint8_t *from_adjacency_host,
                                             const uint8_t *to_adjacency_host, uint64_t *from_out,
                                             uint64_t *to_out);
int32_t drt_digraph_filter_by_mask(drt_digraph_t g, const uint8_t *mask_host, uint64_t mask_len,
                                   int32_t fast_mode);
int32_t drt_digraph_disconnect_nodes(drt_digraph_t g, const uint64_t *nodes_host, uint64_t n,
                                     int32_t fast_mode);
typedef struct drt_digraph_iter *drt_digraph_iter_t;
int32_t drt_digraph_iter_create(drt_digraph_t g, uint64_t from_, uint64_t to, uint64_t depth,
                                int32_t include_from_and_to, drt_digraph_iter_t *out);
int32_t drt_digraph_iter_destroy(drt_digraph_iter_t it);
/* up to max_rows rows into out_host (uint64 [max_rows, path_depth]); *rows_out = rows written
 * (0 = exhausted). */
int32_t drt_digraph_iter_next_chunk(drt_digraph_iter_t it, uint64_t max_rows, uint64_t *out_host,
                                    uint64_t *rows_out);

/* ---------------------------------------------------------------------------------------------
 * (a11) fused trace -- reference: geometry/_solvers.py:499-770 (_trace_path_candidates), hard mask.
 * Candidate source: either an explicit table (device int32 [C,order], negative ids = padding rows,
 * invalid) or a rank interval of the complete graph, unranked on the GPU (no table in HBM).
 * ------------------------------------------------------------------------------------------- */
/* Per-stage counters and HIP-event timers of one drt_trace_paths_compact call (SURVEY.md section 5,
 * "metrics" / "tracing" rows; the dense entry points count `valid` as survivors minus cleared survivors).  Filled when drt_trace_params.stats is non-NULL; costs two extra
 * stream synchronisations, so leave it NULL on the hot path. */
typedef struct drt_trace_stats {
    int64_t candidates;     /* (tx, rx, candidate) rows evaluated by the filter stage */
    int64_t survivors;      /* rows that passed the geometric checks (filter-stage output) */
    int64_t valid;          /* rows that also passed the occlusion stage = valid paths */
    float filter_ms;        /* filter kernel, HIP events on the call's stream */
    float occlusion_ms;     /* occlusion kernel */
    float sort_emit_ms;     /* radix sort of the valid keys + emit kernel */
    int32_t reserved;
} drt_trace_stats;

typedef struct drt_trace_params {
    float epsilon;          /* MT epsilon; reference default 10*eps (_utils.py:1257-1259) */
    float hit_tol;          /* occlusion tolerance; default 100*eps (_utils.py:1418-1420) */
    float min_len;          /* squared-length threshold; default 10*eps (_solvers.py:514-516) */
    int32_t flags;          /* DRT_TRACE_* bits */
    drt_trace_stats *stats; /* host pointer or NULL (drt_trace_paths_compact, drt_trace_paths_dense[_ex]) */
} drt_trace_params;
#define DRT_TRACE_USE_BVH 1 /* occlusion stage walks the mesh LBVH instead of testing every triangle */
#define DRT_TRACE_SKIP_OCCLUSION 2 /* return the candidates that pass the GEOMETRIC checks; the caller tests
                                      occlusion itself (triangle-block sharding: every rank tests the segments
                                      against its block with drt_ray_intersect_any_triangle, then one MAX
                                      all-reduce of u8 per path) */
/* bits of counts_dev[2] written by drt_trace_paths_compact_async */
#define DRT_TRACE_OVERFLOW_SURVIVORS 1 /* more candidates passed the geometric checks than max_survivors */
#define DRT_TRACE_OVERFLOW_PATHS 2     /* more valid paths than max_paths */

typedef struct drt_candidates {
    const int32_t *table;   /* device [num_candidates, order] or NULL */
    int64_t num_candidates; /* rows of `table`, or rank_hi - rank_lo */
    int64_t rank_lo;        /* used when table == NULL */
    int64_t num_nodes;      /* primitives in the complete graph (table == NULL) */
    const int32_t *node_map;/* device [num_nodes] compact node -> primitive, or NULL */
    int32_t order;          /* interactions per path (0..DRT_MAX_ORDER) */
    int32_t reserved;
    /* Pruned PRODUCT space (HybridPathTracer, reference _solvers.py:996-1056: first / last interaction
     * restricted to the primitives visible from the transmitters / receivers), table == NULL and
     * order >= 2, selected by a non-NULL first_map or last_map: position 0 draws from
     * first_map[0..num_first), position order-1 from last_map[0..num_last), the others from
     * node_map[0..num_nodes) (or the identity).  Ranks enumerate F x N^(order-2) x L in lexicographic
     * order -- the DFS order of the reference's DiGraph; tuples with two equal neighbours are not
     * paths of that graph and are skipped (they behave as padding rows). */
    const int32_t *first_map;
    int64_t num_first;
    const int32_t *last_map;
    int64_t num_last;
    /* RAGGED per-pair product spaces -- ONE launch for all (tx, rx) pairs, pair (i, j) having its own
     * first / last sets (drt_trace_paths_compact / _vjp only, order >= 2), selected by pair_offsets != NULL:
     *   first_map / last_map       CSR id arrays (device), rows delimited by
     *   first_offsets[num_tx + 1] / last_offsets[num_rx + 1]   (device int64)
     *   pair_offsets[num_tx * num_rx + 1]   device int64 prefix sums of the pair space sizes
     *                                       F_i * num_nodes^(order-2) * L_j
     *   num_candidates = pair_offsets[num_tx * num_rx] (total rows); rank_lo must be 0;
     *   reserved bit 0 = every pair space is < 2^32 rows (32-bit unranking); bit 1 (order >= 3) = large
 *   pair spaces: stage A runs one lane per PREFIX (first order-1 interactions) with inner loops over
 *   the receivers and their last interactions; num_first must then hold max_i F_i.
     * Keys returned by the compact tracer are then GLOBAL ragged row indices (pair-major). */
    const int64_t *pair_offsets;
    const int64_t *first_offsets;
    const int64_t *last_offsets;
    /* PER-PAIR TABLE (drt_trace_paths_compact / _async / _vjp): table != NULL together with pair_offsets
     * != NULL -- rows [pair_offsets[p], pair_offsets[p+1]) of `table` are the candidates of pair
     * p = tx * num_rx + rx (triangle ids, already even for quads); first / last maps unused.  Keys are
     * global table rows.  This is how drt_trace_paths_beam traces its candidate rows.
     * PACKED KEYS (drt_trace_paths_vjp only): table == NULL and reserved & DRT_CAND_PACKED_KEYS -- a key is
     * (tx * num_rx + rx) * num_nodes^order + sum_j m_j * num_nodes^(order-1-j), m_j = primitive ids: the keys
     * drt_trace_paths_beam returns are self-describing, no table has to outlive the forward call. */
} drt_candidates;

/* Dense reference layout for every (tx, rx, candidate):
 *   vertices [Ntx,Nrx,C,order+2,3] f32 (zeroed where not finite, _solvers.py:696-699)
 *   objects  [Ntx,Nrx,C,order+2] i32  (_solvers.py:723-748)
 *   mask     [Ntx,Nrx,C] u8           (_solvers.py:715-717)
 * workspace: drt_trace_dense_workspace_size bytes.  When the call's work has completed on the stream, its first two
 * 64-bit words hold DEVICE-side counters -- [0] rows that passed the geometric checks, [1] those of them the occlusion
 * stage cleared -- so the number of valid paths is [0] - [1] without a reduction over the mask (the reference's
 * TracedPaths.num_valid_paths, _paths.py:264-272, is what its own harness reads after every call:
 * tests/benchmarks/test_rt.py:151-196). */
size_t drt_trace_dense_workspace_size(int64_t num_tx, int64_t num_rx, int64_t num_candidates);
/* FIXED-CAPACITY survivor queue (round 5; what an XLA binding allocates as a result buffer on every call): the workspace
 * above is the worst case, 8 B per (tx, rx, candidate) row -- 512 MB for a 2^20-row chunk x 64 pairs -- although 1e-5 of
 * the rows survive the geometric checks.  drt_trace_paths_dense_capped takes `max_survivors` instead (< 0: the worst
 * case) and a workspace of drt_trace_dense_capped_workspace_size(max_survivors) = 64 + 8 max_survivors bytes.  Same
 * outputs.  The counter block gains a third 64-bit word: [2] status, DRT_TRACE_OVERFLOW_SURVIVORS when more rows passed
 * the geometric checks than the queue holds -- the occlusion stage then has not seen the rows beyond the capacity, and
 * their mask entries are CLEARED (a row that was never occlusion-tested is not reported as a valid path; a caller that
 * does not read the status word gets fewer paths, never unverified ones): re-run with a larger capacity.  Counter [0] is
 * the true number of survivors either way; valid paths = min([0], max_survivors) - [1]. */
size_t drt_trace_dense_capped_workspace_size(int64_t max_survivors);
int32_t drt_trace_paths_dense(drt_mesh_t mesh, const drt_trace_params *params, const float *tx,
                              int64_t num_tx, const float *rx, int64_t num_rx,
                              const drt_candidates *cands, float *vertices, int32_t *objects,
                              uint8_t *mask, void *workspace, size_t workspace_bytes, void *stream);
/* The same call with the fourth field of the reference's TracedPaths (_solvers.py:751-762):
 *   interaction_types_out [Ntx,Nrx,C,order] i32 = interaction_types_in [C,order] broadcast over (tx, rx), or zeros
 *   (specular reflection) when interaction_types_in is NULL; interaction_types_out may be NULL (not written).
 * Every element of every output is written (callers need not clear them).  The operator is HBM-write bound
 * (81 B per (tx, rx, candidate) at order 2): rows are staged per wave and stored as whole 128-byte lines with
 * 16-byte nontemporal stores whenever num_candidates * (row bytes) is a multiple of 16 for the array (always for
 * the vertices at orders 2 and 6; num_candidates % 4 == 0 covers vertices, objects and types at every order,
 * % 16 the mask as well); other shapes take a 4-byte coalesced path.  Same results either way. */
int32_t drt_trace_paths_dense_ex(drt_mesh_t mesh, const drt_trace_params *params, const float *tx,
                                 int64_t num_tx, const float *rx, int64_t num_rx,
                                 const drt_candidates *cands, const int32_t *interaction_types_in,
                                 float *vertices, int32_t *objects, uint8_t *mask,
                                 int32_t *interaction_types_out, void *workspace, size_t workspace_bytes,
                                 void *stream);
int32_t drt_trace_paths_dense_capped(drt_mesh_t mesh, const drt_trace_params *params, const float *tx,
                                 int64_t num_tx, const float *rx, int64_t num_rx,
                                 const drt_candidates *cands, const int32_t *interaction_types_in,
                                 float *vertices, int32_t *objects, uint8_t *mask,
                                 int32_t *interaction_types_out, int64_t max_survivors, void *workspace, size_t workspace_bytes,
                                 void *stream);

/* Compacted output: only valid paths, in the order of TracedPaths.masked_vertices
 * (geometry/_paths.py:274-297: row-major over [tx, rx, candidate]).
 *   keys     [max_paths] i64 : flat index (tx*Nrx + rx)*C + candidate_row of each valid path
 *   vertices [max_paths,order+2,3], objects [max_paths,order+2]
 * *num_valid_host receives the number of valid paths (the call synchronises the stream).
 * Returns DRT_E_CAPACITY (and the required count in *num_valid_host) if max_paths is too small. */
size_t drt_trace_compact_workspace_size(int64_t max_survivors, int64_t max_paths);
int32_t drt_trace_paths_compact(drt_mesh_t mesh, const drt_trace_params *params, const float *tx,
                                int64_t num_tx, const float *rx, int64_t num_rx,
                                const drt_candidates *cands, int64_t max_survivors,
                                int64_t max_paths, int64_t *keys, float *vertices,
                                int32_t *objects, int64_t *num_valid_host, void *workspace,
                                size_t workspace_bytes, void *stream);

/* The same trace with STATIC output shapes and NO host synchronisation -- the form a jax.ffi /
 * XLA custom-call handler or a HIP graph needs (the reference's boundary is
 * wp.jax_callable(func, output_dims=...), geometry/_mesh.py:266-276: output sizes fixed at trace
 * time).  Capacities are fixed by the caller; all `max_paths` rows of keys / vertices / objects are
 * written: the first counts_dev[1] rows are the valid paths in masked_vertices order, bit-identical
 * to drt_trace_paths_compact, the others are padding (key -1, vertices 0, objects -1).
 *   counts_dev [4] i64 (DEVICE): [0] candidates that passed the geometric checks, [1] valid paths,
 *                                [2] status word (DRT_TRACE_OVERFLOW_* bits, 0 = complete), [3] 0.
 * On overflow the rows written are valid paths but not all of them / not the first ones: the
 * caller inspects counts_dev[2] whenever it reads the result back and re-runs with larger
 * capacities.  Nothing is allocated, nothing synchronises, the launch sequence does not depend on
 * device data: the call can be captured in a HIP graph and replayed.  With DRT_TRACE_USE_BVH the
 * mesh LBVH must have been built before (drt_mesh_build_bvh).  Workspace as the sync variant. */
int32_t drt_trace_paths_compact_async(drt_mesh_t mesh, const drt_trace_params *params, const float *tx,
                                      int64_t num_tx, const float *rx, int64_t num_rx,
                                      const drt_candidates *cands, int64_t max_survivors,
                                      int64_t max_paths, int64_t *keys, float *vertices,
                                      int32_t *objects, int64_t *counts_dev, void *workspace,
                                      size_t workspace_bytes, void *stream);

/* VJP of the path vertices of `num_paths` traced paths w.r.t. transmitters, receivers and mesh
 * vertices (hand-derived reverse of the image method; the mask is a constant, as in the reference:
 * SURVEY.md section 3.2).  keys are those returned by drt_trace_paths_compact[_async] (or flat dense
 * indices; rows with key -1 are padding and contribute nothing); cotangent [num_paths,order+2,3].  Gradients are ACCUMULATED with atomic adds into
 * grad_tx [Ntx,3], grad_rx [Nrx,3], grad_vertices [Nv,3] (each may be NULL). */
int32_t drt_trace_paths_vjp(drt_mesh_t mesh, const float *tx, int64_t num_tx, const float *rx,
                            int64_t num_rx, const drt_candidates *cands, const int64_t *keys,
                            const float *vertices_cotangent, int64_t num_paths, float *grad_tx,
                            float *grad_rx, float *grad_vertices, void *stream);

/* The same VJP with an option for run-to-run REPRODUCIBLE gradients (hard part 6): with
 * DRT_TRACE_DETERMINISTIC_GRAD in params->flags every path writes its contributions, a stable sort groups them by
 * destination (transmitter / receiver / mesh vertex) and each group is summed in a fixed two-level order (chunks of 256
 * sorted contributions in path order, then the chunks in order: g / 256 + 256 dependent adds for a group of g) -- no
 * float atomics, bit-identical from run to run (and within rounding of the atomic version).  Without the
 * flag (or params == NULL) this is drt_trace_paths_vjp and the workspace is not touched. */
#define DRT_TRACE_DETERMINISTIC_GRAD 4
size_t drt_trace_vjp_workspace_size(int64_t num_paths, int32_t order);
int32_t drt_trace_paths_vjp_ex(drt_mesh_t mesh, const drt_trace_params *params, const float *tx, int64_t num_tx,
                               const float *rx, int64_t num_rx, const drt_candidates *cands, const int64_t *keys,
                               const float *vertices_cotangent, int64_t num_paths, float *grad_tx, float *grad_rx,
                               float *grad_vertices, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Beam-pruned exhaustive tracer: ONE call that returns the valid paths of the exhaustive tracer
 * (reference: Scene.trace_paths -> solver.trace_path_candidates over the FULL candidate list,
 * geometry/_scene.py:650-764, geometry/_solvers.py:803-848, 936-957; static output capacity like
 * wp.jax_callable(output_dims=...), geometry/_mesh.py:266-276) without visiting the
 * n (n-1)^(order-1) candidates per pair: prefixes of mirrors are discarded by NECESSARY geometric
 * conditions of a valid specular path (next primitive inside the pyramids spanned by the image of
 * the transmitter and the mirrors so far; previous and next point on one side of the mirror plane,
 * _solver_image_method.py:443-454) failing by more than a bound on the reference's own float32
 * error -- built per mirror from its incidence geometry, so a mirror seen at grazing incidence
 * switches its own tests off instead of losing a path (csrc/beam.hip).
 * Survivors are traced by the ordinary kernels: same valid paths, same masked_vertices order,
 * bit-identical vertices.  The lossless counterpart of the reference's SAMPLED visibility pruning
 * (HybridPathTracer, _solvers.py:1013-1056).  Orders 0..3.
 *
 *   keys     [max_paths] i64 : (tx * num_rx + rx) * n^order + sum_j m_j * n^(order-1-j), n = primitives,
 *                              m_j = primitive ids (ascending = masked_vertices order); order 0: tx * num_rx + rx
 *   vertices [max_paths,order+2,3] f32, objects [max_paths,order+2] i32 (triangle ids, even for quads)
 *   *num_valid_host: number of valid paths; the call synchronises the stream (list sizes are read back once
 *   per level / slice).  DRT_E_CAPACITY: a capacity of drt_beam_params / max_paths / the workspace is too small
 *   (*num_valid_host then holds the count that did not fit).  workspace: device memory, 16-byte aligned.
 * Gradients: drt_trace_paths_vjp with a drt_candidates of {table NULL, num_nodes = n, order,
 * reserved = DRT_CAND_PACKED_KEYS} and these keys.
 * Everything -- Morton clustering of primitives (cached in the mesh handle, like the LBVH) and receivers,
 * prefix expansion, receiver stage, row sort / decode, trace -- runs inside this call; no torch, no Python.
 * ------------------------------------------------------------------------------------------- */
typedef struct drt_beam_stats {
    int64_t levels[4];        /* prefixes kept at level 1, 2, 3 (this shard) */
    int64_t rows;             /* candidate rows handed to the tracer */
    int64_t slices;           /* slices of the last expansion */
    int64_t valid;            /* valid paths */
    int64_t grazing_prefixes; /* last-level prefixes whose error bound is unbounded: every test of theirs was off
                                 (they were kept, never dropped) -- informational */
    float unit_m;             /* u = kappa * ulp(largest coordinate magnitude), metres */
    float magnitude;          /* that magnitude */
    int32_t pair_mode;        /* 1: a triangle mesh searched over the primitives of the pairing pass (levels[] count
                                 prefixes of those primitives) */
    int32_t paired_primitives; /* pair mode: primitives that are coplanar PAIRS (the others are single triangles) */
    float expand_last_ms;     /* HIP-event time of the last expansion's kernels (all slices), */
    float emit_ms;            /*   of the receiver stage, */
    float trace_ms;           /*   of the fused trace of the candidate rows (sort / decode included) */
    float next_probe_prefixes; /* the smallest slice size any slice of this call suggested (prefixes per slice of the last
                                  expansion, from the measured fan-out; 0 = no slice ran): pass it as drt_beam_params.probe_prefixes
                                  of the NEXT call on a similar scene (a training loop that moves its transmitters) and
                                  that call starts with full-size slices instead of a 4096-prefix probe.  Only a hint:
                                  a slice that overflows is retried smaller, results never depend on it. */
} drt_beam_stats;

#define DRT_BEAM_EXPAND_PLAIN 1   /* expansion: every (prefix, primitive) pair tested, no cluster culling */
#define DRT_BEAM_EMIT_PLAIN 2     /* receiver stage: lane = prefix walks the receivers (after a vote on their clusters' boxes) */
#define DRT_BEAM_EMIT_CLUSTERED 4 /* receiver stage: Morton clusters of 64 even below 128 receivers */
#define DRT_BEAM_NO_PAIRS 8        /* triangle meshes: search triangle by triangle even when the pairing pass found pairs */
#define DRT_BEAM_EXPAND_FUSED 32  /* orders 2, 3: the last expansion as ONE kernel (round 4's form) instead of box stage +
                                     per-primitive stage as two launches (round 5; same records either way: A/B and cross-check) */
#define DRT_BEAM_ROWS_PLAIN 16    /* coplanar-pair mode: trace the 2^order triangle rows of a primitive row one by one instead
                                     of as one DRT_CAND_PAIR_BLOCKS block (A/B and cross-check; same result) */
/* (the mappings return the same rows; default: clustered expansion, clustered receivers from 128 on).
 * COPLANAR PAIRS (round 4; per primitive and for triangle soups since round 5, ABI 7): on a triangle mesh
 * (assume_quads == 0) two triangles A = (v0, v1, v2) and B = (v0, v2, v3) -- ANYWHERE in the triangle array -- with equal
 * (==) unit normals, the same first vertex and the same mask value are the same mirror for the reference's arithmetic
 * (plane point and normal are all the image method reads of a triangle, _solvers.py:552-562); when v0 v1 v2 v3 is convex
 * they are searched as ONE primitive with one 4-face pyramid.  drt_mesh_build_beam_clusters runs the pairing pass (a
 * matching along the fans around shared first vertices: the two halves of a wall, consecutive ears of a roof polygon);
 * triangles without a partner stay single-triangle primitives in the same list; when at least 30 % of the triangles
 * found a partner the prefix search runs over these P < n primitives, and every surviving primitive row is split into
 * its (at most 2^order) triangle rows for the exact trace.  Same paths, same keys / order / vertex bits as the
 * triangle-by-triangle search (DRT_BEAM_NO_PAIRS); levels[] and grazing_prefixes of drt_beam_stats then count
 * prefixes of primitives.  The reference's bruxelles.obj (14 206 triangles): 5 829 pairs, 8 377 primitives. */

typedef struct drt_beam_params {
    float kappa;            /* error unit u = kappa * ulp(M); <= 0: default 64 = the worst-case rounding count of
                               DESIGN.md section 9 (measured errors are 5-50x smaller: oracle/studies/beam_error_model.py;
                               configs[3]: 0.83 s at 64, 0.63 s at 16, same paths) */
    int32_t flags;          /* DRT_BEAM_* */
    /* list capacities; <= 0: sized from the scene (hard bounds num_tx * n^2 etc. and a fan-out estimate; at most
     * 2^26 / 2^27 / 2^26 / 2^22, which is what 10k+ triangle cities resolve to; a 12-triangle box at order 3 needs
     * < 1 MB).  Results never depend on them: the call slices its lists to fit. */
    int64_t max_entries;    /* level-2 prefix list (order 3), 32 B each */
    int64_t max_records;    /* records of one expansion slice, 8 B each */
    int64_t max_rows;       /* candidate rows of one slice, 8 + 8 + 4 order B each */
    int64_t max_survivors;  /* survivor queue of the fused trace */
    int64_t probe_prefixes; /* size of the first slice of the last expansion; <= 0: 4096 */
    int64_t shard_rank;     /* multi-GPU split: keep the level-1 prefixes (tx, m) with */
    int64_t shard_world;    /*   (tx * n + m) % shard_world == shard_rank; <= 1: everything */
    drt_beam_stats *stats;  /* host pointer or NULL */
} drt_beam_params;
#define DRT_CAND_PACKED_KEYS 4 /* drt_candidates.reserved bit: keys ARE the candidates (see above) */
/* drt_candidates.reserved bit, PER-PAIR TABLE only (table != NULL, pair_offsets != NULL, mesh without assume_quads):
 * COPLANAR-PAIR BLOCKS.  The table consists of blocks of 2^order consecutive rows; block b enumerates the triangle
 * choices of ONE sequence of primitives (q_0 .. q_{order-1}), each a pair of triangles (first, second) or a single
 * triangle: row b 2^order + c names, at mirror j, the first (bit_j(c) = 0) or second (1) triangle of q_j, with
 * bit_j(c) = (c >> (order-1-j)) & 1.  A row that is not a candidate (it names the missing second triangle of a single,
 * or one triangle twice in a row) is a PADDING row: all its ids are negative -- -1 for a triangle that does not exist,
 * -2 - id for one that does, so that the block's first and last row always show both triangles of every primitive; a
 * whole padding block is all -1.  Every pair_offsets entry is a multiple of 2^order; and the two triangles of every
 * pair are THE SAME MIRROR: equal (==) unit normals, equal first vertices, equal mask values (what
 * drt_mesh_build_beam_clusters establishes before drt_trace_paths_beam searches a triangle mesh over its pairs).
 * All rows of a block then share every image and reflection point -- the mirror (first vertex, normal) is the
 * only thing the image method reads of a triangle, _solvers.py:552-562 -- and differ in the inside tests alone: the
 * filter stage evaluates the chain ONCE per block and Moller-Trumbore against both triangles of each pair.  Same
 * survivors, same keys (global table rows), same vertices as without the bit (those are computed per row from the
 * row's own triangles). */
#define DRT_CAND_PAIR_BLOCKS 8

/* ---------------------------------------------------------------------------------------------
 * Capture-safe stable radix sort of device u64 keys, optionally with a u32 payload (csrc/radix_sort.hip): what the
 * asynchronous entry points sort their rows with, exported for hosts that build their own capturable pipelines
 * (library radix sorts reset their state with hipMemsetAsync, and memset nodes of a captured HIP graph do not replay
 * reliably on ROCm 7.x).  Kernels only, no allocation, no synchronisation; bits [begin_bit, end_bit) of the keys
 * decide, 8 per pass; keys_in / values_in are left intact; n < 2^31.
 * ------------------------------------------------------------------------------------------- */
size_t drt_sort_u64_workspace_size(int64_t n, int32_t with_values);
int32_t drt_sort_u64(const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *values_in, uint32_t *values_out, int64_t n,
                     int32_t begin_bit, int32_t end_bit, void *workspace, size_t workspace_bytes, void *stream);

/* Morton clusters of the mesh's primitives (allocates, synchronises; implied by drt_trace_paths_beam).  On a triangle
 * mesh the first call also runs the pairing pass.  The handle keeps up to two sets of clusters -- the pairing pass's
 * primitives and (`allow_pairs` = 0, what a DRT_BEAM_NO_PAIRS search uses) the plain triangles -- each built once and
 * never freed or moved before drt_mesh_destroy: a captured HIP graph may hold their addresses.
 * THREADS: these calls (and the first drt_trace_paths_beam / drt_mesh_build_bvh on a handle, which imply them) WRITE
 * to the handle; they must not run concurrently with any other call on the same handle.  Once built, any number of host
 * threads may trace through one handle on their own streams (tests/abi/abi_threads.cpp). */
int32_t drt_mesh_build_beam_clusters(drt_mesh_t mesh, void *stream);
int32_t drt_mesh_build_beam_clusters_ex(drt_mesh_t mesh, int32_t allow_pairs, void *stream);
/* What the pairing pass found: returns -1 (not run yet), 0 (too few pairs: the search runs triangle by triangle) or
 * 1 (pair mode), and the number of primitives / of pairs among them (0 unless pair mode).  Not a status code. */
int32_t drt_mesh_beam_pairing(drt_mesh_t mesh, int64_t *num_primitives, int64_t *num_pairs);
/* pair mode: the primitive table, i32 [num_primitives, 2] (DEVICE): the two triangles of each primitive, second = -1
 * for a single triangle; every triangle of the mesh appears exactly once */
int32_t drt_mesh_beam_pairing_table(drt_mesh_t mesh, int32_t *table_out, int64_t num_primitives, void *stream);
size_t drt_trace_beam_workspace_size(int64_t num_tx, int64_t num_rx, int64_t num_primitives, int32_t order,
                                     const drt_beam_params *beam, int64_t max_paths);
int32_t drt_trace_paths_beam(drt_mesh_t mesh, const drt_trace_params *params, const drt_beam_params *beam,
                             const float *tx, int64_t num_tx, const float *rx, int64_t num_rx, int32_t order,
                             int64_t max_paths, int64_t *keys, float *vertices, int32_t *objects,
                             int64_t *num_valid_host, void *workspace, size_t workspace_bytes, void *stream);

/* The same search with STATIC shapes and NO host synchronisation or allocation: the form a jax.ffi / XLA custom-call
 * handler or a HIP graph needs (reference boundary: wp.jax_callable(func, output_dims=...), geometry/_mesh.py:266-276,
 * 3082-3092), like drt_trace_paths_compact_async.  ONE pass with fixed capacities instead of slices sized from counts
 * read back: the lists are sized by `beam` (or the scene-sized defaults of drt_trace_beam_workspace_size -- the same
 * workspace query), every stage is launched over the capacity of its input list and takes the length from a device
 * counter, the scene scalars are computed on the device.  Orders 0..3.
 *   keys / vertices / objects : all max_paths rows are written; the first counts_dev[1] are the valid paths in
 *       masked_vertices order, bit-identical to drt_trace_paths_beam; the others are padding (key -1, vertices 0, objects -1)
 *   counts_dev [4] i64 (DEVICE): [0] rows that passed the geometric checks, [1] valid paths, [2] status word
 *       (DRT_TRACE_OVERFLOW_* | DRT_BEAM_OVERFLOW_*, 0 = complete), [3] candidate rows handed to the tracer
 * The workspace (and the outputs) belong to ONE call in flight: a second call on another stream must not start before the
 * first has finished -- the stages address their lists through device-side counters in the workspace.
 * Preconditions (they allocate / synchronise, so they cannot happen here): drt_mesh_build_beam_clusters(mesh), and
 * drt_mesh_build_bvh(mesh) with DRT_TRACE_USE_BVH.  On overflow the rows written are valid paths, but not all of them:
 * re-run with larger capacities (or call drt_trace_paths_beam, which slices). */
#define DRT_BEAM_OVERFLOW_ENTRIES 4  /* level-2 prefix list (order 3) beyond max_entries */
#define DRT_BEAM_OVERFLOW_RECORDS 8  /* records of the last expansion beyond max_records */
#define DRT_BEAM_OVERFLOW_ROWS 16    /* candidate rows beyond max_rows (max_rows >> order pair rows in coplanar-pair mode) */
int32_t drt_trace_paths_beam_async(drt_mesh_t mesh, const drt_trace_params *params, const drt_beam_params *beam,
                                   const float *tx, int64_t num_tx, const float *rx, int64_t num_rx, int32_t order,
                                   int64_t max_paths, int64_t *keys, float *vertices, int32_t *objects,
                                   int64_t *counts_dev, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * (f2) visibility pruning PER (transmitter, receiver) pair -- reference: HybridPathTracer, geometry/_solvers.py:960-1176
 * (which merges the visible sets over all end points, :969-973; per pair is the MI355X extension).  Pair (i, j) traces
 * F_i x N^(order-2) x L_j: first interaction among the primitives visible from transmitter i, last one among those
 * visible from receiver j, middle ones among the active primitives; all pairs in one ragged launch of the compact
 * tracer.  Inputs: vis_tx u8 [num_tx, T], vis_rx u8 [num_rx, T] (device; non-zero = triangle seen, e.g. from
 * drt_triangles_visible_from_vertex); quads count as seen when either triangle is (:1024-1031), masked primitives are
 * dropped (:1038-1042).  The CSR sets, their offsets and the pair offsets are built by kernels in the workspace (no
 * host enumeration; one read-back of the num_tx + num_rx set sizes).  Outputs as drt_trace_paths_compact, except
 * keys = PACKED keys (tx num_rx + rx) n^order + sum_j m_j n^(order-1-j) (n primitives): self-describing, so
 * drt_trace_paths_vjp takes them with DRT_CAND_PACKED_KEYS and needs none of the sets.  *num_evaluated_host (may be
 * NULL): rows of all pair spaces.  order >= 2.  flags: 0 = choose, DRT_HYBRID_RAGGED = lane per (pair, candidate)
 * row, DRT_HYBRID_PREFIX (order >= 3) = lane per prefix of order-1 interactions (huge pair spaces). */
#define DRT_HYBRID_PREFIX 1
#define DRT_HYBRID_RAGGED 2
size_t drt_trace_hybrid_pairs_workspace_size(int64_t num_tx, int64_t num_rx, int64_t num_primitives,
                                             int64_t max_survivors, int64_t max_paths);
int32_t drt_trace_paths_hybrid_pairs(drt_mesh_t mesh, const drt_trace_params *params, const float *tx, int64_t num_tx,
                                     const float *rx, int64_t num_rx, int32_t order, const uint8_t *vis_tx,
                                     const uint8_t *vis_rx, int32_t flags, int64_t max_survivors, int64_t max_paths,
                                     int64_t *keys, float *vertices, int32_t *objects, int64_t *num_valid_host,
                                     int64_t *num_evaluated_host, void *workspace, size_t workspace_bytes,
                                     void *stream);

/* ---------------------------------------------------------------------------------------------
 * (e) collectives of the path on RCCL, no torch in the process.  One process per
 * GPU; rank 0 draws an id (drt_comm_unique_id) and hands its 128 bytes to the other ranks out of band (the
 * host's rendezvous: a file, MPI, torch.distributed's store ...); every rank then calls drt_comm_init with
 * ITS device current.  Collectives are in place, on the caller's stream, asynchronous.
 *   drt_allreduce_min_u64 : packed first-hit keys of drt_first_hit_keys (triangle-block sharding), 8 B per ray
 *   drt_allreduce_max_u8  : blocked flags of the triangle-block tracer, 1 B per surviving candidate
 *   drt_allreduce_sum_f32 : gradients w.r.t. transmitters / receivers / mesh vertices
 *   drt_allgather_bytes   : fixed-size blocks (counts, then padded compact records); recv = world * bytes_per_rank
 * librccl.so is opened on first use (DRT_E_UNSUPPORTED when it cannot be): no link-time dependency.
 * ------------------------------------------------------------------------------------------- */
typedef struct drt_comm *drt_comm_t;
#define DRT_COMM_ID_BYTES 128
int32_t drt_comm_unique_id(uint8_t *id_out_host /* [DRT_COMM_ID_BYTES] */);
int32_t drt_comm_init(const uint8_t *unique_id_host, int32_t rank, int32_t world, drt_comm_t *comm_out);
int32_t drt_comm_destroy(drt_comm_t comm);
int32_t drt_comm_rank(drt_comm_t comm);
int32_t drt_comm_world(drt_comm_t comm);
int32_t drt_allreduce_min_u64(drt_comm_t comm, uint64_t *buf, int64_t n, void *stream);
int32_t drt_allreduce_max_u8(drt_comm_t comm, uint8_t *buf, int64_t n, void *stream);
int32_t drt_allreduce_sum_f32(drt_comm_t comm, float *buf, int64_t n, void *stream);
int32_t drt_allgather_bytes(drt_comm_t comm, const void *send, void *recv, int64_t bytes_per_rank, void *stream);

/* ---------------------------------------------------------------------------------------------
 * (f4) smoothed ("soft mask") mode -- reference: differt/src/differt/utils.py:70-89
 * (smoothing_function = sigmoid(x * alpha)); every bool of the hard mode becomes a float32
 * confidence in [0,1], differentiable in rays / end points / mesh vertices.
 *   Moller-Trumbore           geometry/_utils.py:1279-1320   (min of the six sigmoids)
 *   any triangle              geometry/_utils.py:1436-1537   (per tile of `batch_size` triangles: sum
 *                             of min(hit, sigmoid((1-hit_tol-t) alpha)) over the active triangles,
 *                             tiles folded with clip(left+right, max=1); batch_size <= 0 = None)
 *   same side                 geometry/_solver_image_method.py:450-453
 *   tracer                    geometry/_solvers.py:499-770, smoothed branches :599-713;
 *                             mask = min(inside, same_side, 1-blocked, 1-too_small, finite) * active
 * `dense` != 0: rays [R,3] x triangles [T,3,3] -> [R,T]; 0: paired, num_triangles == num_rays.
 * All *_vjp entry points ACCUMULATE (atomic adds) into zero-initialised gradient buffers, any of
 * which may be NULL; the min / max route cotangents to the first extremal term.
 * ------------------------------------------------------------------------------------------- */
int32_t drt_ray_intersect_triangle_smooth(const float *ray_origins, const float *ray_directions,
                                          int64_t num_rays, const float *triangle_vertices,
                                          int64_t num_triangles, int32_t dense, float epsilon,
                                          float smoothing_factor, float *t_out, float *hit_out,
                                          void *stream);
int32_t drt_ray_intersect_triangle_smooth_vjp(const float *ray_origins, const float *ray_directions,
                                              int64_t num_rays, const float *triangle_vertices,
                                              int64_t num_triangles, int32_t dense, float epsilon,
                                              float smoothing_factor, const float *t_cotangent,
                                              const float *hit_cotangent, float *grad_origins,
                                              float *grad_directions, float *grad_triangle_vertices,
                                              void *stream);
int32_t drt_ray_intersect_any_triangle_smooth(const float *ray_origins, const float *ray_directions,
                                              int64_t num_rays, const float *triangle_vertices,
                                              int64_t num_triangles, int64_t tv_ray_stride,
                                              const uint8_t *active, int64_t active_ray_stride,
                                              float epsilon, float hit_tol, float smoothing_factor,
                                              int64_t batch_size, float *out, void *stream);
/* grad_origins / grad_directions [R,3] are WRITTEN for rays with a non-zero cotangent (pass zeroed
 * buffers); grad_triangle_vertices has the layout of triangle_vertices (shared or per ray). */
int32_t drt_ray_intersect_any_triangle_smooth_vjp(const float *ray_origins, const float *ray_directions,
                                                  int64_t num_rays, const float *triangle_vertices,
                                                  int64_t num_triangles, int64_t tv_ray_stride,
                                                  const uint8_t *active, int64_t active_ray_stride,
                                                  float epsilon, float hit_tol, float smoothing_factor,
                                                  int64_t batch_size, const float *out_cotangent,
                                                  float *grad_origins, float *grad_directions,
                                                  float *grad_triangle_vertices, void *stream);
int32_t drt_consecutive_vertices_same_side_smooth(const float *vertices, const float *mirror_vertices,
                                                  const float *mirror_normals, int64_t batch,
                                                  int32_t num_mirrors, float smoothing_factor,
                                                  float *out, void *stream);
/* Dense layout of drt_trace_paths_dense with a float32 mask [Ntx,Nrx,C]; no workspace. */
int32_t drt_trace_paths_dense_smooth(drt_mesh_t mesh, const drt_trace_params *params,
                                     float smoothing_factor, int64_t batch_size, const float *tx,
                                     int64_t num_tx, const float *rx, int64_t num_rx,
                                     const drt_candidates *cands, float *vertices, int32_t *objects,
                                     float *mask, void *stream);
/* Cotangents of the dense outputs (vertices [Ntx,Nrx,C,order+2,3] and/or mask [Ntx,Nrx,C], either may
 * be NULL) -> grad_tx [Ntx,3], grad_rx [Nrx,3], grad_vertices [Nv,3]. */
int32_t drt_trace_paths_dense_smooth_vjp(drt_mesh_t mesh, const drt_trace_params *params,
                                         float smoothing_factor, int64_t batch_size, const float *tx,
                                         int64_t num_tx, const float *rx, int64_t num_rx,
                                         const drt_candidates *cands, const float *vertices_cotangent,
                                         const float *mask_cotangent, float *grad_tx, float *grad_rx,
                                         float *grad_vertices, void *stream);

/* ---------------------------------------------------------------------------------------------
 * (f4, second half) EM post-processing of traced paths.  Complex numbers are interleaved float32
 * pairs (re, im).  Reference: geometry/_utils.py:150-181 (path_length), em/_utils.py:84-303
 * (sp_directions, sp_rotation_matrix), em/_fresnel.py:47-214 (fresnel_coefficients),
 * plugins/deepmimo.py:334-404, 480-482, 533-711 (per-path channel coefficient as exported to DeepMIMO:
 * isotropic antennas, far field, all interactions specular reflections on half spaces
 * (thickness < 0) or slabs (thickness >= 0)).
 * ------------------------------------------------------------------------------------------- */
typedef struct drt_em_params {
    double frequency;        /* Hz */
    int32_t tx_polarization; /* 0 = "V", 1 = "H", 2 = tx_vector (plugins/deepmimo.py:568-589) */
    float tx_vector[3];
    int32_t rx_polarization; /* same encoding (plugins/deepmimo.py:645-662) */
    float rx_vector[3];
} drt_em_params;

int32_t drt_path_length(const float *paths, int64_t batch, int32_t path_length, float *out, void *stream);
/* em/_utils.py:14-44, :345-367 (db != 0: decibels), em/_fresnel.py:10-44 (principal sqrt of complex
 * epsilon_r * mu_r, [batch,2] -> [batch,2]) */
int32_t drt_length_to_delay(const float *length, const float *speed, int64_t batch, float *out, void *stream);
int32_t drt_fspl(const float *d, const float *f, int64_t batch, int32_t db, float *out, void *stream);
int32_t drt_refractive_index(const float *epsilon_r, int64_t batch, float *out, void *stream);
int32_t drt_sp_directions(const float *k_i, const float *k_r, const float *normals, int64_t batch,
                          float *e_i_s, float *e_i_p, float *e_r_s, float *e_r_p, void *stream);
/* out [batch,2,2] = [[<b_s,a_s>, <b_s,a_p>], [<b_p,a_s>, <b_p,a_p>]] */
int32_t drt_sp_rotation_matrix(const float *e_a_s, const float *e_a_p, const float *e_b_s,
                               const float *e_b_p, int64_t batch, float *out, void *stream);
/* n_r complex [batch,2], cos_theta_i [batch] -> r_s, r_p, t_s, t_p complex [batch,2] */
int32_t drt_fresnel_coefficients(const float *n_r, const float *cos_theta_i, int64_t batch, float *r_s,
                                 float *r_p, float *t_s, float *t_p, void *stream);
/* HOST pointers: sqrt(eta_r - j sigma / (omega eps0)) per material, complex [M,2]. */
int32_t drt_complex_refractive_index(const float *eta_r, const float *conductivity,
                                     int64_t num_materials, double frequency, float *n_complex_out);
/* vertices [N,order+2,3], objects [N,order+2] (triangle ids in columns 1..order), mesh normals [T,3],
 * face_materials i32[T] -> material index, n_complex [M,2], thickness [M] (negative = half space).
 * Outputs per path: a complex [N,2] (incl. the lambda/4pi factor), power [dBW], phase [deg],
 * length [m], delay [s], angles of arrival / departure [deg] (azimuth, elevation = polar angle). */
int32_t drt_paths_channel(const float *vertices, const int32_t *objects, int64_t num_paths, int32_t order,
                          const float *normals, const int32_t *face_materials, int64_t num_triangles,
                          const float *n_complex, const float *thickness, int64_t num_materials,
                          const drt_em_params *params, float *a, float *power, float *phase,
                          float *length, float *delay, float *aoa_az, float *aoa_el, float *aod_az,
                          float *aod_el, void *stream);
/* VJP of drt_paths_channel with respect to the path vertices.  cotangents [N,10] in the output order
 * (a.re, a.im, power, phase, length, delay, aoa_az, aoa_el, aod_az, aod_el); grad_vertices
 * [N,order+2,3] is WRITTEN.  Computed by forward-mode duals inside one lane per path (3 (order+2)
 * evaluations); the mesh normals and the material constants are constants of the differentiation. */
int32_t drt_paths_channel_vjp(const float *vertices, const int32_t *objects, int64_t num_paths,
                              int32_t order, const float *normals, const int32_t *face_materials,
                              int64_t num_triangles, const float *n_complex, const float *thickness,
                              int64_t num_materials, const drt_em_params *params,
                              const float *cotangents, float *grad_vertices, void *stream);

/* ---------------------------------------------------------------------------------------------
 * (a15) cell ids of equal rows -- reference: geometry/_paths.py:21-38 (`_cell_ids`), behind
 * TracedPaths.group_by_objects :378-421, multipath_cells :331-376, merge_cell_ids :41-74 and the
 * "first occurrence" of mask_duplicate_objects :196-252.
 *   ids_out[r] = min { i : rows[i,:] == rows[r,:] }      rows: i32[num_rows, width], row-major
 * ------------------------------------------------------------------------------------------- */
size_t drt_row_cell_ids_workspace_size(int64_t num_rows);
int32_t drt_row_cell_ids(const int32_t *rows, int64_t num_rows, int32_t width, int32_t *ids_out,
                         void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFERT_AMD_H */
