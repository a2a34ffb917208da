// mesh.hpp -- the device-resident mesh record behind drt_mesh_t.
#pragma once

#include <cstdint>

struct drt_mesh {
    int64_t num_vertices = 0;
    int64_t num_triangles = 0;
    int32_t assume_quads = 0;
    int32_t has_mask = 0;
    // all device pointers, owned by the handle
    float *vertices = nullptr;     // [Nv,3]   copy of the caller's vertices
    int32_t *triangles = nullptr;  // [T,3]
    float *tri_verts = nullptr;    // [T,3,3]  gathered triangle vertices (reference Mesh.triangle_vertices)
    float *normals = nullptr;      // [T,3]    reference Mesh.normals
    float *shape = nullptr;        // [T]      largest 1/sin(corner angle) per triangle (beam.hip error bounds)
    uint8_t *mask = nullptr;       // [T] or nullptr (all active)
    void *bvh_nodes = nullptr;     // LBVH (csrc/bvh.hip), built lazily by drt_mesh_build_bvh
    uint32_t *bvh_leaf_ids = nullptr;  // [T] triangle ids in Morton order (leaf ranges of the nodes index it)
    // Primitive clusters of the beam-pruned tracer (csrc/beam.hip), built lazily by
    // drt_mesh_build_beam_clusters: primitives (triangles, or quads = triangle pairs) sorted along a Morton
    // curve, 64 per cluster.  One allocation (`beam_blob`), the others point into it.
    void *beam_blob = nullptr;
    int32_t *beam_order = nullptr;   // [P]            primitive id at sorted position
    float *beam_verts = nullptr;     // [Pp,3*scale,3] vertices in sorted order (Pp = clusters * 64, padded)
    float *beam_normals = nullptr;   // [Pp,scale,3]   unit normals of the primitive's triangles
    float *beam_sigma = nullptr;     // [Pp]           shape factor (see beam.hip), max over its triangles
    float *beam_planes = nullptr;    // [Pp*scale,4]   (n, <n, v0>) per triangle
    float *beam_uplanes = nullptr;   // [Pp*scale,4]   per cluster: its DISTINCT planes first (count in beam_boxes[.,7])
    float *beam_boxes = nullptr;     // [clusters,8]   lo[3], hi[3], max sigma, number of distinct planes
    float *beam_subboxes = nullptr;  // [clusters,4,6] lo[3], hi[3] of each group of 16 consecutive primitives
    int64_t beam_clusters = 0;
    float beam_max_abs = 0.0f;       // largest |coordinate| of the mesh vertices
    int32_t beam_scale = 0;          // what the clusters were built for: primitive shape (1, 2, 4: beam.hip struct Shape)
                                     // + 16 when a triangle mesh is searched over its coplanar pairs
    int32_t beam_quad4 = -1;         // every pair (2i, 2i+1) is a convex planar fan quad (shape 4): -1 / 0 / 1
    int32_t beam_pairs = -1;         // triangle mesh whose triangles (2i, 2i+1) share their mirror plane
                                     // (same unit normal, same first vertex, same mask): -1 not examined, 0 no, 1 yes
                                     // -- the pruned search then runs over the n/2 coplanar PAIRS (beam.hip)
};
