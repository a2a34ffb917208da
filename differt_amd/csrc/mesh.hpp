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
    // drt_mesh_build_beam_clusters: primitives sorted along a Morton curve, 64 per cluster.  TWO slots, both kept
    // once built (a captured HIP graph holds the pointers by value: nothing is ever freed before drt_mesh_destroy):
    //   beam[0]  the caller's primitives: triangles, or the quads (2i, 2i+1) of an assume_quads mesh
    //   beam[1]  a TRIANGLE mesh searched over the primitives of the pairing pass (`pair_*` below)
    struct BeamCache {
        void *blob = nullptr;        // one allocation, the others point into it
        int32_t *order = nullptr;    // [P]            primitive id at sorted position
        float *verts = nullptr;      // [Pp,NV,3]      vertices in sorted order (Pp = clusters * 64, padded)
        float *sigma = nullptr;      // [Pp]           shape factor (see beam.hip), max over its triangles
        float *planes = nullptr;     // [Pp*NP,4]      (n, <n, v0>) per plane
        float *uplanes = nullptr;    // [Pp*NP,4]      per cluster: its DISTINCT planes first (count in boxes[.,7])
        float *boxes = nullptr;      // [clusters,8]   lo[3], hi[3], max sigma, number of distinct planes
        float *subboxes = nullptr;   // [clusters,4,6] lo[3], hi[3] of each group of 16 consecutive primitives
        int64_t clusters = 0;
        int32_t kind = 0;            // primitive shape the clusters were built for (1, 2, 4: beam.hip struct Shape)
    } beam[2];
    float beam_max_abs = 0.0f;       // largest |coordinate| of the mesh vertices
    int32_t beam_quad4 = -1;         // assume_quads: every quad (2i, 2i+1) is a convex planar fan quad (shape 4): -1 / 0 / 1
    // The pairing pass over a triangle soup (beam.hip, pair_triangles): primitive p = the convex planar fan quad of
    // triangles (pair_tri[2p], pair_tri[2p+1]) that are THE SAME MIRROR for the reference (equal unit normals, first
    // vertices and mask values), or the single triangle pair_tri[2p] (pair_tri[2p+1] = -1).  The `pair_tv / _normals /
    // _shape / _mask` arrays lay the primitives out as a virtual mesh of 2 P triangles (a single's second triangle is
    // the degenerate (v0, v2, v2) with the first one's normal, shape factor and mask), which is what the kernels read.
    int32_t pair_state = -1;         // -1 not examined, 0 too few pairs to pay (search triangle by triangle), 1 built
    int64_t pair_prims = 0;          // P
    int64_t pair_quads = 0;          // primitives that are pairs
    void *pair_blob = nullptr;
    int32_t *pair_tri = nullptr;     // [P,2]
    float *pair_tv = nullptr;        // [2P,3,3]
    float *pair_normals = nullptr;   // [2P,3]
    float *pair_shape = nullptr;     // [2P]
    uint8_t *pair_mask = nullptr;    // [2P] or nullptr
};
