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
    uint8_t *mask = nullptr;       // [T] or nullptr (all active)
    void *bvh_nodes = nullptr;     // LBVH (csrc/bvh.hip), built lazily by drt_mesh_build_bvh
};
