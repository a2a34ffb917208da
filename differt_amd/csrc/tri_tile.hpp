// tri_tile.hpp -- the LDS triangle tile shared by every brute-force kernel (any-hit, first-hit,
// visibility, occlusion stage of the tracer): 256 triangles as 48-byte (v0, e1, e2, active) records.
// A lane that walks the tile reads record j with three broadcast-friendly ds_read_b128; with one lane
// per triangle (stride 48 B) the four 16-lane groups of ds_read_b128 fall on distinct bank quads.
#pragma once

#include "geom.hpp"

namespace drt {

constexpr int kTile = 256;  // triangles per LDS tile: 256 * 48 B = 12 KiB

struct __attribute__((aligned(16))) TriRec {
    float v0x, v0y, v0z, e1x;
    float e1y, e1z, e2x, e2y;
    float e2z;
    uint32_t active;
    uint32_t pad0, pad1;
};
static_assert(sizeof(TriRec) == 48, "TriRec must be 3 x 16 B");

// thread t of a (>= 256-thread) block stages triangle base + t (edges precomputed once per tile)
__device__ __forceinline__ void stage_tile(TriRec *lds, const float *__restrict__ tv,
                                           const uint8_t *__restrict__ active, int64_t base,
                                           int64_t end) {
    const int64_t j = base + threadIdx.x;
    if (threadIdx.x < kTile && j < end) {
        const TriE tr = load_tri(tv + 9 * j);
        TriRec rec;
        rec.v0x = tr.v0.x; rec.v0y = tr.v0.y; rec.v0z = tr.v0.z;
        rec.e1x = tr.e1.x; rec.e1y = tr.e1.y; rec.e1z = tr.e1.z;
        rec.e2x = tr.e2.x; rec.e2y = tr.e2.y; rec.e2z = tr.e2.z;
        rec.active = active ? (uint32_t)active[j] : 1u;
        rec.pad0 = rec.pad1 = 0;
        lds[threadIdx.x] = rec;
    }
}

__device__ __forceinline__ TriE rec_tri(const TriRec &r) {
    return TriE{V3{r.v0x, r.v0y, r.v0z}, V3{r.e1x, r.e1y, r.e1z}, V3{r.e2x, r.e2y, r.e2z}};
}

}  // namespace drt
