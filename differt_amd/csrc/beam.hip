// beam.hip -- conservative ("beam") pruning of the exhaustive candidate space, GPU resident.
//
// Reference context: the exhaustive tracer enumerates n (n-1)^(k-1) candidates per (tx, rx) pair
// (geometry/_solvers.py:803-848) and the hybrid tracer prunes them with SAMPLED visibility
// (_solvers.py:1013-1056), which is lossy.  This file prunes with a geometric argument instead.
//
// A specular path tx -> P_1 in m_1 -> ... -> P_k in m_k -> rx unfolds into straight lines through
// the images I_j of the transmitter (I_0 = tx, I_j = mirror image of I_{j-1} in the plane of m_j):
// P_{j+1} lies on the ray from I_j through P_j, i.e. inside the pyramid with apex I_j spanned by the
// primitive m_j, and rx lies inside the pyramid (I_k, m_k).  Moreover the reference's same-side
// check (_solver_image_method.py:443-454) needs P_{j-1} and P_{j+1} on one side of the plane of m_j.
// Both are NECESSARY conditions of a valid path, so a prefix (m_1..m_j) can be discarded together
// with all of its n^(k-j) extensions when
//   (S) the previous point set (tx or the primitive m_{j-1}) lies strictly on one side of the plane of
//       m_j and the next primitive strictly on the other, or
//   (B) every vertex of the next primitive lies strictly outside one face plane of the pyramid
//       (I_j, m_j) (for quads: of both triangles' pyramids),
// "strictly" meaning by more than a margin derived from E, a bound on the position error of the
// reference's own float32 reflection points (DESIGN.md section 9 derives E from the scene magnitude,
// the order and the smallest incidence cosine covered by the guarantee): plane-side tests use 4 E
// (two points, each within E of its primitive, plus the rounding of the dot products); pyramid faces
// use E (1 + |x - I_j| / h_j) with h_j the distance of the apex from the mirror plane -- an error E at
// the mirror opens the pyramid by the angle E / h_j, and an apex (nearly) in the mirror plane
// switches the test off by itself.  The tests only ever REMOVE candidates
// that the reference arithmetic rejects; what survives is evaluated by the ordinary trace kernels
// with the reference arithmetic, so results are those of the exhaustive tracer.
//
// Pipeline (device lists, wave-ballot compaction, no host enumeration):
//   beam_seed    level-1 prefixes (tx, m_1) for all active primitives
//   beam_expand  level-j prefixes x primitives, tests (S) and (B) -> 8-byte (prefix, primitive) records
//   beam_finish  records -> level-(j+1) prefixes (intermediate levels only)
//   beam_emit    level-k prefixes (or level-(k-1) prefixes + records) x receivers -> packed rows
//                ((tx nrx + rx) n^k + sum_j m_j n^(k-1-j))
#include "bvh.hpp"
#include "common.hpp"
#include "geom.hpp"
#include "mesh.hpp"

#pragma clang fp contract(off)

namespace drt {
// Dot product with fused multiply-adds (3 instructions instead of 5).  The beam tests are necessary
// conditions with explicit margins, not parity arithmetic: one rounding instead of three per product-sum is
// only more accurate, and every mapping of the expansion / receiver stage uses this same function, so
// their survivors stay identical.
__device__ __forceinline__ float fdot(V3 a, V3 b) { return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, a.z * b.z)); }
// |w| for the distance-proportional part of a margin: v_sqrt_f32 (1 ulp) nudged up, instead of the ~12
// instruction correctly rounded square root per (prefix, receiver)
__device__ __forceinline__ float margin_len(V3 w) { return __builtin_amdgcn_sqrtf(fdot(w, w)) * 1.000001f; }
}  // namespace drt

namespace drt {

struct BeamEntry {  // == drt_beam_entry (32 bytes)
    int32_t tx;
    int32_t id[3];
    float apex[3];
    int32_t side_prev;  // side of the previous point set w.r.t. the plane of the last mirror: +1 / -1 / 0 (near, straddling)
};
static_assert(sizeof(BeamEntry) == 32, "BeamEntry layout");

struct BeamMesh {
    const float *tv;       // [T,3,3]
    const float *normals;  // [T,3]
    const uint8_t *mask;   // [T] or null
    int64_t nprim;
    int32_t scale;  // triangles per primitive (2 with assume_quads)
};

__device__ __forceinline__ bool prim_active(const BeamMesh &M, int64_t p) {
    if (!M.mask) return true;
    const int64_t f = p * M.scale;
    return M.mask[f] != 0 && (M.scale == 1 || M.mask[f + 1] != 0);
}

// mirror plane of a primitive: first vertex and normal of its first triangle (_solvers.py:552-562)
__device__ __forceinline__ void prim_plane(const BeamMesh &M, int64_t p, V3 &pt, V3 &n) {
    const int64_t f = p * M.scale;
    pt = ld3(M.tv + 9 * f);
    n = ld3(M.normals + 3 * f);
}

__device__ __forceinline__ int side_of_range(float dmin, float dmax, float E) {
    return (dmin > E) ? 1 : ((dmax < -E) ? -1 : 0);
}

// side of all vertices of primitive p w.r.t. plane (pt, n)
__device__ __forceinline__ int side_of_prim(const BeamMesh &M, int64_t p, V3 pt, V3 n, float E) {
    float dmin = kInf, dmax = -kInf;
    const float *v = M.tv + 9 * p * M.scale;
    for (int i = 0; i < 3 * M.scale; ++i) {
        const float d = fdot(ld3(v + 3 * i) - pt, n);
        dmin = fminf(dmin, d);
        dmax = fmaxf(dmax, d);
    }
    if (!(dmin == dmin) || !(dmax == dmax)) return 0;  // NaN geometry: never prune
    return side_of_range(dmin, dmax, E);
}

// inward unit normals of the three face planes of the pyramid (apex I, triangle u v w); a degenerate
// face (apex on the edge line, or apex in the triangle's plane) gets a zero normal: it never separates
struct Pyramid {
    V3 n[3];
};
__device__ __forceinline__ V3 face_normal(V3 I, V3 a, V3 b, V3 third) {
    const V3 N = cross(a - I, b - I);
    const float len = __builtin_sqrtf(fdot(N, N));
    const float s = fdot(third - I, N);
    if (!(len > 0.0f) || !(s == s) || s == 0.0f || !is_finite(len)) return V3{0, 0, 0};
    const float inv = ((s > 0.0f) ? 1.0f : -1.0f) / len;
    return N * inv;
}
__device__ __forceinline__ Pyramid make_pyramid(V3 I, const float *tri9) {
    const V3 u = ld3(tri9), v = ld3(tri9 + 3), w = ld3(tri9 + 6);
    Pyramid P;
    P.n[0] = face_normal(I, u, v, w);
    P.n[1] = face_normal(I, v, w, u);
    P.n[2] = face_normal(I, w, u, v);
    return P;
}

// Pyramid (apex I) over the triangle `t` of primitive `p` after reflecting it in the planes of the
// primitives refl[0..nrefl) in turn ("unfolding"): a specular path is a straight line from the last
// image of the transmitter that crosses the unfolded images of ALL earlier mirrors, not only the last
// one.  Also returns 1 / distance of the apex from the unfolded triangle's plane (for the margin).
__device__ __forceinline__ Pyramid unfolded_pyramid(const BeamMesh &M, V3 I, int64_t p, int t, const int32_t *refl,
                                                    int nrefl, float &inv_h) {
    const float *tri = M.tv + 9 * (p * M.scale + t);
    V3 v[3] = {ld3(tri), ld3(tri + 3), ld3(tri + 6)};
    for (int r = 0; r < nrefl; ++r) {
        V3 pt, n;
        prim_plane(M, refl[r], pt, n);
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = image_of_vertex(v[k], pt, n);
    }
    Pyramid P;
    P.n[0] = face_normal(I, v[0], v[1], v[2]);
    P.n[1] = face_normal(I, v[1], v[2], v[0]);
    P.n[2] = face_normal(I, v[2], v[0], v[1]);
    const V3 c = cross(v[1] - v[0], v[2] - v[0]);
    const float len = __builtin_sqrtf(fdot(c, c));
    const float h = (len > 0.0f) ? __builtin_fabsf(fdot(I - v[0], c)) / len : 0.0f;
    inv_h = (h > 0.0f) ? 1.0f / h : kInf;
    return P;
}

constexpr int kBeamTile = 128;  // primitives per LDS tile (x up to 6 vertices x 12 B = 9 KiB)

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void beam_seed_kernel(BeamMesh M, const float *__restrict__ tx, int64_t ntx,
                                                        float E, BeamEntry *__restrict__ out, int64_t cap,
                                                        unsigned long long *__restrict__ count) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool in = g < ntx * M.nprim;
    const int64_t it = in ? g / M.nprim : 0, a = in ? g - it * M.nprim : 0;
    const bool keep = in && prim_active(M, a);
    BeamEntry e{};
    if (keep) {
        V3 pt, n;
        prim_plane(M, a, pt, n);
        const V3 t = ld3(tx + 3 * it);
        const V3 I = image_of_vertex(t, pt, n);
        const float d = fdot(t - pt, n);
        e.tx = (int32_t)it;
        e.id[0] = (int32_t)a;
        e.id[1] = e.id[2] = -1;
        e.apex[0] = I.x;
        e.apex[1] = I.y;
        e.apex[2] = I.z;
        e.side_prev = (d == d) ? side_of_range(d, d, 4.0f * E) : 0;
    }
    const unsigned long long vote = __ballot(keep);
    if (vote) {
        unsigned long long base = 0;
        if (lane == __builtin_ctzll(vote)) base = atomicAdd(count, (unsigned long long)__popcll(vote));
        base = __shfl(base, __builtin_ctzll(vote), 64);
        if (keep) {
            const unsigned long long slot = base + (unsigned long long)__popcll(vote & ((1ull << lane) - 1ull));
            if ((int64_t)slot < cap) out[slot] = e;
        }
    }
}

constexpr int kBeamWaveBuf = 192;  // records staged per wave before one flush (>= 128: a flush moves 64+)
constexpr int kBeamWaveBufBig = 1024;  // the same for kernels that emit ~1e10 records (one atomic per ~960)

// wave-private LDS staging buffer -> output list: ONE global atomic for `n` records
__device__ __forceinline__ void beam_flush(const unsigned long long *buf, int n, int lane,
                                           unsigned long long *__restrict__ out, int64_t cap,
                                           unsigned long long *__restrict__ count) {
    unsigned long long b0 = 0;
    if (lane == 0) b0 = atomicAdd(count, (unsigned long long)n);
    b0 = __shfl(b0, 0, 64);
    for (int i = lane; i < n; i += 64) {
        const unsigned long long slot = b0 + (unsigned long long)i;
        if ((int64_t)slot < cap) out[slot] = buf[i];
    }
}

// lane = prefix; the block walks all primitives through LDS tiles
template <int SCALE>
__global__ __launch_bounds__(256) void beam_expand_kernel(BeamMesh M, const BeamEntry *__restrict__ in, int64_t n_in,
                                                          int level, float E, unsigned long long *__restrict__ out,
                                                          int64_t cap, unsigned long long *__restrict__ count,
                                                          int64_t prims_per_split) {
    __shared__ float lds_v[kBeamTile][3 * SCALE][3];
    __shared__ uint8_t lds_act[kBeamTile];
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool have = g < n_in;
    BeamEntry e{};
    if (have) e = in[g];
    const int32_t m = have ? e.id[level - 1] : 0;
    const V3 I = V3{e.apex[0], e.apex[1], e.apex[2]};
    V3 pm{0, 0, 0}, nm{0, 0, 1};
    Pyramid pyr[SCALE];
    Pyramid pyr0[SCALE];  // level 2: the first mirror id[0] unfolded in the plane of m (zero normals otherwise)
    float inv_h = kInf;   // 1 / distance of the apex from the plane of m (inf: the pyramid test never prunes)
    float inv_h0[SCALE];
#pragma unroll
    for (int t = 0; t < SCALE; ++t) {
        pyr[t] = Pyramid{};
        pyr0[t] = Pyramid{};
        inv_h0[t] = kInf;
    }
    if (have) {
        prim_plane(M, m, pm, nm);
        const float h = __builtin_fabsf(fdot(I - pm, nm));
        inv_h = (h > 0.0f) ? 1.0f / h : kInf;
#pragma unroll
        for (int t = 0; t < SCALE; ++t) pyr[t] = make_pyramid(I, M.tv + 9 * ((int64_t)m * SCALE + t));
        if (level == 2) {
#pragma unroll
            for (int t = 0; t < SCALE; ++t) pyr0[t] = unfolded_pyramid(M, I, e.id[0], t, &e.id[1], 1, inv_h0[t]);
        }
    }
    __shared__ unsigned long long wbuf[4][kBeamWaveBuf];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int wcount = 0;  // wave-uniform: records waiting in wbuf[wave]
    // few prefixes x many primitives (configs[4]: 2e5 x 2e5) would leave most CUs idle with one block per
    // 256 prefixes: blockIdx.y splits the primitive range so that the launch holds >= ~2048 blocks
    const int64_t prim_begin = (int64_t)blockIdx.y * prims_per_split;
    const int64_t prim_end = (prim_begin + prims_per_split < M.nprim) ? prim_begin + prims_per_split : M.nprim;
    for (int64_t base = prim_begin; base < prim_end; base += kBeamTile) {
        __syncthreads();
        for (int i = threadIdx.x; i < kBeamTile * 3 * SCALE; i += 256) {
            const int64_t p = base + i / (3 * SCALE);
            const int vtx = i % (3 * SCALE);
            V3 v{0, 0, 0};
            if (p < prim_end) v = ld3(M.tv + 9 * p * SCALE + 3 * vtx);
            lds_v[i / (3 * SCALE)][vtx][0] = v.x;
            lds_v[i / (3 * SCALE)][vtx][1] = v.y;
            lds_v[i / (3 * SCALE)][vtx][2] = v.z;
        }
        if (threadIdx.x < kBeamTile) {
            const int64_t p = base + threadIdx.x;
            lds_act[threadIdx.x] = (uint8_t)(p < prim_end && prim_active(M, p));
        }
        __syncthreads();
        const int nt = (int)((prim_end - base < kBeamTile) ? prim_end - base : kBeamTile);
        for (int j = 0; j < nt; ++j) {
            if (!lds_act[j]) continue;  // wave-uniform
            const int32_t c = (int32_t)(base + j);
            // (S) sides of c w.r.t. the plane of m, (B) all vertices outside one face of every pyramid of m
            float dmin = kInf, dmax = -kInf;
            bool out_face[SCALE][3], out_face0[SCALE][3];
#pragma unroll
            for (int t = 0; t < SCALE; ++t)
#pragma unroll
                for (int f = 0; f < 3; ++f) out_face[t][f] = out_face0[t][f] = true;
            bool nan = false;
#pragma unroll
            for (int vtx = 0; vtx < 3 * SCALE; ++vtx) {
                const V3 x = V3{lds_v[j][vtx][0], lds_v[j][vtx][1], lds_v[j][vtx][2]};
                const float d = fdot(x - pm, nm);
                nan = nan || !(d == d);
                dmin = fminf(dmin, d);
                dmax = fmaxf(dmax, d);
                const V3 w = x - I;
                // |w|_1 >= |w|_2: a slightly larger margin (conservative), no square root in the loop
                const float wl = (__builtin_fabsf(w.x) + __builtin_fabsf(w.y)) + __builtin_fabsf(w.z);
                const float thr = -(E + E * (wl * inv_h));  // -inf / NaN: never separates
#pragma unroll
                for (int t = 0; t < SCALE; ++t) {
                    const float thr0 = -(E + E * (wl * inv_h0[t]));
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        const float s = fdot(w, pyr[t].n[f]);
                        out_face[t][f] = out_face[t][f] && (s < thr);  // NaN compares false
                        const float s0 = fdot(w, pyr0[t].n[f]);
                        out_face0[t][f] = out_face0[t][f] && (s0 < thr0);
                    }
                }
            }
            bool separated = true, separated0 = true;
#pragma unroll
            for (int t = 0; t < SCALE; ++t) {
                separated = separated && (out_face[t][0] || out_face[t][1] || out_face[t][2]);
                separated0 = separated0 && (out_face0[t][0] || out_face0[t][1] || out_face0[t][2]);
            }
            separated = separated || separated0;  // outside the cone of the last mirror OR of the unfolded first one
            const int side_c = nan ? 0 : side_of_range(dmin, dmax, 4.0f * E);
            const bool keep = have && (c != m) && !separated && !(e.side_prev * side_c == -1);
            // survivors leave as 8-byte (source prefix, primitive) records, staged per wave in LDS and
            // flushed 64+ at a time: one global atomic per FLUSH.  One atomic per iteration meant 7e9
            // atomics on ONE address for configs[3] (40 s: the L2 atomic unit, not the arithmetic, set
            // the pace); building the child prefix here (dependent global loads under divergence) was
            // worse still.
            const unsigned long long vote = __ballot(keep);
            if (vote) {
                if (keep) {
                    const int slot = wcount + __popcll(vote & ((1ull << lane) - 1ull));
                    wbuf[wave][slot] = ((unsigned long long)(uint32_t)g << 32) | (uint32_t)c;
                }
                wcount += __popcll(vote);
                if (wcount > kBeamWaveBuf - 64) {  // room for one more full ballot is gone: flush
                    beam_flush(wbuf[wave], wcount, lane, out, cap, count);
                    wcount = 0;
                }
            }
        }
    }
    if (wcount > 0) beam_flush(wbuf[wave], wcount, lane, out, cap, count);
}

// ---------------------------------------------------------------------------------------------
// The same expansion as a walk over the mesh LBVH (csrc/bvh.hip): lane = prefix, a subtree is skipped
// when its (padded) box fails the SAME tests as a primitive would -- every point of the box outside one
// face plane of every pyramid of a cone, or the box strictly on the wrong side of the mirror plane.
// The box versions use the box's support along the plane normal and the largest margin inside the box,
// so "box pruned" implies "every primitive inside pruned": the survivors are exactly those of the
// brute-force kernel, found in O(survivors x depth) instead of O(primitives) per prefix.
// With quads a primitive can be reached through either of its triangles: the second triangle emits only
// if the first one's own box is pruned (duplicates left by the padding of stored boxes are removed when
// the rows are sorted).
// ---------------------------------------------------------------------------------------------
template <int SCALE>
struct BeamCtx {
    V3 I, pm, nm;
    float inv_h, inv_h0[SCALE], E;
    Pyramid pyr[SCALE], pyr0[SCALE];
    int side_prev;

    __device__ __forceinline__ bool cone_outside_box(const Pyramid (&P)[SCALE], const float (&ih)[SCALE], V3 w, V3 e,
                                                     float wl) const {
        bool sep = true;
#pragma unroll
        for (int t = 0; t < SCALE; ++t) {
            const float thr = -(E + E * (wl * ih[t]));
            bool st = false;
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const V3 n = P[t].n[f];
                const float smax = fdot(w, n) + ((__builtin_fabsf(n.x) * e.x + __builtin_fabsf(n.y) * e.y) +
                                                __builtin_fabsf(n.z) * e.z);
                st = st || (smax < thr);
            }
            sep = sep && st;
        }
        return sep;
    }

    // true: no primitive inside [lo, hi] can survive
    __device__ __forceinline__ bool box_pruned(const float *lo, const float *hi) const {
        const V3 c = V3{0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
        const V3 e = V3{0.5f * (hi[0] - lo[0]), 0.5f * (hi[1] - lo[1]), 0.5f * (hi[2] - lo[2])};
        if (!(e.x >= 0.0f) || !(e.y >= 0.0f) || !(e.z >= 0.0f)) return false;  // NaN / empty box: keep
        if (side_prev != 0) {
            const float dc = fdot(c - pm, nm);
            const float r = (__builtin_fabsf(nm.x) * e.x + __builtin_fabsf(nm.y) * e.y) + __builtin_fabsf(nm.z) * e.z;
            const int sb = (dc == dc) ? side_of_range(dc - r, dc + r, 4.0f * E) : 0;
            if (side_prev * sb == -1) return true;
        }
        const V3 w = c - I;
        // largest |x - I|_1 inside the box -> the largest (most demanding) margin
        const float wl = ((__builtin_fabsf(w.x) + __builtin_fabsf(w.y)) + __builtin_fabsf(w.z)) + ((e.x + e.y) + e.z);
        float ih[SCALE];
#pragma unroll
        for (int t = 0; t < SCALE; ++t) ih[t] = inv_h;
        return cone_outside_box(pyr, ih, w, e, wl) || cone_outside_box(pyr0, inv_h0, w, e, wl);
    }

    // the per-vertex test of the brute-force kernel for primitive c
    __device__ __forceinline__ bool prim_survives(const BeamMesh &M, int64_t c) const {
        const float *v = M.tv + 9 * c * SCALE;
        V3 vx[3 * SCALE];
#pragma unroll
        for (int vtx = 0; vtx < 3 * SCALE; ++vtx) vx[vtx] = ld3(v + 3 * vtx);
        return prim_survives_v(vx);
    }
    __device__ __forceinline__ bool prim_survives_v(const V3 (&vx)[3 * SCALE]) const {
        float dmin = kInf, dmax = -kInf;
        bool out_face[SCALE][3], out_face0[SCALE][3];
#pragma unroll
        for (int t = 0; t < SCALE; ++t)
#pragma unroll
            for (int f = 0; f < 3; ++f) out_face[t][f] = out_face0[t][f] = true;
        bool nan = false;
#pragma unroll
        for (int vtx = 0; vtx < 3 * SCALE; ++vtx) {
            const V3 x = vx[vtx];
            const float d = fdot(x - pm, nm);
            nan = nan || !(d == d);
            dmin = fminf(dmin, d);
            dmax = fmaxf(dmax, d);
            const V3 w = x - I;
            const float wl = (__builtin_fabsf(w.x) + __builtin_fabsf(w.y)) + __builtin_fabsf(w.z);
            const float thr = -(E + E * (wl * inv_h));
#pragma unroll
            for (int t = 0; t < SCALE; ++t) {
                const float thr0 = -(E + E * (wl * inv_h0[t]));
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    out_face[t][f] = out_face[t][f] && (fdot(w, pyr[t].n[f]) < thr);
                    out_face0[t][f] = out_face0[t][f] && (fdot(w, pyr0[t].n[f]) < thr0);
                }
            }
        }
        bool separated = true, separated0 = true;
#pragma unroll
        for (int t = 0; t < SCALE; ++t) {
            separated = separated && (out_face[t][0] || out_face[t][1] || out_face[t][2]);
            separated0 = separated0 && (out_face0[t][0] || out_face0[t][1] || out_face0[t][2]);
        }
        const int side_c = nan ? 0 : side_of_range(dmin, dmax, 4.0f * E);
        return !(separated || separated0) && !(side_prev * side_c == -1);
    }
};

template <int SCALE>
__global__ __launch_bounds__(256) void beam_expand_bvh_kernel(BeamMesh M, const BvhNode *__restrict__ nodes, int64_t T,
                                                              const BeamEntry *__restrict__ in, int64_t n_in, int level,
                                                              float E, unsigned long long *__restrict__ out,
                                                              int64_t cap, unsigned long long *__restrict__ count) {
    __shared__ unsigned long long wbuf[4][kBeamWaveBuf];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool have = g < n_in;
    BeamEntry e{};
    if (have) e = in[g];
    const int32_t m = have ? e.id[level - 1] : 0;
    BeamCtx<SCALE> ctx;
    ctx.E = E;
    ctx.I = V3{e.apex[0], e.apex[1], e.apex[2]};
    ctx.pm = V3{0, 0, 0};
    ctx.nm = V3{0, 0, 1};
    ctx.inv_h = kInf;
    ctx.side_prev = e.side_prev;
#pragma unroll
    for (int t = 0; t < SCALE; ++t) {
        ctx.pyr[t] = Pyramid{};
        ctx.pyr0[t] = Pyramid{};
        ctx.inv_h0[t] = kInf;
    }
    if (have) {
        prim_plane(M, m, ctx.pm, ctx.nm);
        const float h = __builtin_fabsf(fdot(ctx.I - ctx.pm, ctx.nm));
        ctx.inv_h = (h > 0.0f) ? 1.0f / h : kInf;
#pragma unroll
        for (int t = 0; t < SCALE; ++t) ctx.pyr[t] = make_pyramid(ctx.I, M.tv + 9 * ((int64_t)m * SCALE + t));
        if (level == 2) {
#pragma unroll
            for (int t = 0; t < SCALE; ++t) ctx.pyr0[t] = unfolded_pyramid(M, ctx.I, e.id[0], t, &e.id[1], 1, ctx.inv_h0[t]);
        }
    }
    int32_t stack[kBvhStack];
    int sp = 0;
    int32_t node = (T == 1) ? ~0 : 0;
    bool active = have;
    int wcount = 0;
    // wave-synchronous walk: every active lane handles ONE node per trip, survivors of the trip are
    // appended with one ballot
    while (__any(active)) {
        bool keep = false;
        int32_t kc = 0;
        if (active) {
            if (node < 0) {  // leaf = triangle ~node of primitive c
                const int32_t tri = ~node;
                const int32_t c = tri / SCALE;
                bool first = true;
                if (SCALE == 2 && (tri & 1)) {
                    // the quad's first triangle gets there too unless its own box is pruned
                    const float *v = M.tv + 9 * (int64_t)(tri - 1);
                    float lo[3], hi[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        lo[k] = fminf(v[k], fminf(v[3 + k], v[6 + k]));
                        hi[k] = fmaxf(v[k], fmaxf(v[3 + k], v[6 + k]));
                    }
                    first = ctx.box_pruned(lo, hi);
                }
                if (first && c != m && prim_active(M, c) && ctx.prim_survives(M, c)) {
                    keep = true;
                    kc = c;
                }
                if (sp == 0) active = false; else node = stack[--sp];
            } else {
                const BvhNode nd = nodes[node];
                const bool gl = !ctx.box_pruned(nd.llo, nd.lhi);
                const bool gr = !ctx.box_pruned(nd.rlo, nd.rhi);
                if (gl && gr) {
                    if (sp < kBvhStack) stack[sp++] = nd.right;
                    node = nd.left;
                } else if (gl) {
                    node = nd.left;
                } else if (gr) {
                    node = nd.right;
                } else if (sp == 0) {
                    active = false;
                } else {
                    node = stack[--sp];
                }
            }
        }
        const unsigned long long vote = __ballot(keep);
        if (vote) {
            if (keep) {
                const int slot = wcount + __popcll(vote & ((1ull << lane) - 1ull));
                wbuf[wave][slot] = ((unsigned long long)(uint32_t)g << 32) | (uint32_t)kc;
            }
            wcount += __popcll(vote);
            if (wcount > kBeamWaveBuf - 64) {
                beam_flush(wbuf[wave], wcount, lane, out, cap, count);
                wcount = 0;
            }
        }
    }
    if (wcount > 0) beam_flush(wbuf[wave], wcount, lane, out, cap, count);
}

// ---------------------------------------------------------------------------------------------
// Transposed expansion: lane = PRIMITIVE (64 consecutive primitives per wave are spatially coherent in any
// sensible mesh), prefixes are wave-uniform: a block stages the derived data of 256 prefixes in LDS (apex,
// mirror plane, pyramids), every wave then walks them and first tests the bounding SPHERE of its 64
// primitives against the prefix's cones and mirror plane -- one uniform decision that skips the 64
// per-primitive tests for most (prefix, wave) pairs.  With lanes = prefixes (beam_expand_kernel) such a
// pre-test cannot pay: 64 unrelated cones almost never agree.  Same survivors, by construction: the sphere
// test is the box test of the LBVH variant with a ball instead of a box.
// ---------------------------------------------------------------------------------------------
template <int SCALE>
struct alignas(16) BeamPrefD {  // derived data of one prefix, as staged in LDS
    float I[3], pm[3], nm[3];
    float inv_h;
    int32_t side_prev, m;
    float pyr[SCALE][9], pyr0[SCALE][9];
    float inv_h0[SCALE];
    float pad[(4 - ((12 + 19 * SCALE) & 3)) & 3];
};

__device__ __forceinline__ float wave_sum_f(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}
__device__ __forceinline__ float wave_max_f(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
    return x;
}

template <int SCALE>
__global__ __launch_bounds__(256) void beam_expand_t_kernel(BeamMesh M, const BeamEntry *__restrict__ in, int64_t n_in,
                                                            int level, float E, unsigned long long *__restrict__ out,
                                                            int64_t cap, unsigned long long *__restrict__ count,
                                                            int64_t prefixes_per_split) {
    __shared__ BeamPrefD<SCALE> pd[256];
    __shared__ unsigned long long wbuf[4][kBeamWaveBuf];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool lane_ok = c < M.nprim && prim_active(M, c);
    V3 vx[3 * SCALE];
#pragma unroll
    for (int k = 0; k < 3 * SCALE; ++k) vx[k] = lane_ok ? ld3(M.tv + 9 * c * SCALE + 3 * k) : V3{0, 0, 0};
    // bounding sphere of the wave's primitives (centre = mean vertex of the active lanes)
    float cnt = lane_ok ? (float)(3 * SCALE) : 0.0f;
    V3 sum{0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3 * SCALE; ++k) sum = sum + vx[k];
    cnt = wave_sum_f(cnt);
    const float inv_cnt = (cnt > 0.0f) ? 1.0f / cnt : 0.0f;
    const V3 sc = V3{wave_sum_f(lane_ok ? sum.x : 0.0f) * inv_cnt, wave_sum_f(lane_ok ? sum.y : 0.0f) * inv_cnt,
                     wave_sum_f(lane_ok ? sum.z : 0.0f) * inv_cnt};
    float r2 = 0.0f;
#pragma unroll
    for (int k = 0; k < 3 * SCALE; ++k) {
        const V3 dv = vx[k] - sc;
        r2 = fmaxf(r2, lane_ok ? fdot(dv, dv) : 0.0f);
    }
    // radius rounded up generously (sqrt + a relative pad); NaN geometry -> NaN radius -> never culls
    const float sr = __builtin_sqrtf(wave_max_f(r2)) * 1.0001f + 1e-30f;
    const bool wave_any = cnt > 0.0f;

    const int64_t p_begin = (int64_t)blockIdx.y * prefixes_per_split;
    const int64_t p_end = (p_begin + prefixes_per_split < n_in) ? p_begin + prefixes_per_split : n_in;
    int wcount = 0;
    for (int64_t base = p_begin; base < p_end; base += 256) {
        __syncthreads();
        {  // stage the derived data of prefix base + threadIdx.x
            const int64_t g = base + threadIdx.x;
            BeamPrefD<SCALE> d{};
            d.m = -1;
            d.inv_h = kInf;
#pragma unroll
            for (int t = 0; t < SCALE; ++t) d.inv_h0[t] = kInf;
            if (g < p_end) {
                const BeamEntry e = in[g];
                const int32_t m = (level == 1) ? e.id[0] : e.id[1];
                const V3 I = V3{e.apex[0], e.apex[1], e.apex[2]};
                V3 pm, nm;
                prim_plane(M, m, pm, nm);
                const float h = __builtin_fabsf(fdot(I - pm, nm));
                d.I[0] = I.x; d.I[1] = I.y; d.I[2] = I.z;
                d.pm[0] = pm.x; d.pm[1] = pm.y; d.pm[2] = pm.z;
                d.nm[0] = nm.x; d.nm[1] = nm.y; d.nm[2] = nm.z;
                d.inv_h = (h > 0.0f) ? 1.0f / h : kInf;
                d.side_prev = e.side_prev;
                d.m = m;
#pragma unroll
                for (int t = 0; t < SCALE; ++t) {
                    const Pyramid P = make_pyramid(I, M.tv + 9 * ((int64_t)m * SCALE + t));
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        d.pyr[t][3 * f] = P.n[f].x; d.pyr[t][3 * f + 1] = P.n[f].y; d.pyr[t][3 * f + 2] = P.n[f].z;
                    }
                    if (level == 2) {
                        const int32_t refl = e.id[1];
                        const Pyramid Q = unfolded_pyramid(M, I, e.id[0], t, &refl, 1, d.inv_h0[t]);
#pragma unroll
                        for (int f = 0; f < 3; ++f) {
                            d.pyr0[t][3 * f] = Q.n[f].x; d.pyr0[t][3 * f + 1] = Q.n[f].y; d.pyr0[t][3 * f + 2] = Q.n[f].z;
                        }
                    }
                }
            }
            pd[threadIdx.x] = d;
        }
        __syncthreads();
        const int nt = (int)((p_end - base < 256) ? p_end - base : 256);
        if (!wave_any) continue;  // wave-uniform; the barriers above are outside this branch
        for (int j = 0; j < nt; ++j) {
            const BeamPrefD<SCALE> &d = pd[j];  // broadcast reads
            const V3 I = V3{d.I[0], d.I[1], d.I[2]}, pm = V3{d.pm[0], d.pm[1], d.pm[2]}, nm = V3{d.nm[0], d.nm[1], d.nm[2]};
            const float inv_h = d.inv_h;
            const int side_prev = d.side_prev;
            // ---- uniform: bounding sphere of the wave's primitives vs this prefix ----
            {
                const V3 w = sc - I;
                const float wl = ((__builtin_fabsf(w.x) + __builtin_fabsf(w.y)) + __builtin_fabsf(w.z)) + 1.7320509f * sr;
                bool sep = true, sep0 = true;
#pragma unroll
                for (int t = 0; t < SCALE; ++t) {
                    const float thr = -(E + E * (wl * inv_h)), thr0 = -(E + E * (wl * d.inv_h0[t]));
                    bool st = false, st0 = false;
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        st = st || (fdot(w, V3{d.pyr[t][3 * f], d.pyr[t][3 * f + 1], d.pyr[t][3 * f + 2]}) + sr < thr);
                        st0 = st0 || (fdot(w, V3{d.pyr0[t][3 * f], d.pyr0[t][3 * f + 1], d.pyr0[t][3 * f + 2]}) + sr < thr0);
                    }
                    sep = sep && st;
                    sep0 = sep0 && st0;
                }
                bool cull = sep || sep0;
                if (side_prev != 0) {
                    const float dc = fdot(sc - pm, nm);
                    const int sb = (dc == dc) ? side_of_range(dc - sr, dc + sr, 4.0f * E) : 0;
                    cull = cull || (side_prev * sb == -1);
                }
                // identical on every lane; make it a scalar branch for the compiler
                if (__builtin_amdgcn_readfirstlane((int)cull)) continue;
            }
            // ---- per lane: the primitive test ----
            float dmin = kInf, dmax = -kInf;
            bool out_face[SCALE][3], out_face0[SCALE][3];
#pragma unroll
            for (int t = 0; t < SCALE; ++t)
#pragma unroll
                for (int f = 0; f < 3; ++f) out_face[t][f] = out_face0[t][f] = true;
            bool nan = false;
#pragma unroll
            for (int k = 0; k < 3 * SCALE; ++k) {
                const V3 x = vx[k];
                const float dd = fdot(x - pm, nm);
                nan = nan || !(dd == dd);
                dmin = fminf(dmin, dd);
                dmax = fmaxf(dmax, dd);
                const V3 w = x - I;
                const float wl = (__builtin_fabsf(w.x) + __builtin_fabsf(w.y)) + __builtin_fabsf(w.z);
                const float thr = -(E + E * (wl * inv_h));
#pragma unroll
                for (int t = 0; t < SCALE; ++t) {
                    const float thr0 = -(E + E * (wl * d.inv_h0[t]));
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        out_face[t][f] = out_face[t][f] &&
                                         (fdot(w, V3{d.pyr[t][3 * f], d.pyr[t][3 * f + 1], d.pyr[t][3 * f + 2]}) < thr);
                        out_face0[t][f] = out_face0[t][f] &&
                                          (fdot(w, V3{d.pyr0[t][3 * f], d.pyr0[t][3 * f + 1], d.pyr0[t][3 * f + 2]}) < thr0);
                    }
                }
            }
            bool separated = true, separated0 = true;
#pragma unroll
            for (int t = 0; t < SCALE; ++t) {
                separated = separated && (out_face[t][0] || out_face[t][1] || out_face[t][2]);
                separated0 = separated0 && (out_face0[t][0] || out_face0[t][1] || out_face0[t][2]);
            }
            const int side_c = nan ? 0 : side_of_range(dmin, dmax, 4.0f * E);
            const bool keep = lane_ok && ((int32_t)c != d.m) && !(separated || separated0) && !(side_prev * side_c == -1);
            const unsigned long long vote = __ballot(keep);
            if (vote) {
                if (keep) {
                    const int slot = wcount + __popcll(vote & ((1ull << lane) - 1ull));
                    wbuf[wave][slot] = ((unsigned long long)(uint32_t)(base + j) << 32) | (uint32_t)c;
                }
                wcount += __popcll(vote);
                if (wcount > kBeamWaveBuf - 64) {
                    beam_flush(wbuf[wave], wcount, lane, out, cap, count);
                    wcount = 0;
                }
            }
        }
    }
    if (wcount > 0) beam_flush(wbuf[wave], wcount, lane, out, cap, count);
}

// (source prefix, primitive) record -> child prefix: image of the apex in the new mirror, side of the
// parent mirror w.r.t. the new mirror's plane
__device__ __forceinline__ BeamEntry beam_child(const BeamMesh &M, const BeamEntry &e, int level, int32_t c, float E) {
    V3 pc, nc;
    prim_plane(M, c, pc, nc);
    const V3 I2 = image_of_vertex(V3{e.apex[0], e.apex[1], e.apex[2]}, pc, nc);
    BeamEntry o = e;
    o.id[level] = c;
    o.apex[0] = I2.x;
    o.apex[1] = I2.y;
    o.apex[2] = I2.z;
    o.side_prev = side_of_prim(M, e.id[level - 1], pc, nc, 4.0f * E);
    return o;
}

__global__ __launch_bounds__(256) void beam_finish_kernel(BeamMesh M, const BeamEntry *__restrict__ src,
                                                          const unsigned long long *__restrict__ rec, int64_t n,
                                                          int level, float E, BeamEntry *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long r = rec[i];
    out[i] = beam_child(M, src[r >> 32], level, (int32_t)(uint32_t)r, E);
}

// lane = level-k prefix, loop over the receivers
template <int SCALE>
__global__ __launch_bounds__(256) void beam_emit_kernel(BeamMesh M, const BeamEntry *__restrict__ in,
                                                        const unsigned long long *__restrict__ rec, int64_t n_in,
                                                        int order, const float *__restrict__ rx, int64_t nrx, float E,
                                                        long long *__restrict__ rows, int64_t cap,
                                                        unsigned long long *__restrict__ count) {
    __shared__ unsigned long long wbuf[4][kBeamWaveBuf];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int wcount = 0;  // wave-uniform: rows waiting in wbuf[wave]
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool have = g < n_in;
    BeamEntry e{};
    if (have) {
        if (rec) {  // (level order-1 prefix, last primitive) record: build the level-`order` prefix here
            const unsigned long long r = rec[g];
            e = beam_child(M, in[r >> 32], order - 1, (int32_t)(uint32_t)r, E);
        } else {
            e = in[g];
        }
    }
    const int32_t c = have ? e.id[order - 1] : 0;
    const V3 I = V3{e.apex[0], e.apex[1], e.apex[2]};
    V3 pc{0, 0, 0}, nc{0, 0, 1};
    // pyramids (apex = last image) over the last mirror and over every earlier mirror unfolded through the
    // later ones: the receiver must see ALL of them in line -- the exact-geometry form of "every reflection
    // point lies inside its primitive"
    Pyramid pyr[3][SCALE];
    float inv_h[3][SCALE];
    long long tail = 0;  // sum_j id_j n^(k-1-j)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int t = 0; t < SCALE; ++t) {
            pyr[j][t] = Pyramid{};
            inv_h[j][t] = kInf;
        }
    if (have) {
        prim_plane(M, c, pc, nc);
        for (int j = 0; j < order; ++j) {
#pragma unroll
            for (int t = 0; t < SCALE; ++t)
                pyr[j][t] = unfolded_pyramid(M, I, e.id[j], t, &e.id[j + 1], order - 1 - j, inv_h[j][t]);
            tail = tail * (long long)M.nprim + (long long)e.id[j];
        }
    }
    long long npow = 1;
    for (int j = 0; j < order; ++j) npow *= (long long)M.nprim;
    // receivers are wave-uniform scalar loads; the next one is in flight during this one's tests (the counters
    // showed 41 % of the wave-cycles in s_waitcnt with a load + wait per iteration)
    const float *prx = rx;
    V3 r_next = ld3(prx);
    const int nrx32 = (int)nrx;  // < 2^31: the 62-bit row key bounds it
    for (int ir = 0; ir < nrx32; ++ir) {
        const V3 r = r_next;
        prx += (ir + 1 < nrx32) ? 3 : 0;
        r_next = ld3(prx);
        const float d = fdot(r - pc, nc);
        // wrong side of the last mirror: side_prev * d < -4E (side_prev in {-1, 0, +1}; 0 or a NaN distance never
        // rejects) -- the same decision as side_prev * side_of_range(d, d, 4E) == -1 in one multiply + compare
        const bool wrong_side = (float)e.side_prev * d < -4.0f * E;
        const V3 w = r - I;
        const float wl = margin_len(w);
        // the pyramids in turn, earliest mirror first (unfolded farthest from the apex = the narrowest cone);
        // the wave leaves the receiver as soon as none of its 64 prefixes is still inside (same tests, same
        // result: a prefix that fails one pyramid is dropped whatever the others say)
        bool alive = have && !wrong_side;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j < order) {
                if (!__any(alive)) break;
                bool inside_any = false;
#pragma unroll
                for (int t = 0; t < SCALE; ++t) {
                    const float thr = -(E + E * (wl * inv_h[j][t]));
                    bool inside = true;
#pragma unroll
                    for (int f = 0; f < 3; ++f) inside = inside && !(fdot(w, pyr[j][t].n[f]) < thr);
                    inside_any = inside_any || inside;
                }
                alive = alive && inside_any;
            }
        }
        const bool keep = alive;
        // rows leave through the wave's LDS staging buffer, one global atomic per FLUSH: rows are sparse (mostly
        // one lane per ballot), so an atomic per ballot was ~1.5e8 same-address atomics at configs[3]
        const unsigned long long vote = __ballot(keep);
        if (vote) {
            if (keep) {
                const int slot = wcount + __popcll(vote & ((1ull << lane) - 1ull));
                wbuf[wave][slot] = (unsigned long long)(((long long)e.tx * (long long)nrx + (long long)ir) * npow + tail);
            }
            wcount += __popcll(vote);
            if (wcount > kBeamWaveBuf - 64) {
                beam_flush(wbuf[wave], wcount, lane, reinterpret_cast<unsigned long long *>(rows), cap, count);
                wcount = 0;
            }
        }
    }
    if (wcount > 0) beam_flush(wbuf[wave], wcount, lane, reinterpret_cast<unsigned long long *>(rows), cap, count);
}

// value of lane `l` (wave-uniform index) on every lane, through v_readlane: no LDS round trip, no wait
__device__ __forceinline__ float lane_bcast(float x, int l) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(x), l));
}
__device__ __forceinline__ V3 lane_bcast(V3 v, int l) { return V3{lane_bcast(v.x, l), lane_bcast(v.y, l), lane_bcast(v.z, l)}; }
template <int SCALE>
__device__ __forceinline__ BeamCtx<SCALE> lane_bcast(const BeamCtx<SCALE> &c, int l) {
    BeamCtx<SCALE> o;
    o.I = lane_bcast(c.I, l);
    o.pm = lane_bcast(c.pm, l);
    o.nm = lane_bcast(c.nm, l);
    o.inv_h = lane_bcast(c.inv_h, l);
    o.E = c.E;
    o.side_prev = __builtin_amdgcn_readlane(c.side_prev, l);
#pragma unroll
    for (int t = 0; t < SCALE; ++t) {
        o.inv_h0[t] = lane_bcast(c.inv_h0[t], l);
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            o.pyr[t].n[f] = lane_bcast(c.pyr[t].n[f], l);
            o.pyr0[t].n[f] = lane_bcast(c.pyr0[t].n[f], l);
        }
    }
    return o;
}

// ---------------------------------------------------------------------------------------------
// Expansion with cluster-level culling and transposed survivors (the structure of
// beam_emit_clustered_kernel): the primitives arrive sorted along a Morton curve (`prim_order`) in clusters
// of 64 with an axis-aligned box each.  lane = prefix tests each cluster's box with BeamCtx::box_pruned (the
// box form of the very tests a primitive gets: box pruned => every primitive inside pruned, as in the LBVH
// walk); the surviving (prefix, cluster) pairs are then tested per primitive with the prefix's context
// broadcast lane-to-wave (v_readlane) and lane = primitive of the cluster.  Same survivors as the other mappings (tested);
// per prefix the work drops from one 150-instruction test per primitive to one ~60-instruction box test per
// 64 primitives plus full-lane tests of the clusters its cones actually reach.
// ---------------------------------------------------------------------------------------------
template <int SCALE>
__global__ __launch_bounds__(128) void beam_expand_clustered_kernel(
    BeamMesh M, const BeamEntry *__restrict__ in, int64_t n_in, int level, float E, unsigned long long *__restrict__ out,
    int64_t cap, unsigned long long *__restrict__ count, const int32_t *__restrict__ prim_order,
    const float *__restrict__ sorted_vertices, const float *__restrict__ boxes, int64_t nclusters,
    int64_t clusters_per_split) {
    // 8 KiB per wave: a flush every ~960 records.  With 192 (~130 per flush) the 8.4e7 flush atomics per step of
    // configs[3] -- all on ONE address, ~1.75e8/s -- were half of the kernel's time
    __shared__ unsigned long long wbuf[2][kBeamWaveBufBig];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t g = (int64_t)blockIdx.x * 128 + threadIdx.x;
    const bool have = g < n_in;
    BeamEntry e{};
    if (have) e = in[g];
    const int32_t m = have ? e.id[level - 1] : -1;
    BeamCtx<SCALE> ctx;
    ctx.E = E;
    ctx.I = V3{e.apex[0], e.apex[1], e.apex[2]};
    ctx.pm = V3{0, 0, 0};
    ctx.nm = V3{0, 0, 1};
    ctx.inv_h = kInf;
    ctx.side_prev = e.side_prev;
#pragma unroll
    for (int t = 0; t < SCALE; ++t) {
        ctx.pyr[t] = Pyramid{};
        ctx.pyr0[t] = Pyramid{};
        ctx.inv_h0[t] = kInf;
    }
    if (have) {
        prim_plane(M, m, ctx.pm, ctx.nm);
        const float h = __builtin_fabsf(fdot(ctx.I - ctx.pm, ctx.nm));
        ctx.inv_h = (h > 0.0f) ? 1.0f / h : kInf;
#pragma unroll
        for (int t = 0; t < SCALE; ++t) ctx.pyr[t] = make_pyramid(ctx.I, M.tv + 9 * ((int64_t)m * SCALE + t));
        if (level == 2) {
#pragma unroll
            for (int t = 0; t < SCALE; ++t) ctx.pyr0[t] = unfolded_pyramid(M, ctx.I, e.id[0], t, &e.id[1], 1, ctx.inv_h0[t]);
        }
    }
    const int64_t cl_begin = (int64_t)blockIdx.y * clusters_per_split;
    const int64_t cl_end = (cl_begin + clusters_per_split < nclusters) ? cl_begin + clusters_per_split : nclusters;
    const unsigned long long gbase = (unsigned long long)((int64_t)blockIdx.x * 128 + wave * 64);
    int wcount = 0;
    // software pipeline: the next cluster's box (scalar loads), primitive id and vertices (sorted copy, no
    // indirection) are in flight while this cluster is tested -- fetched whether or not the cluster will be
    // hit (the mesh lives in L2); with the loads issued only after a hit the kernel sat in s_waitcnt
    float nb[6];
    int32_t p_next;
    V3 vx_next[3 * SCALE];
    auto fetch = [&](int64_t c) {
        const int64_t cc = (c < cl_end) ? c : cl_end - 1;  // the last trip re-reads its own cluster
#pragma unroll
        for (int k = 0; k < 6; ++k) nb[k] = boxes[6 * cc + k];
        const int64_t pos = cc * 64 + lane;
        const int64_t pc = (pos < M.nprim) ? pos : M.nprim - 1;
        p_next = (pos < M.nprim) ? prim_order[pc] : -1;
#pragma unroll
        for (int k = 0; k < 3 * SCALE; ++k) vx_next[k] = ld3(sorted_vertices + 9 * pc * SCALE + 3 * k);
    };
    if (cl_begin < cl_end) fetch(cl_begin);
    for (int64_t cl = cl_begin; cl < cl_end; ++cl) {
        const float lo[3] = {nb[0], nb[1], nb[2]}, hi[3] = {nb[3], nb[4], nb[5]};
        const int32_t p = p_next;
        V3 vx[3 * SCALE];
#pragma unroll
        for (int k = 0; k < 3 * SCALE; ++k) vx[k] = vx_next[k];
        fetch(cl + 1);
        unsigned long long todo = __ballot(have && !ctx.box_pruned(lo, hi));
        if (todo == 0) continue;
        // ---- transposed: lane = primitive of the cluster ----
        const bool act = p >= 0 && prim_active(M, p);
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            // the prefix of lane l on every lane: v_readlane of its registers (the LDS broadcast this replaces
            // left the kernel 76 % of its wave-cycles in s_waitcnt at 20 % VALU issue, profiles/r02/beam.md)
            const BeamCtx<SCALE> cx = lane_bcast<SCALE>(ctx, l);
            const bool keep = act && (p != __builtin_amdgcn_readlane(m, l)) && cx.prim_survives_v(vx);
            const unsigned long long vote = __ballot(keep);
            if (vote) {
                if (keep) {
                    const int slot = wcount + __popcll(vote & ((1ull << lane) - 1ull));
                    wbuf[wave][slot] = ((gbase + (unsigned long long)l) << 32) | (uint32_t)p;
                }
                wcount += __popcll(vote);
                if (wcount > kBeamWaveBufBig - 64) {
                    beam_flush(wbuf[wave], wcount, lane, out, cap, count);
                    wcount = 0;
                }
            }
        }
    }
    if (wcount > 0) beam_flush(wbuf[wave], wcount, lane, out, cap, count);
}

// beam_emit for MANY receivers (configs[4]: 1024): the receivers arrive sorted along a Morton curve in
// clusters of 64 with an axis-aligned bounding box each (a receiver grid is flat: a ball would be a poor
// bound).  lane = prefix as above, but a lane first tests each cluster's box against its pyramids / mirror
// plane (the same inequalities with the box's extent along the face normal added, so a
// culled cluster holds no receiver the per-receiver test would keep), and only the (prefix, cluster) pairs
// that survive are tested per receiver -- TRANSPOSED: the prefix's data is broadcast from LDS and lane =
// receiver of the cluster, so those tests run with full lanes instead of once per wave-any.  Same per-receiver
// arithmetic as beam_emit_kernel -> the same set of rows (measured on configs[4]: emit 551 -> see
// profiles/r02/beam.md).
template <int SCALE>
struct alignas(16) BeamEmitD {
    float I[3];
    int32_t side_prev;
    float pc[3];
    int32_t tx;
    float nc[3];
    int32_t pad;
    long long tail;
    float inv_h[3][SCALE];
    float pyr[3][SCALE][9];
};

template <int SCALE>
__global__ __launch_bounds__(128) void beam_emit_clustered_kernel(
    BeamMesh M, const BeamEntry *__restrict__ in, const unsigned long long *__restrict__ rec, int64_t n_in, int order,
    const float *__restrict__ rx_sorted, const int32_t *__restrict__ rx_index, const float *__restrict__ boxes,
    int64_t nrx, float E, long long *__restrict__ rows, int64_t cap, unsigned long long *__restrict__ count) {
    __shared__ unsigned long long wbuf[2][kBeamWaveBuf];
    int wcount = 0;  // wave-uniform: rows waiting in wbuf[wave]
    __shared__ BeamEmitD<SCALE> lds[128];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t g = (int64_t)blockIdx.x * 128 + threadIdx.x;
    const bool have = g < n_in;
    BeamEntry e{};
    if (have) {
        if (rec) {
            const unsigned long long r = rec[g];
            e = beam_child(M, in[r >> 32], order - 1, (int32_t)(uint32_t)r, E);
        } else {
            e = in[g];
        }
    }
    const int32_t c = have ? e.id[order - 1] : 0;
    const V3 I = V3{e.apex[0], e.apex[1], e.apex[2]};
    V3 pc{0, 0, 0}, nc{0, 0, 1};
    Pyramid pyr[3][SCALE];
    float inv_h[3][SCALE];
    long long tail = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int t = 0; t < SCALE; ++t) {
            pyr[j][t] = Pyramid{};
            inv_h[j][t] = kInf;
        }
    if (have) {
        prim_plane(M, c, pc, nc);
        for (int j = 0; j < order; ++j) {
#pragma unroll
            for (int t = 0; t < SCALE; ++t)
                pyr[j][t] = unfolded_pyramid(M, I, e.id[j], t, &e.id[j + 1], order - 1 - j, inv_h[j][t]);
            tail = tail * (long long)M.nprim + (long long)e.id[j];
        }
    }
    {  // this lane's data, for the transposed per-receiver tests (read back by its own wave only)
        BeamEmitD<SCALE> &d = lds[threadIdx.x];
        d.I[0] = I.x; d.I[1] = I.y; d.I[2] = I.z;
        d.pc[0] = pc.x; d.pc[1] = pc.y; d.pc[2] = pc.z;
        d.nc[0] = nc.x; d.nc[1] = nc.y; d.nc[2] = nc.z;
        d.side_prev = e.side_prev;
        d.tx = e.tx;
        d.pad = 0;
        d.tail = tail;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int t = 0; t < SCALE; ++t) {
                d.inv_h[j][t] = inv_h[j][t];
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    d.pyr[j][t][3 * f] = pyr[j][t].n[f].x;
                    d.pyr[j][t][3 * f + 1] = pyr[j][t].n[f].y;
                    d.pyr[j][t][3 * f + 2] = pyr[j][t].n[f].z;
                }
            }
    }
    __syncthreads();
    long long npow = 1;
    for (int j = 0; j < order; ++j) npow *= (long long)M.nprim;
    const int64_t nclusters = (nrx + 63) / 64;
    for (int64_t cl = 0; cl < nclusters; ++cl) {
        // ---- per lane (= prefix): can ANY receiver of the cluster pass?  box (centre, half extents), wave-uniform ----
        const V3 sc = ld3(boxes + 6 * cl);
        // half extents, generously padded (rounding of the box and of the tests below)
        const V3 hx = V3{boxes[6 * cl + 3] * 1.0001f + E, boxes[6 * cl + 4] * 1.0001f + E, boxes[6 * cl + 5] * 1.0001f + E};
        bool maybe = have;
        {
            const V3 w = sc - I;
            // >= |r - I| for every receiver r of the cluster
            const float wl = margin_len(w) + margin_len(hx);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j < order) {
                    bool sep_all = true;
#pragma unroll
                    for (int t = 0; t < SCALE; ++t) {
                        const float thr = -(E + E * (wl * inv_h[j][t]));
                        bool sep = false;
#pragma unroll
                        for (int f = 0; f < 3; ++f) {
                            const V3 nf = pyr[j][t].n[f];  // max over the box of <x - I, n> = <c - I, n> + <|n|, h>
                            const float ext = (__builtin_fabsf(nf.x) * hx.x + __builtin_fabsf(nf.y) * hx.y) + __builtin_fabsf(nf.z) * hx.z;
                            sep = sep || (fdot(w, nf) + ext < thr);
                        }
                        sep_all = sep_all && sep;
                    }
                    maybe = maybe && !sep_all;
                }
            }
            const float dc = fdot(sc - pc, nc);
            const float de = (__builtin_fabsf(nc.x) * hx.x + __builtin_fabsf(nc.y) * hx.y) + __builtin_fabsf(nc.z) * hx.z;
            const int sb = (dc == dc) ? side_of_range(dc - de, dc + de, 4.0f * E) : 0;
            maybe = maybe && !(e.side_prev * sb == -1);
        }
        unsigned long long todo = __ballot(maybe);
        if (todo == 0) continue;
        // ---- transposed: lane = receiver of the cluster, prefix broadcast from LDS ----
        const int64_t pos = cl * 64 + lane;
        const bool have_r = pos < nrx;
        const V3 r = have_r ? ld3(rx_sorted + 3 * pos) : V3{0, 0, 0};
        const long long ir = have_r ? (long long)rx_index[pos] : 0;
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const BeamEmitD<SCALE> &d = lds[wave * 64 + l];  // wave-uniform address: broadcast reads
            const V3 dI = V3{d.I[0], d.I[1], d.I[2]};
            const float dd = fdot(r - V3{d.pc[0], d.pc[1], d.pc[2]}, V3{d.nc[0], d.nc[1], d.nc[2]});
            const bool wrong_side = (float)d.side_prev * dd < -4.0f * E;  // as beam_emit_kernel
            const V3 w = r - dI;
            const float wl = margin_len(w);  // as beam_emit_kernel
            bool inside_all = true;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j < order) {
                    bool inside_any = false;
#pragma unroll
                    for (int t = 0; t < SCALE; ++t) {
                        const float thr = -(E + E * (wl * d.inv_h[j][t]));
                        bool inside = true;
#pragma unroll
                        for (int f = 0; f < 3; ++f)
                            inside = inside && !(fdot(w, V3{d.pyr[j][t][3 * f], d.pyr[j][t][3 * f + 1], d.pyr[j][t][3 * f + 2]}) < thr);
                        inside_any = inside_any || inside;
                    }
                    inside_all = inside_all && inside_any;
                }
            }
            const bool keep = have_r && inside_all && !wrong_side;
            const unsigned long long vote = __ballot(keep);
            if (vote) {  // staged per wave, one global atomic per flush (see beam_emit_kernel)
                if (keep) {
                    const int slot = wcount + __popcll(vote & ((1ull << lane) - 1ull));
                    wbuf[wave][slot] = (unsigned long long)(((long long)d.tx * (long long)nrx + ir) * npow + d.tail);
                }
                wcount += __popcll(vote);
                if (wcount > kBeamWaveBuf - 64) {
                    beam_flush(wbuf[wave], wcount, lane, reinterpret_cast<unsigned long long *>(rows), cap, count);
                    wcount = 0;
                }
            }
        }
    }
    if (wcount > 0) beam_flush(wbuf[wave], wcount, lane, reinterpret_cast<unsigned long long *>(rows), cap, count);
}

static BeamMesh beam_mesh(drt_mesh_t m) {
    BeamMesh M;
    M.tv = m->tri_verts;
    M.normals = m->normals;
    M.mask = m->has_mask ? m->mask : nullptr;
    M.scale = m->assume_quads ? 2 : 1;
    M.nprim = m->num_triangles / M.scale;
    return M;
}

}  // namespace drt

using namespace drt;

extern "C" {

int32_t drt_beam_seed(drt_mesh_t mesh, const float *tx, int64_t ntx, float margin, drt_beam_entry *out,
                      int64_t capacity, int64_t *count_dev, void *stream) {
    DRT_REQUIRE(mesh && count_dev, "null argument");
    DRT_REQUIRE(ntx >= 0 && capacity >= 0 && margin >= 0.0f, "bad argument");
    const BeamMesh M = beam_mesh(mesh);
    const int64_t n = ntx * M.nprim;
    if (n == 0) return DRT_OK;
    DRT_REQUIRE(tx && (out || capacity == 0), "null pointer");
    hipLaunchKernelGGL(beam_seed_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), M, tx, ntx,
                       margin, reinterpret_cast<BeamEntry *>(out), capacity,
                       reinterpret_cast<unsigned long long *>(count_dev));
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_beam_expand(drt_mesh_t mesh, const drt_beam_entry *in, int64_t n_in, int32_t level, float margin,
                        int32_t use_bvh, uint64_t *out, int64_t capacity, int64_t *count_dev, void *stream) {
    DRT_REQUIRE(mesh && count_dev, "null argument");
    DRT_REQUIRE(n_in >= 0 && capacity >= 0 && margin >= 0.0f, "bad argument");
    DRT_REQUIRE(level >= 1 && level <= 2, "expansion goes from level 1 or 2 (orders up to 3)");
    const BeamMesh M = beam_mesh(mesh);
    if (n_in == 0 || M.nprim == 0) return DRT_OK;
    DRT_REQUIRE(in && (out || capacity == 0), "null pointer");
    if (use_bvh == 0) {  // default: lane = primitive, prefixes staged in LDS, wave-level sphere culling
        const int64_t bx = ceil_div(M.nprim, 256);
        int64_t by = ceil_div(4096, bx);
        const int64_t ptiles = ceil_div(n_in, 256);
        if (by > ptiles) by = ptiles;
        if (by > 65535) by = 65535;
        if (by < 1) by = 1;
        const int64_t pps = ceil_div(ptiles, by) * 256;
        by = ceil_div(n_in, pps);
        const dim3 gt((unsigned)bx, (unsigned)by);
        if (M.scale == 2)
            hipLaunchKernelGGL(beam_expand_t_kernel<2>, gt, dim3(256), 0, as_stream(stream), M,
                               reinterpret_cast<const BeamEntry *>(in), n_in, (int)level, margin,
                               reinterpret_cast<unsigned long long *>(out), capacity,
                               reinterpret_cast<unsigned long long *>(count_dev), pps);
        else
            hipLaunchKernelGGL(beam_expand_t_kernel<1>, gt, dim3(256), 0, as_stream(stream), M,
                               reinterpret_cast<const BeamEntry *>(in), n_in, (int)level, margin,
                               reinterpret_cast<unsigned long long *>(out), capacity,
                               reinterpret_cast<unsigned long long *>(count_dev), pps);
        DRT_LAUNCH_CHECK();
        return DRT_OK;
    }
    if (use_bvh == 1) {
        int32_t rc = drt_mesh_build_bvh(mesh, stream);
        if (rc != DRT_OK) return rc;
        const auto *nodes = reinterpret_cast<const BvhNode *>(mesh->bvh_nodes);
        const dim3 g1((unsigned)ceil_div(n_in, 256));
        if (M.scale == 2)
            hipLaunchKernelGGL(beam_expand_bvh_kernel<2>, g1, dim3(256), 0, as_stream(stream), M, nodes,
                               mesh->num_triangles, reinterpret_cast<const BeamEntry *>(in), n_in, (int)level, margin,
                               reinterpret_cast<unsigned long long *>(out), capacity,
                               reinterpret_cast<unsigned long long *>(count_dev));
        else
            hipLaunchKernelGGL(beam_expand_bvh_kernel<1>, g1, dim3(256), 0, as_stream(stream), M, nodes,
                               mesh->num_triangles, reinterpret_cast<const BeamEntry *>(in), n_in, (int)level, margin,
                               reinterpret_cast<unsigned long long *>(out), capacity,
                               reinterpret_cast<unsigned long long *>(count_dev));
        DRT_LAUNCH_CHECK();
        return DRT_OK;
    }
    const int64_t bx = ceil_div(n_in, 256), tiles = ceil_div(M.nprim, kBeamTile);
    int64_t by = ceil_div(2048, bx);
    if (by > tiles) by = tiles;
    if (by > 65535) by = 65535;
    if (by < 1) by = 1;
    const int64_t pps = ceil_div(tiles, by) * kBeamTile;  // whole tiles per split
    by = ceil_div(M.nprim, pps);
    const dim3 grid((unsigned)bx, (unsigned)by);
    if (M.scale == 2)
        hipLaunchKernelGGL(beam_expand_kernel<2>, grid, dim3(256), 0, as_stream(stream), M,
                           reinterpret_cast<const BeamEntry *>(in), n_in, (int)level, margin,
                           reinterpret_cast<unsigned long long *>(out), capacity,
                           reinterpret_cast<unsigned long long *>(count_dev), pps);
    else
        hipLaunchKernelGGL(beam_expand_kernel<1>, grid, dim3(256), 0, as_stream(stream), M,
                           reinterpret_cast<const BeamEntry *>(in), n_in, (int)level, margin,
                           reinterpret_cast<unsigned long long *>(out), capacity,
                           reinterpret_cast<unsigned long long *>(count_dev), pps);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_beam_finish(drt_mesh_t mesh, const drt_beam_entry *src, const uint64_t *records, int64_t n,
                        int32_t level, float margin, drt_beam_entry *out, void *stream) {
    DRT_REQUIRE(mesh, "null argument");
    DRT_REQUIRE(n >= 0 && level >= 1 && level <= 2 && margin >= 0.0f, "bad argument");
    if (n == 0) return DRT_OK;
    DRT_REQUIRE(src && records && out, "null pointer");
    hipLaunchKernelGGL(beam_finish_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream),
                       beam_mesh(mesh), reinterpret_cast<const BeamEntry *>(src),
                       reinterpret_cast<const unsigned long long *>(records), n, (int)level, margin,
                       reinterpret_cast<BeamEntry *>(out));
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_beam_emit(drt_mesh_t mesh, const drt_beam_entry *in, const uint64_t *records, int64_t n_in,
                      int32_t order, const float *rx, int64_t nrx, int64_t ntx, float margin, int64_t *rows_out,
                      int64_t capacity, int64_t *count_dev, void *stream) {
    DRT_REQUIRE(mesh && count_dev, "null argument");
    DRT_REQUIRE(n_in >= 0 && nrx >= 0 && ntx >= 0 && capacity >= 0 && margin >= 0.0f, "bad argument");
    DRT_REQUIRE(order >= 1 && order <= 3, "beam pruning covers orders 1..3");
    const BeamMesh M = beam_mesh(mesh);
    // packed rows must fit 62 bits: (ntx nrx) n^order
    unsigned __int128 total = (unsigned __int128)(ntx > 0 ? ntx : 1) * (unsigned __int128)(nrx > 0 ? nrx : 1);
    for (int j = 0; j < order; ++j) total *= (unsigned __int128)(M.nprim > 0 ? M.nprim : 1);
    DRT_REQUIRE(total < ((unsigned __int128)1 << 62), "tx * rx * primitives^order does not fit a 62-bit row key");
    if (n_in == 0 || nrx == 0) return DRT_OK;
    DRT_REQUIRE(in && rx && (rows_out || capacity == 0), "null pointer");
    DRT_REQUIRE(!records || order >= 2, "records address level order-1 prefixes: order >= 2");
    const dim3 grid((unsigned)ceil_div(n_in, 256));
    const auto *rec = reinterpret_cast<const unsigned long long *>(records);
    if (M.scale == 2)
        hipLaunchKernelGGL(beam_emit_kernel<2>, grid, dim3(256), 0, as_stream(stream), M,
                           reinterpret_cast<const BeamEntry *>(in), rec, n_in, (int)order, rx, nrx, margin,
                           reinterpret_cast<long long *>(rows_out), capacity,
                           reinterpret_cast<unsigned long long *>(count_dev));
    else
        hipLaunchKernelGGL(beam_emit_kernel<1>, grid, dim3(256), 0, as_stream(stream), M,
                           reinterpret_cast<const BeamEntry *>(in), rec, n_in, (int)order, rx, nrx, margin,
                           reinterpret_cast<long long *>(rows_out), capacity,
                           reinterpret_cast<unsigned long long *>(count_dev));
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_beam_expand_clustered(drt_mesh_t mesh, const drt_beam_entry *in, int64_t n_in, int32_t level,
                                  float margin, const int32_t *prim_order, const float *sorted_vertices,
                                  const float *cluster_boxes, int64_t num_clusters, uint64_t *out, int64_t capacity, int64_t *count_dev,
                                  void *stream) {
    DRT_REQUIRE(mesh && count_dev, "null argument");
    DRT_REQUIRE(n_in >= 0 && capacity >= 0 && margin >= 0.0f && num_clusters >= 0, "bad argument");
    DRT_REQUIRE(level >= 1 && level <= 2, "expansion goes from level 1 or 2 (orders up to 3)");
    const BeamMesh M = beam_mesh(mesh);
    if (n_in == 0 || M.nprim == 0) return DRT_OK;
    DRT_REQUIRE(in && prim_order && sorted_vertices && cluster_boxes && (out || capacity == 0), "null pointer");
    DRT_REQUIRE(num_clusters == ceil_div(M.nprim, (int64_t)64), "one cluster per 64 primitives of prim_order");
    DRT_REQUIRE(n_in < (1ll << 32), "record format holds 32-bit prefix indices");
    const int64_t bx = ceil_div(n_in, 128);
    int64_t by = ceil_div(2048, bx);  // few prefixes: split the cluster range so that the launch fills the chip
    if (by > num_clusters) by = num_clusters;
    if (by > 65535) by = 65535;
    if (by < 1) by = 1;
    const int64_t cps = ceil_div(num_clusters, by);
    by = ceil_div(num_clusters, cps);
    const dim3 grid((unsigned)bx, (unsigned)by);
    if (M.scale == 2)
        hipLaunchKernelGGL(beam_expand_clustered_kernel<2>, grid, dim3(128), 0, as_stream(stream), M,
                           reinterpret_cast<const BeamEntry *>(in), n_in, (int)level, margin,
                           reinterpret_cast<unsigned long long *>(out), capacity,
                           reinterpret_cast<unsigned long long *>(count_dev), prim_order, sorted_vertices, cluster_boxes, num_clusters, cps);
    else
        hipLaunchKernelGGL(beam_expand_clustered_kernel<1>, grid, dim3(128), 0, as_stream(stream), M,
                           reinterpret_cast<const BeamEntry *>(in), n_in, (int)level, margin,
                           reinterpret_cast<unsigned long long *>(out), capacity,
                           reinterpret_cast<unsigned long long *>(count_dev), prim_order, sorted_vertices, cluster_boxes, num_clusters, cps);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_beam_emit_clustered(drt_mesh_t mesh, const drt_beam_entry *in, const uint64_t *records, int64_t n_in,
                                int32_t order, const float *rx_sorted, const int32_t *rx_index, const float *boxes,
                                int64_t nrx, int64_t ntx, float margin, int64_t *rows_out, int64_t capacity,
                                int64_t *count_dev, void *stream) {
    DRT_REQUIRE(mesh && count_dev, "null argument");
    DRT_REQUIRE(n_in >= 0 && nrx >= 0 && ntx >= 0 && capacity >= 0 && margin >= 0.0f, "bad argument");
    DRT_REQUIRE(order >= 1 && order <= 3, "beam pruning covers orders 1..3");
    const BeamMesh M = beam_mesh(mesh);
    unsigned __int128 total = (unsigned __int128)(ntx > 0 ? ntx : 1) * (unsigned __int128)(nrx > 0 ? nrx : 1);
    for (int j = 0; j < order; ++j) total *= (unsigned __int128)(M.nprim > 0 ? M.nprim : 1);
    DRT_REQUIRE(total < ((unsigned __int128)1 << 62), "tx * rx * primitives^order does not fit a 62-bit row key");
    if (n_in == 0 || nrx == 0) return DRT_OK;
    DRT_REQUIRE(in && rx_sorted && rx_index && boxes && (rows_out || capacity == 0), "null pointer");
    DRT_REQUIRE(!records || order >= 2, "records address level order-1 prefixes: order >= 2");
    const dim3 grid((unsigned)ceil_div(n_in, 128));
    const auto *rec = reinterpret_cast<const unsigned long long *>(records);
    if (M.scale == 2)
        hipLaunchKernelGGL(beam_emit_clustered_kernel<2>, grid, dim3(128), 0, as_stream(stream), M,
                           reinterpret_cast<const BeamEntry *>(in), rec, n_in, (int)order, rx_sorted, rx_index, boxes,
                           nrx, margin, reinterpret_cast<long long *>(rows_out), capacity,
                           reinterpret_cast<unsigned long long *>(count_dev));
    else
        hipLaunchKernelGGL(beam_emit_clustered_kernel<1>, grid, dim3(128), 0, as_stream(stream), M,
                           reinterpret_cast<const BeamEntry *>(in), rec, n_in, (int)order, rx_sorted, rx_index, boxes,
                           nrx, margin, reinterpret_cast<long long *>(rows_out), capacity,
                           reinterpret_cast<unsigned long long *>(count_dev));
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

}  // extern "C"
