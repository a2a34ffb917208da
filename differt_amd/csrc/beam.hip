// beam.hip -- conservative ("beam") pruning of the exhaustive candidate space, GPU resident, behind ONE
// entry point: drt_trace_paths_beam (clustering, prefix expansion, receiver stage, row sort / decode,
// fused trace; no host-side glue between the stages).
//
// Reference context: the exhaustive tracer enumerates n (n-1)^(k-1) candidates per (tx, rx) pair
// (geometry/_solvers.py:803-848, traced by :936-957) and the hybrid tracer prunes them with SAMPLED
// visibility (_solvers.py:1013-1056), which is lossy.  This file prunes with a geometric argument.
//
// A specular path tx -> P_1 in m_1 -> ... -> P_k in m_k -> rx unfolds into straight lines through
// the images I_j of the transmitter (I_0 = tx, I_j = mirror image of I_{j-1} in the plane of m_j):
// P_{j+1} lies on the ray from I_j through P_j, i.e. inside the pyramid with apex I_j spanned by the
// primitive m_j -- and inside the pyramids over every EARLIER mirror unfolded through the later planes --
// and rx lies inside all pyramids with apex I_k.  Moreover the reference's same-side check
// (_solver_image_method.py:443-454) needs P_{j-1} and P_{j+1} on one side of the plane of m_j.  Both are
// NECESSARY conditions of a valid path, so a prefix (m_1..m_j) is discarded with all of its extensions when
//   (S) the previous point set (tx or the primitive m_{j-1}) lies strictly on one side of the plane of
//       m_j and the next primitive strictly on the other, or
//   (B) every vertex of the next primitive lies strictly outside one face plane of one of the pyramids.
//
// "Strictly" = by more than an error bound of the REFERENCE's float32 arithmetic, built per mirror from
// LOCAL quantities (DESIGN.md section 9 has the argument): the reference computes its reflection points
// backwards, P~_j = line(P~_{j+1}, I_j) /\ plane(m_j) (_solver_image_method.py:152-203) and accepts a
// candidate only if Moller-Trumbore finds each P~_j inside m_j (_solvers.py:598-642).  Each of those
// steps is a float32 ray / plane intersection whose position error is at most
//       eps_j = u * sigma_j * D_j / h_j,        u = kappa * ulp(M)
// (M = largest coordinate magnitude, sigma_j = 1 / sin of the sharpest corner of m_j, D_j = largest distance
// of a vertex of m_j from the apex I_{j-1}, h_j = distance of that apex from the plane of m_j: D_j / h_j
// bounds 1 / cos(incidence) for every ray from the apex to the primitive).  Consecutive computed points
// are tied to each other by these LOCAL errors -- the chain is never compared with an exact path -- so the
// bound of a test is a SUM S_j = eps_1 + ... + eps_j over the prefix (plus eps of the candidate being
// tested), not a product of 1 / cos over the whole path: there is no user-chosen smallest incidence
// cosine any more.  A mirror seen at grazing incidence (h_j -> 0) makes its own eps_j -- and with it every
// test it takes part in -- unbounded: the test switches itself off instead of dropping a path.
//   side tests     : previous set by S_{j-1} + 2u, candidate by eps_c + 2u, receivers by 2u
//   pyramid faces  : vertex x of candidate c is outside face f when
//                    <x - I, n_f> < -( 2 eps_c + u + g |x - I|_1 ),  g = 1.01 * 2 S / (h_P - 2 S) + 2e-6
//                    (h_P = distance of the apex from the (unfolded) mirror's plane; h_P <= 2.1 S: off)
// The tests only ever REMOVE candidates that the reference arithmetic rejects; what survives is evaluated
// by the ordinary trace kernels with the reference arithmetic, so results are those of the exhaustive tracer.
//
// Pipeline (device lists, wave-ballot compaction, no host enumeration):
//   seed      level-1 prefixes (tx, m_1) for all active primitives (optionally one shard of them)
//   expand    level-j prefixes x primitives -> 8-byte (prefix, primitive) records; cluster culling:
//             primitives in Morton clusters of 64 with boxes, lane = prefix tests the box, surviving
//             (prefix, cluster) pairs are tested with lane = primitive and the prefix broadcast
//   finish    records -> level-(j+1) prefixes (intermediate levels only)
//   emit      level-k prefixes (or level-(k-1) prefixes + records) x receivers -> packed rows
//             ((tx nrx + rx) n^k + sum_j m_j n^(k-1-j)); receivers in Morton clusters from 128 on
//   rows      radix sort, duplicate marking, decode to a per-pair table, fused compact trace
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <vector>

#include "common.hpp"
#include "geom.hpp"
#include "beam_margins.hpp"
#include "mesh.hpp"
#include "sort_safe.hpp"

#pragma clang fp contract(off)

namespace drt {
// Dot product with fused multiply-adds (3 instructions instead of 5).  The beam tests are necessary
// conditions with explicit margins, not parity arithmetic: one rounding instead of three per product-sum is
// only more accurate, and every mapping of the expansion / receiver stage uses these same functions, so
// their survivors stay identical.
__device__ __forceinline__ float fdot(V3 a, V3 b) { return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, a.z * b.z)); }
// |w|, rounded up: v_sqrt_f32 (1 ulp) nudged, instead of the ~12-instruction correctly rounded square root
__device__ __forceinline__ float margin_len(V3 w) { return __builtin_amdgcn_sqrtf(fdot(w, w)) * margins::kLenRoundUp; }
__device__ __forceinline__ float l1_len(V3 w) { return (__builtin_fabsf(w.x) + __builtin_fabsf(w.y)) + __builtin_fabsf(w.z); }

struct BeamEntry {  // 32 bytes
    uint32_t tx_side;  // bits 0..29 transmitter, bits 30..31 = side + 1 (side of the previous point set w.r.t.
                       // the plane of the last mirror: +1 / -1 / 0 = near or straddling)
    int32_t id[3];     // primitive ids of the prefix, unused slots -1
    float apex[3];     // image of the transmitter through the prefix's mirrors
    float esum;        // S = sum of the per-mirror error bounds eps_j of the prefix (+inf: every test off)
};
static_assert(sizeof(BeamEntry) == 32, "BeamEntry layout");
__device__ __forceinline__ int entry_tx(const BeamEntry &e) { return (int)(e.tx_side & 0x3fffffffu); }
__device__ __forceinline__ int entry_side(const BeamEntry &e) { return (int)(e.tx_side >> 30) - 1; }
__device__ __forceinline__ uint32_t pack_tx_side(int tx, int side) { return (uint32_t)tx | ((uint32_t)(side + 1) << 30); }

struct BeamMesh {
    const float *tv;       // [T,3,3]
    const float *normals;  // [T,3]
    const float *shape;    // [T]
    const uint8_t *mask;   // [T] or null
    int64_t nprim;
    int32_t scale;  // triangles per primitive (2 with assume_quads, or for the coplanar pairs of a triangle mesh)
    int32_t kind;   // primitive shape the kernels are instantiated for: 1, 2 or 4 (struct Shape)
    float inv_2m;   // 1 / (2 M), M = largest coordinate magnitude of mesh, transmitters and receivers
    int32_t self_loops;  // coplanar-pair mode of a TRIANGLE mesh: a pair may follow itself (its two triangles are
                         // different candidates of the triangle-level space; assume_quads has no such candidate)
    const int32_t *pair_tri;  // [nprim,2] pair mode: the real triangles of a primitive (second -1: a single triangle, which
                              // may NOT follow itself); null otherwise
};
// may primitive p follow a prefix whose last mirror is m?
__device__ __forceinline__ bool may_follow(const BeamMesh &M, int32_t p, int32_t m) {
    return p != m || (M.self_loops && M.pair_tri[2 * (int64_t)p + 1] >= 0);
}
// The error unit u = kappa ulp(M) assumes operands within 2 M (differences of scene points, images one reflection
// away).  Images of images can reach (2k+1) M, and float32 rounding grows with the operand: a prefix whose apex lies
// beyond 2 M scales its unit by |apex|_inf / (2 M).
__device__ __forceinline__ float mag_scale(const BeamMesh &M, V3 I) {
    const float m = fmaxf(__builtin_fabsf(I.x), fmaxf(__builtin_fabsf(I.y), __builtin_fabsf(I.z)));
    return fmaxf(1.0f, m * M.inv_2m);  // NaN apex: 1 (its tests are off anyway)
}

// The LAST expansion may drop a child before its record is written: a child none of whose receivers can lie inside
// its narrowest pyramid (the first mirror, unfolded through every later one) yields no row in the receiver stage.
// box = bounds of all receivers (finite ones; `on` = 0 when some receiver is not finite, or at earlier levels).
struct RxAll {
    float lo[3], hi[3];
    int32_t on;
    float ulp_m;  // ulp of the largest coordinate magnitude M (the error unit without its kappa)
    // centre and half extents of the box as box_pruned forms them (0.5 (lo + hi), (hi - lo) kBoxHalfExtent), computed ONCE
    // where the box is (rx_all_finish): kernel arguments live in scalar registers, and the child filter reads them as scalar
    // operands -- computed in the kernel they were loop-invariant VECTOR values that the compiler kept in six registers
    // across the whole expansion (three of them in scratch at the pairs kernel's 80-register budget)
    float ce[3], he[3];
};
__host__ __device__ inline void rx_all_finish(RxAll &r) {
    for (int k = 0; k < 3; ++k) {
        r.ce[k] = 0.5f * (r.lo[k] + r.hi[k]);
        r.he[k] = (r.hi[k] - r.lo[k]) * margins::kBoxHalfExtent;
    }
}
// Scene-dependent scalars of a call, for the entry point that may not read anything back
// (drt_trace_paths_beam_async): computed on the device by beam_dyn_kernel -- the same expressions the synchronous
// entry point evaluates on the host -- and list sizes that live in device counters.  A kernel given a BeamDev with
// non-null members takes u / inv_2m / the receivers' box from `dyn` and the smaller of `*n` and its size argument.
struct BeamDyn {
    float u, inv_2m;
    RxAll rxall;
};
struct BeamDev {
    const BeamDyn *dyn;
    const unsigned long long *n;
    int64_t n_off = 0;  // the launch covers entries [n_off, n_off + its size argument) of the list whose length is *n
};
__device__ __forceinline__ void beam_dev_apply(const BeamDev &dv, BeamMesh &M, float &u, int64_t &n) {
    if (dv.dyn) {
        u = dv.dyn->u;
        M.inv_2m = dv.dyn->inv_2m;
    }
    if (dv.n) {
        int64_t nd = (int64_t)*dv.n - dv.n_off;
        nd = nd < 0 ? 0 : nd;
        n = nd < n ? nd : n;
    }
}

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
    return (dmin > E) ? 1 : ((dmax < -E) ? -1 : 0);  // E = +inf or NaN: 0
}

// (n, d = <n, v0>) of a triangle and the distance of a point from that plane -- ONE expression for the
// cluster-level bound and the per-primitive value, so that the former really is the minimum of the latter
__device__ __forceinline__ float plane_offset(V3 n, V3 v0) { return fdot(n, v0); }
__device__ __forceinline__ float plane_dist(V3 I, V3 n, float d) {
    return __builtin_fabsf(__builtin_fmaf(n.x, I.x, __builtin_fmaf(n.y, I.y, __builtin_fmaf(n.z, I.z, -d))));
}

// eps = u sigma D / h: position error bound of the reference's float32 reflection point on a mirror seen
// from an apex at distance h from its plane, D = largest apex-vertex distance.  h <= u sigma (apex in the
// plane up to the arithmetic's resolution), NaN or overflow: +inf, which switches every test that uses it off.
__device__ __forceinline__ float beam_eps(float u, float sigma, float D, float h) {
    const float us = u * sigma;
    const float e = us * (D * __builtin_amdgcn_rcpf(h)) * margins::kEpsRoundUp;  // v_rcp_f32: 1 ulp, inside the round-up
    return (h > us && e < kInf) ? e : kInf;
}

// side of all vertices of primitive p w.r.t. plane (pt, n), with margin E
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

// error bound eps of primitive p as the NEXT mirror of a prefix whose apex is I (global loads: seed / finish)
__device__ __forceinline__ float prim_eps_global(const BeamMesh &M, int64_t p, V3 I, float u) {
    float D = 0.0f, h = kInf, sg = 0.0f;
    for (int t = 0; t < M.scale; ++t) {
        const int64_t f = p * M.scale + t;
        const float *v = M.tv + 9 * f;
        const V3 v0 = ld3(v), n = ld3(M.normals + 3 * f);
        h = fminf(h, plane_dist(I, n, plane_offset(n, v0)));
        sg = fmaxf(sg, M.shape[f]);
#pragma unroll
        for (int k = 0; k < 3; ++k) D = fmaxf(D, margin_len(ld3(v + 3 * k) - I));
    }
    return beam_eps(u, sg, D, h);
}

// A pyramid (apex I over a triangle): inward unit normals of its three face planes -- a degenerate face
// (apex on the edge line or in the triangle's plane) gets a zero normal and never separates -- and, per face,
// the slope g of its margin.  Seen from the apex, the reference's decisions about this mirror are uncertain by a
// LATERAL distance delta (the along-ray error of a reflection point, which incidence amplifies, does not move
// the line apex -> point): a line through the apex that passes within delta of the face's edge makes an angle
// of at most delta / (rho - delta) with the face plane, rho = distance of the apex from the EDGE LINE.
// ---- primitive shapes ------------------------------------------------------------------------------------------
// The kernels are templated on the SHAPE of a primitive (template parameter SCALE, for history):
//   1  one triangle                       3 vertices, one 3-face pyramid, one plane
//   2  two arbitrary triangles            6 vertices, two 3-face pyramids (a point is "inside" when inside either),
//      (assume_quads as the reference      two planes
//      defines it, _solvers.py:526-627)
//   4  a CONVEX PLANAR FAN QUAD           4 vertices, ONE 4-face pyramid, one plane -- both triangles lie in one plane
//      (v0 v1 v2) + (v0 v2 v3)            with equal unit normals and first vertices, share the diagonal v0-v2 and form
//                                         a convex quadrilateral (quad_shape_kernel checks every primitive; the walls and
//                                         roofs of box meshes).  The cone over the quad IS the union of the two
//                                         triangles' cones, and whenever the two 3-face tests separate a candidate an
//                                         OUTER face does (the two diagonal half-spaces are complementary): same pruning
//                                         power for 16 instead of 36 vertex-face products per pyramid.
// In the mesh arrays shapes 2 and 4 both occupy two consecutive triangles.
template <int S> struct Shape {
    static_assert(S == 1 || S == 2 || S == 4, "primitive shape");
    static constexpr int NV = (S == 1) ? 3 : (S == 2 ? 6 : 4);   // vertices of a primitive
    static constexpr int NP = (S == 2) ? 2 : 1;                  // pyramids per mirror = planes per primitive
    static constexpr int NF = (S == 4) ? 4 : 3;                  // faces of a pyramid
    static constexpr int TPP = (S == 1) ? 1 : 2;                 // triangles per primitive in the mesh arrays
};
// vertex k of a primitive from its triangles' vertex arrays tv[TPP][3][3] (shape 4: v3 = third vertex of the 2nd triangle)
template <int S>
__device__ __forceinline__ V3 shape_vertex(const float *__restrict__ tv, int k) {
    if (S == 4) return (k < 3) ? ld3(tv + 3 * k) : ld3(tv + 9 + 6);
    return ld3(tv + 3 * k);  // shapes 1 and 2: the 3 or 6 vertices in storage order
}

template <int NF>
struct PyrN {
    V3 n[NF];
    float g[NF];  // a face whose test is off (apex within the arithmetic's resolution of the edge line, degenerate
                  // face, unbounded tolerance) has n = 0 and g = 0: its value <x - I, n> + g |x - I| is 0, never below a
                  // negative threshold -- no guard compare anywhere
};
// minimum that IGNORES NaN operands (v_min3_f32 / v_max3_f32): a NaN never separates
__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
template <int NF>
__device__ __forceinline__ float min_faces(const float (&v)[NF]) {
    float m = min3f(v[0], v[1], v[2]);
    if constexpr (NF == 4) m = __builtin_fminf(m, v[3]);
    return m;
}
// `third` (and, for a quadrilateral, `third2`): the polygon's other vertices; they fix which side of the face is
// "inside".  A quadrilateral passes BOTH: a convex quad has them on the same side (their sum has that sign), and a
// triangle laid out as the degenerate quad (v0, v1, v2, v2) -- a single triangle among the coplanar pairs of a
// triangle soup, drt_mesh::pair_* -- has one of the two ON the face's own edge, where the triple product is rounding
// noise next to the other one's value.
// CHILD: the same face with the child filter's constants (csrc/beam_margins.hpp: rho rounded down further, switched off
// earlier, a larger rounding allowance in the slope) -- on the same inputs a CHILD face that is on implies the ordinary face
// is on, with the same normal and a slope at least as large.
template <bool TWO, bool CHILD = false>
__device__ __forceinline__ void pyr_face(V3 I, V3 a, V3 b, V3 third, V3 third2, float delta, bool apex_off_plane, V3 &n_out,
                                         float &g_out) {
    constexpr float kRhoDown = CHILD ? margins::kChildRhoRoundDown : margins::kRhoRoundDown;
    constexpr float kOff = CHILD ? margins::kChildFaceOffRatio : margins::kFaceOffRatio;
    constexpr float kRound = CHILD ? margins::kChildSlopeRounding : margins::kSlopeRounding;
    // v_sqrt_f32 / v_rcp_f32 (1 ulp) instead of the correctly rounded forms (~12 instructions each, five per face,
    // three faces per pyramid, up to three pyramids per prefix): these are margins, and the 2e-6 |x - I|_1 in the
    // slope covers the rounding of the normalisation and of <x - I, n_f>; rho is rounded DOWN by the 0.9999
    const V3 N = cross(a - I, b - I);
    const float len = __builtin_amdgcn_sqrtf(fdot(N, N));
    float s = fdot(third - I, N);
    if constexpr (TWO) s += fdot(third2 - I, N);
    const V3 e = b - a;
    const float el = __builtin_amdgcn_sqrtf(fdot(e, e));
    const float rho = (el > 0.0f) ? kRhoDown * len * __builtin_amdgcn_rcpf(el) : 0.0f;
    const float g = margins::kSlopeFactor * delta * __builtin_amdgcn_rcpf(rho - delta) + kRound;
    const bool on = apex_off_plane && (rho > kOff * delta) && (len > 0.0f) && (s == s) && (s != 0.0f) && is_finite(len) && (g < kInf);
    // Round 6: normal and slope are stored DIVIDED by (1 + kSpreadFactor g), kSpreadFactor >= sqrt 3: moving a point by r moves
    // the face expression <w, n> + g |w|_1 by at most r (1 + sqrt(3) g), so a threshold that must cover a positional error r
    // has to grow with the slope; scaling the face instead keeps every test a comparison with the plain threshold
    // (s v < -thr  <=>  v < -thr (1 + sqrt(3) g)) at no cost per test, and a steep face (apex near the edge line) stays usable.
    const float spread = __builtin_amdgcn_rcpf(__builtin_fmaf(margins::kSpreadFactor, g, 1.0f)) * kRhoDown;  // (rounded down; further for the child filter)
    const float inv = __builtin_amdgcn_rcpf(len) * spread;
    const float sc = on ? ((s > 0.0f) ? inv : -inv) : 0.0f;
    n_out = on ? N * sc : V3{0, 0, 0};  // (0 * inf = NaN for a huge N: the select, not the product, zeroes it)
    g_out = on ? g * spread : 0.0f;
}
// pyramid over the polygon v[0..NF): face f spans the edge v[f] -> v[f+1]; the vertex after that edge fixes "inside"
// distance of the apex from the polygon's plane (through v0, v1, v2); NaN for a degenerate polygon
__device__ __forceinline__ float apex_plane_distance(V3 I, V3 v0, V3 v1, V3 v2) {
    const V3 Np = cross(v1 - v0, v2 - v0);
    return __builtin_fabsf(fdot(I - v0, Np)) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(fdot(Np, Np)));
}
// A pyramid has an inside only while its apex is CLEARLY off the polygon's plane.  An apex within the lateral tolerance
// delta of that plane -- a transmitter placed in a wall plane up to rounding, seen at 89 degrees from the reflection
// point 2 mm away -- may lie on either side of it as far as the reference's arithmetic can tell, the cone over the
// polygon flips with the side, and the triple product that orients each face is rounding noise: the whole pyramid is
// OFF (never separates).  Round 5: found by the ROTATED stress cities (20 lost paths in 620 076 scenes, every one a
// transmitter within 0.2 ulp(M) of a wall plane 3-4 km from the origin; tests/golden/beam_cases/flat_pyramid_*.npz) --
// on axis-aligned walls the triple product of such an apex is exactly 0 and the face test `s != 0` caught it, which is
// why four rounds of box cities never saw it.  (The header of this file always said "h_P <= 2.1 S: off"; the
// implementation had kept only the distance from the edge LINES.)
template <int NF, bool CHILD = false>
__device__ __forceinline__ PyrN<NF> make_pyr(V3 I, const V3 (&v)[NF], float delta) {
    PyrN<NF> P;
    constexpr float kDown = CHILD ? margins::kChildRhoRoundDown : margins::kRhoRoundDown;
    constexpr float kOff = CHILD ? margins::kChildPlaneOffRatio : margins::kPlaneOffRatio;
    const bool apex_off_plane = apex_plane_distance(I, v[0], v[1], v[2]) * kDown > kOff * delta;  // (NaN: off)
#pragma unroll
    for (int f = 0; f < NF; ++f)
        pyr_face<NF == 4, CHILD>(I, v[f], v[(f + 1) % NF], v[(f + 2) % NF], v[(f + 3) % NF], delta, apex_off_plane, P.n[f], P.g[f]);
    return P;
}

// Everything a prefix needs to test candidates: apex, plane of its last mirror, side of the previous point
// set, and its LEVEL pyramids -- [LEVEL-1] over the last mirror, [j] over mirror j unfolded (reflected) through
// the planes of the mirrors after it: a specular path is a straight line from the last image that crosses the
// unfolded images of ALL earlier mirrors.
template <int SCALE, int LEVEL>
struct BeamCtx {
    V3 I, pm, nm;
    float u;    // kappa * ulp(M)
    int side_prev;
    PyrN<Shape<SCALE>::NF> pyr[LEVEL][Shape<SCALE>::NP];
};

template <int SCALE, int LEVEL>
__device__ __forceinline__ void ctx_clear(BeamCtx<SCALE, LEVEL> &c) {
#pragma unroll
    for (int j = 0; j < LEVEL; ++j)
#pragma unroll
        for (int t = 0; t < Shape<SCALE>::NP; ++t)
#pragma unroll
            for (int f = 0; f < Shape<SCALE>::NF; ++f) {
                c.pyr[j][t].n[f] = V3{0, 0, 0};
                c.pyr[j][t].g[f] = 0.0f;
            }
}

template <int SCALE, int LEVEL>
__device__ __forceinline__ void build_ctx(const BeamMesh &M, const BeamEntry &e, float u, bool have,
                                          BeamCtx<SCALE, LEVEL> &c) {
    using Sh = Shape<SCALE>;
    c.I = V3{e.apex[0], e.apex[1], e.apex[2]};
    u = u * mag_scale(M, c.I);
    c.u = u;
    c.pm = V3{0, 0, 0};
    c.nm = V3{0, 0, 1};
    c.side_prev = have ? entry_side(e) : 0;
    ctx_clear<SCALE, LEVEL>(c);
    if (!have) return;
    prim_plane(M, e.id[LEVEL - 1], c.pm, c.nm);
    // lateral tolerance of the prefix: u sigma per mirror (Moller-Trumbore's edge tests + the lateral rounding of
    // the reflection points), summed over the mirrors the unfolding passes; doubled (both end points of a segment)
    float lat = 0.0f;
#pragma unroll
    for (int j = 0; j < LEVEL; ++j) {
        float sg = M.shape[(int64_t)e.id[j] * Sh::TPP];
        if (Sh::TPP == 2) sg = fmaxf(sg, M.shape[(int64_t)e.id[j] * Sh::TPP + 1]);
        lat += u * (margins::kLateralSigma * sg + margins::kLateralConst);
    }
    const float delta = lat;  // sum over the mirrors of u (kLateralSigma sigma_l + kLateralConst); +inf for a degenerate mirror: every face off
#pragma unroll
    for (int j = 0; j < LEVEL; ++j) {
        const float *tv = M.tv + 9 * ((int64_t)e.id[j] * Sh::TPP);
#pragma unroll
        for (int t = 0; t < Sh::NP; ++t) {
            V3 v[Sh::NF];
#pragma unroll
            for (int k = 0; k < Sh::NF; ++k) v[k] = shape_vertex<SCALE>(tv, t * Sh::NF + k);
#pragma unroll
            for (int r = j + 1; r < LEVEL; ++r) {
                V3 pt, n;
                prim_plane(M, e.id[r], pt, n);
#pragma unroll
                for (int k = 0; k < Sh::NF; ++k) v[k] = image_of_vertex(v[k], pt, n);
            }
            c.pyr[j][t] = make_pyr<Sh::NF>(c.I, v, delta);
        }
    }
}

#if defined(DRT_LAB) && defined(BEAM_LAB_COUNT)
// [0] box tests, [1] surviving pairs, [2] pairs with >= 1 child, [3] children, [4..6] pairs by bound, [7] sub-boxes passing,
// [8 + 2 s] passes that reach stage s of the per-primitive test (0: first pyramid, 1: second, ...), [9 + 2 s] lanes still alive there
__device__ unsigned long long beam_dbg[16];
#endif
// What prim_pruned reads of a prefix, in the order it reads it: the head (apex, last mirror's plane, unit, side of the
// previous point set) first, the pyramids one at a time, last mirror first.  Two views of one BeamCtx:
//   CtxOwn    the lane's own prefix (plain expansion: lane = prefix);
//   CtxBcast  the prefix of lane `l` of the wave, broadcast lane-to-wave with v_readlane AS IT IS NEEDED: a pass of
//             the transposed stage that ends at the side test or after the first pyramid (most do: 44 % of the
//             (prefix, cluster) pairs that reach it have no child at all) fetches 11 or 27 words instead of all 43.
//             The reads are `asm volatile`: the compiler otherwise hoists every v_readlane to the top of the pass
//             (they have no side effects) and spills the scalar registers that hold them.
__device__ __forceinline__ float lane_bcast_pinned(float x, int l) {
    uint32_t r;
    asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(r) : "v"(x), "s"(l));
    return __uint_as_float(r);
}
__device__ __forceinline__ V3 lane_bcast_pinned(V3 v, int l) {
    return V3{lane_bcast_pinned(v.x, l), lane_bcast_pinned(v.y, l), lane_bcast_pinned(v.z, l)};
}
template <int SCALE, int LEVEL>
struct CtxOwn {
    static constexpr bool kUniformPrefix = false;  // lane = prefix
    const BeamCtx<SCALE, LEVEL> &c;
    __device__ __forceinline__ V3 I() const { return c.I; }
    __device__ __forceinline__ V3 pm() const { return c.pm; }
    __device__ __forceinline__ V3 nm() const { return c.nm; }
    __device__ __forceinline__ float u() const { return c.u; }
    __device__ __forceinline__ int side_prev() const { return c.side_prev; }
    __device__ __forceinline__ PyrN<Shape<SCALE>::NF> pyr(int j, int t) const { return c.pyr[j][t]; }
};
template <int SCALE, int LEVEL>
struct CtxBcast {
    static constexpr bool kUniformPrefix = true;  // one prefix per pass
    const BeamCtx<SCALE, LEVEL> &c;
    int l;  // wave-uniform
    __device__ __forceinline__ V3 I() const { return lane_bcast_pinned(c.I, l); }
    __device__ __forceinline__ V3 pm() const { return lane_bcast_pinned(c.pm, l); }
    __device__ __forceinline__ V3 nm() const { return lane_bcast_pinned(c.nm, l); }
    __device__ __forceinline__ float u() const { return lane_bcast_pinned(c.u, l); }
    __device__ __forceinline__ int side_prev() const { return (int)__float_as_uint(lane_bcast_pinned(__uint_as_float((uint32_t)c.side_prev), l)); }
    __device__ __forceinline__ PyrN<Shape<SCALE>::NF> pyr(int j, int t) const {
        PyrN<Shape<SCALE>::NF> P;
#pragma unroll
        for (int f = 0; f < Shape<SCALE>::NF; ++f) {
            P.g[f] = lane_bcast_pinned(c.pyr[j][t].g[f], l);
            P.n[f] = lane_bcast_pinned(c.pyr[j][t].n[f], l);
        }
        return P;
    }
};

// true: primitive (vertices vx, per-triangle planes pl = (n, d), shape factor sigma) cannot follow this prefix.
// The side test and the pyramids in turn, the prefix's own (last) mirror first; the wave leaves as soon as none of
// its lanes is still a candidate (same tests, same result for every lane that matters: a lane that failed one
// test is pruned whatever the others say) -- 44 % of the (prefix, cluster) pairs that pass the box tests have no
// child at all (debug counters, profiles/r03/beam.md).
// separated by the pyramids P[0..NP) of ONE mirror (apex cI): for EACH pyramid SOME face has all vertices outside, i.e. the
// smallest of its face maxima (over the vertices, of <x - I, n_f> + g |x - I|_1) is below the threshold `base`
template <int SCALE>
__device__ __forceinline__ bool pyramids_separate(const PyrN<Shape<SCALE>::NF> (&P)[Shape<SCALE>::NP], V3 cI,
                                                  const V3 (&vx)[Shape<SCALE>::NV], float base) {
    using Sh = Shape<SCALE>;
    float worst = -kInf;  // max over the pyramids of min over the faces
#pragma unroll
    for (int t = 0; t < Sh::NP; ++t) {
        float m[Sh::NF];
#pragma unroll
        for (int f = 0; f < Sh::NF; ++f) m[f] = -kInf;
#pragma unroll
        for (int k = 0; k < Sh::NV; ++k) {
            const V3 w = vx[k] - cI;
            const float wl = l1_len(w);  // |w|_1 >= |w|_2: a slightly larger margin, no square root per face
#pragma unroll
            for (int f = 0; f < Sh::NF; ++f) m[f] = fmaxf(m[f], __builtin_fmaf(P[t].g[f], wl, fdot(w, P[t].n[f])));
        }
        worst = fmaxf(worst, min_faces<Sh::NF>(m));
    }
    return worst < base;
}

// FIRST STAGE of the per-primitive test: the side test, the candidate's own error bound and the pyramid over the
// prefix's last mirror, in ONE walk over the vertices (counters, order 3 of configs[3]: 96 % of the passes of the
// transposed stage reach that pyramid -- the side test alone empties 4 % of them -- so testing for an early exit before
// it cost more than it saved, and the differences x - I were computed twice).  Returns "pruned"; `base_out` = the
// threshold of the pyramid tests of the remaining mirrors, -inf (nothing separates) for NaN geometry or a candidate seen
// at grazing incidence.
template <int SCALE, int LEVEL, class View>
__device__ __forceinline__ bool prim_stage1(const View &cv, const V3 (&vx)[Shape<SCALE>::NV],
                                            const float (&pl)[Shape<SCALE>::NP][4], float sigma, V3 &cI_out, float &base_out) {
    using Sh = Shape<SCALE>;
    const V3 cI = cv.I(), cpm = cv.pm(), cnm = cv.nm();
    const float cu = cv.u();
    const int cside = cv.side_prev();
#if defined(DRT_LAB) && defined(BEAM_LAB_FIRST_PYRAMID_0)
    constexpr int J1 = 0;  // lab: the narrowest pyramid (earliest mirror) first -- measured: 4.7e9 instead of 1.8e9 lanes survive it
#else
    constexpr int J1 = LEVEL - 1;
#endif
    PyrN<Sh::NF> P1[Sh::NP];
#pragma unroll
    for (int t = 0; t < Sh::NP; ++t) P1[t] = cv.pyr(J1, t);
    float m1[Sh::NP][Sh::NF];
#pragma unroll
    for (int t = 0; t < Sh::NP; ++t)
#pragma unroll
        for (int f = 0; f < Sh::NF; ++f) m1[t][f] = -kInf;
    // Round 6: the distances from the last mirror's plane are computed only when the side test can reject -- side_prev != 0,
    // which is uniform over the wave in the transposed mappings (the prefix of a pass is one prefix): 28 of a pass's 184
    // instructions.  (Measured and NOT kept: D as the largest L1 distance instead of the Euclidean one -- 16 instructions
    // fewer per pass, but the larger bound keeps 3-13 % more records and the step got slower, profiles/r06/beam.md.)
    float dmin = kInf, dmax = -kInf, D2 = 0.0f;
    bool nan = false;
#pragma unroll
    for (int k = 0; k < Sh::NV; ++k) {
        const V3 w = vx[k] - cI;
        const float wl = l1_len(w);  // |w|_1 >= |w|_2: a slightly larger margin, no square root per face
        nan = nan || !(wl == wl);    // NaN vertex or apex (the maxima below ignore NaNs): never prune
        D2 = fmaxf(D2, fdot(w, w));
#pragma unroll
        for (int t = 0; t < Sh::NP; ++t)
#pragma unroll
            for (int f = 0; f < Sh::NF; ++f)
                m1[t][f] = fmaxf(m1[t][f], __builtin_fmaf(P1[t].g[f], wl, fdot(w, P1[t].n[f])));
    }
    if (!View::kUniformPrefix || cside != 0) {  // (a per-lane prefix -- the plain mapping -- computes them unconditionally)
#pragma unroll
        for (int k = 0; k < Sh::NV; ++k) {
            const float d = fdot(vx[k] - cpm, cnm);
            nan = nan || !(d == d);  // NaN plane: never prune
            dmin = fminf(dmin, d);
            dmax = fmaxf(dmax, d);
        }
    }
    float h = kInf;
#pragma unroll
    for (int t = 0; t < Sh::NP; ++t) h = fminf(h, plane_dist(cI, V3{pl[t][0], pl[t][1], pl[t][2]}, pl[t][3]));
#if defined(DRT_LAB) && defined(BEAM_LAB_NO_PRIM_EPS)
    const float eps_c = 0.0f * (sigma + D2 + h);
#else
    const float eps_c = beam_eps(cu, sigma, __builtin_amdgcn_sqrtf(D2) * margins::kLenRoundUp, h);
#endif
    const float base = -(margins::kFaceEpsFactor * eps_c + margins::kFaceUnits * cu);  // -inf for a candidate seen at grazing incidence: nothing separates
    const int side_c = side_of_range(dmin, dmax, margins::kSideEpsFactor * eps_c + margins::kSideUnits * cu);
    bool pruned = !nan && (cside * side_c == -1);
    float worst = -kInf;  // max over the pyramids of min over the faces
#pragma unroll
    for (int t = 0; t < Sh::NP; ++t) worst = fmaxf(worst, min_faces<Sh::NF>(m1[t]));
    pruned = pruned || (!nan && worst < base);
    cI_out = cI;
    base_out = nan ? -kInf : base;
    return pruned;
}

// true: primitive (vertices vx, per-triangle planes pl = (n, d), shape factor sigma) cannot follow this prefix: the
// first stage, then the earlier mirrors' pyramids in turn; the wave leaves as soon as none of its lanes is still a
// candidate (same tests, same result for every lane that matters: a lane that failed one test is pruned whatever the
// others say).
template <int SCALE, int LEVEL, class View>
__device__ __forceinline__ bool prim_pruned_view(const View &cv, const V3 (&vx)[Shape<SCALE>::NV],
                                                 const float (&pl)[Shape<SCALE>::NP][4], float sigma, bool lane_on) {
    using Sh = Shape<SCALE>;
    V3 cI;
    float base;
    bool pruned = prim_stage1<SCALE, LEVEL>(cv, vx, pl, sigma, cI, base);
#pragma unroll
    for (int jj = LEVEL - 2; jj >= 0; --jj) {
#if defined(DRT_LAB) && defined(BEAM_LAB_FIRST_PYRAMID_0)
        const int j = jj + 1;
#else
        const int j = jj;
#endif
        if (!__any(lane_on && !pruned)) return true;  // nobody left in this wave: the caller keeps no lane
#if defined(DRT_LAB) && defined(BEAM_LAB_COUNT)
        {
            const unsigned long long av = __ballot(lane_on && !pruned);
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(&beam_dbg[8 + 2 * (LEVEL - 1 - jj)], 1ull);
                atomicAdd(&beam_dbg[9 + 2 * (LEVEL - 1 - jj)], (unsigned long long)__popcll(av));
            }
        }
#endif
        PyrN<Sh::NF> P[Sh::NP];
#pragma unroll
        for (int t = 0; t < Sh::NP; ++t) P[t] = cv.pyr(j, t);
        pruned = pruned || pyramids_separate<SCALE>(P, cI, vx, base);
    }
    return pruned;
}
template <int SCALE, int LEVEL>
__device__ __forceinline__ bool prim_pruned(const BeamCtx<SCALE, LEVEL> &c, const V3 (&vx)[Shape<SCALE>::NV],
                                            const float (&pl)[Shape<SCALE>::NP][4], float sigma, bool lane_on = true) {
    return prim_pruned_view<SCALE, LEVEL>(CtxOwn<SCALE, LEVEL>{c}, vx, pl, sigma, lane_on);
}

// true: NO primitive inside the box [lo, hi] whose own bound is <= eps_max can follow this prefix (the box form
// of prim_pruned: support of the box along each normal, the largest margin inside the box)
template <int SCALE, int LEVEL>
__device__ __forceinline__ bool box_pruned(const BeamCtx<SCALE, LEVEL> &c, const float (&lo)[3], const float (&hi)[3],
                                           float eps_max) {
    const V3 ce = V3{0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
    // half extents, rounded up (the centre itself is rounded)
    const V3 e = V3{(hi[0] - lo[0]) * margins::kBoxHalfExtent, (hi[1] - lo[1]) * margins::kBoxHalfExtent, (hi[2] - lo[2]) * margins::kBoxHalfExtent};
    if (!(e.x >= 0.0f) || !(e.y >= 0.0f) || !(e.z >= 0.0f)) return false;  // NaN / empty box: keep
    if (c.side_prev != 0) {
        const float dc = fdot(ce - c.pm, c.nm);
        const float r = (__builtin_fabsf(c.nm.x) * e.x + __builtin_fabsf(c.nm.y) * e.y) + __builtin_fabsf(c.nm.z) * e.z;
        const int sb = (dc == dc) ? side_of_range(dc - r, dc + r, margins::kSideEpsFactor * eps_max + (margins::kSideUnits + margins::kBoxExtraUnits) * c.u) : 0;
        if (c.side_prev * sb == -1) return true;
    }
    const V3 w = ce - c.I;
    const float wl = l1_len(w) + ((e.x + e.y) + e.z);  // largest |x - I|_1 inside the box
    // (kBoxExtraUnits: a box's centre, half extents and support are rounded on their own -- at 8e4 m from the origin the centre
    // of a 170-m box is off by half an ulp(M) = 4 mm, more than the relative round-up of the half extents -- so a box test
    // is half a unit MORE conservative than the point / vertex test it stands for: "box pruned => everything inside pruned"
    // holds under rounding, not only in exact arithmetic.  Found by round 6's soups: 2 rows of 59 024 between the receiver
    // stage's two mappings in one of 246 877 cross-checks, a scene 8.4e4 m from the origin.)
    const float base = -(margins::kFaceEpsFactor * eps_max + (margins::kFaceUnits + margins::kBoxExtraUnits) * c.u);
    bool separated = false;
#pragma unroll
    for (int j = 0; j < LEVEL; ++j) {
        bool all_t = true;
#pragma unroll
        for (int t = 0; t < Shape<SCALE>::NP; ++t) {
            const PyrN<Shape<SCALE>::NF> &P = c.pyr[j][t];
            float v[Shape<SCALE>::NF];
#pragma unroll
            for (int f = 0; f < Shape<SCALE>::NF; ++f) {
                const V3 n = P.n[f];
                const float smax = fdot(w, n) + ((__builtin_fabsf(n.x) * e.x + __builtin_fabsf(n.y) * e.y) +
                                                __builtin_fabsf(n.z) * e.z);
                v[f] = __builtin_fmaf(P.g[f], wl, smax);
            }
            all_t = all_t && (min_faces<Shape<SCALE>::NF>(v) < base);
        }
        separated = separated || all_t;
    }
    return separated;
}

// Wave-uniform reads of tables that no kernel of the call writes (the cluster boxes of a mesh): through the CONSTANT address
// space, so that they are scalar loads whatever else the kernel does.  Next to an LDS-DMA builtin in the same loop the
// compiler cannot prove them unclobbered and makes them vector loads with a full vmcnt(0) wait each -- 12 per cluster trip
// of every clustered expansion kernel, three serialized memory latencies per trip.
typedef const __attribute__((address_space(4))) float ro_float;
__device__ __forceinline__ ro_float *ro(const float *p) { return (ro_float *)p; }
typedef float ro_v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ro_v4f_t ro4(const float *p, int64_t i) {  // p 16-byte aligned, i in units of 16 bytes
    return ((const __attribute__((address_space(4))) ro_v4f_t *)p)[i];
}

// value of lane `l` (wave-uniform index) on every lane, through v_readlane: no LDS round trip, no wait
__device__ __forceinline__ float lane_bcast(float x, int l) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(x), l));
}
__device__ __forceinline__ V3 lane_bcast(V3 v, int l) { return V3{lane_bcast(v.x, l), lane_bcast(v.y, l), lane_bcast(v.z, l)}; }
template <int SCALE, int LEVEL>
__device__ __forceinline__ BeamCtx<SCALE, LEVEL> lane_bcast(const BeamCtx<SCALE, LEVEL> &c, int l) {
    BeamCtx<SCALE, LEVEL> o;
    o.I = lane_bcast(c.I, l);
    o.pm = lane_bcast(c.pm, l);
    o.nm = lane_bcast(c.nm, l);
    o.u = lane_bcast(c.u, l);  // per prefix since the magnitude rescaling
    o.side_prev = __builtin_amdgcn_readlane(c.side_prev, l);
#pragma unroll
    for (int j = 0; j < LEVEL; ++j)
#pragma unroll
        for (int t = 0; t < Shape<SCALE>::NP; ++t) {
#pragma unroll
            for (int f = 0; f < Shape<SCALE>::NF; ++f) {
                o.pyr[j][t].g[f] = lane_bcast(c.pyr[j][t].g[f], l);
                o.pyr[j][t].n[f] = lane_bcast(c.pyr[j][t].n[f], l);
            }
        }
    return o;
}

// ---------------------------------------------------------------------------------------------
// level-1 prefixes; `shard_world` > 1 keeps those with (tx * n + m) % shard_world == shard_rank (by CONTENT:
// the list is compacted with atomics and has no stable order) -- the multi-GPU split: every valid path has
// exactly one level-1 prefix
__global__ __launch_bounds__(256) void beam_seed_kernel(BeamMesh M, const float *__restrict__ tx, int64_t ntx,
                                                        float u, int64_t shard_rank, int64_t shard_world,
                                                        BeamEntry *__restrict__ out, int64_t cap,
                                                        unsigned long long *__restrict__ count, BeamDev dv) {
    {
        int64_t unused = 0;
        beam_dev_apply(dv, M, u, unused);
    }
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool in = g < ntx * M.nprim;
    const int64_t it = in ? g / M.nprim : 0, a = in ? g - it * M.nprim : 0;
    const bool keep = in && prim_active(M, a) && (shard_world <= 1 || g % shard_world == shard_rank);
    BeamEntry e{};
    if (keep) {
        V3 pt, n;
        prim_plane(M, a, pt, n);
        const V3 t = ld3(tx + 3 * it);
        const V3 I = image_of_vertex(t, pt, n);
        const float d = fdot(t - pt, n);
        e.tx_side = pack_tx_side((int)it, (d == d) ? side_of_range(d, d, margins::kSideUnits * u) : 0);  // the transmitter is exact
        e.id[0] = (int32_t)a;
        e.id[1] = e.id[2] = -1;
        e.apex[0] = I.x;
        e.apex[1] = I.y;
        e.apex[2] = I.z;
        e.esum = prim_eps_global(M, a, t, u * mag_scale(M, I));
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

constexpr int kBeamTile = 128;         // primitives per LDS tile of the plain expansion
// Threads per workgroup of the clustered expansion.  Its waves never synchronise, and a workgroup's resources are held
// until its LAST wave ends: with two waves per workgroup the faster one's slot idles while its partner finishes (the
// survivors per wave vary widely) -- one wave per workgroup: configs[3] last expansion 157.4 -> 154.0 ms, same box.
#if !(defined(DRT_LAB) && defined(BEAM_EXPAND_WG))
#undef BEAM_EXPAND_WG
#define BEAM_EXPAND_WG 64  // (a lab knob like the others: only a -DDRT_LAB build may change it)
#endif
constexpr int kExpandWG = BEAM_EXPAND_WG;
constexpr int kBeamWaveBuf = 192;      // records staged per wave before one flush (>= 128: a flush moves 64+)
#ifndef BEAM_WAVE_BUF_BIG
#define BEAM_WAVE_BUF_BIG 512
#endif
constexpr int kBeamWaveBufBig = BEAM_WAVE_BUF_BIG;  // the same for kernels that emit ~1e10 records (one atomic per ~450; 1024 measured 2 % slower: LDS occupancy)

// wave-private LDS staging buffer -> output list: ONE global atomic for `n` records (one atomic per ballot
// meant 7e9 same-address atomics at configs[3]: the L2 atomic unit, not the arithmetic, set the pace)
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

// append the lanes with `keep` to the wave's staging buffer, flushing when another full ballot may not fit
template <int CAP>
__device__ __forceinline__ void beam_stage(bool keep, unsigned long long value, unsigned long long *wb, int &wcount,
                                           int lane, unsigned long long *__restrict__ out, int64_t cap,
                                           unsigned long long *__restrict__ count) {
    const unsigned long long vote = __ballot(keep);
    if (vote) {
        if (keep) wb[wcount + __popcll(vote & ((1ull << lane) - 1ull))] = value;
        wcount += __popcll(vote);
        if (wcount > CAP - 64) {
            beam_flush(wb, wcount, lane, out, cap, count);
            wcount = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Plain expansion: lane = prefix, the block walks all primitives through LDS tiles (every pair is tested).
// The reference mapping for the clustered kernel below: identical survivors.
template <int SCALE, int LEVEL>
__global__ __launch_bounds__(256) void beam_expand_kernel(BeamMesh M, const BeamEntry *__restrict__ in, int64_t n_in,
                                                          float u, unsigned long long *__restrict__ out, int64_t cap,
                                                          unsigned long long *__restrict__ count,
                                                          int64_t prims_per_split, BeamDev dv) {
    using Sh = Shape<SCALE>;
    beam_dev_apply(dv, M, u, n_in);
    if ((int64_t)blockIdx.x * 256 >= n_in) return;  // a grid sized for the list's CAPACITY (async entry point)
    __shared__ float lds_v[kBeamTile][Sh::NV][3];
    __shared__ float lds_pl[kBeamTile][Sh::NP][4];
    __shared__ float lds_sg[kBeamTile];
    __shared__ uint8_t lds_act[kBeamTile];
    __shared__ unsigned long long wbuf[4][kBeamWaveBuf];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool have = g < n_in;
    BeamEntry e{};
    if (have) e = in[g];
    const int32_t m = have ? e.id[LEVEL - 1] : -1;
    BeamCtx<SCALE, LEVEL> ctx;
    build_ctx<SCALE, LEVEL>(M, e, u, have, ctx);
    int wcount = 0;
    // few prefixes x many primitives would leave most CUs idle with one block per 256 prefixes: blockIdx.y
    // splits the primitive range
    const int64_t prim_begin = (int64_t)blockIdx.y * prims_per_split;
    const int64_t prim_end = (prim_begin + prims_per_split < M.nprim) ? prim_begin + prims_per_split : M.nprim;
    for (int64_t base = prim_begin; base < prim_end; base += kBeamTile) {
        __syncthreads();
        for (int i = threadIdx.x; i < kBeamTile * Sh::NV; i += 256) {
            const int64_t p = base + i / Sh::NV;
            const int vtx = i % Sh::NV;
            V3 v{0, 0, 0};
            if (p < prim_end) v = shape_vertex<SCALE>(M.tv + 9 * p * Sh::TPP, vtx);
            lds_v[i / Sh::NV][vtx][0] = v.x;
            lds_v[i / Sh::NV][vtx][1] = v.y;
            lds_v[i / Sh::NV][vtx][2] = v.z;
        }
        for (int i = threadIdx.x; i < kBeamTile * Sh::NP; i += 256) {
            const int64_t p = base + i / Sh::NP;
            const int t = i % Sh::NP;
            V3 n{0, 0, 1};
            float d = 0.0f;
            if (p < prim_end) {
                n = ld3(M.normals + 3 * (p * Sh::TPP + t));
                d = plane_offset(n, ld3(M.tv + 9 * (p * Sh::TPP + t)));
            }
            lds_pl[i / Sh::NP][t][0] = n.x;
            lds_pl[i / Sh::NP][t][1] = n.y;
            lds_pl[i / Sh::NP][t][2] = n.z;
            lds_pl[i / Sh::NP][t][3] = d;
        }
        if (threadIdx.x < kBeamTile) {
            const int64_t p = base + threadIdx.x;
            const bool ok = p < prim_end;
            lds_act[threadIdx.x] = (uint8_t)(ok && prim_active(M, p));
            float sg = 1.0f;
            if (ok) {
                sg = M.shape[p * Sh::TPP];
                if (Sh::TPP == 2) sg = fmaxf(sg, M.shape[p * Sh::TPP + 1]);
            }
            lds_sg[threadIdx.x] = sg;
        }
        __syncthreads();
        const int nt = (int)((prim_end - base < kBeamTile) ? prim_end - base : kBeamTile);
        for (int j = 0; j < nt; ++j) {
            if (!lds_act[j]) continue;  // wave-uniform
            const int32_t c = (int32_t)(base + j);
            V3 vx[Sh::NV];
            float pl[Sh::NP][4];
#pragma unroll
            for (int k = 0; k < Sh::NV; ++k) vx[k] = V3{lds_v[j][k][0], lds_v[j][k][1], lds_v[j][k][2]};
#pragma unroll
            for (int t = 0; t < Sh::NP; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) pl[t][q] = lds_pl[j][t][q];
            const bool cand = have && may_follow(M, c, m);
            const bool keep = cand && !prim_pruned<SCALE, LEVEL>(ctx, vx, pl, lds_sg[j], cand);
            beam_stage<kBeamWaveBuf>(keep, ((unsigned long long)(uint32_t)g << 32) | (uint32_t)c, wbuf[wave], wcount,
                                     lane, out, cap, count);
        }
    }
    if (wcount > 0) beam_flush(wbuf[wave], wcount, lane, out, cap, count);
}

// ---------------------------------------------------------------------------------------------
// Expansion with cluster-level culling and transposed survivors: the primitives arrive sorted along a Morton
// curve in clusters of 64 (drt_mesh::beam_*) with an axis-aligned box, the planes of their triangles and their
// largest shape factor.  lane = prefix tests each cluster's box with box_pruned, using an upper bound of the
// candidates' own error bounds over the cluster (largest box distance / smallest plane distance); the surviving
// (prefix, cluster) pairs are then tested per primitive with the prefix's context broadcast lane-to-wave
// (v_readlane) and lane = primitive of the cluster.  Same survivors as the plain kernel (tested); per prefix
// the work drops from one ~150-instruction test per primitive to one box test per 64 primitives plus full-lane
// tests of the clusters its cones actually reach.
// ---------------------------------------------------------------------------------------------
struct BeamClusters {
    const int32_t *order;
    const float *verts, *planes, *uplanes, *sigma, *boxes, *subboxes;
    int64_t nclusters;
};

// What the child filter of the LAST expansion needs from the PARENT prefix besides its apex: the vertices of its FIRST mirror
// unfolded through its later mirrors -- build_ctx's own sequence of image_of_vertex calls, so the very float values the receiver
// stage (build_ctx_from) reaches by the same calls -- and the sum of its mirrors' shape factors.
template <int SCALE, int LEVEL>
__device__ __forceinline__ void first_mirror_unfolded(const BeamMesh &M, const BeamEntry &e, bool have,
                                                      V3 (&v)[Shape<SCALE>::NP][Shape<SCALE>::NF], float &sig_sum) {
    using Sh = Shape<SCALE>;
#pragma unroll
    for (int t = 0; t < Sh::NP; ++t)
#pragma unroll
        for (int k = 0; k < Sh::NF; ++k) v[t][k] = V3{0, 0, 0};
    sig_sum = kInf;  // (a lane without a prefix: every face of its "child" is off)
    if (!have) return;
    sig_sum = 0.0f;
#pragma unroll
    for (int j = 0; j < LEVEL; ++j) {
        float sg = M.shape[(int64_t)e.id[j] * Sh::TPP];
        if (Sh::TPP == 2) sg = fmaxf(sg, M.shape[(int64_t)e.id[j] * Sh::TPP + 1]);
        sig_sum += sg;
    }
    const float *tv = M.tv + 9 * ((int64_t)e.id[0] * Sh::TPP);
#pragma unroll
    for (int t = 0; t < Sh::NP; ++t) {
#pragma unroll
        for (int k = 0; k < Sh::NF; ++k) v[t][k] = shape_vertex<SCALE>(tv, t * Sh::NF + k);
#pragma unroll
        for (int r = 1; r < LEVEL; ++r) {
            V3 pt, n;
            prim_plane(M, e.id[r], pt, n);
#pragma unroll
            for (int k = 0; k < Sh::NF; ++k) v[t][k] = image_of_vertex(v[t][k], pt, n);
        }
    }
}
// true: NO point of the receivers' box can lie inside the narrowest pyramid of the child (parent prefix + the new mirror c
// with plane point pc, normal nc and shape factor sig_c) -- the receiver stage would reject every receiver for that pyramid
// alone.  Round 6: the child's pyramid is built HERE exactly as the receiver stage builds it -- apex = image_of_vertex(parent's
// apex), polygon = the parent's unfolded first mirror reflected once more with the same function, make_pyr on those values --
// so both stages hold THE SAME face normals and edge distances bit for bit, and "what this filter drops, the receiver stage
// drops" is monotonicity in the constants: a lateral tolerance rounded up (kChildDeltaRoundUp: the two stages sum the shape
// factors in different orders), faces and the pyramid switched off earlier, a larger rounding allowance in the slope, the
// box's support instead of a receiver's own value, and a threshold kChildFaceUnits > kFaceUnits.  (Until round 5 the filter
// reflected the parent's face NORMALS instead: a different route to the same planes, whose rounding difference the triangle
// soups of round 6 caught dropping 1-2 children per 5 000 scenes that the receiver stage kept.)
template <int SCALE>
__device__ __forceinline__ bool child_misses_receivers(const RxAll &rx, const BeamMesh &M, float u0, V3 I2,
                                                       const V3 (&vpar)[Shape<SCALE>::NP][Shape<SCALE>::NF], float sig_parent,
                                                       V3 pc, V3 nc, float sig_c, int mirrors) {
    using Sh = Shape<SCALE>;
    const V3 I3 = image_of_vertex(I2, pc, nc);  // (beam_child_from's expression)
    const float uc = u0 * mag_scale(M, I3);
    // (the receiver stage sums u (kLateralSigma sigma_l + kLateralConst) mirror by mirror; here the parent's sigmas arrive as one
    // sum and `mirrors` counts them, the new one included: rounded up past any order of summation)
    const float delta = uc * (margins::kLateralSigma * (sig_parent + sig_c) + margins::kLateralConst * (float)mirrors) * margins::kChildDeltaRoundUp;  // +inf for a degenerate mirror: every face off
    const V3 ce = V3{rx.ce[0], rx.ce[1], rx.ce[2]}, he = V3{rx.he[0], rx.he[1], rx.he[2]};  // (rx_all_finish)
    if (!(he.x >= 0.0f) || !(he.y >= 0.0f) || !(he.z >= 0.0f)) return false;
    const V3 w = ce - I3;
    const float wl = l1_len(w) + ((he.x + he.y) + he.z);
    const float thr = -margins::kChildFaceUnits * uc;
    bool all_t = true;
#pragma unroll
    for (int t = 0; t < Sh::NP; ++t) {
        V3 v[Sh::NF];
#pragma unroll
        for (int k = 0; k < Sh::NF; ++k) v[k] = image_of_vertex(vpar[t][k], pc, nc);
        // make_pyr<NF, true>, face by face (each face's value as soon as its normal exists: nothing of the pyramid stays live)
        const bool apex_off_plane = apex_plane_distance(I3, v[0], v[1], v[2]) * margins::kChildRhoRoundDown > margins::kChildPlaneOffRatio * delta;  // (NaN: off)
        float val[Sh::NF];
#pragma unroll
        for (int f = 0; f < Sh::NF; ++f) {
            V3 n;
            float g;
            pyr_face<Sh::NF == 4, true>(I3, v[f], v[(f + 1) % Sh::NF], v[(f + 2) % Sh::NF], v[(f + 3) % Sh::NF], delta, apex_off_plane, n, g);
            const float smax = fdot(w, n) + ((__builtin_fabsf(n.x) * he.x + __builtin_fabsf(n.y) * he.y) + __builtin_fabsf(n.z) * he.z);
            val[f] = __builtin_fmaf(g, wl, smax);  // (a face that is off: n = 0, g = 0 -> 0, never below the threshold)
        }
        all_t = all_t && (min_faces<Sh::NF>(val) < thr);
    }
    return all_t;
}

#ifndef BEAM_EXPAND_WAVES
#define BEAM_EXPAND_WAVES 0
#endif
#if BEAM_EXPAND_WAVES > 0
#define BEAM_EXPAND_OCC __attribute__((amdgpu_waves_per_eu(BEAM_EXPAND_WAVES, BEAM_EXPAND_WAVES)))
#else
#define BEAM_EXPAND_OCC
#endif
#ifndef BEAM_Q4_WAVES
#define BEAM_Q4_WAVES 4  // 128 VGPRs, no scratch: 335 against 351 ms per two steps of configs[3] (profiles/r04/beam.md)
#endif
#if BEAM_Q4_WAVES > 0
#define BEAM_Q4_OCC __attribute__((amdgpu_waves_per_eu(BEAM_Q4_WAVES, BEAM_Q4_WAVES)))
#else
#define BEAM_Q4_OCC
#endif
template <int SCALE, int LEVEL, bool FILTER>
__device__ __forceinline__ void expand_clustered_body(
    const BeamMesh &M, const BeamClusters &C, const BeamEntry *__restrict__ in, int64_t n_in, float u,
    unsigned long long *__restrict__ out, int64_t cap, unsigned long long *__restrict__ count,
    int64_t clusters_per_split, const RxAll &rxall) {
    // 8 KiB per wave: a flush every ~960 records (with 192 the flush atomics -- all on ONE address -- were half
    // of the kernel's time at configs[3])
    __shared__ unsigned long long wbuf[kExpandWG / 64][kBeamWaveBufBig];
    // the cluster's triangle planes, staged per wave: lane k brings plane k with one coalesced load (in flight one
    // cluster ahead), the 64 prefixes of the wave then read them back as LDS broadcasts.  As 64 scalar loads per
    // (wave, cluster) this loop was half of the kernel's time (profiles/r03/beam.md): 157 KiB of planes per wave
    // do not live in the 16-KiB scalar cache
    using Sh = Shape<SCALE>;
    __shared__ __attribute__((aligned(16))) float4 lds_planes[kExpandWG / 64][64 * Sh::NP];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if ((int64_t)blockIdx.x * kExpandWG >= n_in) return;  // a grid sized for the list's CAPACITY (async entry point)
    const int64_t g = (int64_t)blockIdx.x * kExpandWG + threadIdx.x;
    const bool have = g < n_in;
    BeamEntry e{};
    if (have) e = in[g];
    const int32_t m = have ? e.id[LEVEL - 1] : -1;
    BeamCtx<SCALE, LEVEL> ctx;
    build_ctx<SCALE, LEVEL>(M, e, u, have, ctx);
    // last expansion: children are tested against the receivers' box before their records are written -- not in
    // the transposed loop (one test per pass would cost as much as the pass), but on FULL waves of kept children:
    // kept lanes park their record in a wave-private LDS queue, and whenever 64 wait, lane = parked child gathers its
    // parent's apex / narrowest pyramid from the parent's lane (ds_bpermute) and decides.  At configs[3] three quarters of
    // the children go no further.
    // Level 2 (order 3) parks EARLIER: after the first stage of the per-primitive test.  The pyramid over the prefix's
    // first mirror is the very pyramid the child filter starts from, and in the transposed loop it ran on 12 of 64 lanes
    // (counters, configs[3]: 1.55e8 of 2.6e8 passes reach it with 1.8e9 lanes alive): lane = parked candidate re-reads
    // its vertices / plane from the sorted arrays (by position), runs that pyramid with its own threshold and, if it is
    // still a candidate, the receiver-box test -- both on full waves, the parent's data gathered once for both.
    constexpr bool kTwoStage = FILTER && LEVEL == 2;
    __shared__ unsigned long long raw_rec[kExpandWG / 64][128];
    __shared__ float raw_f[kExpandWG / 64][2][128];  // (sorted position of the candidate, threshold of the two-stage form)
    constexpr bool filter_on = FILTER;
    // what the child filter needs of the PARENT besides its apex (its first mirror's vertices unfolded through its later
    // mirrors, and the sum of the shape factors: first_mirror_unfolded): read once per 64 parked children, by the child's
    // lane from the parent's slot -- kept in LDS, not in registers that would be live across the whole cluster loop
    __shared__ float lds_par[kExpandWG / 64][Sh::NP * Sh::NF * 3 + 1][64];
    if (filter_on) {
        V3 vpar[Sh::NP][Sh::NF];
        float sig_sum = kInf;
        first_mirror_unfolded<SCALE, LEVEL>(M, e, have, vpar, sig_sum);
#pragma unroll
        for (int t = 0; t < Sh::NP; ++t)
#pragma unroll
            for (int k = 0; k < Sh::NF; ++k) {
                lds_par[wave][(t * Sh::NF + k) * 3 + 0][lane] = vpar[t][k].x;
                lds_par[wave][(t * Sh::NF + k) * 3 + 1][lane] = vpar[t][k].y;
                lds_par[wave][(t * Sh::NF + k) * 3 + 2][lane] = vpar[t][k].z;
            }
        lds_par[wave][Sh::NP * Sh::NF * 3][lane] = sig_sum;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    int rawcount = 0;
    const int64_t cl_begin = (int64_t)blockIdx.y * clusters_per_split;
    const int64_t cl_end = (cl_begin + clusters_per_split < C.nclusters) ? cl_begin + clusters_per_split : C.nclusters;
    const unsigned long long gbase = (unsigned long long)((int64_t)blockIdx.x * kExpandWG + wave * 64);
    int wcount = 0;
    // the last n parked children (n <= 64): lane = child
    auto filter_parked = [&](int n) {
        const int j = rawcount - n + lane;
        bool mine = lane < n;
        const unsigned long long rec = mine ? raw_rec[wave][j] : 0ull;
        const int l = (int)((rec >> 32) - gbase) & 63;  // the parent's lane
        const V3 I2 = V3{__shfl(ctx.I.x, l, 64), __shfl(ctx.I.y, l, 64), __shfl(ctx.I.z, l, 64)};
        const float sp = lds_par[wave][Sh::NP * Sh::NF * 3][l];
        V3 vpar[Sh::NP][Sh::NF];
#pragma unroll
        for (int t = 0; t < Sh::NP; ++t)
#pragma unroll
            for (int k = 0; k < Sh::NF; ++k)
                vpar[t][k] = V3{lds_par[wave][(t * Sh::NF + k) * 3 + 0][l], lds_par[wave][(t * Sh::NF + k) * 3 + 1][l],
                                lds_par[wave][(t * Sh::NF + k) * 3 + 2][l]};
        // the candidate's mirror (first vertex, normal) and shape factor, from the sorted arrays by position
        const int64_t pos = mine ? (int64_t)__float_as_uint(raw_f[wave][0][j]) : 0;  // (position 0 always exists)
        const V3 pc = ld3(C.verts + 3 * (pos * Sh::NV));
        const float4 q = reinterpret_cast<const float4 *>(C.planes)[pos * Sh::NP];
        const V3 nc = V3{q.x, q.y, q.z};
        const float sgc = C.sigma[pos];
        if constexpr (kTwoStage) {
            const float base = raw_f[wave][1][j & 127];
            V3 vq[Sh::NV];
#pragma unroll
            for (int k = 0; k < Sh::NV; ++k) vq[k] = ld3(C.verts + 3 * (pos * Sh::NV + k));
            PyrN<Sh::NF> P0[Sh::NP];
#pragma unroll
            for (int t = 0; t < Sh::NP; ++t)
#pragma unroll
                for (int f = 0; f < Sh::NF; ++f) {
                    P0[t].n[f] = V3{__shfl(ctx.pyr[0][t].n[f].x, l, 64), __shfl(ctx.pyr[0][t].n[f].y, l, 64),
                                    __shfl(ctx.pyr[0][t].n[f].z, l, 64)};
                    P0[t].g[f] = __shfl(ctx.pyr[0][t].g[f], l, 64);
                }
            mine = mine && !pyramids_separate<SCALE>(P0, I2, vq, base);
        }
        // (rxall.on == 0 -- a non-finite receiver, known only on the device in the async entry point: every child passes)
        const bool pass = mine && !(rxall.on && child_misses_receivers<SCALE>(rxall, M, u, I2, vpar, sp, pc, nc, sgc, LEVEL + 1));
        rawcount -= n;
        beam_stage<kBeamWaveBufBig>(pass, rec, wbuf[wave], wcount, lane, out, cap, count);
    };
    // The cluster's primitive ids, vertices and planes (sorted copy, no indirection; fetched whether or not the cluster
    // will be hit: the mesh lives in L2).  kAhead: one cluster AHEAD, in flight while this cluster is tested.  The
    // order-3 quad kernel instead loads the CURRENT cluster's right after the plane-distance loop: the ~350 VALU
    // instructions of the box stage hide the L2 latency, and the 17 registers of the look-ahead copy were the ones the
    // allocator spilled (16 VGPRs, 60 B of scratch per lane in round 4).
    constexpr bool kAhead = !(SCALE == 4 && LEVEL >= 2);
    int32_t p_ld = -1;
    V3 vx_ld[Sh::NV];
    float pl_ld[Sh::NP][4];  // (plain floats: float4 arrays captured by the lambda stayed in scratch)
    auto fetch = [&](int64_t c) {
        const int64_t cc = (c < cl_end) ? c : cl_end - 1;  // the last trip re-reads its own cluster
        const int64_t pos = cc * 64 + lane;
        p_ld = (pos < M.nprim) ? C.order[pos] : -1;
#pragma unroll
        for (int k = 0; k < Sh::NV; ++k) vx_ld[k] = ld3(C.verts + 3 * (pos * Sh::NV + k));  // padded to whole clusters
#pragma unroll
        for (int t = 0; t < Sh::NP; ++t) {
            const float4 a = reinterpret_cast<const float4 *>(C.planes)[pos * Sh::NP + t];
            pl_ld[t][0] = a.x; pl_ld[t][1] = a.y; pl_ld[t][2] = a.z; pl_ld[t][3] = a.w;
        }
    };
    // the cluster's distinct planes go global -> LDS directly (global_load_lds_dwordx4: 1 KiB per wave instruction, LDS
    // address = wave base + lane * 16, no staging registers): issued as soon as the previous cluster's planes have been
    // read, landed by the time the next trip needs them
    auto fetch_planes = [&](int64_t c) {
        const int64_t cc = (c < cl_end) ? c : cl_end - 1;
#pragma unroll
        for (int t = 0; t < Sh::NP; ++t)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(reinterpret_cast<const float4 *>(C.uplanes) +
                                                                  (cc * 64 * Sh::NP + t * 64 + lane)),
                (__attribute__((address_space(3))) void *)(&lds_planes[wave][t * 64]), 16, 0, 0);
    };
    if (cl_begin < cl_end) {
        if (kAhead) fetch(cl_begin);
        fetch_planes(cl_begin);
    }
    for (int64_t cl = cl_begin; cl < cl_end; ++cl) {
        int32_t p = -1;
        V3 vx[Sh::NV];
        float pl[Sh::NP][4];
        if (kAhead) {
            p = p_ld;
#pragma unroll
            for (int k = 0; k < Sh::NV; ++k) vx[k] = vx_ld[k];
#pragma unroll
            for (int t = 0; t < Sh::NP; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) pl[t][q] = pl_ld[t][q];
        }
        // ---- lane = prefix: box of the cluster (wave-uniform scalar loads) ----
        const ro_v4f_t bx0 = ro4(C.boxes, 2 * cl), bx1 = ro4(C.boxes, 2 * cl + 1);  // (lo, hi.x | hi.yz, sigma, planes)
        const float bx[8] = {bx0.x, bx0.y, bx0.z, bx0.w, bx1.x, bx1.y, bx1.z, bx1.w};
        const float lo[3] = {bx[0], bx[1], bx[2]}, hi[3] = {bx[3], bx[4], bx[5]};
        // bound of the candidates' own eps over the cluster: smallest plane distance of the apex over the
        // cluster's triangles (the SAME expression the per-primitive test evaluates), farthest box corner
        float hmin = kInf;
#if !(defined(DRT_LAB) && defined(BEAM_LAB_NO_HMIN))
        // this cluster's planes (global_load_lds, issued a trip ago) have landed: every vector-memory operation in flight
        // here is at least that old (this trip's loads are issued below, after the loop that reads the planes)
        __builtin_amdgcn_s_waitcnt(/*vmcnt 0; expcnt, lgkmcnt: no wait*/ 0x0f70);
        __builtin_amdgcn_wave_barrier();
        // non-negative floats order like their bit patterns, NaN patterns lie above +inf: an unsigned integer minimum
        // is fminf here, without the NaN-quieting and unordered-compare code the float form expands to (measured:
        // that expansion, a branch per two planes, made this loop half of the kernel)
        uint32_t hbits = 0x7f800000u;
        const int nplanes = __builtin_amdgcn_readfirstlane((int)bx[7]);  // distinct planes of the cluster, first in the list
        // four planes per trip to the LDS (one wait for four reads; one at a time, each plane cost a whole LDS latency for
        // five instructions of arithmetic); the minimum does not depend on the order
        auto plane = [&](const float4 q) {  // (same address on every lane: broadcast)
            float sd = __builtin_fmaf(q.x, ctx.I.x, __builtin_fmaf(q.y, ctx.I.y, __builtin_fmaf(q.z, ctx.I.z, -q.w)));
            asm volatile("" : "+v"(sd));  // keeps the compiler from pairing planes into v_pk_fma_f32 + v_mov shuffles
            const uint32_t b = __float_as_uint(sd) & 0x7fffffffu;
            hbits = (b < hbits) ? b : hbits;
        };
        int k = 0;
        for (; k + 4 <= nplanes; k += 4) {
            const float4 q0 = lds_planes[wave][k], q1 = lds_planes[wave][k + 1], q2 = lds_planes[wave][k + 2],
                         q3 = lds_planes[wave][k + 3];
            plane(q0);
            plane(q1);
            plane(q2);
            plane(q3);
        }
        for (; k < nplanes; ++k) plane(lds_planes[wave][k]);
        hmin = __uint_as_float(hbits);
        __builtin_amdgcn_wave_barrier();  // reads done before the next cluster's planes arrive
        fetch_planes(cl + 1);
#endif
        fetch(kAhead ? cl + 1 : cl);
        if (!kAhead) {
            p = p_ld;
#pragma unroll
            for (int k = 0; k < Sh::NV; ++k) vx[k] = vx_ld[k];
#pragma unroll
            for (int t = 0; t < Sh::NP; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) pl[t][q] = pl_ld[t][q];
        }
        const V3 far = V3{fmaxf(__builtin_fabsf(ctx.I.x - lo[0]), __builtin_fabsf(ctx.I.x - hi[0])),
                          fmaxf(__builtin_fabsf(ctx.I.y - lo[1]), __builtin_fabsf(ctx.I.y - hi[1])),
                          fmaxf(__builtin_fabsf(ctx.I.z - lo[2]), __builtin_fabsf(ctx.I.z - hi[2]))};
#if defined(DRT_LAB) && defined(BEAM_LAB_NO_HMIN)
        const float eps_max = 0.0f;
#else
        const float eps_max = beam_eps(ctx.u, bx[6], margin_len(far) * margins::kBoxFarRoundUp, hmin);  // ctx.u: the prefix's own unit
#endif
        bool alive = have && !box_pruned<SCALE, LEVEL>(ctx, lo, hi, eps_max);
        // second level for the survivors: the four sub-boxes of 16 consecutive primitives (one or two buildings of a
        // city) -- a cluster's box is mostly streets, and half of the (prefix, cluster) pairs that passed it had no
        // child at all (debug counters, profiles/r03/beam.md).  The same test with the same (cluster-wide) bound.
        if (__any(alive)) {
            ro_v4f_t sq4[6];  // all four sub-boxes in one trip to memory (loaded one by one each costs its own latency)
#pragma unroll
            for (int k = 0; k < 6; ++k) sq4[k] = ro4(C.subboxes, 6 * cl + k);
            const float sb[24] = {sq4[0].x, sq4[0].y, sq4[0].z, sq4[0].w, sq4[1].x, sq4[1].y, sq4[1].z, sq4[1].w,
                                  sq4[2].x, sq4[2].y, sq4[2].z, sq4[2].w, sq4[3].x, sq4[3].y, sq4[3].z, sq4[3].w,
                                  sq4[4].x, sq4[4].y, sq4[4].z, sq4[4].w, sq4[5].x, sq4[5].y, sq4[5].z, sq4[5].w};
            bool sub = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float slo[3] = {sb[6 * q], sb[6 * q + 1], sb[6 * q + 2]}, shi[3] = {sb[6 * q + 3], sb[6 * q + 4], sb[6 * q + 5]};
#if defined(DRT_LAB) && defined(BEAM_LAB_COUNT)
                const bool sq = !box_pruned<SCALE, LEVEL>(ctx, slo, shi, eps_max);
                sub = sub || sq;
                const unsigned long long sv = __ballot(alive && sq);
                if (lane == 0) atomicAdd(&beam_dbg[7], (unsigned long long)__popcll(sv));
#else
                sub = sub || !box_pruned<SCALE, LEVEL>(ctx, slo, shi, eps_max);
#endif
            }
            alive = alive && sub;
        }
        unsigned long long todo = __ballot(alive);
#if defined(DRT_LAB) && defined(BEAM_LAB_COUNT)
        if (lane == 0) {
            atomicAdd(&beam_dbg[0], (unsigned long long)__popcll(__ballot(have)));
            atomicAdd(&beam_dbg[1], (unsigned long long)__popcll(todo));
        }
        {
            const unsigned long long einf = __ballot(alive && !(eps_max < kInf)), ebig = __ballot(alive && eps_max > 1.0f && eps_max < kInf),
                                     emid = __ballot(alive && eps_max > 0.1f && eps_max <= 1.0f);
            if (lane == 0) {
                atomicAdd(&beam_dbg[4], (unsigned long long)__popcll(einf));
                atomicAdd(&beam_dbg[5], (unsigned long long)__popcll(ebig));
                atomicAdd(&beam_dbg[6], (unsigned long long)__popcll(emid));
            }
        }
#endif
        if (todo == 0) continue;
        // ---- transposed: lane = primitive of the cluster ----
        const int64_t pos = cl * 64 + lane;
        const bool act = p >= 0 && prim_active(M, p);
        const bool self_ok = p >= 0 && may_follow(M, p, p);
        const float sg = C.sigma[pos];
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const bool cand = act && (p != __builtin_amdgcn_readlane(m, l) || self_ok);
            bool keep;
            float base1 = 0.0f;
            if constexpr (kTwoStage) {
                V3 unused;
                keep = cand && !prim_stage1<SCALE, LEVEL>(CtxBcast<SCALE, LEVEL>{ctx, l}, vx, pl, sg, unused, base1);
            } else {
                keep = cand && !prim_pruned_view<SCALE, LEVEL>(CtxBcast<SCALE, LEVEL>{ctx, l}, vx, pl, sg, cand);
            }
#if defined(DRT_LAB) && defined(BEAM_LAB_COUNT)
            {
                const unsigned long long kv = __ballot(keep);
                if (lane == 0) {
                    if (kv) atomicAdd(&beam_dbg[2], 1ull);
                    atomicAdd(&beam_dbg[3], (unsigned long long)__popcll(kv));
                }
            }
#endif
            const unsigned long long record = ((gbase + (unsigned long long)l) << 32) | (uint32_t)p;
            if constexpr (filter_on) {
                const unsigned long long vote = __ballot(keep);
                if (vote) {
                    if (keep) {
                        const int j = rawcount + __popcll(vote & ((1ull << lane) - 1ull));
                        raw_rec[wave][j] = record;
                        raw_f[wave][0][j] = __uint_as_float((uint32_t)pos);  // cl * 64 + lane < 2^31 primitives
                        if constexpr (kTwoStage) raw_f[wave][1][j] = base1;
                    }
                    rawcount += __popcll(vote);
                    if (rawcount >= 64) filter_parked(64);
                }
            } else {
                beam_stage<kBeamWaveBufBig>(keep, record, wbuf[wave], wcount, lane, out, cap, count);
            }
        }
    }
    if (filter_on && rawcount > 0) filter_parked(rawcount);
    if (wcount > 0) beam_flush(wbuf[wave], wcount, lane, out, cap, count);
}
template <int SCALE, int LEVEL>
__global__ __launch_bounds__(kExpandWG) BEAM_EXPAND_OCC void beam_expand_clustered_kernel(
    BeamMesh M, BeamClusters C, const BeamEntry *__restrict__ in, int64_t n_in, float u,
    unsigned long long *__restrict__ out, int64_t cap, unsigned long long *__restrict__ count,
    int64_t clusters_per_split, BeamDev dv) {
    beam_dev_apply(dv, M, u, n_in);
    expand_clustered_body<SCALE, LEVEL, false>(M, C, in, n_in, u, out, cap, count, clusters_per_split, RxAll{});
}
// the LAST expansion, with the receiver-box child filter.  One triangle per primitive: 130 VGPRs without the
// occupancy attribute, 128 and no scratch with it (4 waves per SIMD: 0.69 against 0.76 s at configs[3]); quads keep
// the compiler's own choice (the attribute would spill there).
template <int LEVEL>
__global__ __launch_bounds__(kExpandWG) __attribute__((amdgpu_waves_per_eu(4, 4))) void beam_expand_clustered_last_kernel_s1(
    BeamMesh M, BeamClusters C, const BeamEntry *__restrict__ in, int64_t n_in, float u,
    unsigned long long *__restrict__ out, int64_t cap, unsigned long long *__restrict__ count,
    int64_t clusters_per_split, RxAll rxall, BeamDev dv) {
    beam_dev_apply(dv, M, u, n_in);
    if (dv.dyn) rxall = dv.dyn->rxall;
    expand_clustered_body<1, LEVEL, true>(M, C, in, n_in, u, out, cap, count, clusters_per_split, rxall);
}
#ifndef BEAM_S2_WAVES
#define BEAM_S2_WAVES 0  // (3 would spill 48 / 132 B per lane; the general two-triangle form is the fallback for quads that are not
                         // convex planar fans -- those take the q4 kernels at 4 waves per SIMD)
#endif
#if BEAM_S2_WAVES > 0
#define BEAM_S2_OCC __attribute__((amdgpu_waves_per_eu(BEAM_S2_WAVES, BEAM_S2_WAVES)))
#else
#define BEAM_S2_OCC
#endif
template <int LEVEL>
__global__ __launch_bounds__(kExpandWG) BEAM_S2_OCC void beam_expand_clustered_last_kernel_s2(
    BeamMesh M, BeamClusters C, const BeamEntry *__restrict__ in, int64_t n_in, float u,
    unsigned long long *__restrict__ out, int64_t cap, unsigned long long *__restrict__ count,
    int64_t clusters_per_split, RxAll rxall, BeamDev dv) {
    beam_dev_apply(dv, M, u, n_in);
    if (dv.dyn) rxall = dv.dyn->rxall;
    expand_clustered_body<2, LEVEL, true>(M, C, in, n_in, u, out, cap, count, clusters_per_split, rxall);
}
// convex planar fan quads (shape 4): one 4-face pyramid and four vertices per primitive
template <int LEVEL>
__global__ __launch_bounds__(kExpandWG) BEAM_Q4_OCC void beam_expand_clustered_last_kernel_q4(
    BeamMesh M, BeamClusters C, const BeamEntry *__restrict__ in, int64_t n_in, float u,
    unsigned long long *__restrict__ out, int64_t cap, unsigned long long *__restrict__ count,
    int64_t clusters_per_split, RxAll rxall, BeamDev dv) {
    beam_dev_apply(dv, M, u, n_in);
    if (dv.dyn) rxall = dv.dyn->rxall;
    expand_clustered_body<4, LEVEL, true>(M, C, in, n_in, u, out, cap, count, clusters_per_split, rxall);
}

// ---------------------------------------------------------------------------------------------
// The last expansion of orders 2 and 3 (levels 1 and 2) in TWO kernels (round 5; the numbers below: order 3).  The fused kernel above keeps the 64 prefix contexts of a wave in
// 43 VGPRs per lane for its box stage (lane = prefix) and reads them back lane-to-wave with 27-43 v_readlane per pass of
// its transposed stage (lane = primitive): 128 VGPRs, 4 waves per SIMD, 0.55 of the VALU issue ceiling with 43 % of the
// wave-cycles waiting.  The two stages want different register files, so they are two launches:
//   beam_boxes_kernel         lane = prefix: builds each prefix's context ONCE, writes it to a table in global memory
//                             (kCtxWords floats per prefix, L2-resident for the wave that reads it back), runs the box
//                             stage over every cluster and writes the 64-bit "which prefixes reach this cluster" mask
//                             per (wave of prefixes, cluster);
//   beam_expand_pairs_kernel  lane = primitive: walks the clusters whose mask is not empty; the prefix of a pass comes
//                             through the SCALAR unit (s_load of its table entry: no v_readlane, no VALU port), the
//                             cluster's vertices are loaded only for clusters that are hit; first stage of the
//                             per-primitive test, survivors parked, second pyramid + receiver-box filter on full waves
//                             (the parent's entry read per lane from the table) -- about 60 VGPRs, 8 waves per SIMD.
// Same tests, same functions, same records as the fused kernel (DRT_BEAM_EXPAND_FUSED keeps that one: cross-check).
// ---------------------------------------------------------------------------------------------
template <int SCALE, int LEVEL>
struct CtxTab {
    using Sh = Shape<SCALE>;
    // float offsets of one entry; every group starts on a 16-byte boundary
    static constexpr int kI = 0, kU = 3, kPm = 4, kSide = 7, kNm = 8, kSig = 11, kM = 12;  // head: 16 floats
    static constexpr int kPyr = 16;                                    // [LEVEL][NP][NF] x (n.x, n.y, n.z, g)
    static constexpr int kV0 = kPyr + 4 * LEVEL * Sh::NP * Sh::NF;     // [NP][NF][3] the FIRST mirror's vertices, unfolded (first_mirror_unfolded)
    static constexpr int kWords = (kV0 + 3 * Sh::NP * Sh::NF + 3) / 4 * 4;
};

template <int SCALE, int LEVEL>
__global__ __launch_bounds__(kExpandWG) void beam_boxes_kernel(BeamMesh M, BeamClusters C, const BeamEntry *__restrict__ in,
                                                               int64_t n_in, float u, float *__restrict__ ctxtab,
                                                               unsigned long long *__restrict__ masks,
                                                               int64_t clusters_per_split, RxAll rxall, BeamDev dv) {
    using Sh = Shape<SCALE>;
    using Tab = CtxTab<SCALE, LEVEL>;
    // one mask row (C.nclusters words) per WORKGROUP, written by its only wave (beam_layout sizes the buffer as ctx_cap / 64
    // rows): with more waves per workgroup they would overwrite each other's words
    static_assert(kExpandWG == 64, "the two-kernel expansion keeps one cluster-mask row per workgroup of ONE wave");
    beam_dev_apply(dv, M, u, n_in);
    if (dv.dyn) rxall = dv.dyn->rxall;
    __shared__ __attribute__((aligned(16))) float4 lds_planes[kExpandWG / 64][64 * Sh::NP];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if ((int64_t)blockIdx.x * kExpandWG >= n_in) return;  // a grid sized for the list's CAPACITY (async entry point)
    const int64_t g = (int64_t)blockIdx.x * kExpandWG + threadIdx.x;
    const bool have = g < n_in;
    BeamEntry e{};
    if (have) e = in[g];
    BeamCtx<SCALE, LEVEL> ctx;
    build_ctx<SCALE, LEVEL>(M, e, u, have, ctx);
    if (blockIdx.y == 0) {  // the table entry of this prefix (every lane writes one: lanes beyond the list hold "off" contexts)
        V3 vpar[Sh::NP][Sh::NF];
        float sig_sum = kInf;
        first_mirror_unfolded<SCALE, LEVEL>(M, e, have, vpar, sig_sum);
        float4 *t = reinterpret_cast<float4 *>(ctxtab + g * Tab::kWords);
        t[0] = float4{ctx.I.x, ctx.I.y, ctx.I.z, ctx.u};
        t[1] = float4{ctx.pm.x, ctx.pm.y, ctx.pm.z, __uint_as_float((uint32_t)ctx.side_prev)};
        t[2] = float4{ctx.nm.x, ctx.nm.y, ctx.nm.z, sig_sum};
        t[3] = float4{__uint_as_float((uint32_t)(have ? e.id[LEVEL - 1] : -1)), 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < LEVEL; ++j)
#pragma unroll
            for (int tt = 0; tt < Sh::NP; ++tt)
#pragma unroll
                for (int f = 0; f < Sh::NF; ++f) {
                    const PyrN<Sh::NF> &P = ctx.pyr[j][tt];
                    t[Tab::kPyr / 4 + (j * Sh::NP + tt) * Sh::NF + f] = float4{P.n[f].x, P.n[f].y, P.n[f].z, P.g[f]};
                }
        float *tf = ctxtab + g * Tab::kWords;
#pragma unroll
        for (int tt = 0; tt < Sh::NP; ++tt)
#pragma unroll
            for (int k = 0; k < Sh::NF; ++k) st3(tf + Tab::kV0 + 3 * (tt * Sh::NF + k), vpar[tt][k]);
    }
    const int64_t cl_begin = (int64_t)blockIdx.y * clusters_per_split;
    const int64_t cl_end = (cl_begin + clusters_per_split < C.nclusters) ? cl_begin + clusters_per_split : C.nclusters;
    auto fetch_planes = [&](int64_t c) {
        const int64_t cc = (c < cl_end) ? c : cl_end - 1;
#pragma unroll
        for (int t = 0; t < Sh::NP; ++t)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(reinterpret_cast<const float4 *>(C.uplanes) +
                                                                  (cc * 64 * Sh::NP + t * 64 + lane)),
                (__attribute__((address_space(3))) void *)(&lds_planes[wave][t * 64]), 16, 0, 0);
    };
    if (cl_begin < cl_end) fetch_planes(cl_begin);
    unsigned long long *mrow = masks + (int64_t)blockIdx.x * C.nclusters;
    // the box and the four sub-boxes of a trip are loaded during the PREVIOUS trip's tests (scalar loads, 32 registers,
    // issued after the plane loop: LDS and scalar memory share one counter, so a scalar load in flight would make the
    // loop's first ds_read wait for it): no memory latency between the plane loop and the tests
    ro_v4f_t nb[2]{}, nq[6]{};
    auto fetch_boxes = [&](int64_t c) {
        nb[0] = ro4(C.boxes, 2 * c);
        nb[1] = ro4(C.boxes, 2 * c + 1);
#pragma unroll
        for (int k = 0; k < 6; ++k) nq[k] = ro4(C.subboxes, 6 * c + k);
    };
    if (cl_begin < cl_end) fetch_boxes(cl_begin);
    for (int64_t cl = cl_begin; cl < cl_end; ++cl) {
        // (the box stage of expand_clustered_body, statement for statement: the same `todo`)
        const ro_v4f_t b0 = nb[0], b1 = nb[1];
        const ro_v4f_t sq[6] = {nq[0], nq[1], nq[2], nq[3], nq[4], nq[5]};
        const float lo[3] = {b0.x, b0.y, b0.z}, hi[3] = {b0.w, b1.x, b1.y};
        __builtin_amdgcn_s_waitcnt(/*vmcnt 0; expcnt, lgkmcnt: no wait*/ 0x0f70);
        __builtin_amdgcn_wave_barrier();
        uint32_t hbits = 0x7f800000u;
        const int nplanes = __builtin_amdgcn_readfirstlane((int)b1.w);
        // four planes per trip to the LDS (one wait for four reads: read one at a time, each plane cost a whole LDS latency
        // for five instructions of arithmetic); the minimum does not depend on the order
        auto plane = [&](const float4 q) {
            float sd = __builtin_fmaf(q.x, ctx.I.x, __builtin_fmaf(q.y, ctx.I.y, __builtin_fmaf(q.z, ctx.I.z, -q.w)));
            asm volatile("" : "+v"(sd));
            const uint32_t b = __float_as_uint(sd) & 0x7fffffffu;
            hbits = (b < hbits) ? b : hbits;
        };
        int k = 0;
        for (; k + 4 <= nplanes; k += 4) {
            const float4 q0 = lds_planes[wave][k], q1 = lds_planes[wave][k + 1], q2 = lds_planes[wave][k + 2],
                         q3 = lds_planes[wave][k + 3];
            plane(q0);
            plane(q1);
            plane(q2);
            plane(q3);
        }
        for (; k < nplanes; ++k) plane(lds_planes[wave][k]);
        const float hmin = __uint_as_float(hbits);
        __builtin_amdgcn_wave_barrier();
        fetch_planes(cl + 1);
        fetch_boxes((cl + 1 < cl_end) ? cl + 1 : cl);
        const V3 far = V3{fmaxf(__builtin_fabsf(ctx.I.x - lo[0]), __builtin_fabsf(ctx.I.x - hi[0])),
                          fmaxf(__builtin_fabsf(ctx.I.y - lo[1]), __builtin_fabsf(ctx.I.y - hi[1])),
                          fmaxf(__builtin_fabsf(ctx.I.z - lo[2]), __builtin_fabsf(ctx.I.z - hi[2]))};
        const float eps_max = beam_eps(ctx.u, b1.z, margin_len(far) * margins::kBoxFarRoundUp, hmin);
        bool alive = have && !box_pruned<SCALE, LEVEL>(ctx, lo, hi, eps_max);
        if (__any(alive)) {
            const float sb[24] = {sq[0].x, sq[0].y, sq[0].z, sq[0].w, sq[1].x, sq[1].y, sq[1].z, sq[1].w,
                                  sq[2].x, sq[2].y, sq[2].z, sq[2].w, sq[3].x, sq[3].y, sq[3].z, sq[3].w,
                                  sq[4].x, sq[4].y, sq[4].z, sq[4].w, sq[5].x, sq[5].y, sq[5].z, sq[5].w};
            bool sub = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float slo[3] = {sb[6 * q], sb[6 * q + 1], sb[6 * q + 2]}, shi[3] = {sb[6 * q + 3], sb[6 * q + 4], sb[6 * q + 5]};
                sub = sub || !box_pruned<SCALE, LEVEL>(ctx, slo, shi, eps_max);
            }
            alive = alive && sub;
        }
        const unsigned long long todo = __ballot(alive);
        if (lane == 0) mrow[cl] = todo;
    }
}

// the prefix of a pass, read from its table entry through the scalar unit (wave-uniform address, read-only memory)
template <int SCALE, int LEVEL>
struct CtxScalar {
    static constexpr bool kUniformPrefix = true;  // one prefix per pass
    using Tab = CtxTab<SCALE, LEVEL>;
    const float *__restrict__ t;  // wave-uniform
    __device__ __forceinline__ V3 I() const { return V3{t[Tab::kI], t[Tab::kI + 1], t[Tab::kI + 2]}; }
    __device__ __forceinline__ V3 pm() const { return V3{t[Tab::kPm], t[Tab::kPm + 1], t[Tab::kPm + 2]}; }
    __device__ __forceinline__ V3 nm() const { return V3{t[Tab::kNm], t[Tab::kNm + 1], t[Tab::kNm + 2]}; }
    __device__ __forceinline__ float u() const { return t[Tab::kU]; }
    __device__ __forceinline__ int side_prev() const { return (int)__float_as_uint(t[Tab::kSide]); }
    __device__ __forceinline__ PyrN<Shape<SCALE>::NF> pyr(int j, int tt) const {
        PyrN<Shape<SCALE>::NF> P;
#pragma unroll
        for (int f = 0; f < Shape<SCALE>::NF; ++f) {
            const float *q = t + Tab::kPyr + 4 * ((j * Shape<SCALE>::NP + tt) * Shape<SCALE>::NF + f);
            P.n[f] = V3{q[0], q[1], q[2]};
            P.g[f] = q[3];
        }
        return P;
    }
};

#ifndef BEAM_PAIRS_WAVES_MIN
#define BEAM_PAIRS_WAVES_MIN 6
#endif
#ifndef BEAM_PAIRS_WAVES_MAX
#define BEAM_PAIRS_WAVES_MAX 8
#endif
template <int SCALE, int LEVEL>
__global__ __launch_bounds__(kExpandWG) __attribute__((amdgpu_waves_per_eu(BEAM_PAIRS_WAVES_MIN, BEAM_PAIRS_WAVES_MAX))) void beam_expand_pairs_kernel(
    BeamMesh M, BeamClusters C, const float *__restrict__ ctxtab, const unsigned long long *__restrict__ masks, int64_t n_in,
    float u, unsigned long long *__restrict__ out, int64_t cap, unsigned long long *__restrict__ count,
    int64_t clusters_per_split, RxAll rxall, BeamDev dv, int64_t rec_off) {
    // (rec_off: position of this launch's first prefix in the list the records index -- a launch covers one chunk of it)
    static_assert(LEVEL == 1 || LEVEL == 2, "the two-kernel expansion is the LAST expansion of orders 2 and 3");
    static_assert(kExpandWG == 64, "the two-kernel expansion keeps one cluster-mask row per workgroup of ONE wave");
    using Sh = Shape<SCALE>;
    using Tab = CtxTab<SCALE, LEVEL>;
    beam_dev_apply(dv, M, u, n_in);
    if (dv.dyn) rxall = dv.dyn->rxall;
    __shared__ unsigned long long wbuf[kExpandWG / 64][kBeamWaveBufBig];
    __shared__ unsigned long long raw_rec[kExpandWG / 64][128];
    __shared__ float raw_f[kExpandWG / 64][2][128];  // (sorted position, threshold)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if ((int64_t)blockIdx.x * kExpandWG >= n_in) return;
    const unsigned long long gbase = (unsigned long long)((int64_t)blockIdx.x * kExpandWG + wave * 64);
    const float *__restrict__ tab = ctxtab + (int64_t)gbase * Tab::kWords;  // the 64 entries of this wave's prefixes
    const unsigned long long *__restrict__ mrow = masks + (int64_t)blockIdx.x * C.nclusters;
    int rawcount = 0, wcount = 0;
    // the last n parked candidates (n <= 64): lane = candidate; its parent's entry comes per lane from the table
    auto filter_parked = [&](int n) {
        const int j = rawcount - n + lane;
        bool mine = lane < n;
        const unsigned long long rec = mine ? raw_rec[wave][j] : 0ull;
        const int l = (int)((rec >> 32) - (unsigned long long)rec_off - gbase) & 63;
        const float *__restrict__ pe = tab + (int64_t)l * Tab::kWords;
        const float4 h0 = reinterpret_cast<const float4 *>(pe)[0];
        const V3 I2 = V3{h0.x, h0.y, h0.z};
        const float sp = pe[Tab::kSig];
        V3 vpar[Sh::NP][Sh::NF];
#pragma unroll
        for (int t = 0; t < Sh::NP; ++t)
#pragma unroll
            for (int k = 0; k < Sh::NF; ++k) vpar[t][k] = ld3(pe + Tab::kV0 + 3 * (t * Sh::NF + k));
        const int64_t pos = mine ? (int64_t)__float_as_uint(raw_f[wave][0][j]) : 0;  // (position 0 always exists)
        V3 vq[Sh::NV];
#pragma unroll
        for (int k = 0; k < Sh::NV; ++k) vq[k] = ld3(C.verts + 3 * (pos * Sh::NV + k));
        const float4 q = reinterpret_cast<const float4 *>(C.planes)[pos * Sh::NP];
        const V3 nc = V3{q.x, q.y, q.z};
        const float sgc = C.sigma[pos];
        if constexpr (LEVEL >= 2) {  // (level 1: the first stage was the whole test)
            const float base = raw_f[wave][1][j & 127];
            PyrN<Sh::NF> P0[Sh::NP];
#pragma unroll
            for (int t = 0; t < Sh::NP; ++t)
#pragma unroll
                for (int f = 0; f < Sh::NF; ++f) {
                    const float4 pq = reinterpret_cast<const float4 *>(pe + Tab::kPyr)[t * Sh::NF + f];  // pyramid 0
                    P0[t].n[f] = V3{pq.x, pq.y, pq.z};
                    P0[t].g[f] = pq.w;
                }
            mine = mine && !pyramids_separate<SCALE>(P0, I2, vq, base);
        }
        // (the candidate's mirror: first vertex and normal of its first triangle, prim_plane's pair)
        const bool pass = mine && !(rxall.on && child_misses_receivers<SCALE>(rxall, M, u, I2, vpar, sp, vq[0], nc, sgc, LEVEL + 1));
        rawcount -= n;
        beam_stage<kBeamWaveBufBig>(pass, rec, wbuf[wave], wcount, lane, out, cap, count);
    };
    const int64_t cl_begin = (int64_t)blockIdx.y * clusters_per_split;
    const int64_t cl_end = (cl_begin + clusters_per_split < C.nclusters) ? cl_begin + clusters_per_split : C.nclusters;
    for (int64_t cl = cl_begin; cl < cl_end; ++cl) {
        unsigned long long todo = mrow[cl];  // (scalar load)
        if (todo == 0) continue;
        const int64_t pos = cl * 64 + lane;
        const int32_t p = (pos < M.nprim) ? C.order[pos] : -1;
        V3 vx[Sh::NV];
        float pl[Sh::NP][4];
#pragma unroll
        for (int k = 0; k < Sh::NV; ++k) vx[k] = ld3(C.verts + 3 * (pos * Sh::NV + k));  // padded to whole clusters
#pragma unroll
        for (int t = 0; t < Sh::NP; ++t) {
            const float4 a = reinterpret_cast<const float4 *>(C.planes)[pos * Sh::NP + t];
            pl[t][0] = a.x; pl[t][1] = a.y; pl[t][2] = a.z; pl[t][3] = a.w;
        }
        const bool act = p >= 0 && prim_active(M, p);
        const bool self_ok = p >= 0 && may_follow(M, p, p);
        const float sg = C.sigma[pos];
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const float *__restrict__ te = tab + (int64_t)l * Tab::kWords;  // wave-uniform: scalar loads
            const int32_t m = (int32_t)__float_as_uint(te[Tab::kM]);
            const bool cand = act && (p != m || self_ok);
            V3 unused;
            float base1 = 0.0f;
            // (not `cand && ...`: the branch around the test would put the entry's scalar loads behind the wait for `m` --
            // two memory latencies per pass instead of one; a wave without any candidate lane is a padded cluster)
            const bool pruned1 = prim_stage1<SCALE, LEVEL>(CtxScalar<SCALE, LEVEL>{te}, vx, pl, sg, unused, base1);
            const bool keep = cand & !pruned1;
            const unsigned long long record = (((unsigned long long)rec_off + gbase + (unsigned long long)l) << 32) | (uint32_t)p;
            const unsigned long long vote = __ballot(keep);
            if (vote) {
                if (keep) {
                    const int j = rawcount + __popcll(vote & ((1ull << lane) - 1ull));
                    raw_rec[wave][j] = record;
                    raw_f[wave][0][j] = __uint_as_float((uint32_t)pos);
                    raw_f[wave][1][j] = base1;
                }
                rawcount += __popcll(vote);
                if (rawcount >= 64) filter_parked(64);
            }
        }
    }
    if (rawcount > 0) filter_parked(rawcount);
    if (wcount > 0) beam_flush(wbuf[wave], wcount, lane, out, cap, count);
}

// (source prefix, primitive) record -> child prefix: image of the apex in the new mirror, side of the parent's
// last mirror w.r.t. the new mirror's plane, error sum extended by the new mirror's own bound
template <int LEVEL>  // level of the PARENT
__device__ __forceinline__ BeamEntry beam_child(const BeamMesh &M, const BeamEntry &e, int32_t c, float u) {
    V3 pc, nc;
    prim_plane(M, c, pc, nc);
    const V3 I = V3{e.apex[0], e.apex[1], e.apex[2]};
    const V3 I2 = image_of_vertex(I, pc, nc);
    BeamEntry o = e;
    o.id[LEVEL] = c;
    o.apex[0] = I2.x;
    o.apex[1] = I2.y;
    o.apex[2] = I2.z;
    const float us = u * fmaxf(mag_scale(M, I), mag_scale(M, I2));
    o.esum = e.esum + prim_eps_global(M, c, I, us);
    // the previous reflection point lies within S_parent of the parent's last mirror
    o.tx_side = pack_tx_side(entry_tx(e), side_of_prim(M, e.id[LEVEL - 1], pc, nc, margins::kSideEpsFactor * e.esum + margins::kSideUnits * us));
    return o;
}

template <int LEVEL>
__global__ __launch_bounds__(256) void beam_finish_kernel(BeamMesh M, const BeamEntry *__restrict__ src,
                                                          const unsigned long long *__restrict__ rec, int64_t n,
                                                          float u, BeamEntry *__restrict__ out, BeamDev dv) {
    beam_dev_apply(dv, M, u, n);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long r = rec[i];
    out[i] = beam_child<LEVEL>(M, src[r >> 32], (int32_t)(uint32_t)r, u);
}

// ---- the same set-up from geometry loaded ONCE -----------------------------------------------------------------
// beam_child + build_ctx read a mirror's vertices up to three times (plane, error bound, pyramid) in as many
// dependent phases; the receiver stage, whose set-up is a third of its time, loads every mirror once -- the new
// mirror's data together with the source prefix, the others as soon as that prefix is there -- and feeds the SAME
// expressions (checked against the loading forms by the mapping tests: identical rows).
template <int SCALE>
struct PrimGeom {
    V3 v[Shape<SCALE>::NV];
    V3 n[Shape<SCALE>::NP];
    float sg;  // largest shape factor of the primitive's triangles
};
template <int SCALE>
__device__ __forceinline__ PrimGeom<SCALE> load_prim(const BeamMesh &M, int64_t p) {
    using Sh = Shape<SCALE>;
    PrimGeom<SCALE> g;
    const int64_t f0 = p * Sh::TPP;
#pragma unroll
    for (int k = 0; k < Sh::NV; ++k) g.v[k] = shape_vertex<SCALE>(M.tv + 9 * f0, k);
#pragma unroll
    for (int t = 0; t < Sh::NP; ++t) g.n[t] = ld3(M.normals + 3 * (f0 + t));
    g.sg = M.shape[f0];
    if (Sh::TPP == 2) g.sg = fmaxf(g.sg, M.shape[f0 + 1]);
    return g;
}
template <int SCALE>  // prim_eps_global (plane t of shapes 1 / 2 passes through vertex 3 t; shape 4: one plane, through v0)
__device__ __forceinline__ float prim_eps_from(const PrimGeom<SCALE> &g, V3 I, float u) {
    using Sh = Shape<SCALE>;
    float D = 0.0f, h = kInf;
#pragma unroll
    for (int t = 0; t < Sh::NP; ++t) h = fminf(h, plane_dist(I, g.n[t], plane_offset(g.n[t], g.v[3 * t])));
#pragma unroll
    for (int k = 0; k < Sh::NV; ++k) D = fmaxf(D, margin_len(g.v[k] - I));
    return beam_eps(u, g.sg, D, h);
}
template <int SCALE>  // side_of_prim
__device__ __forceinline__ int side_from(const PrimGeom<SCALE> &g, V3 pt, V3 n, float E) {
    float dmin = kInf, dmax = -kInf;
#pragma unroll
    for (int k = 0; k < Shape<SCALE>::NV; ++k) {
        const float d = fdot(g.v[k] - pt, n);
        dmin = fminf(dmin, d);
        dmax = fmaxf(dmax, d);
    }
    if (!(dmin == dmin) || !(dmax == dmax)) return 0;
    return side_of_range(dmin, dmax, E);
}
template <int SCALE, int LEVEL>  // beam_child; gc = the new mirror, glast = the parent's last mirror
__device__ __forceinline__ BeamEntry beam_child_from(const BeamMesh &M, const BeamEntry &e, int32_t c,
                                                     const PrimGeom<SCALE> &gc, const PrimGeom<SCALE> &glast, float u) {
    const V3 pc = gc.v[0], nc = gc.n[0];
    const V3 I = V3{e.apex[0], e.apex[1], e.apex[2]};
    const V3 I2 = image_of_vertex(I, pc, nc);
    BeamEntry o = e;
    o.id[LEVEL] = c;
    o.apex[0] = I2.x;
    o.apex[1] = I2.y;
    o.apex[2] = I2.z;
    const float us = u * fmaxf(mag_scale(M, I), mag_scale(M, I2));
    o.esum = e.esum + prim_eps_from<SCALE>(gc, I, us);
    o.tx_side = pack_tx_side(entry_tx(e), side_from<SCALE>(glast, pc, nc, margins::kSideEpsFactor * e.esum + margins::kSideUnits * us));
    return o;
}
template <int SCALE, int LEVEL>  // build_ctx
__device__ __forceinline__ void build_ctx_from(const BeamMesh &M, const BeamEntry &e, const PrimGeom<SCALE> (&geo)[LEVEL],
                                               float u, bool have, BeamCtx<SCALE, LEVEL> &c) {
    using Sh = Shape<SCALE>;
    c.I = V3{e.apex[0], e.apex[1], e.apex[2]};
    u = u * mag_scale(M, c.I);
    c.u = u;
    c.pm = V3{0, 0, 0};
    c.nm = V3{0, 0, 1};
    c.side_prev = have ? entry_side(e) : 0;
    ctx_clear<SCALE, LEVEL>(c);
    if (!have) return;
    c.pm = geo[LEVEL - 1].v[0];
    c.nm = geo[LEVEL - 1].n[0];
    float lat = 0.0f;
#pragma unroll
    for (int j = 0; j < LEVEL; ++j) lat += u * (margins::kLateralSigma * geo[j].sg + margins::kLateralConst);
    const float delta = lat;
#pragma unroll
    for (int j = 0; j < LEVEL; ++j) {
#pragma unroll
        for (int t = 0; t < Sh::NP; ++t) {
            V3 v[Sh::NF];
#pragma unroll
            for (int k = 0; k < Sh::NF; ++k) v[k] = geo[j].v[t * Sh::NF + k];
#pragma unroll
            for (int r = j + 1; r < LEVEL; ++r) {
#pragma unroll
                for (int k = 0; k < Sh::NF; ++k) v[k] = image_of_vertex(v[k], geo[r].v[0], geo[r].n[0]);
            }
            c.pyr[j][t] = make_pyr<Sh::NF>(c.I, v, delta);
        }
    }
}

// prefix g of the receiver stage (a level-ORDER entry, or a (level ORDER-1 prefix, last primitive) record whose
// child is built here) and its context, every mirror loaded once
template <int SCALE, int ORDER>
__device__ __forceinline__ void emit_setup(const BeamMesh &M, const BeamEntry *__restrict__ in,
                                           const unsigned long long *__restrict__ rec, int64_t g, bool have, float u,
                                           BeamEntry &e, BeamCtx<SCALE, ORDER> &ctx) {
    PrimGeom<SCALE> geo[ORDER] = {};
    if (have) {
        bool from_record = false;
        if constexpr (ORDER >= 2) {
            if (rec) {
                from_record = true;
                const unsigned long long r = rec[g];
                const int32_t c = (int32_t)(uint32_t)r;
                geo[ORDER - 1] = load_prim<SCALE>(M, c);  // (does not wait for the source prefix)
                const BeamEntry par = in[r >> 32];
#pragma unroll
                for (int j = 0; j < ORDER - 1; ++j) geo[j] = load_prim<SCALE>(M, par.id[j]);
                e = beam_child_from<SCALE, ORDER - 1>(M, par, c, geo[ORDER - 1], geo[ORDER - 2], u);
            }
        }
        if (!from_record) {
            e = in[g];
#pragma unroll
            for (int j = 0; j < ORDER; ++j) geo[j] = load_prim<SCALE>(M, e.id[j]);
        }
    }
    build_ctx_from<SCALE, ORDER>(M, e, geo, u, have, ctx);
}

// receiver r vs prefix: on the wrong side of the last mirror, or outside one of the pyramids?  The pyramids in
// turn, earliest mirror first (unfolded farthest from the apex = the narrowest cone); with WAVE_EXIT the wave leaves
// the receiver as soon as none of its 64 lanes is still inside (same tests, same result: a lane that fails one
// pyramid is dropped whatever the others say).
// first half of the receiver test: the narrowest pyramid (index 0: the earliest mirror, unfolded farthest from the
// apex) -- the test that rejects most receivers, alone so that the plain emit kernel's common trip is short
template <int SCALE, int ORDER>
__device__ __forceinline__ bool receiver_first(const BeamCtx<SCALE, ORDER> &c, V3 r, bool lane_on) {
    const V3 w = r - c.I;
    const float wl = l1_len(w);
    bool inside_any = false;
#pragma unroll
    for (int t = 0; t < Shape<SCALE>::NP; ++t) {
        const PyrN<Shape<SCALE>::NF> &P = c.pyr[0][t];
        float fv[Shape<SCALE>::NF];
#pragma unroll
        for (int f = 0; f < Shape<SCALE>::NF; ++f) fv[f] = __builtin_fmaf(P.g[f], wl, fdot(w, P.n[f]));
        inside_any = inside_any | !(min_faces<Shape<SCALE>::NF>(fv) < -margins::kFaceUnits * c.u);
    }
    return lane_on & inside_any;  // (& not &&: no branch around a handful of instructions)
}

// second half: the side of the last mirror, then the remaining pyramids; with WAVE_EXIT the wave leaves as soon as
// none of its 64 lanes is still inside (same tests, same result: a lane that fails one test is dropped whatever
// the others say)
template <int SCALE, int ORDER, bool WAVE_EXIT>
__device__ __forceinline__ bool receiver_rest(const BeamCtx<SCALE, ORDER> &c, V3 r, bool alive) {
    if (WAVE_EXIT && !__any(alive)) return false;
    const V3 w = r - c.I;  // (again: the rare half recomputes five instructions instead of keeping them live)
    const float wl = l1_len(w);
    const float d = fdot(r - c.pm, c.nm);
    // side_prev in {-1, 0, +1}; 0 or a NaN distance never rejects (the receiver is exact: margin 2u)
    alive = alive & !((float)c.side_prev * d < -margins::kSideUnits * c.u);
#pragma unroll
    for (int j = 1; j < ORDER; ++j) {
        if (WAVE_EXIT && !__any(alive)) return false;
        bool inside_any = false;
#pragma unroll
        for (int t = 0; t < Shape<SCALE>::NP; ++t) {
            const PyrN<Shape<SCALE>::NF> &P = c.pyr[j][t];
            float fv[Shape<SCALE>::NF];
#pragma unroll
            for (int f = 0; f < Shape<SCALE>::NF; ++f) fv[f] = __builtin_fmaf(P.g[f], wl, fdot(w, P.n[f]));
            inside_any = inside_any | !(min_faces<Shape<SCALE>::NF>(fv) < -margins::kFaceUnits * c.u);
        }
        alive = alive & inside_any;
    }
    return alive;
}

template <int SCALE, int ORDER, bool WAVE_EXIT>
__device__ __forceinline__ bool receiver_inside(const BeamCtx<SCALE, ORDER> &c, V3 r, bool lane_on) {
    const bool alive = receiver_first<SCALE, ORDER>(c, r, lane_on);
    return receiver_rest<SCALE, ORDER, WAVE_EXIT>(c, r, alive);
}

// lane = level-ORDER prefix, loop over the receivers
#ifndef BEAM_EMIT_WAVES
#define BEAM_EMIT_WAVES 0
#endif
#if BEAM_EMIT_WAVES > 0
#define BEAM_EMIT_OCC __attribute__((amdgpu_waves_per_eu(BEAM_EMIT_WAVES, BEAM_EMIT_WAVES)))
#else
#define BEAM_EMIT_OCC
#endif
template <int SCALE, int ORDER>
__global__ __launch_bounds__(256) BEAM_EMIT_OCC void beam_emit_kernel(BeamMesh M, const BeamEntry *__restrict__ in,
                                                        const unsigned long long *__restrict__ rec, int64_t n_in,
                                                        const float *__restrict__ rx_sorted,
                                                        const int32_t *__restrict__ rx_index,
                                                        const float *__restrict__ rx_boxes, int64_t nrx, float u,
                                                        long long *__restrict__ rows, int64_t cap,
                                                        unsigned long long *__restrict__ count,
                                                        unsigned long long *__restrict__ grazing, BeamDev dv) {
    beam_dev_apply(dv, M, u, n_in);
    __shared__ unsigned long long wbuf[4][kBeamWaveBuf];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int wcount = 0;
    const int lane = threadIdx.x & 63;
    unsigned long long ngraz = 0;  // wave-uniform
    // PERSISTENT: a wave walks groups of 64 prefixes and keeps its staging buffer across them.  A wave emits ~4 rows per
    // group at configs[3]; one group per wave meant one returning atomic per wave on the ONE row counter plus one per
    // grazing prefix on the statistics counter -- 7.9e6 same-address atomics per step, and the L2 atomic unit, not the
    // arithmetic, set the kernel's 37 ms (the lesson of beam_flush, one level up).
    for (int64_t g0 = (int64_t)blockIdx.x * 256; g0 < n_in; g0 += (int64_t)gridDim.x * 256) {
    const int64_t g = g0 + threadIdx.x;
    const bool have = g < n_in;
    BeamEntry e{};
    BeamCtx<SCALE, ORDER> ctx;
    emit_setup<SCALE, ORDER>(M, in, rec, g, have, u, e, ctx);
    ngraz += (unsigned long long)__popcll(__ballot(have && !(e.esum < kInf)));  // every test of this prefix is off (informational)
    const int nrx32 = (int)nrx;  // < 2^31: the 62-bit row key bounds it
    const int nclusters = (nrx32 + 63) / 64;
    long long tail = 0;  // sum_j id_j n^(k-1-j)
    long long npow = 1;
#pragma unroll
    for (int j = 0; j < ORDER; ++j) {
        tail = tail * (long long)M.nprim + (long long)(have ? e.id[j] : 0);
        npow *= (long long)M.nprim;
    }
    const long long pair0 = (long long)entry_tx(e) * (long long)nrx;
    // The receivers are read from the Morton-sorted copy (clusters of 64 with a bounding box each, padded to whole
    // clusters with copies of the last receiver; wave-uniform scalar loads).  Per cluster: ONE box test per lane
    // (box_pruned with eps = 0: receivers are exact points) and a wave vote -- at configs[3] only 18 % of the
    // prefixes, 36 % of the waves, reach the box of all 64 receivers with every pyramid.  Inside a cluster, four
    // receivers per trip (three 16-byte loads, the next trip's in flight): the first-pyramid tests of all four,
    // ONE wave vote, and only the trips where some lane is still inside some receiver go on.  Same per-receiver
    // arithmetic as receiver_inside -> the same rows.
    constexpr int G = 4;
    struct Trip {
        float4 v[3];  // x0 y0 z0 x1 | y1 z1 x2 y2 | z2 x3 y3 z3
    };
    auto load_trip = [&](int t) {
        const float4 *p4 = reinterpret_cast<const float4 *>(rx_sorted) + 3 * (int64_t)t;
        return Trip{{p4[0], p4[1], p4[2]}};
    };
    const int ntrips = (nrx32 + G - 1) / G;
    for (int cl = 0; cl < nclusters; ++cl) {
        const float *bx = rx_boxes + 6 * cl;
        const float lo[3] = {bx[0], bx[1], bx[2]}, hi[3] = {bx[3], bx[4], bx[5]};
        if (!__any(have && !box_pruned<SCALE, ORDER>(ctx, lo, hi, 0.0f))) continue;
        const int t_end = ((cl + 1) * (64 / G) < ntrips) ? (cl + 1) * (64 / G) : ntrips;
        Trip nxt = load_trip(cl * (64 / G));
        for (int t = cl * (64 / G); t < t_end; ++t) {
            const Trip cur = nxt;
            nxt = load_trip((t + 1 < t_end) ? t + 1 : t);
            const V3 r[G] = {V3{cur.v[0].x, cur.v[0].y, cur.v[0].z}, V3{cur.v[0].w, cur.v[1].x, cur.v[1].y},
                             V3{cur.v[1].z, cur.v[1].w, cur.v[2].x}, V3{cur.v[2].y, cur.v[2].z, cur.v[2].w}};
            bool a[G];
            bool any_lane = false;
#pragma unroll
            for (int q = 0; q < G; ++q) {
                a[q] = receiver_first<SCALE, ORDER>(ctx, r[q], have & (G * t + q < nrx32));
                any_lane = any_lane | a[q];
            }
            if (!__any(any_lane)) continue;
#pragma unroll
            for (int q = 0; q < G; ++q) {
                if (!__any(a[q])) continue;
                const bool keep = receiver_rest<SCALE, ORDER, true>(ctx, r[q], a[q]);
                const long long id = (long long)rx_index[G * t + q];
                beam_stage<kBeamWaveBuf>(keep, (unsigned long long)((pair0 + id) * npow + tail), wbuf[wave], wcount, lane,
                                         reinterpret_cast<unsigned long long *>(rows), cap, count);
            }
        }
    }
    }  // groups
    if (wcount > 0) beam_flush(wbuf[wave], wcount, lane, reinterpret_cast<unsigned long long *>(rows), cap, count);
    if (ngraz && lane == 0) atomicAdd(grazing, ngraz);
}

// beam_emit for MANY receivers: the receivers arrive sorted along a Morton curve in clusters of 64 with an
// axis-aligned bounding box each (lo, hi).  lane = prefix first tests each cluster's box against its pyramids /
// mirror plane (box_pruned with eps = 0: receivers are exact points), and only the (prefix, cluster) pairs that
// survive are tested per receiver -- TRANSPOSED: lane = receiver of the cluster, prefix broadcast lane-to-wave.
// Same per-receiver test as beam_emit_kernel -> the same set of rows.
template <int SCALE, int ORDER>
__global__ __launch_bounds__(128) void beam_emit_clustered_kernel(
    BeamMesh M, const BeamEntry *__restrict__ in, const unsigned long long *__restrict__ rec, int64_t n_in,
    const float *__restrict__ rx_sorted, const int32_t *__restrict__ rx_index, const float *__restrict__ boxes,
    int64_t nrx, float u, long long *__restrict__ rows, int64_t cap, unsigned long long *__restrict__ count,
    unsigned long long *__restrict__ grazing, BeamDev dv) {
    beam_dev_apply(dv, M, u, n_in);
    __shared__ unsigned long long wbuf[2][kBeamWaveBuf];
    int wcount = 0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned long long ngraz = 0;
    for (int64_t g0 = (int64_t)blockIdx.x * 128; g0 < n_in; g0 += (int64_t)gridDim.x * 128) {  // persistent, see beam_emit_kernel
    const int64_t g = g0 + threadIdx.x;
    const bool have = g < n_in;
    BeamEntry e{};
    BeamCtx<SCALE, ORDER> ctx;
    emit_setup<SCALE, ORDER>(M, in, rec, g, have, u, e, ctx);
    long long tail = 0, npow = 1;
#pragma unroll
    for (int j = 0; j < ORDER; ++j) {
        tail = tail * (long long)M.nprim + (long long)(have ? e.id[j] : 0);
        npow *= (long long)M.nprim;
    }
    ngraz += (unsigned long long)__popcll(__ballot(have && !(e.esum < kInf)));
    const int tx = entry_tx(e);
    const int64_t nclusters = (nrx + 63) / 64;
    for (int64_t cl = 0; cl < nclusters; ++cl) {
        const float *bx = boxes + 6 * cl;
        const float lo[3] = {bx[0], bx[1], bx[2]}, hi[3] = {bx[3], bx[4], bx[5]};
        unsigned long long todo = __ballot(have && !box_pruned<SCALE, ORDER>(ctx, lo, hi, 0.0f));
        if (todo == 0) continue;
        const int64_t pos = cl * 64 + lane;
        const bool have_r = pos < nrx;
        const V3 r = have_r ? ld3(rx_sorted + 3 * pos) : V3{0, 0, 0};
        const long long ir = have_r ? (long long)rx_index[pos] : 0;
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const BeamCtx<SCALE, ORDER> cx = lane_bcast<SCALE, ORDER>(ctx, l);
            const long long ltail = ((long long)__builtin_amdgcn_readlane((int)(tail >> 32), l) << 32) |
                                    (long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tail, l);
            const long long ltx = (long long)__builtin_amdgcn_readlane(tx, l);
            const bool keep = receiver_inside<SCALE, ORDER, true>(cx, r, have_r);
            beam_stage<kBeamWaveBuf>(keep, (unsigned long long)((ltx * (long long)nrx + ir) * npow + ltail), wbuf[wave],
                                     wcount, lane, reinterpret_cast<unsigned long long *>(rows), cap, count);
        }
    }
    }  // groups
    if (wcount > 0) beam_flush(wbuf[wave], wcount, lane, reinterpret_cast<unsigned long long *>(rows), cap, count);
    if (ngraz && lane == 0) atomicAdd(grazing, ngraz);
}

// ---------------------------------------------------------------------------------------------
// Morton clustering (primitives: cached on the mesh; receivers: per call, in the workspace)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t spread10(uint32_t v) {  // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// bounds[0..2] = min, [3..5] = max (ordered uints), [6] = max |coordinate| (float bits, non-negative)
__global__ __launch_bounds__(256) void point_bounds_kernel(const float *__restrict__ pts, int64_t n, int32_t group,
                                                           uint32_t *__restrict__ bounds) {
    // one item = the mean of `group` consecutive points (group = vertices per primitive; 1 for receivers)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    V3 s{0, 0, 0};
    float mag = 0.0f;
    bool fin = true;
    for (int k = 0; k < group; ++k) {
        const V3 p = ld3(pts + 3 * (i * group + k));
        s = s + p;
        mag = fmaxf(mag, fmaxf(__builtin_fabsf(p.x), fmaxf(__builtin_fabsf(p.y), __builtin_fabsf(p.z))));
        fin = fin && is_finite(p.x) && is_finite(p.y) && is_finite(p.z);
    }
    const float inv = 1.0f / (float)group;
    const float c[3] = {s.x * inv, s.y * inv, s.z * inv};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (is_finite(c[k])) {
            atomicMin(bounds + k, float_to_ordered(c[k]));
            atomicMax(bounds + 3 + k, float_to_ordered(c[k]));
        }
    }
    if (mag == mag) atomicMax(bounds + 6, __float_as_uint(mag));  // NaN coordinates: ignored here, never pruned later
    if (!fin) atomicOr(bounds + 7, 1u);                           // some point is not finite
}

__global__ __launch_bounds__(256) void morton_kernel(const float *__restrict__ pts, int64_t n, int32_t group,
                                                     const uint32_t *__restrict__ bounds, uint32_t *__restrict__ keys,
                                                     uint32_t *__restrict__ ids) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float lo[3], span = 1e-30f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = ordered_to_float(bounds[k]);
        span = fmaxf(span, ordered_to_float(bounds[3 + k]) - lo[k]);  // ONE scale for all axes: a flat set clusters in its plane
    }
    V3 s{0, 0, 0};
    for (int k = 0; k < group; ++k) s = s + ld3(pts + 3 * (i * group + k));
    const float inv = 1.0f / (float)group;
    const float c[3] = {s.x * inv, s.y * inv, s.z * inv};
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float f = (c[k] - lo[k]) / span * 1023.0f;
        f = (f == f) ? fminf(fmaxf(f, 0.0f), 1023.0f) : 0.0f;
        q[k] = (uint32_t)f;
    }
    keys[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    ids[i] = (uint32_t)i;
}

// one wave per cluster of 64 sorted primitives: gathers vertices / planes / shape factors in sorted order
// (padding lanes replicate the cluster's last primitive) and reduces the box over ALL vertices
template <int SCALE>
__global__ __launch_bounds__(64) void prim_cluster_kernel(BeamMesh M, const uint32_t *__restrict__ sorted_ids,
                                                          int32_t *__restrict__ order, float *__restrict__ verts,
                                                          float *__restrict__ planes, float *__restrict__ uplanes,
                                                          float *__restrict__ sigma, float *__restrict__ boxes,
                                                          float *__restrict__ subboxes) {
    const int64_t cl = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t pos = cl * 64 + lane;
    const int64_t src = (pos < M.nprim) ? pos : M.nprim - 1;
    const int64_t p = (int64_t)sorted_ids[src];
    if (pos < M.nprim) order[pos] = (int32_t)p;
    using Sh = Shape<SCALE>;
    float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
    float sg = 0.0f;
    float myq[Sh::NP][4];
    const int64_t f0 = p * Sh::TPP;
#pragma unroll
    for (int t = 0; t < Sh::TPP; ++t) sg = fmaxf(sg, M.shape[f0 + t]);
#pragma unroll
    for (int t = 0; t < Sh::NP; ++t) {
        const int64_t f = f0 + t;
        const V3 n = ld3(M.normals + 3 * f);
        const V3 v0 = ld3(M.tv + 9 * f);
        float *q = planes + 4 * (pos * Sh::NP + t);
        myq[t][0] = n.x; myq[t][1] = n.y; myq[t][2] = n.z; myq[t][3] = plane_offset(n, v0);
        q[0] = myq[t][0]; q[1] = myq[t][1]; q[2] = myq[t][2]; q[3] = myq[t][3];
    }
#pragma unroll
    for (int k = 0; k < Sh::NV; ++k) {
        const V3 v = shape_vertex<SCALE>(M.tv + 9 * f0, k);
        st3(verts + 3 * (pos * Sh::NV + k), v);
        lo[0] = fminf(lo[0], v.x); hi[0] = fmaxf(hi[0], v.x);
        lo[1] = fminf(lo[1], v.y); hi[1] = fmaxf(hi[1], v.y);
        lo[2] = fminf(lo[2], v.z); hi[2] = fmaxf(hi[2], v.z);
    }
    sigma[pos] = sg;
    // DISTINCT planes of the cluster, compacted to the front of `uplanes` (the two triangles of a box face, the
    // faces of one wall line ... share a plane bit for bit): the expansion's bound of the candidates' own error
    // needs the smallest plane distance of an apex over the cluster, one evaluation per distinct plane.  A
    // non-finite plane makes the cluster's shape factor +inf (never box-pruned).
    int ndistinct = 0;
    bool bad_plane = false;
#pragma unroll
    for (int t = 0; t < Sh::NP; ++t) {
        const float qx = myq[t][0], qy = myq[t][1], qz = myq[t][2], qw = myq[t][3];
        bad_plane = bad_plane || !is_finite(qx) || !is_finite(qy) || !is_finite(qz) || !is_finite(qw);
        bool dup = false;
        for (int l = 0; l < 64; ++l) {
#pragma unroll
            for (int t2 = 0; t2 < Sh::NP; ++t2) {
                const float ox = __shfl(myq[t2][0], l, 64), oy = __shfl(myq[t2][1], l, 64);
                const float oz = __shfl(myq[t2][2], l, 64), ow = __shfl(myq[t2][3], l, 64);
                // list order: (triangle t of every lane) before (triangle t + 1 of every lane)
                const bool earlier = (t2 * 64 + l) < (t * 64 + lane);
                // the same plane, or the same plane with the opposite orientation (the distance is an absolute value)
                const bool same = (ox == qx && oy == qy && oz == qz && ow == qw) || (ox == -qx && oy == -qy && oz == -qz && ow == -qw);
                dup = dup || (earlier && same);
            }
        }
        const unsigned long long keepm = __ballot(!dup);
        if (!dup) {
            float *o = uplanes + 4 * (cl * 64 * Sh::NP + ndistinct + __popcll(keepm & ((1ull << lane) - 1ull)));
            o[0] = qx; o[1] = qy; o[2] = qz; o[3] = qw;
        }
        ndistinct += __popcll(keepm);
    }
    if (__any(bad_plane)) sg = kInf;
    {  // boxes of the four groups of 16 consecutive primitives
        float slo[3] = {lo[0], lo[1], lo[2]}, shi[3] = {hi[0], hi[1], hi[2]};
#pragma unroll
        for (int off = 8; off > 0; off >>= 1)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                slo[k] = fminf(slo[k], __shfl_xor(slo[k], off, 64));
                shi[k] = fmaxf(shi[k], __shfl_xor(shi[k], off, 64));
            }
        if ((lane & 15) == 0) {
            float *b = subboxes + 24 * cl + 6 * (lane >> 4);
            b[0] = slo[0]; b[1] = slo[1]; b[2] = slo[2];
            b[3] = shi[0]; b[4] = shi[1]; b[5] = shi[2];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
        }
        // NaN-propagating max for the shape factor is not needed: shape is never NaN (mesh_prepare_kernel)
        sg = fmaxf(sg, __shfl_xor(sg, off, 64));
    }
    if (lane == 0) {
        float *b = boxes + 8 * cl;
        // fminf / fmaxf drop NaNs: a NaN vertex leaves its box finite, but such a primitive is never pruned by
        // its own test and a NaN coordinate makes no guarantee meaningful anyway; keep the box honest for infs
        b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2];
        b[3] = hi[0]; b[4] = hi[1]; b[5] = hi[2];
        b[6] = sg;
        b[7] = (float)ndistinct;
    }
}

// receivers: sorted copy, original indices, box (lo, hi) per cluster of 64
__global__ __launch_bounds__(64) void rx_cluster_kernel(const float *__restrict__ rx, int64_t nrx,
                                                        const uint32_t *__restrict__ sorted_ids,
                                                        float *__restrict__ rx_sorted, int32_t *__restrict__ rx_index,
                                                        float *__restrict__ boxes) {
    const int64_t cl = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t pos = cl * 64 + lane;
    const int64_t src = (pos < nrx) ? pos : nrx - 1;
    const int64_t i = (int64_t)sorted_ids[src];
    const V3 r = ld3(rx + 3 * i);
    // the tail of the last cluster holds copies of the last receiver (the plain emit kernel reads whole trips)
    st3(rx_sorted + 3 * pos, r);
    rx_index[pos] = (int32_t)i;
    float lo[3] = {r.x, r.y, r.z}, hi[3] = {r.x, r.y, r.z};
    bool nan = !(r.x == r.x) || !(r.y == r.y) || !(r.z == r.z);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
        }
    const bool any_nan = __any(nan);
    if (lane == 0) {
        float *b = boxes + 6 * cl;
        // a NaN receiver makes the cluster's box NaN: box_pruned then keeps the cluster (per-receiver tests decide)
        const float bad = __builtin_nanf("");
        for (int k = 0; k < 3; ++k) {
            b[k] = any_nan ? bad : lo[k];
            b[3 + k] = any_nan ? bad : hi[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// rows -> per-pair candidate table of the compact tracer
// ---------------------------------------------------------------------------------------------
constexpr unsigned long long kRowSentinel = 1ull << 62;  // row keys are < 2^62 (checked by the entry points)

// sorted packed rows -> table i32[rows, ORDER] (triangle ids; a repeated row becomes a padding row of -1)
template <int ORDER>
__global__ __launch_bounds__(256) void rows_decode_kernel(const unsigned long long *__restrict__ rows, int64_t n,
                                                          unsigned long long nprim, int32_t scale,
                                                          int32_t *__restrict__ table) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = rows[i];
    // (keys at or above kRowSentinel pad a fixed-capacity list: drt_trace_paths_beam_async)
    const bool dup = (i > 0 && rows[i - 1] == key) || key >= kRowSentinel;
    unsigned long long rest = key;
    int32_t id[ORDER];
#pragma unroll
    for (int j = ORDER - 1; j >= 0; --j) {
        const unsigned long long q = rest / nprim;
        id[j] = (int32_t)(rest - q * nprim) * scale;
        rest = q;
    }
#pragma unroll
    for (int j = 0; j < ORDER; ++j) table[i * ORDER + j] = dup ? -1 : id[j];
}

// offsets[p] = first sorted row of pair p (lower bound of p * n^ORDER), p = 0 .. npairs
__global__ __launch_bounds__(256) void pair_offsets_kernel(const unsigned long long *__restrict__ rows, int64_t n,
                                                           unsigned long long npow, int64_t npairs,
                                                           long long *__restrict__ offsets, int64_t mult) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p > npairs) return;
    const unsigned long long target = (unsigned long long)p * npow;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (rows[mid] < target) lo = mid + 1; else hi = mid;
    }
    offsets[p] = lo * mult;  // (mult: table rows per sorted row, 2^ORDER in coplanar-pair mode)
}

// ---- primitive shapes of an assume_quads mesh ------------------------------------------------------------------
// flag[0] stays 1 when every quad (2i, 2i+1) is a CONVEX PLANAR FAN QUAD (struct Shape, shape 4): equal (==) unit
// normals and first vertices (float equality: +0 == -0 -- the cross product of axis-aligned edges yields zeros of
// either sign -- and a NaN equals nothing), second triangle = (v0, v2, v3) of the first one's (v0, v1, v2), every corner
// of v0 v1 v2 v3 turning the way the normal says by at least sin = 1e-3 (a clear margin: a corner that is flat or reflex
// within rounding falls back to the two-triangle form, shape 2).
__device__ __forceinline__ bool fan_quad_convex(V3 n, const V3 (&q)[4]) {
    bool quad = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const V3 e0 = q[(k + 1) % 4] - q[k], e1 = q[(k + 2) % 4] - q[(k + 1) % 4];
        const float turn = fdot(cross(e0, e1), n);
        const float scale = __builtin_sqrtf(fdot(e0, e0) * fdot(e1, e1));
        quad = quad && (turn > margins::kQuadConvexSin * scale) && is_finite(scale);
    }
    return quad;
}
__global__ __launch_bounds__(256) void quad_shape_kernel(const float *__restrict__ tv, const float *__restrict__ normals,
                                                         int64_t nquads, uint32_t *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nquads) return;
    const float *a = normals + 6 * i, *b = a + 3;
    const float *va = tv + 18 * i, *vb = va + 9;
    const bool same_plane = a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && va[0] == vb[0] && va[1] == vb[1] && va[2] == vb[2];
    bool quad = same_plane && va[6] == vb[3] && va[7] == vb[4] && va[8] == vb[5];  // shared diagonal v0 - v2
    if (quad) {
        const V3 q[4] = {ld3(va), ld3(va + 3), ld3(va + 6), ld3(vb + 6)};
        quad = fan_quad_convex(ld3(a), q);
    }
    if (!quad) atomicAnd(flag, 0u);
}

// ---- coplanar pairs of a triangle soup -------------------------------------------------------------------------
// Two triangles A = (v0, v1, v2) and B = (v0, v2, v3) with EQUAL (==) unit normals, the same first vertex and the same
// mask value are THE SAME MIRROR for the reference (plane point and normal are all the image method reads of a triangle,
// _solvers.py:552-562; the sign of a zero component changes no value there, only the sign of a zero result): every
// image and reflection point of a candidate has the same value whichever of the two it names, and only the inside test
// tells them apart.  When moreover v0 v1 v2 v3 is convex (fan_quad_convex) the pair is searched as ONE shape-4 primitive.
// Round 4 took the pairs (2i, 2i+1) and gave up on the whole mesh when one of them failed; the reference's real meshes
// (tests/golden/bruxelles.npz: walls are fans of two, roofs are ear-clipped polygons, in no particular order) pair
// 36 of 7 103 that way.  pair_candidate_kernel answers, for EVERY triangle A, which triangle B follows it around the
// shared first vertex (succ[A] = the lowest such B, or -1) -- 5 829 pairs on bruxelles after the host's matching.
// Keys are bit patterns with -0 folded into +0 (float equality); a NaN coordinate or normal pairs with nothing.
struct PairKey {
    uint32_t w[6];  // v0, then the diagonal's other end (A: v2, B: v1)
};
__device__ __forceinline__ uint32_t fold_zero(float x) { return __float_as_uint(x + 0.0f); }
__global__ __launch_bounds__(256) void pair_keys_kernel(const float *__restrict__ tv, int64_t T, uint64_t *__restrict__ hash_second,
                                                        uint32_t *__restrict__ ids) {
    // hash of (v0, v1): the key a triangle is found under as the SECOND triangle of a fan
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const float *v = tv + 9 * t;
    uint64_t h = 0x9e3779b97f4a7c15ull;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        h ^= (uint64_t)fold_zero(v[k]);
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 32;
    }
    hash_second[t] = h;
    ids[t] = (uint32_t)t;
}
// succ[a] = lowest b != a with hash/second-key equal to a's (v0, v2), equal normals, masks, convex union
__global__ __launch_bounds__(256) void pair_candidate_kernel(const float *__restrict__ tv, const float *__restrict__ normals,
                                                             const uint8_t *__restrict__ mask, int64_t T,
                                                             const uint64_t *__restrict__ sorted_hash,
                                                             const uint32_t *__restrict__ sorted_ids,
                                                             int32_t *__restrict__ succ) {
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= T) return;
    const float *va = tv + 9 * a;
    uint64_t h = 0x9e3779b97f4a7c15ull;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        h ^= (uint64_t)fold_zero(va[k < 3 ? k : k + 3]);  // (v0, v2)
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 32;
    }
    int64_t lo = 0, hi = T;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sorted_hash[mid] < h) lo = mid + 1; else hi = mid;
    }
    int32_t best = -1;
    const V3 na = ld3(normals + 3 * a);
    for (int64_t i = lo; i < T && sorted_hash[i] == h; ++i) {
        const int64_t b = (int64_t)sorted_ids[i];
        if (b == a) continue;
        const float *vb = tv + 9 * b;
        const V3 nb = ld3(normals + 3 * b);
        bool ok = va[0] == vb[0] && va[1] == vb[1] && va[2] == vb[2] && va[6] == vb[3] && va[7] == vb[4] && va[8] == vb[5] &&
                  na.x == nb.x && na.y == nb.y && na.z == nb.z;
        if (mask) ok = ok && ((mask[a] != 0) == (mask[b] != 0));
        if (ok) {
            const V3 q[4] = {ld3(va), ld3(va + 3), ld3(va + 6), ld3(vb + 6)};
            ok = fan_quad_convex(na, q);
        }
        if (ok && (best < 0 || (int32_t)b < best)) best = (int32_t)b;
    }
    succ[a] = best;
}
// the virtual mesh of 2 P triangles the kernels read in pair mode (drt_mesh::pair_*)
__global__ __launch_bounds__(256) void pair_gather_kernel(const float *__restrict__ tv, const float *__restrict__ normals,
                                                          const float *__restrict__ shape, const uint8_t *__restrict__ mask,
                                                          const int32_t *__restrict__ pair_tri, int64_t P,
                                                          float *__restrict__ ptv, float *__restrict__ pn,
                                                          float *__restrict__ ps, uint8_t *__restrict__ pm) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int64_t t0 = pair_tri[2 * p], t1 = pair_tri[2 * p + 1];
    const float *a = tv + 9 * t0;
#pragma unroll
    for (int k = 0; k < 9; ++k) ptv[18 * p + k] = a[k];
    if (t1 >= 0) {
        const float *b = tv + 9 * t1;
#pragma unroll
        for (int k = 0; k < 9; ++k) ptv[18 * p + 9 + k] = b[k];
    } else {  // (v0, v2, v2): shape_vertex<4> reads the quad (v0, v1, v2, v2)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ptv[18 * p + 9 + k] = a[k];
            ptv[18 * p + 12 + k] = a[6 + k];
            ptv[18 * p + 15 + k] = a[6 + k];
        }
    }
    const int64_t s1 = t1 >= 0 ? t1 : t0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        pn[6 * p + k] = normals[3 * t0 + k];
        pn[6 * p + 3 + k] = normals[3 * s1 + k];
    }
    ps[2 * p] = shape[t0];
    ps[2 * p + 1] = shape[s1];
    if (pm) {
        pm[2 * p] = mask[t0];
        pm[2 * p + 1] = mask[s1];
    }
}

// sorted packed PRIMITIVE rows ((tx nrx + rx) nq^K + sum q_j nq^(K-1-j)) of pair mode -> the 2^K triangle rows of each,
// in place of rows_decode: table row [i 2^K + c] names triangle pair_tri[q_j][bit_j(c)] at mirror j, and tri_keys[i 2^K + c]
// is its triangle-level packed key (over the T triangles of the mesh), what drt_trace_paths_beam returns.  Padding:
//   a repeated primitive row (or a sentinel of the fixed-capacity list): the whole block is -1;
//   a combination that names the absent second triangle of a single, or one triangle twice in a row (not a candidate of
//   the reference's graph, graph.rs:400-470): the row is padding for every consumer (all ids < 0), but an id that exists
//   is stored as -2 - id, so that the pair-block filter (trace_filter_pairblocks_kernel) can still read both triangles
//   of every primitive from the block's first and last row.
template <int ORDER>
__global__ __launch_bounds__(256) void rows_expand_pairs_kernel(const unsigned long long *__restrict__ rows, int64_t n,
                                                                unsigned long long nq, const int32_t *__restrict__ pair_tri,
                                                                unsigned long long T, int32_t *__restrict__ table,
                                                                unsigned long long *__restrict__ tri_keys) {
    constexpr int COMBOS = 1 << ORDER;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n * COMBOS) return;
    const int64_t i = g >> ORDER;
    const int c = (int)(g & (COMBOS - 1));
    const unsigned long long key = rows[i];
    const bool dead = (i > 0 && rows[i - 1] == key) || key >= kRowSentinel;  // the whole block
    bool bad = dead;
    unsigned long long rest = key;
    int32_t id[ORDER];
#pragma unroll
    for (int j = ORDER - 1; j >= 0; --j) {
        const unsigned long long q = rest / nq;
        id[j] = dead ? -1 : pair_tri[2 * (rest - q * nq) + ((c >> (ORDER - 1 - j)) & 1)];
        bad = bad || id[j] < 0;
        rest = q;
    }
#pragma unroll
    for (int j = 1; j < ORDER; ++j) bad = bad || (id[j] == id[j - 1]);
    unsigned long long tk = rest;  // the (tx, rx) pair index
#pragma unroll
    for (int j = 0; j < ORDER; ++j) tk = tk * T + (unsigned long long)(id[j] < 0 ? 0 : id[j]);
#pragma unroll
    for (int j = 0; j < ORDER; ++j) table[g * ORDER + j] = !bad ? id[j] : ((dead || id[j] < 0) ? -1 : -2 - id[j]);
    tri_keys[g] = bad ? ~0ull : tk;
}

// keys of the slice's trace are rows of its table: back to packed rows
__global__ __launch_bounds__(256) void keys_to_rows_kernel(const long long *__restrict__ keys, int64_t n,
                                                           const unsigned long long *__restrict__ rows,
                                                           long long *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (long long)rows[keys[i]];
}

// ---- fixed-capacity, no-readback plumbing of drt_trace_paths_beam_async ------------------------------------------
// scene scalars on the device: what drt_trace_paths_beam computes on the host from the same bounds
__global__ void beam_dyn_kernel(const uint32_t *__restrict__ rx_bounds, const uint32_t *__restrict__ tx_bounds,
                                float mesh_max_abs, float kappa, int32_t filter_allowed, BeamDyn *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float mag = fmaxf(__uint_as_float(rx_bounds[6]), __uint_as_float(tx_bounds[6]));
    mag = fmaxf(fmaxf(mag, mesh_max_abs), 1e-30f);
    BeamDyn d;
    d.rxall = RxAll{{0, 0, 0}, {0, 0, 0}, 0, 0.0f, {0, 0, 0}, {0, 0, 0}};
    if (rx_bounds[7] == 0u && filter_allowed) {
        bool ok = true;
        for (int k = 0; k < 3; ++k) {
            d.rxall.lo[k] = ordered_to_float(rx_bounds[k]);
            d.rxall.hi[k] = ordered_to_float(rx_bounds[3 + k]);
            ok = ok && is_finite(d.rxall.lo[k]) && is_finite(d.rxall.hi[k]) && d.rxall.lo[k] <= d.rxall.hi[k];
        }
        d.rxall.on = ok ? 1 : 0;
    }
    int ex = 0;
    (void)__builtin_frexpf(mag, &ex);  // mag = f * 2^ex, f in [0.5, 1)
    const float ulp = __builtin_ldexpf(1.0f, ex - 1 - 23);
    d.u = kappa * ulp;
    d.inv_2m = 0.5f / mag;
    d.rxall.ulp_m = ulp;
    rx_all_finish(d.rxall);
    *out = d;
}

// rows[min(*count, cap) .. cap) = sentinel: the static-size sort moves them behind every real row
__global__ __launch_bounds__(256) void rows_pad_kernel(unsigned long long *__restrict__ rows,
                                                       const unsigned long long *__restrict__ count, int64_t cap) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = ((int64_t)*count < cap) ? (int64_t)*count : cap;
    if (i >= n && i < cap) rows[i] = kRowSentinel;
}

// table rows of the padded trace -> packed keys; padding stays -1
__global__ __launch_bounds__(256) void keys_to_rows_padded_kernel(const long long *__restrict__ keys, int64_t n,
                                                                  const unsigned long long *__restrict__ rows,
                                                                  long long *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long k = keys[i];
    out[i] = (k < 0) ? -1ll : (long long)rows[k];
}

// counts_dev[2] |= the pruned search's own overflow bits, counts_dev[3] = candidate rows traced
// The asynchronous entry's list counters, each on its OWN 128-byte line (index in 8-byte words): every wave of a stage reads
// the length of its input list (BeamDev::n) at its start while the waves of the same stage add to the length of their
// output list -- on one line the reads queued behind the atomics (configs[2]: 50 000 one-wave workgroups, the order-2
// last expansion 1.18 ms in the asynchronous entry against 0.58 ms in the synchronous one, same instructions).
constexpr int kCtrGrazing = 1, kCtrLevel1 = 16, kCtrLevel2 = 32, kCtrRecords = 48, kCtrRows = 64, kCtrBytes = 640, kDynOffset = 768;
__global__ void beam_counts_kernel(const unsigned long long *__restrict__ c, int64_t cap_entries, int64_t cap_records,
                                   int64_t cap_rows, int32_t row_shift, long long *__restrict__ counts) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long st = counts[2];
    if ((int64_t)c[kCtrLevel2] > cap_entries) st |= DRT_BEAM_OVERFLOW_ENTRIES;
    if ((int64_t)c[kCtrRecords] > cap_records) st |= DRT_BEAM_OVERFLOW_RECORDS;
    if ((int64_t)c[kCtrRows] > cap_rows) st |= DRT_BEAM_OVERFLOW_ROWS;
    counts[2] = st;
    const long long r = ((int64_t)c[kCtrRows] < cap_rows) ? (long long)c[kCtrRows] : (long long)cap_rows;
    counts[3] = r << row_shift;
}

// (a copy KERNEL: memcpy / memset nodes of a captured graph are not trusted on this runtime, see fill_bytes_async)
__global__ __launch_bounds__(256) void copy_u32_kernel(const uint32_t *__restrict__ in, int64_t n, uint32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

__global__ __launch_bounds__(256) void iota_kernel(uint32_t *__restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (uint32_t)i;
}

// out[i, :] = in[perm[i], :] for rows of `width` 4-byte words
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint32_t *__restrict__ in, const uint32_t *__restrict__ perm,
                                                          int64_t n, int32_t width, uint32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * width) return;
    const int64_t r = i / width, c = i - r * width;
    out[i] = in[(int64_t)perm[r] * width + c];
}

static BeamMesh beam_mesh(drt_mesh_t m) {
    BeamMesh M;
    M.tv = m->tri_verts;
    M.normals = m->normals;
    M.shape = m->shape;
    M.mask = m->has_mask ? m->mask : nullptr;
    M.scale = m->assume_quads ? 2 : 1;
    M.kind = m->assume_quads ? ((m->beam_quad4 == 1) ? 4 : 2) : 1;
    M.nprim = m->num_triangles / M.scale;
    M.inv_2m = 0.0f;  // set by the driver once the scene magnitude is known (0: no rescaling)
    M.self_loops = 0;
    M.pair_tri = nullptr;
    return M;
}
// the same TRIANGLE mesh searched over the primitives of the pairing pass (drt_mesh::pair_state == 1): the virtual
// mesh of 2 P triangles, every primitive a shape-4 quad (a single triangle: the degenerate quad v0 v1 v2 v2)
static BeamMesh beam_mesh_pairs(drt_mesh_t m) {
    BeamMesh M;
    M.tv = m->pair_tv;
    M.normals = m->pair_normals;
    M.shape = m->pair_shape;
    M.mask = m->has_mask ? m->pair_mask : nullptr;
    M.scale = 2;
    M.kind = 4;
    M.nprim = m->pair_prims;
    M.inv_2m = 0.0f;
    M.self_loops = 1;
    M.pair_tri = m->pair_tri;
    return M;
}
static BeamClusters beam_clusters_of(const drt_mesh::BeamCache &b) {
    BeamClusters C;
    C.order = b.order;
    C.verts = b.verts;
    C.planes = b.planes;
    C.uplanes = b.uplanes;
    C.sigma = b.sigma;
    C.boxes = b.boxes;
    C.subboxes = b.subboxes;
    C.nclusters = b.clusters;
    return C;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// (temporary storage: enough for rocPRIM's default configuration AND for the capture-safe one of sort_safe.hpp -- the
// synchronous and the asynchronous entry point share one workspace layout)
static size_t sort_pairs_temp_bytes(int64_t n) {
    if (n <= 0) return 0;
    size_t bytes = 0, safe = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (size_t)n, 0, 32, nullptr);
    (void)rocprim::radix_sort_pairs<CaptureSafeSort>(nullptr, safe, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                                     (uint32_t *)nullptr, (size_t)n, 0, 32, nullptr);
    return std::max(bytes, safe);
}
static size_t sort_keys64_temp_bytes(int64_t n) {
    if (n <= 0) return 0;
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                   (size_t)n, 0, 64, nullptr);
    return std::max(bytes, capture_safe_sort_temp_bytes(n, false));
}
static size_t sort_pairs64_temp_bytes(int64_t n) {
    if (n <= 0) return 0;
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                    (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0, 64, nullptr);
    return std::max(bytes, capture_safe_sort_temp_bytes(n, true));
}

// Morton order of `n` items (means of `group` consecutive points): sorted ids in `*ids_out`, bounds
// (min[3], max[3] as ordered uints, max |coordinate| as float bits) in `*bounds_out`; scratch carved from `tmp`
static size_t morton_scratch_bytes(int64_t n) {
    if (n < 1) n = 1;
    return align_up((size_t)n * 4, 256) * 4 + 256 + align_up(sort_pairs_temp_bytes(n), 256);
}
static int32_t morton_order(const float *pts, int64_t n, int32_t group, char *tmp, uint32_t **ids_out,
                            uint32_t **bounds_out, hipStream_t s) {
    const size_t a = align_up((size_t)(n > 0 ? n : 1) * 4, 256);
    uint32_t *keys = reinterpret_cast<uint32_t *>(tmp);
    uint32_t *ids = reinterpret_cast<uint32_t *>(tmp + a);
    uint32_t *keys_sorted = reinterpret_cast<uint32_t *>(tmp + 2 * a);
    uint32_t *ids_sorted = reinterpret_cast<uint32_t *>(tmp + 3 * a);
    uint32_t *bounds = reinterpret_cast<uint32_t *>(tmp + 4 * a);
    char *sort_tmp = tmp + 4 * a + 256;
    // min slots start at 0xffffffff, max slots (and the magnitude) at 0
    DRT_HIP(fill_bytes_async(bounds, 0xff, 12, s));
    DRT_HIP(fill_bytes_async(bounds + 3, 0, 20, s));
    *ids_out = ids_sorted;
    *bounds_out = bounds;
    if (n <= 0) return DRT_OK;
    const dim3 grid((unsigned)ceil_div(n, 256));
    hipLaunchKernelGGL(point_bounds_kernel, grid, dim3(256), 0, s, pts, n, group, bounds);
    hipLaunchKernelGGL(morton_kernel, grid, dim3(256), 0, s, pts, n, group, bounds, keys, ids);
    DRT_LAUNCH_CHECK();
    size_t tb = sort_pairs_temp_bytes(n);
    // (capture-safe configuration: the receivers' order is computed inside drt_trace_paths_beam_async)
    DRT_HIP(rocprim::radix_sort_pairs<CaptureSafeSort>(sort_tmp, tb, keys, keys_sorted, ids, ids_sorted, (size_t)n, 0, 30, s));
    return DRT_OK;
}

}  // namespace drt

using namespace drt;

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
namespace {

struct BeamLayout {  // byte offsets into the caller's workspace
    size_t counters, rx_sorted, rx_index, rx_boxes, morton, entries1, entries2, records, rows, rows_sorted, table,
        pair_offsets, sort_tmp, trace_ws, trace_ws_bytes, slice_keys, merge_keys, merge_perm, merge_iota, merge_rows,
        ctx_table, ctx_masks, total;
};

struct BeamSizes {
    int64_t max_entries, max_records, max_rows, max_survivors;
    int64_t ctx_cap;  // prefixes per launch of the two-kernel last expansion (context table + cluster masks); 0: order < 2
};
constexpr int64_t kCtxWordsMax = 64;  // the largest entry of the shapes that take the two-kernel form (shape known only once the clusters exist)
static_assert(CtxTab<1, 1>::kWords <= kCtxWordsMax && CtxTab<1, 2>::kWords <= kCtxWordsMax && CtxTab<4, 1>::kWords <= kCtxWordsMax &&
                  CtxTab<4, 2>::kWords <= kCtxWordsMax,
              "context table entries");

// Default list capacities, sized from the scene (results never depend on them: a slice that overflows is retried
// smaller, drt_trace_paths_beam; round 3 took 2^26 / 2^27 / 2^26 / 2^22 whatever the scene = 5.5 GiB of workspace
// for a 12-triangle box).  Two ingredients:
//   * hard bounds: a level-2 list has at most ntx * n^2 prefixes, an expansion slice at most (its prefixes) * n
//     records, a slice's rows at most records * nrx, the survivors at most the rows;
//   * a fan-out estimate for scenes in between: the level-2 list of a city mesh holds F = 2-8 % of the n children of a
//     level-1 prefix (configs[2]: 480 of 10 000; configs[4]: 2 600 of 200 000) -- est = ntx * n * max(256, n / 8),
//     capacities 4x that, between 2^18 and the round-3 values.
// configs[2] / [3] / [4] resolve to exactly the round-3 capacities.
static int64_t sat_mul(int64_t a, int64_t b) {
    if (a <= 0 || b <= 0) return 0;
    return (a > (int64_t)(1ll << 62) / b) ? (int64_t)(1ll << 62) : a * b;
}
static int64_t pow2_at_least(int64_t v) {
    int64_t p = 1;
    while (p < v && p < (int64_t)(1ll << 62)) p <<= 1;
    return p;
}
static int64_t clamp64(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

static BeamSizes beam_sizes(const drt_beam_params *bp, int64_t ntx, int64_t nrx, int64_t nprim, int32_t order) {
    const int64_t n1 = sat_mul(ntx > 0 ? ntx : 1, nprim > 0 ? nprim : 1);          // level-1 prefixes at most
    const int64_t n2 = sat_mul(n1, nprim > 0 ? nprim : 1);                          // level-2 prefixes at most
    const int64_t fan = std::min<int64_t>(nprim > 0 ? nprim : 1, std::max<int64_t>(256, nprim / 8));
    const int64_t est = sat_mul(n1, fan);                                           // expected level-2 size
    const int64_t rx1 = nrx > 0 ? nrx : 1;
    BeamSizes z;
    const int64_t d_entries = std::min(clamp64(pow2_at_least(sat_mul(est, 2)), (int64_t)1 << 16, (int64_t)1 << 26), std::max<int64_t>(n2, 64));
    z.max_entries = (bp && bp->max_entries > 0) ? bp->max_entries : d_entries;
    // records of one slice: order 2 expands level-1 prefixes (<= n2 in all); order 3 expands slices of the level-2
    // list, each prefix into at most nprim records
    const int64_t rec_bound = (order >= 3) ? sat_mul(z.max_entries, nprim > 0 ? nprim : 1) : n2;
    const int64_t d_records = std::min(clamp64(pow2_at_least(sat_mul(est, 4)), (int64_t)1 << 18, (int64_t)1 << 27), std::max<int64_t>(rec_bound, 64));
    z.max_records = (bp && bp->max_records > 0) ? bp->max_records : d_records;
    // (hard bound of the rows of one slice, times 2^order: in coplanar-pair mode a row of primitives becomes 2^order triangle
    // rows of the table, and a soup of mostly single triangles has nearly as many primitives as triangles -- without the
    // factor a 81-triangle scene whose every row survives asked for 176 table rows of 162: round 6's soup stress)
    const int64_t row_bound = sat_mul(sat_mul(order >= 2 ? z.max_records : n1, rx1), (int64_t)1 << (order > 0 ? order : 0));
    const int64_t d_rows = std::min(clamp64(z.max_records / 2, (int64_t)1 << 18, (int64_t)1 << 26), std::max<int64_t>(row_bound, 64));
    z.max_rows = (bp && bp->max_rows > 0) ? bp->max_rows : d_rows;
    z.max_survivors = (bp && bp->max_survivors > 0) ? bp->max_survivors : std::min<int64_t>((int64_t)1 << 22, z.max_rows);
    // order 3: the last expansion runs over slices / chunks of at most ctx_cap level-2 prefixes (288 B of context per
    // prefix, 8 B of mask per (64 prefixes, cluster): at most 2^21 prefixes, masks of at most 256 MiB)
    // (order 2: the same over the level-1 list, at most ntx * n prefixes)
    z.ctx_cap = 0;
    if (order >= 2) {
        const int64_t ncl = ceil_div(nprim > 0 ? nprim : 1, 64);
        const int64_t by_masks = std::max<int64_t>(64, ((int64_t)1 << 25) / ncl * 64);  // 2^28 B / 8 B per mask word
        const int64_t list = (order >= 3) ? std::min(z.max_entries, z.max_records) : (n1 + 63) / 64 * 64;
        z.ctx_cap = std::max<int64_t>(64, std::min({list, (int64_t)1 << 21, by_masks}) / 64 * 64);
    }
    return z;
}

static BeamLayout beam_layout(const BeamSizes &z, int64_t ntx, int64_t nrx, int64_t nprim, int32_t order,
                              int64_t max_paths) {
    BeamLayout L{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += align_up(bytes > 0 ? bytes : 1, 256);
        return at;
    };
    const int64_t k2 = order + 2;
    const int64_t nrx_p = ceil_div(nrx > 0 ? nrx : 1, 64) * 64;
    L.counters = take(1024);  // async entry: one 128-byte line per counter (kCtr* below), the device scalars behind them
    L.rx_sorted = take((size_t)nrx_p * 12);
    L.rx_index = take((size_t)nrx_p * 4);
    L.rx_boxes = take((size_t)(nrx_p / 64) * 24);
    L.morton = take(morton_scratch_bytes(nrx));
    L.entries1 = take((size_t)(ntx * nprim > 0 ? ntx * nprim : 1) * 32);
    L.entries2 = take(order >= 3 ? (size_t)z.max_entries * 32 : 1);
    L.records = take(order >= 2 ? (size_t)z.max_records * 8 : 1);
    L.rows = take((size_t)z.max_rows * 8);
    L.rows_sorted = take((size_t)z.max_rows * 8);
    L.table = take((size_t)z.max_rows * 4 * (size_t)(order > 0 ? order : 1));
    L.pair_offsets = take((size_t)(ntx * nrx + 1) * 8);
    size_t st = sort_keys64_temp_bytes(z.max_rows);
    const size_t st2 = sort_pairs64_temp_bytes(max_paths);
    if (st2 > st) st = st2;
    L.sort_tmp = take(st);
    L.trace_ws_bytes = drt_trace_compact_workspace_size(z.max_survivors, max_paths);
    L.trace_ws = take(L.trace_ws_bytes);
    L.slice_keys = take((size_t)max_paths * 8);
    L.merge_keys = take((size_t)max_paths * 8);
    L.merge_perm = take((size_t)max_paths * 4);
    L.merge_iota = take((size_t)max_paths * 4);
    L.merge_rows = take((size_t)max_paths * (size_t)k2 * 12);
    L.ctx_table = take((size_t)z.ctx_cap * (size_t)kCtxWordsMax * 4);
    L.ctx_masks = take((size_t)(z.ctx_cap / 64) * (size_t)ceil_div(nprim > 0 ? nprim : 1, 64) * 8);
    L.total = off;
    return L;
}

// HIP-event stopwatch for drt_beam_stats (only when the caller asked for stats): start / stop around a group of
// launches on the call's stream, read after the next synchronisation the call makes anyway
struct BeamTimer {
    bool on = false;
    hipStream_t s = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    bool pending = false;
    float *acc = nullptr;
    void init(bool enable, hipStream_t stream, float *target) {
        s = stream;
        acc = target;
        on = enable && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess;
    }
    ~BeamTimer() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
    void start() {
        if (on) (void)hipEventRecord(a, s);
    }
    void stop() {
        if (on) {
            (void)hipEventRecord(b, s);
            pending = true;
        }
    }
    void collect() {  // after a stream synchronisation
        if (on && pending) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, a, b) == hipSuccess) *acc += ms;
            pending = false;
        }
    }
};

static int32_t read_count(const unsigned long long *dev, int64_t *host, hipStream_t s) {
    unsigned long long v = 0;
    DRT_HIP(hipMemcpyAsync(&v, dev, 8, hipMemcpyDeviceToHost, s));
    DRT_HIP(hipStreamSynchronize(s));
    *host = (int64_t)v;
    return DRT_OK;
}

// `dv`: device-side scalars / list size (async entry point; then n_in is the list's CAPACITY and sizes the grid);
// `last`: the last expansion (receiver-box child filter; with dv.dyn the receivers' box comes from the device and a box
// that is off lets every child pass)
// Workgroups (of 2 waves) a clustered expansion aims for by splitting the cluster range over blockIdx.y.  2 048 (one
// round of the chip) left the launch to its slowest waves -- the survivors per (wave, cluster range) vary widely:
// configs[4]'s last expansion (782 x 3 workgroups) 19.4 ms, with 32 768: 12.0 ms (step 0.0317 -> 0.024 s); configs[2] 1.38 ->
// 1.02 ms; configs[3] 163.6 -> 157 ms; 131 072 the same, 524 288 slower again (the prefix contexts are rebuilt per split).
#ifndef BEAM_EXPAND_BLOCKS
#define BEAM_EXPAND_BLOCKS 32768
#endif
constexpr int64_t kBeamExpandBlocks = BEAM_EXPAND_BLOCKS;

// context table + masks of the two-kernel order-3 expansion (null: the fused kernel)
struct SplitWs {
    float *ctxtab = nullptr;
    unsigned long long *masks = nullptr;
    int64_t cap = 0;
};

template <int SCALE, int LEVEL>
static void launch_expand(const BeamMesh &M, const BeamClusters &C, bool clustered, const BeamEntry *in, int64_t n_in,
                          float u, unsigned long long *out, int64_t cap, unsigned long long *count, hipStream_t s,
                          const RxAll &rxall = RxAll{{0, 0, 0}, {0, 0, 0}, 0, 0.0f, {0, 0, 0}, {0, 0, 0}}, BeamDev dv = BeamDev{nullptr, nullptr},
                          bool last = false, SplitWs sw = SplitWs{}) {
    if constexpr (LEVEL <= 2 && SCALE != 2) {  // (shape 2, the rare two-arbitrary-triangles quad, stays on the fused kernel)
        if (clustered && sw.ctxtab && (rxall.on || (last && dv.dyn))) {
            // two launches per chunk of at most sw.cap prefixes: box stage (lane = prefix) -> contexts + masks, then the
            // per-primitive stage (lane = primitive, prefix through the scalar unit); the chunks append to one record list
            for (int64_t i0 = 0; i0 < n_in; i0 += sw.cap) {
                const int64_t n = std::min(sw.cap, n_in - i0);
                const int64_t bx = ceil_div(n, kExpandWG);
                int64_t by = ceil_div(kBeamExpandBlocks * 128 / kExpandWG, bx);
                if (by > C.nclusters) by = C.nclusters;
                if (by > 65535) by = 65535;
                if (by < 1) by = 1;
                const int64_t cps = ceil_div(C.nclusters, by);
                by = ceil_div(C.nclusters, cps);
                BeamDev dc = dv;
                dc.n_off = dv.n_off + i0;
                hipLaunchKernelGGL((beam_boxes_kernel<SCALE, LEVEL>), dim3((unsigned)bx, (unsigned)by), dim3(kExpandWG), 0, s, M, C,
                                   in + i0, n, u, sw.ctxtab, sw.masks, cps, rxall, dc);
                hipLaunchKernelGGL((beam_expand_pairs_kernel<SCALE, LEVEL>), dim3((unsigned)bx, (unsigned)by), dim3(kExpandWG), 0, s, M,
                                   C, sw.ctxtab, sw.masks, n, u, out, cap, count, cps, rxall, dc, i0);
            }
            return;
        }
    }
    if (clustered) {
        const int64_t bx = ceil_div(n_in, kExpandWG);
        int64_t by = ceil_div(kBeamExpandBlocks * 128 / kExpandWG, bx);  // few prefixes: split the cluster range so that the launch fills the chip
        if (by > C.nclusters) by = C.nclusters;
        if (by > 65535) by = 65535;
        if (by < 1) by = 1;
        const int64_t cps = ceil_div(C.nclusters, by);
        by = ceil_div(C.nclusters, cps);
        if (rxall.on || (last && dv.dyn)) {
            if constexpr (SCALE == 1)
                hipLaunchKernelGGL((beam_expand_clustered_last_kernel_s1<LEVEL>), dim3((unsigned)bx, (unsigned)by), dim3(kExpandWG),
                                   0, s, M, C, in, n_in, u, out, cap, count, cps, rxall, dv);
            else if constexpr (SCALE == 2)
                hipLaunchKernelGGL((beam_expand_clustered_last_kernel_s2<LEVEL>), dim3((unsigned)bx, (unsigned)by), dim3(kExpandWG),
                                   0, s, M, C, in, n_in, u, out, cap, count, cps, rxall, dv);
            else
                hipLaunchKernelGGL((beam_expand_clustered_last_kernel_q4<LEVEL>), dim3((unsigned)bx, (unsigned)by), dim3(kExpandWG),
                                   0, s, M, C, in, n_in, u, out, cap, count, cps, rxall, dv);
        } else {
            hipLaunchKernelGGL((beam_expand_clustered_kernel<SCALE, LEVEL>), dim3((unsigned)bx, (unsigned)by), dim3(kExpandWG), 0,
                               s, M, C, in, n_in, u, out, cap, count, cps, dv);
        }
    } else {
        const int64_t bx = ceil_div(n_in, 256), tiles = ceil_div(M.nprim, kBeamTile);
        int64_t by = ceil_div(2048, bx);
        if (by > tiles) by = tiles;
        if (by > 65535) by = 65535;
        if (by < 1) by = 1;
        const int64_t pps = ceil_div(tiles, by) * kBeamTile;  // whole tiles per split
        by = ceil_div(M.nprim, pps);
        hipLaunchKernelGGL((beam_expand_kernel<SCALE, LEVEL>), dim3((unsigned)bx, (unsigned)by), dim3(256), 0, s, M, in,
                           n_in, u, out, cap, count, pps, dv);
    }
}

#ifndef BEAM_EMIT_BLOCKS
#define BEAM_EMIT_BLOCKS (256 * 16)
#endif
constexpr int64_t kBeamEmitBlocks = BEAM_EMIT_BLOCKS;  // workgroups of 256 (twice as many of 128) of the receiver stage

template <int SCALE, int ORDER>
static void launch_emit(const BeamMesh &M, bool clustered, const BeamEntry *in, const unsigned long long *rec,
                        int64_t n_in, const float *rx, const float *rx_sorted, const int32_t *rx_index,
                        const float *rx_boxes, int64_t nrx, float u, long long *rows, int64_t cap,
                        unsigned long long *count, unsigned long long *grazing, hipStream_t s,
                        BeamDev dv = BeamDev{nullptr, nullptr}) {
    // persistent grids: a few workgroups per CU slot, every wave walks many groups of prefixes
    if (clustered)
        hipLaunchKernelGGL((beam_emit_clustered_kernel<SCALE, ORDER>), dim3((unsigned)std::min<int64_t>(ceil_div(n_in, 128), kBeamEmitBlocks * 2)),
                           dim3(128), 0, s, M, in, rec, n_in, rx_sorted, rx_index, rx_boxes, nrx, u, rows, cap, count, grazing, dv);
    else
        hipLaunchKernelGGL((beam_emit_kernel<SCALE, ORDER>), dim3((unsigned)std::min<int64_t>(ceil_div(n_in, 256), kBeamEmitBlocks)),
                           dim3(256), 0, s, M, in, rec, n_in, rx_sorted, rx_index, rx_boxes, nrx, u, rows, cap, count, grazing, dv);
}

#define BEAM_DISPATCH2(SC, K, CALL) /* SC = primitive shape (BeamMesh::kind) */ \
    do {                            \
        if ((SC) == 4) {            \
            if ((K) == 1) CALL(4, 1); else if ((K) == 2) CALL(4, 2); else CALL(4, 3); \
        } else if ((SC) == 2) {     \
            if ((K) == 1) CALL(2, 1); else if ((K) == 2) CALL(2, 2); else CALL(2, 3); \
        } else {                    \
            if ((K) == 1) CALL(1, 1); else if ((K) == 2) CALL(1, 2); else CALL(1, 3); \
        }                           \
    } while (0)

}  // namespace

#if defined(DRT_LAB) && defined(BEAM_LAB_COUNT)
extern "C" void drt_debug_beam_counts(unsigned long long *out, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(drt::beam_dbg), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(drt::beam_dbg), z, sizeof(z));
    }
}
#endif

extern "C" {

// A triangle mesh is searched over the pairing pass's primitives when at least this fraction of its triangles found a
// partner: a single triangle costs more as a degenerate quad (4 vertices x 4 faces per pyramid test instead of 3 x 3),
// and the lists shrink like (P / T)^level.
constexpr double kPairMinFraction = 0.30;

// The pairing pass (once per mesh): candidate successors on the GPU (pair_candidate_kernel), a matching along the fans
// on the host (chains A -> B -> C ... around a shared first vertex: heads first, then whatever is left -- cycles), the
// primitive table and the virtual mesh back on the GPU.
static int32_t pair_triangles(drt_mesh_t mesh, hipStream_t s) {
    // pair_state is written only once the outcome is KNOWN (0: examined, too few pairs; 1: paired): an allocation that fails
    // on the way leaves the handle "not run" (-1), so that a later call tries again instead of silently searching triangle by
    // triangle for the rest of the handle's life
    mesh->pair_state = -1;
    const int64_t T = mesh->num_triangles;
    if (T < 2 || mesh->assume_quads) {
        mesh->pair_state = 0;
        return DRT_OK;
    }
    const size_t a8 = align_up((size_t)T * 8, 256), a4 = align_up((size_t)T * 4, 256);
    size_t tb = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tb, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (size_t)T, 0, 64, nullptr);
    char *tmp = nullptr;
    DRT_HIP(hipMalloc(&tmp, 2 * a8 + 3 * a4 + align_up(tb, 256)));
    auto *hash = reinterpret_cast<uint64_t *>(tmp), *hash_sorted = reinterpret_cast<uint64_t *>(tmp + a8);
    auto *ids = reinterpret_cast<uint32_t *>(tmp + 2 * a8), *ids_sorted = reinterpret_cast<uint32_t *>(tmp + 2 * a8 + a4);
    auto *succ_dev = reinterpret_cast<int32_t *>(tmp + 2 * a8 + 2 * a4);
    char *sort_tmp = tmp + 2 * a8 + 3 * a4;
    const dim3 grid((unsigned)ceil_div(T, 256));
    std::vector<int32_t> succ((size_t)T);
    hipLaunchKernelGGL(pair_keys_kernel, grid, dim3(256), 0, s, mesh->tri_verts, T, hash, ids);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(sort_tmp, tb, hash, hash_sorted, ids, ids_sorted, (size_t)T, 0, 64, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pair_candidate_kernel, grid, dim3(256), 0, s, mesh->tri_verts, mesh->normals,
                           mesh->has_mask ? mesh->mask : nullptr, T, hash_sorted, ids_sorted, succ_dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(succ.data(), succ_dev, (size_t)T * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    if (e != hipSuccess) {
        mesh->pair_state = -1;
        return fail(DRT_E_HIP, "triangle pairing pass failed: %s", hipGetErrorString(e));
    }
    std::vector<int32_t> partner((size_t)T, -1);  // of a FIRST triangle: its second; of a second triangle: -2
    std::vector<uint8_t> has_pred((size_t)T, 0);
    for (int64_t a = 0; a < T; ++a)
        if (succ[a] >= 0) has_pred[succ[a]] = 1;
    int64_t quads = 0;
    auto walk = [&](int64_t a) {
        while (a >= 0 && partner[a] == -1) {
            const int64_t b = succ[a];
            if (b < 0 || partner[b] != -1) break;
            partner[a] = (int32_t)b;
            partner[b] = -2;
            ++quads;
            a = succ[b];
        }
    };
    for (int64_t a = 0; a < T; ++a)
        if (!has_pred[a]) walk(a);
    for (int64_t a = 0; a < T; ++a) walk(a);
    if ((double)(2 * quads) < kPairMinFraction * (double)T) {
        mesh->pair_state = 0;  // examined, too few pairs: triangle by triangle
        return DRT_OK;
    }
    const int64_t P = T - quads;
    std::vector<int32_t> table((size_t)(2 * P));
    int64_t p = 0;
    for (int64_t t = 0; t < T; ++t) {
        if (partner[t] == -2) continue;
        table[2 * p] = (int32_t)t;
        table[2 * p + 1] = partner[t];  // -1: a single triangle
        ++p;
    }
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += align_up(bytes, 256);
        return at;
    };
    const size_t o_tri = take((size_t)P * 8), o_tv = take((size_t)P * 72), o_n = take((size_t)P * 24), o_s = take((size_t)P * 8),
                 o_m = take((size_t)P * 2);
    char *blob = nullptr;
    DRT_HIP(hipMalloc(&blob, off));
    auto *pair_tri = reinterpret_cast<int32_t *>(blob + o_tri);
    e = hipMemcpyAsync(pair_tri, table.data(), (size_t)P * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pair_gather_kernel, dim3((unsigned)ceil_div(P, 256)), dim3(256), 0, s, mesh->tri_verts, mesh->normals,
                           mesh->shape, mesh->has_mask ? mesh->mask : nullptr, pair_tri, P,
                           reinterpret_cast<float *>(blob + o_tv), reinterpret_cast<float *>(blob + o_n),
                           reinterpret_cast<float *>(blob + o_s),
                           mesh->has_mask ? reinterpret_cast<uint8_t *>(blob + o_m) : nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // (`table` is read by the copy until here)
    if (e != hipSuccess) {
        (void)hipFree(blob);
        mesh->pair_state = -1;
        return fail(DRT_E_HIP, "triangle pairing pass failed: %s", hipGetErrorString(e));
    }
    mesh->pair_blob = blob;
    mesh->pair_tri = pair_tri;
    mesh->pair_tv = reinterpret_cast<float *>(blob + o_tv);
    mesh->pair_normals = reinterpret_cast<float *>(blob + o_n);
    mesh->pair_shape = reinterpret_cast<float *>(blob + o_s);
    mesh->pair_mask = reinterpret_cast<uint8_t *>(blob + o_m);
    mesh->pair_prims = P;
    mesh->pair_quads = quads;
    mesh->pair_state = 1;
    return DRT_OK;
}

// Clusters for the search's primitives -- slot 0: the caller's (triangles, or the quads of an assume_quads mesh);
// slot 1 (`allow_pairs`, a triangle mesh, enough pairs): the primitives of the pairing pass.  Each slot is built once
// and kept until drt_mesh_destroy: NOT thread-safe against other calls on the same handle while it builds (header).
// `*slot_out`: the slot a search with these flags uses.
static int32_t build_beam_clusters(drt_mesh_t mesh, bool allow_pairs, void *stream, int *slot_out = nullptr) {
    DRT_REQUIRE(mesh, "mesh is null");
    hipStream_t s = as_stream(stream);
    if (mesh->assume_quads && mesh->beam_quad4 < 0) {  // examine the quads once per mesh
        mesh->beam_quad4 = 0;
        const int64_t T = mesh->num_triangles;
        if (T >= 2) {
            uint32_t *flag = nullptr, h = 0;
            DRT_HIP(hipMalloc(&flag, 4));
            const uint32_t one = 1;
            hipError_t e = hipMemcpyAsync(flag, &one, 4, hipMemcpyHostToDevice, s);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(quad_shape_kernel, dim3((unsigned)ceil_div(T / 2, 256)), dim3(256), 0, s, mesh->tri_verts,
                                   mesh->normals, T / 2, flag);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            (void)hipFree(flag);
            if (e != hipSuccess) {
                mesh->beam_quad4 = -1;
                return fail(DRT_E_HIP, "quad shape check failed: %s", hipGetErrorString(e));
            }
            mesh->beam_quad4 = h ? 1 : 0;
        }
    }
    if (!mesh->assume_quads && allow_pairs && mesh->pair_state < 0) {
        const int32_t rcp = pair_triangles(mesh, s);
        if (rcp != DRT_OK) return rcp;
    }
    const bool pairs = !mesh->assume_quads && allow_pairs && mesh->pair_state == 1;
    const int slot = pairs ? 1 : 0;
    if (slot_out) *slot_out = slot;
    const BeamMesh M = pairs ? beam_mesh_pairs(mesh) : beam_mesh(mesh);
    drt_mesh::BeamCache &B = mesh->beam[slot];
    if (B.blob && B.kind == M.kind) return DRT_OK;
    if (M.nprim == 0) return DRT_OK;
    const int64_t ncl = ceil_div(M.nprim, 64), pp = ncl * 64, sc = M.scale;
    const int64_t nv = (M.kind == 1) ? 3 : (M.kind == 2 ? 6 : 4), np = (M.kind == 2) ? 2 : 1;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += align_up(bytes, 256);
        return at;
    };
    const size_t o_order = take((size_t)M.nprim * 4), o_verts = take((size_t)pp * 12 * nv),
                 o_planes = take((size_t)pp * 16 * np), o_uplanes = take((size_t)pp * 16 * np),
                 o_sigma = take((size_t)pp * 4), o_boxes = take((size_t)ncl * 32), o_subboxes = take((size_t)ncl * 96);
    char *blob = nullptr, *tmp = nullptr;
    DRT_HIP(hipMalloc(&blob, off));
    const size_t tmp_bytes = morton_scratch_bytes(M.nprim);
    if (hipMalloc(&tmp, tmp_bytes) != hipSuccess) {
        (void)hipFree(blob);
        return fail(DRT_E_HIP, "hipMalloc of %zu bytes failed", tmp_bytes);
    }
    uint32_t *ids = nullptr, *bounds = nullptr;
    int32_t rc = morton_order(M.tv, M.nprim, 3 * (int32_t)sc, tmp, &ids, &bounds, s);
    if (rc == DRT_OK) {
        auto *order = reinterpret_cast<int32_t *>(blob + o_order);
        auto *verts = reinterpret_cast<float *>(blob + o_verts);
        auto *planes = reinterpret_cast<float *>(blob + o_planes);
        auto *uplanes = reinterpret_cast<float *>(blob + o_uplanes);
        auto *sigma = reinterpret_cast<float *>(blob + o_sigma);
        auto *boxes = reinterpret_cast<float *>(blob + o_boxes);
        auto *subboxes = reinterpret_cast<float *>(blob + o_subboxes);
        if (M.kind == 4)
            hipLaunchKernelGGL(prim_cluster_kernel<4>, dim3((unsigned)ncl), dim3(64), 0, s, M, ids, order, verts, planes, uplanes, sigma, boxes, subboxes);
        else if (M.kind == 2)
            hipLaunchKernelGGL(prim_cluster_kernel<2>, dim3((unsigned)ncl), dim3(64), 0, s, M, ids, order, verts, planes, uplanes, sigma, boxes, subboxes);
        else
            hipLaunchKernelGGL(prim_cluster_kernel<1>, dim3((unsigned)ncl), dim3(64), 0, s, M, ids, order, verts, planes, uplanes, sigma, boxes, subboxes);
        uint32_t mag_bits = 0;
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&mag_bits, bounds + 6, 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = fail(DRT_E_HIP, "beam cluster build failed: %s", hipGetErrorString(e));
        if (rc == DRT_OK) {
            float mag;
            memcpy(&mag, &mag_bits, 4);
            mesh->beam_max_abs = mag;  // (the same vertices in either slot)
            B.order = order;
            B.verts = verts;
            B.planes = planes;
            B.uplanes = uplanes;
            B.sigma = sigma;
            B.boxes = boxes;
            B.subboxes = subboxes;
            B.clusters = ncl;
            B.kind = M.kind;
            B.blob = blob;
        }
    }
    (void)hipFree(tmp);
    if (rc != DRT_OK) (void)hipFree(blob);
    return rc;
}

int32_t drt_mesh_build_beam_clusters(drt_mesh_t mesh, void *stream) { return build_beam_clusters(mesh, true, stream); }
int32_t drt_mesh_build_beam_clusters_ex(drt_mesh_t mesh, int32_t allow_pairs, void *stream) {
    return build_beam_clusters(mesh, allow_pairs != 0, stream);
}
int32_t drt_mesh_beam_pairing_table(drt_mesh_t mesh, int32_t *table_out, int64_t num_primitives, void *stream) {
    DRT_REQUIRE(mesh && mesh->pair_state == 1, "the mesh is not in pair mode (drt_mesh_beam_pairing)");
    DRT_REQUIRE(table_out && num_primitives == mesh->pair_prims, "table_out must hold [num_primitives, 2] int32");
    DRT_HIP(hipMemcpyAsync(table_out, mesh->pair_tri, (size_t)num_primitives * 8, hipMemcpyDeviceToDevice, as_stream(stream)));
    return DRT_OK;
}
int32_t drt_mesh_beam_pairing(drt_mesh_t mesh, int64_t *num_primitives, int64_t *num_pairs) {
    DRT_REQUIRE(mesh, "mesh is null");
    if (num_primitives) *num_primitives = mesh->pair_state == 1 ? mesh->pair_prims : 0;
    if (num_pairs) *num_pairs = mesh->pair_state == 1 ? mesh->pair_quads : 0;
    return mesh->pair_state;
}

size_t drt_trace_beam_workspace_size(int64_t num_tx, int64_t num_rx, int64_t num_primitives, int32_t order,
                                     const drt_beam_params *bp, int64_t max_paths) {
    if (num_tx < 0) num_tx = 0;
    if (num_rx < 0) num_rx = 0;
    if (num_primitives < 0) num_primitives = 0;
    if (max_paths < 0) max_paths = 0;
    if (order < 0) order = 0;
    if (order > 3) order = 3;
    if (order == 0) return drt_trace_compact_workspace_size(num_tx * num_rx, max_paths);  // line of sight: plain trace
    return beam_layout(beam_sizes(bp, num_tx, num_rx, num_primitives, order), num_tx, num_rx, num_primitives, order, max_paths).total;
}

int32_t drt_trace_paths_beam(drt_mesh_t mesh, const drt_trace_params *pr, const drt_beam_params *bp, const float *tx,
                             int64_t ntx, const float *rx, int64_t nrx, int32_t order, int64_t max_paths,
                             int64_t *keys, float *vertices, int32_t *objects, int64_t *num_valid_host, void *ws,
                             size_t ws_bytes, void *stream) {
    DRT_REQUIRE(mesh && pr && num_valid_host, "null argument");
    *num_valid_host = 0;
    DRT_REQUIRE(ntx >= 0 && nrx >= 0 && max_paths >= 0, "negative size");
    DRT_REQUIRE(order >= 0 && order <= 3, "beam pruning covers orders 0..3");
    DRT_REQUIRE(ntx < (1ll << 30) && nrx < (1ll << 31), "too many transmitters / receivers");
    const float kappa = (bp && bp->kappa > 0.0f) ? bp->kappa : margins::kKappaDefault;
    const int32_t flags = bp ? bp->flags : 0;
    const int64_t shard_world = (bp && bp->shard_world > 1) ? bp->shard_world : 1;
    const int64_t shard_rank = bp ? bp->shard_rank : 0;
    DRT_REQUIRE(shard_rank >= 0 && shard_rank < shard_world, "shard_rank must be in [0, shard_world)");
    drt_beam_stats *st = bp ? bp->stats : nullptr;
    if (st) memset(st, 0, sizeof(*st));
    hipStream_t s = as_stream(stream);
    BeamMesh M = beam_mesh(mesh);  // the caller's view: triangles, or quads with assume_quads (sizes, keys, early outs)
    drt_trace_params tp = *pr;
    tp.stats = nullptr;

    if (order == 0) {  // line of sight: no prefix to prune; rank 0 of a sharded call owns it
        if (shard_rank != 0 || ntx == 0 || nrx == 0) return DRT_OK;
        drt_candidates c{};
        c.num_candidates = 1;
        c.num_nodes = M.nprim > 0 ? M.nprim : 1;
        c.order = 0;
        const size_t need = drt_trace_compact_workspace_size(ntx * nrx, max_paths);
        if (!ws || ws_bytes < need) return fail(DRT_E_CAPACITY, "workspace too small: need %zu bytes", need);
        const int32_t rc0 = drt_trace_paths_compact(mesh, &tp, tx, ntx, rx, nrx, &c, ntx * nrx, max_paths, keys, vertices,
                                                    objects, num_valid_host, ws, ws_bytes, stream);
        if (st) st->valid = *num_valid_host;
        if (rc0 == DRT_E_CAPACITY && *num_valid_host > max_paths)  // same wording as the pruned orders: callers regrow on it
            return fail(DRT_E_CAPACITY, "more than %lld valid paths: raise max_paths", (long long)max_paths);
        return rc0;
    }
    if (M.nprim == 0 || ntx == 0 || nrx == 0) return DRT_OK;
    DRT_REQUIRE(tx && rx, "null pointer");
    DRT_REQUIRE(max_paths == 0 || (keys && vertices && objects), "null output");
    unsigned __int128 total = (unsigned __int128)ntx * (unsigned __int128)nrx;
    unsigned long long npow = 1;
    for (int j = 0; j < order; ++j) {
        total *= (unsigned __int128)M.nprim;
        npow *= (unsigned long long)M.nprim;
    }
    if (total >= ((unsigned __int128)1 << 62))
        return fail(DRT_E_OVERFLOW, "tx * rx * primitives^order does not fit a 62-bit row key");
    int key_bits = 1;
    while (key_bits < 64 && ((unsigned __int128)1 << key_bits) < total) ++key_bits;

    const int64_t nprim_caller = M.nprim;  // the workspace query was made with this count
    const BeamSizes z = beam_sizes(bp, ntx, nrx, nprim_caller, order);
    const BeamLayout L = beam_layout(z, ntx, nrx, nprim_caller, order, max_paths);
    if (!ws || ws_bytes < L.total) return fail(DRT_E_CAPACITY, "workspace too small: need %zu bytes", L.total);
    DRT_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15u) == 0, "workspace must be 16-byte aligned");
    char *base = reinterpret_cast<char *>(ws);
    auto *counters = reinterpret_cast<unsigned long long *>(base + L.counters);  // [0] list count, [1] grazing prefixes
    auto *rx_sorted = reinterpret_cast<float *>(base + L.rx_sorted);
    auto *rx_index = reinterpret_cast<int32_t *>(base + L.rx_index);
    auto *rx_boxes = reinterpret_cast<float *>(base + L.rx_boxes);
    auto *entries1 = reinterpret_cast<BeamEntry *>(base + L.entries1);
    auto *entries2 = reinterpret_cast<BeamEntry *>(base + L.entries2);
    auto *records = reinterpret_cast<unsigned long long *>(base + L.records);
    auto *rows = reinterpret_cast<long long *>(base + L.rows);
    auto *rows_sorted = reinterpret_cast<unsigned long long *>(base + L.rows_sorted);
    auto *table = reinterpret_cast<int32_t *>(base + L.table);
    auto *pair_offsets = reinterpret_cast<long long *>(base + L.pair_offsets);
    char *sort_tmp = base + L.sort_tmp;
    auto *slice_keys = reinterpret_cast<long long *>(base + L.slice_keys);

    int slot = 0;
    int32_t rc = build_beam_clusters(mesh, !(flags & DRT_BEAM_NO_PAIRS), stream, &slot);
    if (rc != DRT_OK) return rc;
    // Coplanar-pair mode: a triangle mesh whose triangles (2i, 2i+1) are the same mirror is searched over its n/2
    // pairs with the quad kernels (pyramid of a pair = union of its triangles' pyramids: what keeps a triangle
    // sequence keeps its pair sequence; a pair may follow itself), a quarter of the level-2 prefixes of a box city;
    // every surviving pair row is then split into its 2^order triangle rows, which the exact trace decides.
    const bool pairs = slot == 1;
    int key_bits_rows = key_bits;  // bits of the keys the row sort sees
    int64_t rows_cap = z.max_rows;
    M = pairs ? beam_mesh_pairs(mesh) : beam_mesh(mesh);  // (the shape of the pairs is known only now)
    if (pairs) {
        npow = 1;
        unsigned __int128 tq = (unsigned __int128)ntx * (unsigned __int128)nrx;
        for (int j = 0; j < order; ++j) {
            npow *= (unsigned long long)M.nprim;
            tq *= (unsigned __int128)M.nprim;
        }
        key_bits_rows = 1;
        while (key_bits_rows < 64 && ((unsigned __int128)1 << key_bits_rows) < tq) ++key_bits_rows;
        rows_cap = std::max<int64_t>(z.max_rows >> order, 1);  // the split needs 2^order table rows per pair row
    }
    const BeamClusters C = beam_clusters_of(mesh->beam[slot]);

    // receivers: Morton clusters (also yields the largest |coordinate| of the receivers); transmitters: bounds only
    uint32_t *rx_ids = nullptr, *rx_bounds = nullptr;
    rc = morton_order(rx, nrx, 1, base + L.morton, &rx_ids, &rx_bounds, s);
    if (rc != DRT_OK) return rc;
    hipLaunchKernelGGL(rx_cluster_kernel, dim3((unsigned)ceil_div(nrx, 64)), dim3(64), 0, s, rx, nrx, rx_ids, rx_sorted,
                       rx_index, rx_boxes);
    // (the transmitters' bounds in slots 8.. of the same scratch: the receivers' own box is needed below)
    uint32_t *tx_bounds = rx_bounds + 8;
    DRT_HIP(fill_bytes_async(tx_bounds, 0xff, 12, s));
    DRT_HIP(fill_bytes_async(tx_bounds + 3, 0, 20, s));
    hipLaunchKernelGGL(point_bounds_kernel, dim3((unsigned)ceil_div(ntx, 256)), dim3(256), 0, s, tx, ntx, 1, tx_bounds);
    DRT_LAUNCH_CHECK();
    uint32_t hb[16] = {0};
    DRT_HIP(hipMemcpyAsync(hb, rx_bounds, sizeof(hb), hipMemcpyDeviceToHost, s));
    DRT_HIP(hipStreamSynchronize(s));
    auto bits_to_float = [](uint32_t b) {
        float f;
        memcpy(&f, &b, 4);
        return f;
    };
    auto ordered_to_float_host = [&](uint32_t v) { return bits_to_float((v & 0x80000000u) ? (v & 0x7fffffffu) : ~v); };
    float mag = std::max(bits_to_float(hb[6]), bits_to_float(hb[14]));
    mag = std::max(std::max(mag, mesh->beam_max_abs), 1e-30f);
    // box of all receivers for the last expansion's child filter: off unless every receiver is finite
    RxAll rxall{{0, 0, 0}, {0, 0, 0}, 0, 0.0f};
    if (hb[7] == 0u && !(flags & DRT_BEAM_EXPAND_PLAIN)) {
        bool ok = true;
        for (int k = 0; k < 3; ++k) {
            rxall.lo[k] = ordered_to_float_host(hb[k]);
            rxall.hi[k] = ordered_to_float_host(hb[3 + k]);
            ok = ok && std::isfinite(rxall.lo[k]) && std::isfinite(rxall.hi[k]) && rxall.lo[k] <= rxall.hi[k];
        }
        rxall.on = ok ? 1 : 0;
    }
    int ex = 0;
    (void)std::frexp(mag, &ex);                  // mag = f * 2^ex, f in [0.5, 1)
    const float u = kappa * std::ldexp(1.0f, ex - 1 - 23);  // kappa * ulp(M)
    M.inv_2m = 0.5f / mag;
    rxall.ulp_m = std::ldexp(1.0f, ex - 1 - 23);
    rx_all_finish(rxall);
    if (st) {
        st->unit_m = u;
        st->magnitude = mag;
    }
    const bool expand_clustered = !(flags & DRT_BEAM_EXPAND_PLAIN);
    const bool emit_clustered = (flags & DRT_BEAM_EMIT_CLUSTERED) || (!(flags & DRT_BEAM_EMIT_PLAIN) && nrx >= 128);
    const bool pair_blocks = !(flags & DRT_BEAM_ROWS_PLAIN);
    SplitWs split_ws;  // orders 2, 3: the last expansion as two kernels (DRT_BEAM_EXPAND_FUSED: the single fused kernel)
    if (order >= 2 && z.ctx_cap > 0 && !(flags & DRT_BEAM_EXPAND_FUSED))
        split_ws = SplitWs{reinterpret_cast<float *>(base + L.ctx_table), reinterpret_cast<unsigned long long *>(base + L.ctx_masks), z.ctx_cap};

    DRT_HIP(fill_bytes_async(counters, 0, 256, s));
    // ---- level 1 ----
    hipLaunchKernelGGL(beam_seed_kernel, dim3((unsigned)ceil_div(ntx * M.nprim, 256)), dim3(256), 0, s, M, tx, ntx, u,
                       shard_rank, shard_world, entries1, ntx * M.nprim, counters, BeamDev{nullptr, nullptr});
    DRT_LAUNCH_CHECK();
    int64_t ncur = 0;
    rc = read_count(counters, &ncur, s);
    if (rc != DRT_OK) return rc;
    if (st) st->levels[0] = ncur;
    DRT_REQUIRE(ncur < (1ll << 32), "record format holds 32-bit prefix indices");

    BeamTimer t_expand, t_emit, t_trace;
    t_expand.init(st != nullptr, s, st ? &st->expand_last_ms : nullptr);
    t_emit.init(st != nullptr, s, st ? &st->emit_ms : nullptr);
    t_trace.init(st != nullptr, s, st ? &st->trace_ms : nullptr);
    // ---- last level: rows of a slice of prefixes -> sort -> table -> trace ----
    int64_t nvalid = 0, slices_with_paths = 0;
    auto process = [&](const BeamEntry *src, const unsigned long long *rec, int64_t nsrc, int64_t *rows_out,
                       bool *fits) -> int32_t {
        *fits = true;
        *rows_out = 0;
        if (nsrc == 0) return DRT_OK;
        DRT_HIP(fill_bytes_async(counters, 0, 8, s));
        t_emit.start();
#define CALL(SC, K) launch_emit<SC, K>(M, emit_clustered, src, rec, nsrc, rx, rx_sorted, rx_index, rx_boxes, nrx, u, rows, rows_cap, counters, counters + 1, s)
        BEAM_DISPATCH2(M.kind, order, CALL);
#undef CALL
        t_emit.stop();
        DRT_LAUNCH_CHECK();
        int64_t r = 0;
        int32_t rc2 = read_count(counters, &r, s);
        if (rc2 != DRT_OK) return rc2;
        t_emit.collect();
        t_expand.collect();
        *rows_out = r;  // in the unit of rows_cap: pair rows in coplanar-pair mode
        if (r > rows_cap) {
            *fits = false;
            return DRT_OK;
        }
        if (r == 0) return DRT_OK;
        t_trace.start();
        size_t tb = sort_keys64_temp_bytes(r);
        DRT_HIP(rocprim::radix_sort_keys(sort_tmp, tb, reinterpret_cast<unsigned long long *>(rows), rows_sorted, (size_t)r,
                                         0, key_bits_rows, s));
        int64_t table_rows = r;
        const unsigned long long *row_keys = rows_sorted;  // table row -> packed key the caller gets
        if (pairs) {
            // every sorted pair row -> its 2^order triangle rows (table) and their triangle-level keys (into `rows`,
            // whose unsorted content is no longer needed); pairs stay grouped, the order inside a pair is fixed later
            table_rows = r << order;
            const dim3 ge((unsigned)ceil_div(table_rows, 256));
            auto *tri_keys = reinterpret_cast<unsigned long long *>(rows);
            if (order == 1) hipLaunchKernelGGL(rows_expand_pairs_kernel<1>, ge, dim3(256), 0, s, rows_sorted, r, (unsigned long long)M.nprim, M.pair_tri, (unsigned long long)mesh->num_triangles, table, tri_keys);
            else if (order == 2) hipLaunchKernelGGL(rows_expand_pairs_kernel<2>, ge, dim3(256), 0, s, rows_sorted, r, (unsigned long long)M.nprim, M.pair_tri, (unsigned long long)mesh->num_triangles, table, tri_keys);
            else hipLaunchKernelGGL(rows_expand_pairs_kernel<3>, ge, dim3(256), 0, s, rows_sorted, r, (unsigned long long)M.nprim, M.pair_tri, (unsigned long long)mesh->num_triangles, table, tri_keys);
            row_keys = tri_keys;
        } else {
            const dim3 gr((unsigned)ceil_div(r, 256));
            if (order == 1) hipLaunchKernelGGL(rows_decode_kernel<1>, gr, dim3(256), 0, s, rows_sorted, r, (unsigned long long)M.nprim, M.scale, table);
            else if (order == 2) hipLaunchKernelGGL(rows_decode_kernel<2>, gr, dim3(256), 0, s, rows_sorted, r, (unsigned long long)M.nprim, M.scale, table);
            else hipLaunchKernelGGL(rows_decode_kernel<3>, gr, dim3(256), 0, s, rows_sorted, r, (unsigned long long)M.nprim, M.scale, table);
        }
        hipLaunchKernelGGL(pair_offsets_kernel, dim3((unsigned)ceil_div(ntx * nrx + 1, 256)), dim3(256), 0, s, rows_sorted, r,
                           npow, ntx * nrx, pair_offsets, pairs ? ((int64_t)1 << order) : (int64_t)1);
        DRT_LAUNCH_CHECK();
        drt_candidates c{};
        c.table = table;
        c.num_candidates = table_rows;
        c.order = order;
        c.pair_offsets = reinterpret_cast<const int64_t *>(pair_offsets);
        if (pairs && pair_blocks) c.reserved |= DRT_CAND_PAIR_BLOCKS;  // the filter stage evaluates each pair row's chain once
        int64_t nv = 0;
        const int64_t k2 = order + 2;
        rc2 = drt_trace_paths_compact(mesh, &tp, tx, ntx, rx, nrx, &c, z.max_survivors, max_paths - nvalid,
                                      reinterpret_cast<int64_t *>(slice_keys), vertices ? vertices + nvalid * k2 * 3 : nullptr,
                                      objects ? objects + nvalid * k2 : nullptr, &nv, base + L.trace_ws, L.trace_ws_bytes, stream);
        t_trace.stop();
        if (t_trace.on) (void)hipStreamSynchronize(s);  // (the trace synchronised already; the stop event is the only thing in flight)
        t_trace.collect();
        if (rc2 == DRT_E_CAPACITY && nv > z.max_survivors) {
            // the survivor queue of the trace overflowed (it reports the survivor count, which a count of valid
            // paths can never exceed): the slice is too large, like one whose rows do not fit
            *fits = false;
            *rows_out = nv;
            return DRT_OK;
        }
        if (rc2 == DRT_E_CAPACITY && nv > max_paths - nvalid) {
            *num_valid_host = nvalid + nv;
            return fail(DRT_E_CAPACITY, "more than %lld valid paths: raise max_paths", (long long)max_paths);
        }
        if (rc2 != DRT_OK) {
            *num_valid_host = nv;
            return rc2;
        }
        if (nv > 0) {
            hipLaunchKernelGGL(keys_to_rows_kernel, dim3((unsigned)ceil_div(nv, 256)), dim3(256), 0, s, slice_keys, nv,
                               row_keys, reinterpret_cast<long long *>(keys) + nvalid);
            DRT_LAUNCH_CHECK();
            nvalid += nv;
            ++slices_with_paths;
        }
        return DRT_OK;
    };

    int64_t total_rows = 0, nslices = 0;
    // the last expansion of the list `cur` (level order-1) in slices sized from the measured fan-out (a small probe
    // slice first), so that the records of a slice and its rows fit their buffers; a slice that overflows is
    // retried smaller.  LEVEL = order - 1.
    int64_t step_hint = (bp && bp->probe_prefixes > 0) ? bp->probe_prefixes : 4096;  // carried from list to list
    double hint_min = 4e18;  // smallest slice size any slice of this call suggested: what the NEXT call can start with
    int64_t done = 0;
    auto last_expansion = [&](const BeamEntry *cur, int64_t ncur, int64_t *records_total) -> int32_t {
        int64_t i0 = 0;
        int64_t step = std::max<int64_t>(std::min<int64_t>(step_hint, ncur), 1);
        while (i0 < ncur) {
            const int64_t i1 = std::min(i0 + step, ncur);
            DRT_HIP(fill_bytes_async(counters, 0, 8, s));
            t_expand.start();
            if (order == 2) {
#define CALL(SC, K) launch_expand<SC, 1>(M, C, expand_clustered, cur + i0, i1 - i0, u, records, z.max_records, counters, s, rxall, BeamDev{nullptr, nullptr}, true, split_ws)
                BEAM_DISPATCH2(M.kind, 1, CALL);
#undef CALL
            } else {
#define CALL(SC, K) launch_expand<SC, 2>(M, C, expand_clustered, cur + i0, i1 - i0, u, records, z.max_records, counters, s, rxall, BeamDev{nullptr, nullptr}, true, split_ws)
                BEAM_DISPATCH2(M.kind, 1, CALL);
#undef CALL
            }
            t_expand.stop();
            DRT_LAUNCH_CHECK();
            int64_t c = 0, r = 0;
            int32_t rc2 = read_count(counters, &c, s);
            if (rc2 != DRT_OK) return rc2;
            t_expand.collect();
            bool fits = c <= z.max_records;
            if (fits) {
                rc2 = process(cur + i0, records, c, &r, &fits);
                if (rc2 != DRT_OK) return rc2;
            }
            if (!fits) {
                if (step == 1) {
                    *num_valid_host = std::max(c, r);
                    return fail(DRT_E_CAPACITY, "one prefix overflows max_records / max_rows / max_survivors (%lld records, %lld rows or survivors)",
                                (long long)c, (long long)r);
                }
                step = std::max<int64_t>(step / 4, 1);
                continue;
            }
            *records_total += c;
            total_rows += pairs ? (r << order) : r;  // rows handed to the tracer
            ++nslices;
            ++done;
            const double per = (double)(i1 - i0);
            // (rows also against the survivor queue of the trace: survivors <= rows, so half of it never overflows)
            // (pair mode: most of a pair row's 2^order triangle rows fail the first inside test -- the survivors are
            // bounded like the pair rows, and an overflow of that queue only makes the slice retry smaller)
            const double row_budget = (double)std::min(rows_cap, z.max_survivors);
            const double fan = std::max({(double)c / per / (double)z.max_records, (double)r / per / row_budget, 1e-18});
            double next = 0.5 / fan;
            hint_min = std::min(hint_min, next);
            if (done > 1) next = std::min(next, 4.0 * (double)step);
            step_hint = (int64_t)std::max(1.0, std::min(next, 4e18));
            step = std::min<int64_t>(step_hint, ncur);
            i0 = i1;
        }
        return DRT_OK;
    };

    if (order == 1) {
        bool fits = true;
        int64_t r = 0;
        rc = process(entries1, nullptr, ncur, &r, &fits);
        if (rc != DRT_OK) return rc;
        if (!fits) {
            *num_valid_host = r;
            return fail(DRT_E_CAPACITY, "%lld candidate rows or survivors: raise max_rows (%lld%s) / max_survivors (%lld)",
                        (long long)r, (long long)z.max_rows, pairs ? ", 2^order table rows per pair row" : "",
                        (long long)z.max_survivors);
        }
        total_rows = pairs ? (r << order) : r;
        nslices = 1;
    } else if (order == 2) {
        int64_t last = 0;
        rc = last_expansion(entries1, ncur, &last);
        if (rc != DRT_OK) return rc;
        if (st) st->levels[1] = last;
    } else {
        // order 3: the level-1 list in slices whose level-2 lists fit max_entries (sized from the measured fan-out
        // like the inner slices), each level-2 list then through the last expansion
        int64_t i0 = 0, level2 = 0, level3 = 0, outer_done = 0;
        const int64_t cap2 = std::min(z.max_entries, z.max_records);
        int64_t step = std::max<int64_t>(std::min<int64_t>(1024, ncur), 1);
        while (i0 < ncur) {
            const int64_t i1 = std::min(i0 + step, ncur);
            DRT_HIP(fill_bytes_async(counters, 0, 8, s));
#define CALL(SC, K) launch_expand<SC, 1>(M, C, expand_clustered, entries1 + i0, i1 - i0, u, records, cap2, counters, s)
            BEAM_DISPATCH2(M.kind, 1, CALL);
#undef CALL
            DRT_LAUNCH_CHECK();
            int64_t c2 = 0;
            rc = read_count(counters, &c2, s);
            if (rc != DRT_OK) return rc;
            if (c2 > cap2) {
                if (step == 1) {
                    *num_valid_host = c2;
                    return fail(DRT_E_CAPACITY, "one level-1 prefix has %lld children: raise max_records / max_entries",
                                (long long)c2);
                }
                step = std::max<int64_t>(step / 4, 1);
                continue;
            }
            if (c2 > 0) {
                hipLaunchKernelGGL(beam_finish_kernel<1>, dim3((unsigned)ceil_div(c2, 256)), dim3(256), 0, s, M, entries1 + i0,
                                   records, c2, u, entries2, BeamDev{nullptr, nullptr});
                DRT_LAUNCH_CHECK();
                rc = last_expansion(entries2, c2, &level3);
                if (rc != DRT_OK) return rc;
            }
            level2 += c2;
            ++outer_done;
            const double fan = std::max((double)c2 / (double)(i1 - i0) / (double)cap2, 1e-18);
            double next = 0.7 / fan;
            if (outer_done > 1) next = std::min(next, 8.0 * (double)step);
            step = (int64_t)std::max(1.0, std::min(next, (double)ncur));
            i0 = i1;
        }
        if (st) {
            st->levels[1] = level2;
            st->levels[2] = level3;
        }
    }

    // ---- slices interleave in key order: one final sort of (key, position) and a gather ----
    if ((slices_with_paths > 1 || pairs) && nvalid > 1) {  // (pair mode: a slice's rows are grouped by pair, not sorted by triangle key)
        const int64_t k2 = order + 2;
        auto *mk = reinterpret_cast<unsigned long long *>(base + L.merge_keys);
        auto *perm = reinterpret_cast<uint32_t *>(base + L.merge_perm);
        auto *iota = reinterpret_cast<uint32_t *>(base + L.merge_iota);
        auto *tmp_rows = reinterpret_cast<uint32_t *>(base + L.merge_rows);
        hipLaunchKernelGGL(iota_kernel, dim3((unsigned)ceil_div(nvalid, 256)), dim3(256), 0, s, iota, nvalid);
        size_t tb = sort_pairs64_temp_bytes(nvalid);
        DRT_HIP(rocprim::radix_sort_pairs(sort_tmp, tb, reinterpret_cast<unsigned long long *>(keys), mk, iota, perm,
                                          (size_t)nvalid, 0, key_bits, s));
        DRT_HIP(hipMemcpyAsync(keys, mk, (size_t)nvalid * 8, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)ceil_div(nvalid * k2 * 3, 256)), dim3(256), 0, s,
                           reinterpret_cast<const uint32_t *>(vertices), perm, nvalid, (int32_t)(k2 * 3), tmp_rows);
        DRT_HIP(hipMemcpyAsync(vertices, tmp_rows, (size_t)nvalid * k2 * 12, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)ceil_div(nvalid * k2, 256)), dim3(256), 0, s,
                           reinterpret_cast<const uint32_t *>(objects), perm, nvalid, (int32_t)k2, tmp_rows);
        DRT_HIP(hipMemcpyAsync(objects, tmp_rows, (size_t)nvalid * k2 * 4, hipMemcpyDeviceToDevice, s));
        DRT_LAUNCH_CHECK();
    }
    *num_valid_host = nvalid;
    if (st) {
        int64_t gz = 0;
        rc = read_count(counters + 1, &gz, s);
        if (rc != DRT_OK) return rc;
        st->grazing_prefixes = gz;
        st->rows = total_rows;
        st->pair_mode = pairs ? 1 : 0;
        st->paired_primitives = pairs ? (int32_t)std::min<int64_t>(mesh->pair_quads, 2147483647) : 0;
        st->slices = nslices;
        st->next_probe_prefixes = (hint_min < 4e18) ? (float)std::max(1.0, std::min(hint_min, 1073741824.0)) : 0.0f;
        st->valid = nvalid;
    }
    return DRT_OK;
}


// ---------------------------------------------------------------------------------------------
// The same search with STATIC shapes: nothing is read back, nothing is allocated, the launch sequence does not depend
// on device data -- what an XLA FFI handler / a HIP graph needs (reference boundary: wp.jax_callable(func,
// output_dims=...), geometry/_mesh.py:266-276, 3082-3092).  ONE pass with the caller's capacities instead of slices
// sized from read-back counts: every list is sized by drt_beam_params (or the scene-sized defaults), every kernel of a
// later stage is launched over the CAPACITY of its input list and takes the list's length from the device counter, the
// scene scalars (error unit, receivers' box) are computed by beam_dyn_kernel, rows beyond the count are sentinels.  An
// overflowing list is reported in counts_dev[2]; the rows written are then valid paths, but not all of them.
int32_t drt_trace_paths_beam_async(drt_mesh_t mesh, const drt_trace_params *pr, const drt_beam_params *bp,
                                   const float *tx, int64_t ntx, const float *rx, int64_t nrx, int32_t order,
                                   int64_t max_paths, int64_t *keys, float *vertices, int32_t *objects,
                                   int64_t *counts_dev, void *ws, size_t ws_bytes, void *stream) {
    DRT_REQUIRE(mesh && pr && counts_dev, "null argument");
    DRT_REQUIRE(ntx >= 0 && nrx >= 0 && max_paths >= 0, "negative size");
    DRT_REQUIRE(order >= 0 && order <= 3, "beam pruning covers orders 0..3");
    DRT_REQUIRE(ntx < (1ll << 30) && nrx < (1ll << 31), "too many transmitters / receivers");
    DRT_REQUIRE(max_paths == 0 || (keys && vertices && objects), "null output");
    const float kappa = (bp && bp->kappa > 0.0f) ? bp->kappa : margins::kKappaDefault;
    const int32_t flags = bp ? bp->flags : 0;
    const int64_t shard_world = (bp && bp->shard_world > 1) ? bp->shard_world : 1;
    const int64_t shard_rank = bp ? bp->shard_rank : 0;
    DRT_REQUIRE(shard_rank >= 0 && shard_rank < shard_world, "shard_rank must be in [0, shard_world)");
    hipStream_t s = as_stream(stream);
    BeamMesh M = beam_mesh(mesh);
    drt_trace_params tp = *pr;
    tp.stats = nullptr;
    const int64_t k2 = order + 2;

    if (order == 0) {  // line of sight: the plain static-shape trace (rank 0 of a sharded call owns it)
        drt_candidates c{};
        c.num_candidates = (shard_rank == 0) ? 1 : 0;
        c.num_nodes = M.nprim > 0 ? M.nprim : 1;
        c.order = 0;
        return drt_trace_paths_compact_async(mesh, &tp, tx, ntx, rx, nrx, &c, ntx * nrx, max_paths, keys, vertices, objects,
                                             counts_dev, ws, ws_bytes, stream);
    }
    DRT_REQUIRE((tx || ntx == 0) && (rx || nrx == 0), "null pointer");
    unsigned __int128 total = (unsigned __int128)ntx * (unsigned __int128)nrx;
    for (int j = 0; j < order; ++j) total *= (unsigned __int128)M.nprim;
    if (total >= ((unsigned __int128)1 << 62))
        return fail(DRT_E_OVERFLOW, "tx * rx * primitives^order does not fit a 62-bit row key");
    const int64_t nprim_caller = M.nprim;
    const BeamSizes z = beam_sizes(bp, ntx, nrx, nprim_caller, order);
    const BeamLayout L = beam_layout(z, ntx, nrx, nprim_caller, order, max_paths);
    if (!ws || ws_bytes < L.total) return fail(DRT_E_CAPACITY, "workspace too small: need %zu bytes", L.total);
    DRT_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15u) == 0, "workspace must be 16-byte aligned");
    // the slot a synchronous call with these flags uses (drt_mesh_build_beam_clusters examines the triangle pairs and
    // builds that one; drt_mesh_build_beam_clusters_ex(mesh, 0, ...) builds the triangle-by-triangle slot)
    const bool pairs = !mesh->assume_quads && !(flags & DRT_BEAM_NO_PAIRS) && mesh->pair_state == 1;
    const int slot = pairs ? 1 : 0;
    DRT_REQUIRE(M.nprim == 0 || mesh->assume_quads || (flags & DRT_BEAM_NO_PAIRS) || mesh->pair_state >= 0,
                "the primitive clusters must exist before a no-allocation call: drt_mesh_build_beam_clusters(mesh) once");
    DRT_REQUIRE(M.nprim == 0 || mesh->beam[slot].blob,
                "the primitive clusters of this search (pairs / DRT_BEAM_NO_PAIRS) must exist before a no-allocation call: "
                "drt_mesh_build_beam_clusters[_ex](mesh) once");
    if ((tp.flags & DRT_TRACE_USE_BVH) && mesh->num_triangles > 0)
        DRT_REQUIRE(drt_mesh_has_bvh(mesh), "DRT_TRACE_USE_BVH: build the LBVH before a no-allocation call (drt_mesh_build_bvh)");
    char *base = reinterpret_cast<char *>(ws);
    auto *counters = reinterpret_cast<unsigned long long *>(base + L.counters);  // kCtrGrazing / Level1 / Level2 / Records / Rows
    auto *dyn = reinterpret_cast<BeamDyn *>(base + L.counters + kDynOffset);
    auto *rx_sorted = reinterpret_cast<float *>(base + L.rx_sorted);
    auto *rx_index = reinterpret_cast<int32_t *>(base + L.rx_index);
    auto *rx_boxes = reinterpret_cast<float *>(base + L.rx_boxes);
    auto *entries1 = reinterpret_cast<BeamEntry *>(base + L.entries1);
    auto *entries2 = reinterpret_cast<BeamEntry *>(base + L.entries2);
    auto *records = reinterpret_cast<unsigned long long *>(base + L.records);
    auto *rows = reinterpret_cast<long long *>(base + L.rows);
    auto *rows_sorted = reinterpret_cast<unsigned long long *>(base + L.rows_sorted);
    auto *table = reinterpret_cast<int32_t *>(base + L.table);
    auto *pair_offsets = reinterpret_cast<long long *>(base + L.pair_offsets);
    char *sort_tmp = base + L.sort_tmp;
    auto *slice_keys = reinterpret_cast<long long *>(base + L.slice_keys);

    if (M.nprim == 0 || ntx == 0 || nrx == 0) {  // nothing to reflect on: all padding
        drt_candidates c{};
        c.table = table;
        c.num_candidates = 0;
        c.order = order;
        c.pair_offsets = reinterpret_cast<const int64_t *>(pair_offsets);
        DRT_HIP(fill_bytes_async(pair_offsets, 0, (size_t)(ntx * nrx + 1) * 8, s));
        return drt_trace_paths_compact_async(mesh, &tp, tx, ntx, rx, nrx, &c, z.max_survivors, max_paths, keys, vertices,
                                             objects, counts_dev, base + L.trace_ws, L.trace_ws_bytes, stream);
    }
    M = pairs ? beam_mesh_pairs(mesh) : beam_mesh(mesh);
    unsigned long long npow = 1;
    for (int j = 0; j < order; ++j) npow *= (unsigned long long)M.nprim;
    const int64_t rows_cap = pairs ? std::max<int64_t>(z.max_rows >> order, 1) : z.max_rows;
    const int64_t table_rows = pairs ? (rows_cap << order) : rows_cap;
    DRT_REQUIRE(ntx * M.nprim < (1ll << 32) && z.max_entries < (1ll << 32), "record format holds 32-bit prefix indices");
    const BeamClusters C = beam_clusters_of(mesh->beam[slot]);

    uint32_t *rx_ids = nullptr, *rx_bounds = nullptr;
    int32_t rc = morton_order(rx, nrx, 1, base + L.morton, &rx_ids, &rx_bounds, s);
    if (rc != DRT_OK) return rc;
    hipLaunchKernelGGL(rx_cluster_kernel, dim3((unsigned)ceil_div(nrx, 64)), dim3(64), 0, s, rx, nrx, rx_ids, rx_sorted,
                       rx_index, rx_boxes);
    uint32_t *tx_bounds = rx_bounds + 8;
    DRT_HIP(fill_bytes_async(tx_bounds, 0xff, 12, s));
    DRT_HIP(fill_bytes_async(tx_bounds + 3, 0, 20, s));
    hipLaunchKernelGGL(point_bounds_kernel, dim3((unsigned)ceil_div(ntx, 256)), dim3(256), 0, s, tx, ntx, 1, tx_bounds);
    const bool expand_clustered = !(flags & DRT_BEAM_EXPAND_PLAIN);
    const bool emit_clustered = (flags & DRT_BEAM_EMIT_CLUSTERED) || (!(flags & DRT_BEAM_EMIT_PLAIN) && nrx >= 128);
    SplitWs split_ws;
    if (order >= 2 && z.ctx_cap > 0 && !(flags & DRT_BEAM_EXPAND_FUSED))
        split_ws = SplitWs{reinterpret_cast<float *>(base + L.ctx_table), reinterpret_cast<unsigned long long *>(base + L.ctx_masks), z.ctx_cap};
    hipLaunchKernelGGL(beam_dyn_kernel, dim3(1), dim3(64), 0, s, rx_bounds, tx_bounds, mesh->beam_max_abs, kappa,
                       expand_clustered ? 1 : 0, dyn);
    DRT_HIP(fill_bytes_async(counters, 0, kCtrBytes, s));
    DRT_LAUNCH_CHECK();
    const float u0 = 0.0f;  // (every kernel takes the unit from `dyn`)
    const RxAll rx_off{{0, 0, 0}, {0, 0, 0}, 0, 0.0f};
    const int64_t cap1 = ntx * M.nprim;
    const int64_t cap2 = std::min(z.max_entries, z.max_records);
    unsigned long long *c1 = counters + kCtrLevel1, *c2 = counters + kCtrLevel2, *c3 = counters + kCtrRecords, *c4 = counters + kCtrRows;
    hipLaunchKernelGGL(beam_seed_kernel, dim3((unsigned)ceil_div(cap1, 256)), dim3(256), 0, s, M, tx, ntx, u0, shard_rank,
                       shard_world, entries1, cap1, c1, BeamDev{dyn, nullptr});
    const BeamEntry *last_src = entries1;  // the list the receiver stage reads (with the records of the last expansion)
    const unsigned long long *last_rec = nullptr;
    int64_t emit_cap = cap1;
    const unsigned long long *emit_count = c1;
    if (order == 2) {
#define CALL(SC, K) launch_expand<SC, 1>(M, C, expand_clustered, entries1, cap1, u0, records, z.max_records, c3, s, rx_off, BeamDev{dyn, c1}, true, split_ws)
        BEAM_DISPATCH2(M.kind, 1, CALL);
#undef CALL
        last_rec = records;
        emit_cap = z.max_records;
        emit_count = c3;
    } else if (order == 3) {
#define CALL(SC, K) launch_expand<SC, 1>(M, C, expand_clustered, entries1, cap1, u0, records, cap2, c2, s, rx_off, BeamDev{dyn, c1}, false)
        BEAM_DISPATCH2(M.kind, 1, CALL);
#undef CALL
        hipLaunchKernelGGL(beam_finish_kernel<1>, dim3((unsigned)ceil_div(cap2, 256)), dim3(256), 0, s, M, entries1, records,
                           cap2, u0, entries2, BeamDev{dyn, c2});
#define CALL(SC, K) launch_expand<SC, 2>(M, C, expand_clustered, entries2, cap2, u0, records, z.max_records, c3, s, rx_off, BeamDev{dyn, c2}, true, split_ws)
        BEAM_DISPATCH2(M.kind, 1, CALL);
#undef CALL
        last_src = entries2;
        last_rec = records;
        emit_cap = z.max_records;
        emit_count = c3;
    }
#define CALL(SC, K) launch_emit<SC, K>(M, emit_clustered, last_src, last_rec, emit_cap, rx, rx_sorted, rx_index, rx_boxes, nrx, u0, rows, rows_cap, c4, counters + kCtrGrazing, s, BeamDev{dyn, emit_count})
    BEAM_DISPATCH2(M.kind, order, CALL);
#undef CALL
    hipLaunchKernelGGL(rows_pad_kernel, dim3((unsigned)ceil_div(rows_cap, 256)), dim3(256), 0, s,
                       reinterpret_cast<unsigned long long *>(rows), c4, rows_cap);
    DRT_LAUNCH_CHECK();
    size_t tb = sort_keys64_temp_bytes(rows_cap);
    DRT_HIP(capture_safe_sort(sort_tmp, tb, reinterpret_cast<unsigned long long *>(rows), rows_sorted, nullptr, nullptr, rows_cap, 0,
                              63, s));  // sort_safe.hpp / radix_sort.hip: kernels only
    const unsigned long long *row_keys = rows_sorted;
    if (pairs) {
        const dim3 ge((unsigned)ceil_div(table_rows, 256));
        auto *tri_keys = reinterpret_cast<unsigned long long *>(rows);
        if (order == 1) hipLaunchKernelGGL(rows_expand_pairs_kernel<1>, ge, dim3(256), 0, s, rows_sorted, rows_cap, (unsigned long long)M.nprim, M.pair_tri, (unsigned long long)mesh->num_triangles, table, tri_keys);
        else if (order == 2) hipLaunchKernelGGL(rows_expand_pairs_kernel<2>, ge, dim3(256), 0, s, rows_sorted, rows_cap, (unsigned long long)M.nprim, M.pair_tri, (unsigned long long)mesh->num_triangles, table, tri_keys);
        else hipLaunchKernelGGL(rows_expand_pairs_kernel<3>, ge, dim3(256), 0, s, rows_sorted, rows_cap, (unsigned long long)M.nprim, M.pair_tri, (unsigned long long)mesh->num_triangles, table, tri_keys);
        row_keys = tri_keys;
    } else {
        const dim3 gr((unsigned)ceil_div(rows_cap, 256));
        if (order == 1) hipLaunchKernelGGL(rows_decode_kernel<1>, gr, dim3(256), 0, s, rows_sorted, rows_cap, (unsigned long long)M.nprim, M.scale, table);
        else if (order == 2) hipLaunchKernelGGL(rows_decode_kernel<2>, gr, dim3(256), 0, s, rows_sorted, rows_cap, (unsigned long long)M.nprim, M.scale, table);
        else hipLaunchKernelGGL(rows_decode_kernel<3>, gr, dim3(256), 0, s, rows_sorted, rows_cap, (unsigned long long)M.nprim, M.scale, table);
    }
    hipLaunchKernelGGL(pair_offsets_kernel, dim3((unsigned)ceil_div(ntx * nrx + 1, 256)), dim3(256), 0, s, rows_sorted, rows_cap,
                       npow, ntx * nrx, pair_offsets, pairs ? ((int64_t)1 << order) : (int64_t)1);
    DRT_LAUNCH_CHECK();
    drt_candidates c{};
    c.table = table;
    c.num_candidates = table_rows;
    c.order = order;
    c.pair_offsets = reinterpret_cast<const int64_t *>(pair_offsets);
    if (pairs && !(flags & DRT_BEAM_ROWS_PLAIN)) c.reserved |= DRT_CAND_PAIR_BLOCKS;
    rc = drt_trace_paths_compact_async(mesh, &tp, tx, ntx, rx, nrx, &c, z.max_survivors, max_paths,
                                       reinterpret_cast<int64_t *>(slice_keys), vertices, objects, counts_dev,
                                       base + L.trace_ws, L.trace_ws_bytes, stream);
    if (rc != DRT_OK) return rc;
    if (max_paths > 0) {
        hipLaunchKernelGGL(keys_to_rows_padded_kernel, dim3((unsigned)ceil_div(max_paths, 256)), dim3(256), 0, s, slice_keys,
                           max_paths, row_keys, reinterpret_cast<long long *>(keys));
        if (pairs && max_paths > 1) {  // rows of a pair are grouped, not sorted by triangle key: one static-size sort
            auto *mk = reinterpret_cast<unsigned long long *>(base + L.merge_keys);
            auto *perm = reinterpret_cast<uint32_t *>(base + L.merge_perm);
            auto *iota = reinterpret_cast<uint32_t *>(base + L.merge_iota);
            auto *tmp_rows = reinterpret_cast<uint32_t *>(base + L.merge_rows);
            hipLaunchKernelGGL(iota_kernel, dim3((unsigned)ceil_div(max_paths, 256)), dim3(256), 0, s, iota, max_paths);
            size_t tb2 = sort_pairs64_temp_bytes(max_paths);
            DRT_HIP(capture_safe_sort(sort_tmp, tb2, reinterpret_cast<unsigned long long *>(keys), mk, iota, perm, max_paths, 0, 64,
                                      s));  // padding keys (-1) sort last
            auto copy = [&](const void *src, void *dst, int64_t words) {
                hipLaunchKernelGGL(copy_u32_kernel, dim3((unsigned)ceil_div(words, 256)), dim3(256), 0, s,
                                   reinterpret_cast<const uint32_t *>(src), words, reinterpret_cast<uint32_t *>(dst));
            };
            copy(mk, keys, max_paths * 2);
            hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)ceil_div(max_paths * k2 * 3, 256)), dim3(256), 0, s,
                               reinterpret_cast<const uint32_t *>(vertices), perm, max_paths, (int32_t)(k2 * 3), tmp_rows);
            copy(tmp_rows, vertices, max_paths * k2 * 3);
            hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)ceil_div(max_paths * k2, 256)), dim3(256), 0, s,
                               reinterpret_cast<const uint32_t *>(objects), perm, max_paths, (int32_t)k2, tmp_rows);
            copy(tmp_rows, objects, max_paths * k2);
        }
    }
    hipLaunchKernelGGL(beam_counts_kernel, dim3(1), dim3(64), 0, s, counters, cap2, z.max_records, rows_cap,
                       pairs ? order : 0, reinterpret_cast<long long *>(counts_dev));
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

}  // extern "C"
