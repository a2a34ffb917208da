// smooth.hip -- the smoothed ("soft mask") mode of the hot path and its VJPs, for gfx950
// (SURVEY.md section 8 row f4).  Every output that is a bool in the hard mode becomes a float32
// confidence in [0, 1] that is differentiable in the rays, the end points and the mesh vertices.
//
//   drt_ray_intersect_triangle_smooth(+_vjp)       reference geometry/_utils.py:1279-1320
//   drt_ray_intersect_any_triangle_smooth(+_vjp)   reference geometry/_utils.py:1436-1537
//   drt_consecutive_vertices_same_side_smooth      reference geometry/_solver_image_method.py:450-453
//   drt_trace_paths_dense_smooth(+_vjp)            reference geometry/_solvers.py:499-770, smoothed
//                                                  branches :599-613, 628-635, 647-653, 664-674,
//                                                  686-689, 701-713
//
// There is no sparsity to exploit here (every candidate has a non-zero confidence and the blocked
// term sums over ALL triangles), so the layout is the reference's dense one: one WAVEFRONT per
// (tx, rx, candidate) path; the image chain and the cheap terms are computed wave-uniformly, the
// lanes stride over the triangles for the blocked sums (butterfly reductions, fixed order).
// The reverse pass recomputes the forward, routes the cotangent of the mask to the single term that
// realises the min (first smallest, like an argmin), and differentiates that term by hand.
#include "common.hpp"
#include "smooth.hpp"
#include "trace_common.hpp"

#pragma clang fp contract(off)

namespace drt {

// ------------------------------------------------------------------------------------------
// (a1 smoothed) Moller-Trumbore, dense [R] x [T] or paired [n]
// ------------------------------------------------------------------------------------------
template <bool DENSE>
__global__ __launch_bounds__(256) void mt_smooth_kernel(const float *__restrict__ o,
                                                        const float *__restrict__ d, int64_t R,
                                                        const float *__restrict__ tv, int64_t T,
                                                        float eps, float alpha, float *__restrict__ t_out,
                                                        float *__restrict__ hit_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = DENSE ? R * T : R;
    if (i >= n) return;
    const int64_t r = DENSE ? i / T : i;
    const int64_t j = DENSE ? i - r * T : i;
    const MtSmooth m = mt_smooth(ld3(o + 3 * r), ld3(d + 3 * r), load_tri(tv + 9 * j), eps, alpha);
    t_out[i] = m.t;
    hit_out[i] = m.hit;
}

__device__ __forceinline__ void atomic_add_tri(float *g, const MtBar &b) {
    atomic_add3(g, b.v0);
    atomic_add3(g + 3, b.v1);
    atomic_add3(g + 6, b.v2);
}

template <bool DENSE>
__global__ __launch_bounds__(256) void mt_smooth_vjp_kernel(
    const float *__restrict__ o, const float *__restrict__ d, int64_t R, const float *__restrict__ tv,
    int64_t T, float eps, float alpha, const float *__restrict__ t_bar,
    const float *__restrict__ hit_bar, float *__restrict__ g_o, float *__restrict__ g_d,
    float *__restrict__ g_tv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = DENSE ? R * T : R;
    if (i >= n) return;
    const int64_t r = DENSE ? i / T : i;
    const int64_t j = DENSE ? i - r * T : i;
    const float tb = t_bar ? t_bar[i] : 0.0f, hb = hit_bar ? hit_bar[i] : 0.0f;
    if (tb == 0.0f && hb == 0.0f) return;
    const float *t9 = tv + 9 * j;
    const MtBar b = mt_smooth_vjp(ld3(o + 3 * r), ld3(d + 3 * r), ld3(t9), ld3(t9 + 3), ld3(t9 + 6), eps,
                                  alpha, tb, hb);
    if (g_o) atomic_add3(g_o + 3 * r, b.o);
    if (g_d) atomic_add3(g_d + 3 * r, b.d);
    if (g_tv) atomic_add_tri(g_tv + 9 * j, b);
}

// ------------------------------------------------------------------------------------------
// (a2 smoothed) clipped sums over the triangles, NSEG rays per wavefront
// ------------------------------------------------------------------------------------------
struct TileIter {
    int64_t T, bs, nb, rem;
    __device__ TileIter(int64_t T_, int64_t batch) : T(T_) {
        bs = (batch <= 0 || batch > T_) ? T_ : batch;  // batch_size = None -> one tile
        if (bs < 1) bs = 1;
        nb = T_ / bs;
        rem = T_ - nb * bs;
    }
    __device__ int64_t count() const { return nb + (rem > 0 ? 1 : 0); }
    __device__ int64_t lo(int64_t i) const { return i < nb ? i * bs : T - rem; }  // remainder tile last
    __device__ int64_t hi(int64_t i) const { return i < nb ? (i + 1) * bs : T; }
};

// acc[j] = fold over tiles of clip(acc + sum_{active i in tile} w(seg j, tri i), max=1)
// (_utils.py:1465-1476, 1518-1537).  Same value in every lane.
template <int NSEG>
__device__ __forceinline__ void blocked_sums(const V3 (&o)[NSEG], const V3 (&d)[NSEG],
                                             const float *__restrict__ tv, int64_t T,
                                             const uint8_t *__restrict__ active, int64_t batch,
                                             float eps, float thr, float alpha, int lane,
                                             float (&acc)[NSEG]) {
#pragma unroll
    for (int j = 0; j < NSEG; ++j) acc[j] = 0.0f;
    const TileIter tiles(T, batch);
    for (int64_t ti = 0; ti < tiles.count(); ++ti) {
        float part[NSEG];
#pragma unroll
        for (int j = 0; j < NSEG; ++j) part[j] = 0.0f;
        for (int64_t i = tiles.lo(ti) + lane; i < tiles.hi(ti); i += kWave) {
            if (active && !active[i]) continue;
            const TriE tr = load_tri(tv + 9 * i);
#pragma unroll
            for (int j = 0; j < NSEG; ++j) part[j] += blocked_weight(o[j], d[j], tr, eps, thr, alpha);
        }
#pragma unroll
        for (int j = 0; j < NSEG; ++j) acc[j] = nmin(acc[j] + wave_sum(part[j]), 1.0f);
    }
}

// Reverse of ONE ray's clipped sum: w_bar per triangle = acc_bar when the sum never clipped.
// Lanes stride over the triangles; per-lane (o, d) cotangents are returned un-reduced.
// `scatter(i, MtBar)` adds the triangle cotangents wherever the caller keeps them.
template <typename Scatter>
__device__ __forceinline__ void blocked_sum_vjp(V3 o, V3 d, const float *__restrict__ tv, int64_t T,
                                                const uint8_t *__restrict__ active, float eps, float thr,
                                                float alpha, int lane, float acc_bar, V3 &o_bar, V3 &d_bar,
                                                Scatter scatter) {
    o_bar = V3{0, 0, 0};
    d_bar = V3{0, 0, 0};
    for (int64_t i = lane; i < T; i += kWave) {
        if (active && !active[i]) continue;
        const float *t9 = tv + 9 * i;
        const MtBar b = blocked_weight_vjp(o, d, ld3(t9), ld3(t9 + 3), ld3(t9 + 6), eps, thr, alpha, acc_bar);
        o_bar = o_bar + b.o;
        d_bar = d_bar + b.d;
        scatter(i, b);
    }
}

__device__ __forceinline__ V3 wave_sum3(V3 v) { return V3{wave_sum(v.x), wave_sum(v.y), wave_sum(v.z)}; }

__global__ __launch_bounds__(256) void any_smooth_kernel(
    const float *__restrict__ o, const float *__restrict__ d, int64_t R, const float *__restrict__ tv,
    int64_t T, int64_t tv_stride, const uint8_t *__restrict__ active, int64_t act_stride, int64_t batch,
    float eps, float thr, float alpha, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const V3 oo[1] = {ld3(o + 3 * r)}, dd[1] = {ld3(d + 3 * r)};
    float acc[1];
    blocked_sums<1>(oo, dd, tv + r * tv_stride, T, active ? active + r * act_stride : nullptr, batch, eps,
                    thr, alpha, lane, acc);
    if (lane == 0) out[r] = acc[0];
}

__global__ __launch_bounds__(256) void any_smooth_vjp_kernel(
    const float *__restrict__ o, const float *__restrict__ d, int64_t R, const float *__restrict__ tv,
    int64_t T, int64_t tv_stride, const uint8_t *__restrict__ active, int64_t act_stride, int64_t batch,
    float eps, float thr, float alpha, const float *__restrict__ out_bar, float *__restrict__ g_o,
    float *__restrict__ g_d, float *__restrict__ g_tv) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float ob = out_bar[r];
    if (ob == 0.0f) return;
    const V3 oo[1] = {ld3(o + 3 * r)}, dd[1] = {ld3(d + 3 * r)};
    const float *tvr = tv + r * tv_stride;
    const uint8_t *act = active ? active + r * act_stride : nullptr;
    float acc[1];
    blocked_sums<1>(oo, dd, tvr, T, act, batch, eps, thr, alpha, lane, acc);
    if (!(acc[0] < 1.0f)) return;  // clipped (or NaN): constant
    V3 ob3, db3;
    float *gt = g_tv ? g_tv + r * tv_stride : nullptr;
    blocked_sum_vjp(oo[0], dd[0], tvr, T, act, eps, thr, alpha, lane, ob, ob3, db3,
                    [&](int64_t i, const MtBar &b) {
                        if (gt) atomic_add_tri(gt + 9 * i, b);
                    });
    ob3 = wave_sum3(ob3);
    db3 = wave_sum3(db3);
    if (lane == 0) {
        if (g_o) st3(g_o + 3 * r, ob3);
        if (g_d) st3(g_d + 3 * r, db3);
    }
}

// ------------------------------------------------------------------------------------------
// same side of the mirror, smoothed (IM:450-453): sigmoid(alpha * sign(dot_prev) * sign(dot_next))
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sign_f(float x) {  // jnp.sign: 0 for +-0, NaN for NaN
    return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : ((x == x) ? 0.0f : x));
}

__device__ __forceinline__ float same_side_smooth(V3 prev, V3 next, V3 p, V3 n, float alpha) {
    return smoothing(sign_f(dot(prev - p, n)) * sign_f(dot(next - p, n)), alpha);
}

__global__ __launch_bounds__(256) void same_side_smooth_kernel(const float *__restrict__ vertices,
                                                               const float *__restrict__ mv,
                                                               const float *__restrict__ mn, int64_t B,
                                                               int K, float alpha,
                                                               float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * K) return;
    const int64_t b = i / K;
    const int j = (int)(i % K);
    out[i] = same_side_smooth(ld3(vertices + 3 * ((K + 2) * b + j)), ld3(vertices + 3 * ((K + 2) * b + j + 2)),
                              ld3(mv + 3 * i), ld3(mn + 3 * i), alpha);
}

// ------------------------------------------------------------------------------------------
// the smoothed tracer: forward state of one path, shared by the forward and reverse kernels
// ------------------------------------------------------------------------------------------
template <int K, bool QUADS>
struct SmoothPath {
    int64_t it, ir;
    int32_t id[KA<K>::n];
    Mirrors<K, QUADS> m;
    V3 full[K + 2];
    V3 o[K + 1], d[K + 1];
    bool finite;
    float inside, valid, blocked, too_small, mask;
    float acc[K + 1];  // clipped blocked sum per segment
    int term;          // which of {0 inside, 1 valid, 2 blocked, 3 too small, 4 finite} realises the min; -1 none
};

template <int K, bool QUADS>
__device__ __forceinline__ void smooth_path_forward(const TraceArgs &a, const CandSrc &cs, int64_t flat,
                                                    float alpha, int64_t batch, int lane,
                                                    SmoothPath<K, QUADS> &s) {
    const int64_t pair = flat / cs.count;
    const int64_t row = flat - pair * cs.count;
    s.it = pair / a.nrx;
    s.ir = pair - s.it * a.nrx;
    load_candidate<K>(cs, row, s.id);
    load_mirrors<K, QUADS>(a, s.id, s.m);
    s.full[0] = ld3(a.tx + 3 * s.it);
    s.full[K + 1] = ld3(a.rx + 3 * s.ir);
    if constexpr (K > 0) {
        V3 path[KA<K>::n];
        image_chain<KA<K>::n>(s.full[0], s.full[K + 1], s.m.p, s.m.n, path);
#pragma unroll
        for (int j = 0; j < K; ++j) s.full[j + 1] = path[j];
    }
    s.finite = path_finite<K>(s.full);
#pragma unroll
    for (int j = 0; j <= K; ++j) {
        s.o[j] = s.full[j];
        s.d[j] = s.full[j + 1] - s.full[j];
    }
    // 3.1 inside the candidate triangles: max over the quad pair (initial 0), min over the order (initial 1)
    float inside = 1.0f, valid = 1.0f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        float h = mt_smooth(s.o[j], s.d[j], s.m.tri[j], a.eps, alpha).hit;
        if (QUADS) h = nmax(nmax(0.0f, h), mt_smooth(s.o[j], s.d[j], s.m.tri2[j], a.eps, alpha).hit);
        inside = nmin(inside, h);
        // 3.2 consecutive vertices on the same side of each mirror
        valid = nmin(valid, same_side_smooth(s.full[j], s.full[j + 2], s.m.p[j], s.m.n[j], alpha));
    }
    // 3.3 blocked by any active triangle: max over the segments (initial 0) of the clipped sums
    blocked_sums<K + 1>(s.o, s.d, a.tri_verts, a.T, a.mask, batch, a.eps, a.thr, alpha, lane, s.acc);
    float blocked = 0.0f, too_small = 0.0f;
#pragma unroll
    for (int j = 0; j <= K; ++j) {
        blocked = nmax(blocked, s.acc[j]);
        // 3.4 too small: squared segment lengths against min_len
        too_small = nmax(too_small, smoothing(a.min_len - dot(s.d[j], s.d[j]), alpha));
    }
    s.inside = inside;
    s.valid = valid;
    s.blocked = blocked;
    s.too_small = too_small;
    const float terms[5] = {inside, valid, 1.0f - blocked, 1.0f - too_small, s.finite ? 1.0f : 0.0f};
    float mk = 1.0f;
    int term = -1;
    bool nan = false;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        nan = nan || (terms[i] != terms[i]);
        if (terms[i] < mk) {
            mk = terms[i];
            term = i;
        }
    }
    s.mask = nan ? __builtin_nanf("") : mk;
    s.term = nan ? -1 : term;
}

template <int K, bool QUADS>
__global__ __launch_bounds__(256) void trace_smooth_kernel(TraceArgs a, CandSrc cs, float alpha,
                                                           int64_t batch, int64_t total,
                                                           float *__restrict__ vertices,
                                                           int32_t *__restrict__ objects,
                                                           float *__restrict__ mask) {
    const int lane = threadIdx.x & 63;
    const int64_t flat = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (flat >= total) return;
    SmoothPath<K, QUADS> s;
    smooth_path_forward<K, QUADS>(a, cs, flat, alpha, batch, lane, s);
    if (lane != 0) return;
    float mk = s.mask;
    if (a.mask) mk = mk * (s.m.active ? 1.0f : 0.0f);  // SV:710-711
    const bool keep = s.m.ok && s.finite;               // SV:696-699 (and padding rows)
    if (!s.m.ok) mk = 0.0f;
    mask[flat] = mk;
#pragma unroll
    for (int j = 0; j < K + 2; ++j) st3(vertices + (flat * (K + 2) + j) * 3, keep ? s.full[j] : V3{0, 0, 0});
    int32_t *ob = objects + flat * (K + 2);
    ob[0] = (int32_t)s.it;
#pragma unroll
    for (int j = 0; j < K; ++j) ob[1 + j] = s.id[j];
    ob[K + 1] = (int32_t)s.ir;
}

template <int K, bool QUADS>
__global__ __launch_bounds__(256) void trace_smooth_vjp_kernel(
    TraceArgs a, CandSrc cs, float alpha, int64_t batch, int64_t total,
    const float *__restrict__ mesh_vertices, const int32_t *__restrict__ mesh_triangles,
    const float *__restrict__ cot_v, const float *__restrict__ cot_m, float *__restrict__ g_tx,
    float *__restrict__ g_rx, float *__restrict__ g_vertices) {
    const int lane = threadIdx.x & 63;
    const int64_t flat = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (flat >= total) return;
    float gm = cot_m ? cot_m[flat] : 0.0f;
    bool any_v = false;
    V3 gf[K + 2];
#pragma unroll
    for (int j = 0; j < K + 2; ++j) {
        gf[j] = cot_v ? ld3(cot_v + (flat * (K + 2) + j) * 3) : V3{0, 0, 0};
        any_v = any_v || gf[j].x != 0.0f || gf[j].y != 0.0f || gf[j].z != 0.0f;
    }
    if (gm == 0.0f && !any_v) return;
    SmoothPath<K, QUADS> s;
    smooth_path_forward<K, QUADS>(a, cs, flat, alpha, batch, lane, s);
    // padding rows and non-finite paths are constants (zeroed vertices; their mask has no finite slope)
    if (!s.m.ok || !s.finite) return;
    if (a.mask && !s.m.active) gm = 0.0f;

    V3 go[K + 1], gd[K + 1];
#pragma unroll
    for (int j = 0; j <= K; ++j) go[j] = gd[j] = V3{0, 0, 0};

    auto scatter_tri = [&](int64_t tri, const MtBar &b) {
        if (!g_vertices) return;
        const int32_t *ix = mesh_triangles + 3 * tri;
        atomic_add3(g_vertices + 3 * (int64_t)ix[0], b.v0);
        atomic_add3(g_vertices + 3 * (int64_t)ix[1], b.v1);
        atomic_add3(g_vertices + 3 * (int64_t)ix[2], b.v2);
    };

    if (gm != 0.0f && s.term == 0) {
        if constexpr (K > 0) {  // inside: the segment realising the min, the quad half realising the max
            int js = -1;
            float best = 1.0f;
            int64_t tri = 0;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                float h = mt_smooth(s.o[j], s.d[j], s.m.tri[j], a.eps, alpha).hit;
                int64_t tj = s.id[j];
                if (QUADS) {
                    const float h2 = mt_smooth(s.o[j], s.d[j], s.m.tri2[j], a.eps, alpha).hit;
                    if (h2 > h) {
                        h = h2;
                        tj = (int64_t)s.id[j] + 1;
                    }
                }
                if (h < best) {
                    best = h;
                    js = j;
                    tri = tj;
                }
            }
            if (js >= 0) {
                V3 oj{0, 0, 0}, dj{0, 0, 0};
#pragma unroll
                for (int j = 0; j < K; ++j)
                    if (j == js) {
                        oj = s.o[j];
                        dj = s.d[j];
                    }
                const float *t9 = a.tri_verts + 9 * tri;
                const MtBar b = mt_smooth_vjp(oj, dj, ld3(t9), ld3(t9 + 3), ld3(t9 + 6), a.eps, alpha, 0.0f, gm);
#pragma unroll
                for (int j = 0; j < K; ++j)
                    if (j == js) {
                        go[j] = b.o;
                        gd[j] = b.d;
                    }
                if (lane == 0) scatter_tri(tri, b);
            }
        }
    } else if (gm != 0.0f && s.term == 2) {  // 1 - blocked: the segment with the largest clipped sum
        int js = -1;
        float best = 0.0f;
#pragma unroll
        for (int j = 0; j <= K; ++j)
            if (s.acc[j] > best) {
                best = s.acc[j];
                js = j;
            }
        if (js >= 0 && best < 1.0f) {
            V3 oj{0, 0, 0}, dj{0, 0, 0};
#pragma unroll
            for (int j = 0; j <= K; ++j)
                if (j == js) {
                    oj = s.o[j];
                    dj = s.d[j];
                }
            V3 ob, db;
            blocked_sum_vjp(oj, dj, a.tri_verts, a.T, a.mask, a.eps, a.thr, alpha, lane, -gm, ob, db,
                            [&](int64_t i, const MtBar &b) { scatter_tri(i, b); });
            ob = wave_sum3(ob);
            db = wave_sum3(db);
#pragma unroll
            for (int j = 0; j <= K; ++j)
                if (j == js) {
                    go[j] = ob;
                    gd[j] = db;
                }
        }
    } else if (gm != 0.0f && s.term == 3) {  // 1 - too_small: the shortest segment
        int js = -1;
        float best = 0.0f;
#pragma unroll
        for (int j = 0; j <= K; ++j) {
            const float v = smoothing(a.min_len - dot(s.d[j], s.d[j]), alpha);
            if (v > best) {
                best = v;
                js = j;
            }
        }
        // d mask / d len2 = +alpha s (1 - s);  len2 = <d, d>
        const float c = 2.0f * (gm * smoothing_grad(best, alpha));
#pragma unroll
        for (int j = 0; j <= K; ++j)
            if (j == js) gd[j] = s.d[j] * c;
    }
    if (lane != 0) return;
    // segments -> path vertices: o_j = X_j, d_j = X_{j+1} - X_j
#pragma unroll
    for (int j = 0; j <= K; ++j) {
        gf[j] = gf[j] + go[j] - gd[j];
        gf[j + 1] = gf[j + 1] + gd[j];
    }
    V3 tx_bar = gf[0], rx_bar = gf[K + 1];
    if constexpr (K > 0) {
        V3 gp[KA<K>::n], pb[KA<K>::n], nb[KA<K>::n], fb, tb;
#pragma unroll
        for (int j = 0; j < K; ++j) gp[j] = gf[j + 1];
        image_chain_vjp<KA<K>::n>(s.full[0], s.full[K + 1], s.m.p, s.m.n, gp, fb, tb, pb, nb);
        tx_bar = tx_bar + fb;
        rx_bar = rx_bar + tb;
        if (g_vertices) {
#pragma unroll
            for (int j = 0; j < K; ++j)
                mirror_vjp_to_mesh(mesh_vertices, mesh_triangles, s.id[j], pb[j], nb[j], g_vertices);
        }
    }
    if (g_tx) atomic_add3(g_tx + 3 * s.it, tx_bar);
    if (g_rx) atomic_add3(g_rx + 3 * s.ir, rx_bar);
}

}  // namespace drt

using namespace drt;

extern "C" {

int32_t drt_ray_intersect_triangle_smooth(const float *o, const float *d, int64_t R, const float *tv,
                                          int64_t T, int32_t dense, float eps, float alpha, float *t_out,
                                          float *hit_out, void *stream) {
    DRT_REQUIRE(R >= 0 && T >= 0, "negative size");
    DRT_REQUIRE(dense || R == T, "paired form needs num_triangles == num_rays");
    const int64_t n = dense ? R * T : R;
    if (n == 0) return DRT_OK;
    DRT_REQUIRE(o && d && tv && t_out && hit_out, "null pointer");
    const dim3 grid((unsigned)ceil_div(n, 256));
    if (dense)
        hipLaunchKernelGGL(mt_smooth_kernel<true>, grid, dim3(256), 0, as_stream(stream), o, d, R, tv, T, eps,
                           alpha, t_out, hit_out);
    else
        hipLaunchKernelGGL(mt_smooth_kernel<false>, grid, dim3(256), 0, as_stream(stream), o, d, R, tv, T, eps,
                           alpha, t_out, hit_out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_ray_intersect_triangle_smooth_vjp(const float *o, const float *d, int64_t R, const float *tv,
                                              int64_t T, int32_t dense, float eps, float alpha,
                                              const float *t_bar, const float *hit_bar, float *g_o,
                                              float *g_d, float *g_tv, void *stream) {
    DRT_REQUIRE(R >= 0 && T >= 0, "negative size");
    DRT_REQUIRE(dense || R == T, "paired form needs num_triangles == num_rays");
    const int64_t n = dense ? R * T : R;
    if (n == 0 || (!t_bar && !hit_bar)) return DRT_OK;
    DRT_REQUIRE(o && d && tv, "null pointer");
    const dim3 grid((unsigned)ceil_div(n, 256));
    if (dense)
        hipLaunchKernelGGL(mt_smooth_vjp_kernel<true>, grid, dim3(256), 0, as_stream(stream), o, d, R, tv, T,
                           eps, alpha, t_bar, hit_bar, g_o, g_d, g_tv);
    else
        hipLaunchKernelGGL(mt_smooth_vjp_kernel<false>, grid, dim3(256), 0, as_stream(stream), o, d, R, tv, T,
                           eps, alpha, t_bar, hit_bar, g_o, g_d, g_tv);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_ray_intersect_any_triangle_smooth(const float *o, const float *d, int64_t R, const float *tv,
                                              int64_t T, int64_t tv_ray_stride, const uint8_t *active,
                                              int64_t active_ray_stride, float eps, float hit_tol,
                                              float alpha, int64_t batch_size, float *out, void *stream) {
    DRT_REQUIRE(R >= 0 && T >= 0, "negative size");
    if (R == 0) return DRT_OK;
    DRT_REQUIRE(out, "out is null");
    if (T == 0) {  // _utils.py:1441-1450
        DRT_HIP(fill_bytes_async(out, 0, (size_t)R * sizeof(float), as_stream(stream)));
        return DRT_OK;
    }
    DRT_REQUIRE(o && d && tv, "null pointer");
    DRT_REQUIRE(tv_ray_stride == 0 || tv_ray_stride == 9 * T, "tv_ray_stride must be 0 or 9*T");
    DRT_REQUIRE(active_ray_stride == 0 || active_ray_stride == T, "active_ray_stride must be 0 or T");
    hipLaunchKernelGGL(any_smooth_kernel, dim3((unsigned)ceil_div(R, 4)), dim3(256), 0, as_stream(stream), o, d,
                       R, tv, T, tv_ray_stride, active, active_ray_stride, batch_size, eps, 1.0f - hit_tol,
                       alpha, out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_ray_intersect_any_triangle_smooth_vjp(const float *o, const float *d, int64_t R,
                                                  const float *tv, int64_t T, int64_t tv_ray_stride,
                                                  const uint8_t *active, int64_t active_ray_stride,
                                                  float eps, float hit_tol, float alpha, int64_t batch_size,
                                                  const float *out_bar, float *g_o, float *g_d, float *g_tv,
                                                  void *stream) {
    DRT_REQUIRE(R >= 0 && T >= 0, "negative size");
    if (R == 0 || T == 0) return DRT_OK;
    DRT_REQUIRE(o && d && tv && out_bar, "null pointer");
    DRT_REQUIRE(tv_ray_stride == 0 || tv_ray_stride == 9 * T, "tv_ray_stride must be 0 or 9*T");
    DRT_REQUIRE(active_ray_stride == 0 || active_ray_stride == T, "active_ray_stride must be 0 or T");
    hipLaunchKernelGGL(any_smooth_vjp_kernel, dim3((unsigned)ceil_div(R, 4)), dim3(256), 0, as_stream(stream), o,
                       d, R, tv, T, tv_ray_stride, active, active_ray_stride, batch_size, eps, 1.0f - hit_tol,
                       alpha, out_bar, g_o, g_d, g_tv);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_consecutive_vertices_same_side_smooth(const float *vertices, const float *mv, const float *mn,
                                                  int64_t B, int32_t k, float alpha, float *out,
                                                  void *stream) {
    DRT_REQUIRE(B >= 0 && k >= 0, "negative size");
    if (B == 0 || k == 0) return DRT_OK;
    DRT_REQUIRE(vertices && mv && mn && out, "null pointer");
    hipLaunchKernelGGL(same_side_smooth_kernel, dim3((unsigned)ceil_div(B * k, 256)), dim3(256), 0,
                       as_stream(stream), vertices, mv, mn, B, (int)k, alpha, out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_trace_paths_dense_smooth(drt_mesh_t mesh, const drt_trace_params *pr, float alpha,
                                     int64_t batch_size, const float *tx, int64_t ntx, const float *rx,
                                     int64_t nrx, const drt_candidates *cands, float *vertices,
                                     int32_t *objects, float *mask, void *stream) {
    DRT_REQUIRE(mesh && pr && cands, "null argument");
    DRT_REQUIRE(ntx >= 0 && nrx >= 0, "negative size");
    const bool quads = mesh->assume_quads != 0;
    CandSrc cs;
    int32_t rc = make_cand_src(cands, quads ? 2 : 1, &cs);
    if (rc != DRT_OK) return rc;
    DRT_REQUIRE(!cs.ragged, "ragged pair spaces have no dense layout");
    const TraceArgs a = make_args(mesh, pr, tx, ntx, rx, nrx);
    const int64_t total = ntx * nrx * cs.count;
    if (total == 0) return DRT_OK;
    DRT_REQUIRE(tx && rx && vertices && objects && mask, "null pointer");
    const dim3 grid((unsigned)ceil_div(total, 4));
#define CALL(K)                                                                                          \
    do {                                                                                                 \
        if (quads)                                                                                       \
            hipLaunchKernelGGL((trace_smooth_kernel<K, true>), grid, dim3(256), 0, as_stream(stream), a, cs, \
                               alpha, batch_size, total, vertices, objects, mask);                       \
        else                                                                                             \
            hipLaunchKernelGGL((trace_smooth_kernel<K, false>), grid, dim3(256), 0, as_stream(stream), a, cs, \
                               alpha, batch_size, total, vertices, objects, mask);                       \
    } while (0)
    DRT_ORDER_SWITCH(cands->order, CALL)
#undef CALL
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_trace_paths_dense_smooth_vjp(drt_mesh_t mesh, const drt_trace_params *pr, float alpha,
                                         int64_t batch_size, const float *tx, int64_t ntx, const float *rx,
                                         int64_t nrx, const drt_candidates *cands,
                                         const float *vertices_cotangent, const float *mask_cotangent,
                                         float *grad_tx, float *grad_rx, float *grad_mesh_vertices,
                                         void *stream) {
    DRT_REQUIRE(mesh && pr && cands, "null argument");
    DRT_REQUIRE(ntx >= 0 && nrx >= 0, "negative size");
    const bool quads = mesh->assume_quads != 0;
    CandSrc cs;
    int32_t rc = make_cand_src(cands, quads ? 2 : 1, &cs);
    if (rc != DRT_OK) return rc;
    DRT_REQUIRE(!cs.ragged, "ragged pair spaces have no dense layout");
    const TraceArgs a = make_args(mesh, pr, tx, ntx, rx, nrx);
    const int64_t total = ntx * nrx * cs.count;
    if (total == 0 || (!vertices_cotangent && !mask_cotangent)) return DRT_OK;
    DRT_REQUIRE(tx && rx, "null pointer");
    const dim3 grid((unsigned)ceil_div(total, 4));
#define CALL(K)                                                                                              \
    do {                                                                                                     \
        if (quads)                                                                                           \
            hipLaunchKernelGGL((trace_smooth_vjp_kernel<K, true>), grid, dim3(256), 0, as_stream(stream), a, cs, \
                               alpha, batch_size, total, mesh->vertices, mesh->triangles, vertices_cotangent, \
                               mask_cotangent, grad_tx, grad_rx, grad_mesh_vertices);                        \
        else                                                                                                 \
            hipLaunchKernelGGL((trace_smooth_vjp_kernel<K, false>), grid, dim3(256), 0, as_stream(stream), a, cs, \
                               alpha, batch_size, total, mesh->vertices, mesh->triangles, vertices_cotangent, \
                               mask_cotangent, grad_tx, grad_rx, grad_mesh_vertices);                        \
    } while (0)
    DRT_ORDER_SWITCH(cands->order, CALL)
#undef CALL
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

}  // extern "C"
