// smooth.hpp -- device arithmetic of the smoothed ("soft mask") mode: every hard comparison of
// the reference becomes sigmoid(alpha * margin), conjunctions become min, disjunctions max / clipped
// sums.  Reference: differt/src/differt/utils.py:70-89 (`smoothing_function`),
// geometry/_utils.py:1279-1320 (Moller-Trumbore), :1465-1476 (any-triangle),
// geometry/_solver_image_method.py:450-453 (same side), geometry/_solvers.py:599-713 (tracer).
//
// Floating-point tolerance path (1e-5 rel, tests/test_smooth_gpu.py): `t` keeps the exact operation
// order of the hard mode (bit-identical), the sigmoids use expf + IEEE division.
#pragma once

#include "geom.hpp"

#pragma clang fp contract(off)

namespace drt {

// jax.nn.sigmoid = lax.logistic, expanded by XLA as 1 / (1 + exp(-x)); exp(+inf) -> 0, no NaN
// for +-inf inputs (utils.py:89, differt/tests/test_utils.py:59-81).
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float smoothing(float x, float alpha) { return sigmoid_f(x * alpha); }
// d/dx smoothing(x, alpha) given its value s
__device__ __forceinline__ float smoothing_grad(float s, float alpha) { return alpha * (s * (1.0f - s)); }

// jnp.minimum / jnp.maximum / reductions propagate NaN
__device__ __forceinline__ float nmin(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }
__device__ __forceinline__ float nmax(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }

// Intermediate values of one smoothed Moller-Trumbore test (_utils.py:1262-1322), kept for the reverse.
struct MtSmooth {
    float t, hit;
    int which;  // the term hit equals: 0 |a|-eps, 1 u, 2 1-u, 3 v, 4 1-(u+v), 5 t-eps; -1 = NaN / initial
};

__device__ __forceinline__ MtSmooth mt_smooth(V3 o, V3 d, const TriE &tr, float eps, float alpha) {
    const V3 h = cross(d, tr.e2);
    const float a0 = dot(h, tr.e1);
    const float a = (a0 == 0.0f) ? kInf : a0;
    float c[6];
    c[0] = smoothing(__builtin_fabsf(a) - eps, alpha);
    const float f = 1.0f / a;
    const V3 s = o - tr.v0;
    const float u = f * dot(s, h);
    c[1] = smoothing(u, alpha);
    c[2] = smoothing(1.0f - u, alpha);
    const V3 q = cross(s, tr.e1);
    const float v = f * dot(q, d);
    c[3] = smoothing(v, alpha);
    c[4] = smoothing(1.0f - (u + v), alpha);
    const float t = f * dot(q, tr.e2);
    c[5] = smoothing(t - eps, alpha);
    MtSmooth r;
    r.t = t;
    float hit = 1.0f;  // min(..., initial=1.0)
    int which = -1;
    bool nan = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        nan = nan || (c[i] != c[i]);
        if (c[i] < hit) {
            hit = c[i];
            which = i;
        }
    }
    r.hit = nan ? __builtin_nanf("") : hit;
    r.which = nan ? -1 : which;
    return r;
}

struct MtBar {
    V3 o, d, v0, v1, v2;
};

// Reverse of mt_smooth for the cotangents (t_bar, hit_bar).  The min routes hit_bar to the first
// smallest term (ties have measure zero; saturated terms carry a zero derivative anyway).
__device__ __forceinline__ MtBar mt_smooth_vjp(V3 o, V3 d, V3 v0, V3 v1, V3 v2, float eps, float alpha,
                                               float t_bar, float hit_bar) {
    const V3 e1 = v1 - v0, e2 = v2 - v0;
    const V3 h = cross(d, e2);
    const float a0 = dot(h, e1);
    const bool zero = (a0 == 0.0f);
    const float a = zero ? kInf : a0;
    const float f = 1.0f / a;
    const V3 s = o - v0;
    const float su = dot(s, h);
    const float u = f * su;
    const V3 q = cross(s, e1);
    const float sv = dot(q, d);
    const float v = f * sv;
    const float st = dot(q, e2);
    const float t = f * st;
    float c[6];
    c[0] = smoothing(__builtin_fabsf(a) - eps, alpha);
    c[1] = smoothing(u, alpha);
    c[2] = smoothing(1.0f - u, alpha);
    c[3] = smoothing(v, alpha);
    c[4] = smoothing(1.0f - (u + v), alpha);
    c[5] = smoothing(t - eps, alpha);
    float hit = 1.0f;
    int which = -1;
    bool nan = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        nan = nan || (c[i] != c[i]);
        if (c[i] < hit) {
            hit = c[i];
            which = i;
        }
    }
    float a_bar = 0.0f, u_bar = 0.0f, v_bar = 0.0f;
    if (!nan && which >= 0 && hit_bar != 0.0f) {
        const float g = hit_bar * smoothing_grad(hit, alpha);
        switch (which) {
            case 0: a_bar = (a < 0.0f) ? -g : g; break;
            case 1: u_bar = g; break;
            case 2: u_bar = -g; break;
            case 3: v_bar = g; break;
            case 4: u_bar = -g; v_bar = -g; break;
            default: t_bar += g; break;
        }
    }
    MtBar r;
    const V3 z{0, 0, 0};
    if (zero) {  // a is the constant +inf, f = 0: u = v = t = 0 * (...) carry no finite derivative
        r.o = r.d = r.v0 = r.v1 = r.v2 = z;
        return r;
    }
    const float f_bar = (u_bar * su + v_bar * sv) + t_bar * st;
    const float su_bar = u_bar * f, sv_bar = v_bar * f, st_bar = t_bar * f;
    a_bar -= f_bar * f * f;                       // f = 1 / a
    V3 h_bar = e1 * a_bar + s * su_bar;           // a = <h, e1>, su = <s, h>
    V3 e1_bar = h * a_bar;
    V3 s_bar = h * su_bar;
    const V3 q_bar = d * sv_bar + e2 * st_bar;    // sv = <q, d>, st = <q, e2>
    V3 d_bar = q * sv_bar;
    V3 e2_bar = q * st_bar;
    s_bar = s_bar + cross(e1, q_bar);             // q = s x e1
    e1_bar = e1_bar + cross(q_bar, s);
    d_bar = d_bar + cross(e2, h_bar);             // h = d x e2
    e2_bar = e2_bar + cross(h_bar, d);
    r.o = s_bar;                                   // s = o - v0, e1 = v1 - v0, e2 = v2 - v0
    r.d = d_bar;
    r.v1 = e1_bar;
    r.v2 = e2_bar;
    r.v0 = z - s_bar - e1_bar - e2_bar;
    return r;
}

// Contribution of one (segment, triangle) pair to the smoothed any-triangle sum
// (_utils.py:1465-1468): min(hit, sigmoid((thr - t) alpha)).
__device__ __forceinline__ float blocked_weight(V3 o, V3 d, const TriE &tr, float eps, float thr,
                                                float alpha) {
    const MtSmooth m = mt_smooth(o, d, tr, eps, alpha);
    return nmin(m.hit, smoothing(thr - m.t, alpha));
}

// Reverse of blocked_weight: w_bar -> cotangents of (o, d, triangle).
__device__ __forceinline__ MtBar blocked_weight_vjp(V3 o, V3 d, V3 v0, V3 v1, V3 v2, float eps, float thr,
                                                    float alpha, float w_bar) {
    const MtSmooth m = mt_smooth(o, d, make_tri(v0, v1, v2), eps, alpha);
    const float st = smoothing(thr - m.t, alpha);
    float t_bar = 0.0f, hit_bar = 0.0f;
    if (m.hit != m.hit || st != st) {
        // NaN: no gradient
    } else if (st < m.hit) {
        t_bar = -w_bar * smoothing_grad(st, alpha);
    } else {
        hit_bar = w_bar;
    }
    return mt_smooth_vjp(o, d, v0, v1, v2, eps, alpha, t_bar, hit_bar);
}

// sum over the wavefront, same value in every lane, fixed (butterfly) order
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

}  // namespace drt
