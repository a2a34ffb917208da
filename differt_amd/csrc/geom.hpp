// geom.hpp -- float32 device arithmetic of the hot path, one IEEE rounding per operation.
//
// Every decision the reference takes (u >= 0, u + v <= 1, t > eps, sign(dot), ...) flips on a
// 1-ulp difference, so these routines fix the operation order and are compiled WITHOUT FMA
// contraction (-ffp-contract=off plus the pragma below) and with correctly rounded / and sqrt
// (-fhip-fp32-correctly-rounded-divide-sqrt, hipcc's default).  3-term sums are associated
// left to right: (x0*y0 + x1*y1) + x2*y2.
//
// Reference formulas: Moller-Trumbore differt/src/differt/geometry/_utils.py:1263-1322;
// image / ray-plane differt/src/differt/geometry/_solver_image_method.py:68-79, 110-135, 152-182.
#pragma once

#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace drt {

struct V3 {
    float x, y, z;
};

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 ld3(const float *p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float *p, V3 a) {
    p[0] = a.x;
    p[1] = a.y;
    p[2] = a.z;
}
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }

__device__ __forceinline__ float dot(V3 a, V3 b) {
    float p0 = a.x * b.x;
    float p1 = a.y * b.y;
    float p2 = a.z * b.z;
    float s = p0 + p1;
    return s + p2;
}

__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

constexpr float kInf = __builtin_inff();

__device__ __forceinline__ bool is_inf(float x) { return __builtin_fabsf(x) == kInf; }
__device__ __forceinline__ bool is_finite(float x) { return __builtin_fabsf(x) < kInf; }

// Correctly rounded reciprocal for Moller-Trumbore's `f = 1 / where(a == 0, inf, a)` in about half
// the issue slots of the compiler's IEEE division expansion (v_div_scale x2, v_rcp, 4 fma, v_mul,
// v_div_fmas, v_div_fixup): v_rcp_f32 + ONE Newton step.  Verified EXHAUSTIVELY on gfx950 against
// `1.0f / a` for all 2^32 bit patterns (scratch/rcp_exhaustive.hip: 0 mismatches in the range where
// a and 1/a are both normal).  If ANY lane of the wave is outside that range (denormal / huge /
// nan), the whole wave takes the exact division instead -- a wave-uniform branch, no exec juggling.
// a == 0 (ray parallel to the triangle plane, common in axis-aligned scenes) stays on the fast
// path: the reference turns it into a = inf, whose reciprocal is +0.
__device__ __forceinline__ float mt_reciprocal(float a0, bool zero) {
    const float aa = __builtin_fabsf(a0);
    const bool ok = zero || (aa >= 0x1p-126f && aa <= 0x1p+126f);
    if (__builtin_expect(__all(ok), 1)) {
        const float r = __builtin_amdgcn_rcpf(a0);
        const float e = __builtin_fmaf(-a0, r, 1.0f);
        const float f = __builtin_fmaf(e, r, r);
        return zero ? 0.0f : f;
    }
    return 1.0f / (zero ? kInf : a0);
}

// A triangle prepared for Moller-Trumbore: v0 and the two edges e1 = v1 - v0, e2 = v2 - v0
// (_utils.py:1269-1271; computing the edges once per triangle is value-identical).
struct TriE {
    V3 v0, e1, e2;
};

__device__ __forceinline__ TriE make_tri(V3 v0, V3 v1, V3 v2) { return TriE{v0, v1 - v0, v2 - v0}; }
__device__ __forceinline__ TriE load_tri(const float *tv) {
    return make_tri(ld3(tv), ld3(tv + 3), ld3(tv + 6));
}

// Lower bound of the fast path's determinant range, folded with the reference's `|a| > eps` test:
// every |a| of the wave must EXCEED max(eps, largest denormal).  Either way |a| >= 2^-126 and |a| > eps
// follow, so the fast path needs no per-test `|a| > eps` compare; a determinant at or below eps sends
// the wave to the literal formula instead (same bits).  A NaN eps gives a NaN threshold: never fast.
__device__ __forceinline__ float mt_fast_threshold(float eps) {
    return (eps < 0x1p-126f) ? 0x1.fffffcp-127f : eps;
}

// _utils.py:1273-1322, hard mode.  Returns hit; t always written (also for misses).
// Fast path when every determinant of the WAVE lies in [2^-126, 2^126] (no zero / denormal / huge /
// non-finite value): reciprocal = v_rcp + one Newton step (exhaustively verified), no `a == 0`
// handling, `u <= 1` dropped (implied by v >= 0 && u + v <= 1: rounding is monotone), `u >= 0 && v >= 0`
// as min(u, v) >= 0 (a NaN in u or v makes u + v NaN, which fails u + v <= 1).  Otherwise the
// reference formula literally.  Same arithmetic, same bits either way (see moller_trumbore_x4).
__device__ __forceinline__ bool moller_trumbore(V3 o, V3 d, const TriE &tr, float eps, float &t_out) {
    const V3 h = cross(d, tr.e2);
    const float a0 = dot(h, tr.e1);
    const V3 s = o - tr.v0;
    const float pu = dot(s, h);
    const V3 q = cross(s, tr.e1);
    const float pv = dot(q, d);
    const float pt = dot(q, tr.e2);
    const float aa = __builtin_fabsf(a0);
    // two ballots of plain compares: the masks meet on the scalar unit (a ballot of `c1 & c2` costs a
    // v_cndmask + v_cmp to rebuild the lane mask)
    const uint64_t okm = __builtin_amdgcn_ballot_w64(aa > mt_fast_threshold(eps)) &
                         __builtin_amdgcn_ballot_w64(aa <= 0x1p+126f);
    if (__builtin_expect(okm == __builtin_amdgcn_read_exec(), 1)) {
        const float r = __builtin_amdgcn_rcpf(a0);
        const float e = __builtin_fmaf(-a0, r, 1.0f);
        const float f = __builtin_fmaf(e, r, r);
        const float u = f * pu;
        const float v = f * pv;
        const float upv = u + v;
        const float t = f * pt;
        t_out = t;
        return (__builtin_fminf(u, v) >= 0.0f) & (upv <= 1.0f) & (t > eps);
    }
    const bool zero = (a0 == 0.0f);                        // a = where(a == 0, inf, a)
    bool hit = (zero ? kInf : aa) > eps;                   // |a| > eps
    const float f = 1.0f / (zero ? kInf : a0);             // f = 1 / a
    const float u = f * pu;
    hit = hit && (u >= 0.0f) && (u <= 1.0f);
    const float v = f * pv;
    const float upv = u + v;
    hit = hit && (v >= 0.0f) && (upv <= 1.0f);
    const float t = f * pt;
    hit = hit && (t > eps);
    t_out = t;
    return hit;
}

// Wave-mask form for the tracer's filter stage: bit l of the result = lane l is in `want` and
// moller_trumbore(o, d, tr, eps) hits.  Same arithmetic; the masks of the single compares meet on the
// scalar unit, and <q, e2> / the `t > eps` test are evaluated only when some wanted lane passes the
// barycentric tests (wave-uniform branch; a rejected candidate's t is never used).
// WANT_T: also return t in `t_out`, meaningful for the lanes of a NON-ZERO result only.
template <bool WANT_T = false>
__device__ __forceinline__ uint64_t moller_trumbore_wave(V3 o, V3 d, const TriE &tr, float eps, uint64_t want,
                                                         float *t_out = nullptr) {
    const V3 h = cross(d, tr.e2);
    const float a0 = dot(h, tr.e1);
    const V3 s = o - tr.v0;
    const float pu = dot(s, h);
    const V3 q = cross(s, tr.e1);
    const float pv = dot(q, d);
    const float aa = __builtin_fabsf(a0);
    const uint64_t okm = __builtin_amdgcn_ballot_w64(aa > mt_fast_threshold(eps)) &
                         __builtin_amdgcn_ballot_w64(aa <= 0x1p+126f);
    // A determinant that is EXACTLY zero (segment in the triangle's plane: two mirrors on one wall plane of an
    // axis-aligned city, 10 % of the waves of configs[2]) is a miss in the reference (a -> inf, f = 0, t = 0 or
    // NaN, `t > eps` false) and on the fast path alike (rcp(0) = inf, e = NaN, u = NaN: every compare false), so
    // it need not send the wave to the literal formula; t of such a lane is never used (it is not a hit).
    bool fast = okm == __builtin_amdgcn_read_exec();
    if (__builtin_expect(!fast, 0))
        fast = (okm | __builtin_amdgcn_ballot_w64(aa == 0.0f)) == __builtin_amdgcn_read_exec();
    if (__builtin_expect(fast, 1)) {
        const float r = __builtin_amdgcn_rcpf(a0);
        const float e = __builtin_fmaf(-a0, r, 1.0f);
        const float f = __builtin_fmaf(e, r, r);
        const float u = f * pu;
        const float v = f * pv;
        const float upv = u + v;
        const uint64_t m = __builtin_amdgcn_ballot_w64(__builtin_fminf(u, v) >= 0.0f) &
                           __builtin_amdgcn_ballot_w64(upv <= 1.0f) & want;
        if (__builtin_expect(m == 0, 1)) return 0;
        const float t = f * dot(q, tr.e2);
        if (WANT_T) *t_out = t;
        return m & __builtin_amdgcn_ballot_w64(t > eps);
    }
#ifdef DRT_MT_DBG  // scratch instrumentation of trace.hip's debug build
    DRT_MT_DBG();
#endif
    const float pt = dot(q, tr.e2);
    const bool zero = (a0 == 0.0f);
    bool hit = (zero ? kInf : aa) > eps;
    const float f = 1.0f / (zero ? kInf : a0);
    const float u = f * pu;
    hit = hit && (u >= 0.0f) && (u <= 1.0f);
    const float v = f * pv;
    const float upv = u + v;
    hit = hit && (v >= 0.0f) && (upv <= 1.0f);
    const float t = f * pt;
    hit = hit && (t > eps);
    if (WANT_T) *t_out = t;
    return __builtin_amdgcn_ballot_w64(hit) & want;
}

// N independent tests of one ray against N triangles, arithmetic identical to moller_trumbore but
// phased so that the reciprocal's wave-uniform branch is taken once for all N (keeps the N
// dependency chains interleaved = instruction-level parallelism for the dense kernel).
template <int N>
__device__ __forceinline__ void moller_trumbore_n(V3 o, V3 d, const TriE (&tr)[N], float eps,
                                                  float (&t_out)[N], bool (&hit_out)[N]) {
    V3 h[N];
    float a0[N], f[N];
    bool zero[N], ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        h[i] = cross(d, tr[i].e2);
        a0[i] = dot(h[i], tr[i].e1);
        zero[i] = (a0[i] == 0.0f);
        const float aa = __builtin_fabsf(a0[i]);
        ok = ok && (zero[i] || (aa >= 0x1p-126f && aa <= 0x1p+126f));
    }
    if (__builtin_expect(__all(ok), 1)) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float r = __builtin_amdgcn_rcpf(a0[i]);
            const float e = __builtin_fmaf(-a0[i], r, 1.0f);
            f[i] = zero[i] ? 0.0f : __builtin_fmaf(e, r, r);
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) f[i] = 1.0f / (zero[i] ? kInf : a0[i]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        bool hit = (zero[i] ? kInf : __builtin_fabsf(a0[i])) > eps;
        const V3 s = o - tr[i].v0;
        const float u = f[i] * dot(s, h[i]);
        hit = hit && (u >= 0.0f) && (u <= 1.0f);
        const V3 q = cross(s, tr[i].e1);
        const float v = f[i] * dot(q, d);
        const float upv = u + v;
        hit = hit && (v >= 0.0f) && (upv <= 1.0f);
        const float t = f[i] * dot(q, tr[i].e2);
        hit = hit && (t > eps);
        t_out[i] = t;
        hit_out[i] = hit;
    }
}

// ------------------------------------------------------------------------------------------
// Dense-kernel formulation: 4 tests of ONE ray against a lane's 4 triangles, bit-identical to
// moller_trumbore() with fewer issue slots (55.75 instead of 62 VALU instructions per test):
//   * mt_phase1: the 41 arithmetic operations that do not depend on the reciprocal; only
//     (a, <s,h>, <q,d>, <q,e2>) stay live, so four tests fit in 16 VGPRs next to the triangles;
//   * fast path, taken when every |a| of the WAVE lies in [2^-126, 2^126] (no zero, denormal, huge
//     or non-finite determinant; one max3/min3 pair + 2 compares per 4 tests): `a == 0` cannot
//     occur, the reciprocal is v_rcp + one Newton step (exhaustively verified, see above);
//     `u <= 1` is implied by `v >= 0 && u + v <= 1` (rounding is monotone: u <= rn(u + v));
//     `u >= 0 && v >= 0` is `min(u, v) >= 0` (a NaN in u or v makes u + v NaN, which fails
//     `u + v <= 1`, so v_min's NaN-skipping cannot turn a miss into a hit);
//   * otherwise the reference formula literally (a = where(a == 0, inf, a); f = 1 / a; all six
//     comparisons) on the same phase-1 values;
//   * the four hit flags become the four bytes of one dword with one v_cndmask each (the mask
//     logic stays on the scalar unit).
// scratch/dense_lab.hip checks the formulation bit for bit against moller_trumbore_n on 6.5e8 tests
// incl. axis-aligned / zero / 1e18 / 1e-30 / NaN / inf directions and coplanar rays.
// ------------------------------------------------------------------------------------------
struct MtPart {
    float a0, pu, pv, pt;
};

__device__ __forceinline__ MtPart mt_phase1(V3 o, V3 d, const TriE &tr) {
    const V3 h = cross(d, tr.e2);
    MtPart p;
    p.a0 = dot(h, tr.e1);
    const V3 s = o - tr.v0;
    p.pu = dot(s, h);
    const V3 q = cross(s, tr.e1);
    p.pv = dot(q, d);
    p.pt = dot(q, tr.e2);
    return p;
}

// t_out[i] and byte i of `hits` = moller_trumbore(o, d, tr[i], eps, t_out[i])
__device__ __forceinline__ void moller_trumbore_x4(V3 o, V3 d, const TriE (&tr)[4], float eps,
                                                   float (&t_out)[4], uint32_t &hits) {
    MtPart p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = mt_phase1(o, d, tr[i]);
    const float mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(p[0].a0), __builtin_fabsf(p[1].a0)),
                                     __builtin_fmaxf(__builtin_fabsf(p[2].a0), __builtin_fabsf(p[3].a0)));
    const float mn = __builtin_fminf(__builtin_fminf(__builtin_fabsf(p[0].a0), __builtin_fabsf(p[1].a0)),
                                     __builtin_fminf(__builtin_fabsf(p[2].a0), __builtin_fabsf(p[3].a0)));
    // fmax / fmin skip a NaN operand: a NaN determinant next to in-range ones stays on the fast
    // path, where rcp / fma propagate it exactly like the division does (t = NaN, hit = false)
    const uint64_t okm = __builtin_amdgcn_ballot_w64(mn > mt_fast_threshold(eps)) &
                         __builtin_amdgcn_ballot_w64(mx <= 0x1p+126f);
    if (__builtin_expect(okm == __builtin_amdgcn_read_exec(), 1)) {
        uint32_t hh = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float r = __builtin_amdgcn_rcpf(p[i].a0);
            const float e = __builtin_fmaf(-p[i].a0, r, 1.0f);
            const float f = __builtin_fmaf(e, r, r);
            const float u = f * p[i].pu;
            const float v = f * p[i].pv;
            const float upv = u + v;
            const float t = f * p[i].pt;
            const bool c1 = __builtin_fminf(u, v) >= 0.0f;
            const bool c2 = upv <= 1.0f;
            const bool c3 = t > eps;
            t_out[i] = t;
            // one v_cndmask per test + 2 ORs per 4 tests (SDWA byte selects measured 8x the issue
            // cost of a plain VALU instruction on gfx950: scratch/valu_mix.hip)
            hh |= (c1 & (c2 & c3)) ? (1u << (8 * i)) : 0u;
        }
        hits = hh;
    } else {
        uint32_t hh = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool zero = (p[i].a0 == 0.0f);
            const float f = 1.0f / (zero ? kInf : p[i].a0);
            bool hit = (zero ? kInf : __builtin_fabsf(p[i].a0)) > eps;
            const float u = f * p[i].pu;
            hit = hit & (u >= 0.0f) & (u <= 1.0f);
            const float v = f * p[i].pv;
            const float upv = u + v;
            hit = hit & (v >= 0.0f) & (upv <= 1.0f);
            const float t = f * p[i].pt;
            hit = hit & (t > eps);
            t_out[i] = t;
            hh |= (uint32_t)hit << (8 * i);
        }
        hits = hh;
    }
}

// Reverse mode of t = f * <q, e2> (hard-mode Moller-Trumbore, _utils.py:1263-1316; the reference gets
// it from jax autodiff, `hit` carries no gradient): cotangents of (o, d, v0, v1, v2) for t_bar.
// a = where(a == 0, inf, a) is a constant branch: f = 0 and no gradient flows through a.
struct MtHardBar {
    V3 o, d, v0, v1, v2;
};
__device__ __forceinline__ MtHardBar mt_t_vjp(V3 o, V3 d, V3 v0, V3 v1, V3 v2, float tbar) {
    const V3 e1 = v1 - v0, e2 = v2 - v0;
    const V3 h = cross(d, e2);
    float a = dot(h, e1);
    const bool degenerate = (a == 0.0f);
    a = degenerate ? kInf : a;
    const float f = 1.0f / a;
    const V3 s = o - v0;
    const V3 q = cross(s, e1);
    const float w = dot(q, e2);
    const float fbar = tbar * w, wbar = tbar * f;  // t = f * w
    const V3 qbar = e2 * wbar;
    V3 e2bar = q * wbar;
    const V3 sbar = cross(e1, qbar);  // q = s x e1
    V3 e1bar = cross(qbar, s);
    const float abar = degenerate ? 0.0f : -(fbar * f) * f;  // f = 1 / a
    const V3 hbar = e1 * abar;                               // a = <h, e1>
    e1bar = e1bar + h * abar;
    MtHardBar r;
    r.d = cross(e2, hbar);  // h = d x e2
    e2bar = e2bar + cross(hbar, d);
    r.o = sbar;  // s = o - v0
    r.v0 = V3{0, 0, 0} - sbar - e1bar - e2bar;
    r.v1 = e1bar;
    r.v2 = e2bar;
    return r;
}

// _solver_image_method.py:73-79: x - (2 * <x - p, n>) * n
__device__ __forceinline__ V3 image_of_vertex(V3 x, V3 p, V3 n) {
    float c = 2.0f * dot(x - p, n);
    return V3{x.x - c * n.x, x.y - c * n.y, x.z - c * n.z};
}

// _solver_image_method.py:116-135
__device__ __forceinline__ V3 ray_plane(V3 o, V3 d, V3 p, V3 n) {
    V3 v = p - o;
    float un = dot(d, n);
    float vn = dot(v, n);
    bool parallel = (un == 0.0f);
    un = parallel ? 1.0f : un;
    float t = vn / un;
    V3 r = V3{o.x + d.x * t, o.y + d.y * t, o.z + d.z * t};
    bool bad = parallel && (vn != 0.0f);
    return bad ? V3{kInf, kInf, kInf} : r;
}

// _solver_image_method.py:160-182 (_backward): one reverse-scan step from `prev` through mirror
// (p, n) towards `image`; infinite components of prev are zeroed for the arithmetic and the
// corresponding output components forced back to +inf.
__device__ __forceinline__ V3 backward_step(V3 prev, V3 image, V3 p, V3 n) {
    bool ix = is_inf(prev.x), iy = is_inf(prev.y), iz = is_inf(prev.z);
    V3 pi = V3{ix ? 0.0f : prev.x, iy ? 0.0f : prev.y, iz ? 0.0f : prev.z};
    V3 x = ray_plane(pi, image - pi, p, n);
    return V3{ix ? kInf : x.x, iy ? kInf : x.y, iz ? kInf : x.z};
}

// jnp.sign semantics for the same-side test (_solver_image_method.py:443-454): nan != nan.
__device__ __forceinline__ bool same_sign(float a, float b) {
    float sa = (a > 0.0f) ? 1.0f : ((a < 0.0f) ? -1.0f : ((a == a) ? 0.0f : a));
    float sb = (b > 0.0f) ? 1.0f : ((b < 0.0f) ? -1.0f : ((b == b) ? 0.0f : b));
    return sa == sb;
}

// monotone map float -> uint32 (total order of finite values and infinities; used for packed
// (t, tie) first-hit keys).
__device__ __forceinline__ uint32_t float_to_ordered(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(uint32_t u) {
    uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(b);
}

}  // namespace drt
