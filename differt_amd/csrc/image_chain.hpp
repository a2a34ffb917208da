// image_chain.hpp -- the image method for ONE path candidate, fully unrolled in registers, and its
// hand-derived reverse (VJP).  Shared by the stand-alone operator (image_method.hip) and by the
// fused trace kernels (trace.hip).
//
// Forward  (reference geometry/_solver_image_method.py:138-203):
//   I_0 = from;  I_j = image(I_{j-1}; p_j, n_j)                 (forward lax.scan  :191-195)
//   X_{K+1} = to; X_j = intersect(X_{j+1}, I_j - X_{j+1}; p_j, n_j)   (reverse lax.scan :196-201)
// Reverse: SURVEY.md appendix A, with the reference's where-guards (:123-135, :165-181) mirrored so
// that gradients stay NaN-free on parallel / infinite configurations.
#pragma once

#include "geom.hpp"

#pragma clang fp contract(off)

namespace drt {

// path[0..K-1] = X_1..X_K (end points excluded), bit-identical to the oracle's operation order.
template <int K>
__device__ __forceinline__ void image_chain(V3 from, V3 to, const V3 (&p)[K], const V3 (&n)[K],
                                            V3 (&path)[K]) {
    V3 img[K];
    V3 prev = from;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        img[j] = image_of_vertex(prev, p[j], n[j]);
        prev = img[j];
    }
    V3 cur = to;
#pragma unroll
    for (int j = K - 1; j >= 0; --j) {
        cur = backward_step(cur, img[j], p[j], n[j]);
        path[j] = cur;
    }
}

// Reverse mode.  g[j] = dL/dX_{j+1} (cotangent of path[j]).  Outputs: cotangents of from, to and of
// every mirror point / normal.  Arithmetic here is not decision-critical (tolerance 1e-5 rel).
template <int K>
__device__ __forceinline__ void image_chain_vjp(V3 from, V3 to, const V3 (&p)[K], const V3 (&n)[K],
                                                const V3 (&g)[K], V3 &from_bar, V3 &to_bar,
                                                V3 (&p_bar)[K], V3 (&n_bar)[K]) {
    // ---- recompute the forward pass, keeping what the reverse needs ----
    V3 img[K];      // I_j
    V3 inc[K];      // I_{j-1} - p_j
    float cc[K];    // c_j = 2 <inc_j, n_j>
    V3 prev = from;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        inc[j] = prev - p[j];
        cc[j] = 2.0f * dot(inc[j], n[j]);
        img[j] = V3{prev.x - cc[j] * n[j].x, prev.y - cc[j] * n[j].y, prev.z - cc[j] * n[j].z};
        prev = img[j];
    }
    V3 pin[K];      // previous intersection with infinities zeroed (ray origin o)
    V3 dir[K];      // I_j - o
    float un[K], tt[K];
    bool par[K], bad[K];
    bool infx[K], infy[K], infz[K];
    V3 cur = to;
#pragma unroll
    for (int j = K - 1; j >= 0; --j) {
        infx[j] = is_inf(cur.x);
        infy[j] = is_inf(cur.y);
        infz[j] = is_inf(cur.z);
        pin[j] = V3{infx[j] ? 0.0f : cur.x, infy[j] ? 0.0f : cur.y, infz[j] ? 0.0f : cur.z};
        dir[j] = img[j] - pin[j];
        const V3 v = p[j] - pin[j];
        float u = dot(dir[j], n[j]);
        const float vn = dot(v, n[j]);
        par[j] = (u == 0.0f);
        u = par[j] ? 1.0f : u;
        un[j] = u;
        tt[j] = vn / u;
        bad[j] = par[j] && (vn != 0.0f);
        V3 x = V3{pin[j].x + dir[j].x * tt[j], pin[j].y + dir[j].y * tt[j], pin[j].z + dir[j].z * tt[j]};
        if (bad[j]) x = V3{kInf, kInf, kInf};
        cur = V3{infx[j] ? kInf : x.x, infy[j] ? kInf : x.y, infz[j] ? kInf : x.z};
    }

    // ---- reverse of the reverse scan (j = 0 .. K-1 in output order X_1 .. X_K) ----
    V3 img_bar[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        img_bar[j] = V3{0, 0, 0};
        p_bar[j] = V3{0, 0, 0};
        n_bar[j] = V3{0, 0, 0};
    }
    V3 carry{0, 0, 0};  // cotangent flowing into X_{j+1} from X_j's computation
#pragma unroll
    for (int j = 0; j < K; ++j) {
        V3 xbar = g[j] + carry;
        // forced-to-inf components are constants
        xbar = V3{infx[j] ? 0.0f : xbar.x, infy[j] ? 0.0f : xbar.y, infz[j] ? 0.0f : xbar.z};
        if (bad[j]) xbar = V3{0, 0, 0};
        // x = o + d * t
        V3 obar = xbar;
        V3 dbar = xbar * tt[j];
        const float tbar = dot(xbar, dir[j]);
        // t = vn / un
        const float vnbar = tbar / un[j];
        const float unbar = par[j] ? 0.0f : -(tbar * tt[j]) / un[j];
        // vn = <p - o, n>;  un = <d, n>
        const V3 v = p[j] - pin[j];
        const V3 vbar = n[j] * vnbar;
        p_bar[j] = p_bar[j] + vbar;
        obar = obar - vbar;
        n_bar[j] = n_bar[j] + v * vnbar + dir[j] * unbar;
        dbar = dbar + n[j] * unbar;
        // d = I_j - o
        img_bar[j] = img_bar[j] + dbar;
        obar = obar - dbar;
        // o = where(isinf(prev), 0, prev)
        carry = V3{infx[j] ? 0.0f : obar.x, infy[j] ? 0.0f : obar.y, infz[j] ? 0.0f : obar.z};
    }
    to_bar = carry;

    // ---- reverse of the forward scan ----
    V3 acc{0, 0, 0};  // cotangent of I_j accumulated from I_{j+1}
#pragma unroll
    for (int j = K - 1; j >= 0; --j) {
        const V3 ib = img_bar[j] + acc;
        // I_j = I_{j-1} - c * n
        const float cbar = -dot(ib, n[j]);
        n_bar[j] = n_bar[j] - ib * cc[j];
        // c = 2 <inc, n>
        const float dotbar = 2.0f * cbar;
        const V3 incbar = n[j] * dotbar;
        n_bar[j] = n_bar[j] + inc[j] * dotbar;
        p_bar[j] = p_bar[j] - incbar;
        acc = ib + incbar;
    }
    from_bar = acc;
}

}  // namespace drt
