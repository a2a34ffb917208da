// lattice.hpp -- Fibonacci-lattice ray directions (reference geometry/_utils.py:369-490), shared by the
// brute-force and the BVH visibility kernels.
#pragma once

#include "geom.hpp"

#pragma clang fp contract(off)

namespace drt {

constexpr float kPi = 3.14159274101257324f;      // float32(pi)
constexpr float kTwoPi = 6.28318548202514648f;   // float32(2*pi)

// direction of lattice point i of n inside the frustum (or on the full sphere when fr == nullptr)
__device__ __forceinline__ V3 lattice_direction(int64_t i_int, int64_t n, const float *fr) {
    const float i = (float)i_int;
    const float inv_phi = 0.6180339887498949f;
    const float m1 = 262144.0f, m2 = 512.0f;
    const float inv_phi_m1 = (float)(0.6180339887498949 * 262144.0 - 162013.0);  // (inv_phi*m1) % 1
    const float inv_phi_m2 = (float)(0.6180339887498949 * 512.0 - 316.0);        // (inv_phi*m2) % 1
    const float q1 = floorf(i / m1);
    const float rem = i - q1 * m1;
    const float q2 = floorf(rem / m2);
    const float r = rem - q2 * m2;
    const float frac = fmodf((q1 * inv_phi_m1 + q2 * inv_phi_m2) + r * inv_phi, 1.0f);
    float lat, lon;
    if (fr) {
        const float p_min = fr[1], a_min = fr[2], p_max = fr[4], a_max = fr[5];
        const float c0 = cosf(p_min), c1 = cosf(p_max);
        const float denom = (n > 1) ? (float)(n - 1) : 1.0f;
        lat = acosf(c0 - (c0 - c1) * (i / denom));
        lon = a_min + (a_max - a_min) * frac;
    } else {
        lat = acosf(1.0f - (2.0f * i) / (float)n);
        lon = kTwoPi * frac;
    }
    const float sp = sinf(lat), cp = cosf(lat);
    return V3{sp * cosf(lon), sp * sinf(lon), cp};
}


void launch_frustum_kernel(const float *view, int64_t B, const float *tv, int64_t T,
                           const uint8_t *active, float *out, hipStream_t s);

}  // namespace drt
