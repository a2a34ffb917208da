// Does FP32 VALU work overlap with the dense operator's [R,T] f32 + u8 store pattern on MI355X?
// Per ray each lane does K independent mul/add pairs (4 chains, like the 4 triangles of a lane in
// mt_dense_kernel) and then stores 16 B of `t` + 4 B of `hit`.  MODE: 0 = VALU only (results sunk
// once at the end), 1 = stores only, 2 = both, 3 = both with buffer stores (SGPR row offset).
// build: hipcc --offload-arch=gfx950 -O3 -o store_overlap store_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u4 make_rsrc(const void *p, uint32_t bytes) {
    uint64_t a = (uint64_t)p;
    u4 r;
    r.x = (uint32_t)a;
    r.y = (uint32_t)(a >> 32) & 0xffffu;
    r.z = bytes;
    r.w = 0x00020000u;  // raw buffer, DATA_FORMAT = 32 (gfx9 family)
    return r;
}

template <int MODE, int K>
__global__ __launch_bounds__(256) void rows(float *t, uint8_t *h, int64_t R, int64_t T, int rpb, float seed) {
    const int64_t j0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (j0 >= T) return;
    const int64_t r0 = (int64_t)blockIdx.y * rpb;
    f4 v = {seed, seed + 1.f, seed + 2.f, (float)threadIdx.x};
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t r = r0; r < r0 + rpb && r < R; ++r) {
        const float s = __builtin_amdgcn_readfirstlane((int)(r & 7)) * 1e-3f + 1.0f;  // wave-uniform, varies per ray
        f4 x = v;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            x.x = x.x * s + 0.5f;
            x.y = x.y * s + 0.25f;
            x.z = x.z * s + 0.125f;
            x.w = x.w * s + 0.0625f;
        }
        if (MODE == 0) {
            acc += x;
        } else if (MODE == 3) {
            // not used here
        } else {
            const f4 out = (MODE == 1) ? v : x;
            __builtin_nontemporal_store(out, (f4 *)(t + r * T + j0));
            const uint32_t hv = (MODE == 1) ? 0x01000100u : (uint32_t)(x.x > 1.0f) | ((uint32_t)(x.y > 1.0f) << 8) |
                                                                ((uint32_t)(x.z > 1.0f) << 16) | ((uint32_t)(x.w > 1.0f) << 24);
            __builtin_nontemporal_store(hv, (uint32_t *)(h + r * T + j0));
        }
    }
    if (MODE == 0 && acc.x + acc.y + acc.z + acc.w == 12345.678f) t[j0] = acc.x;
}

int main() {
    const int64_t R = 65536, T = 10000;
    const size_t bytes = (size_t)R * T * 5;
    char *buf;
    hipMalloc(&buf, bytes);
    float *t = (float *)buf;
    uint8_t *h = (uint8_t *)(buf + (size_t)R * T * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](const char *name, auto fn) {
        fn();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) fn();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %.3f ms\n", name, ms / 10);
    };
    const int rpb = 64;
    const dim3 grid(10, R / rpb);
#define RUN(K)                                                                                         \
    time("K=" #K " valu only", [&] { rows<0, K><<<grid, 256>>>(t, h, R, T, rpb, 1.0f); });             \
    time("K=" #K " valu + stores", [&] { rows<2, K><<<grid, 256>>>(t, h, R, T, rpb, 1.0f); });
    time("stores only", [&] { rows<1, 0><<<grid, 256>>>(t, h, R, T, rpb, 1.0f); });
    RUN(8)
    RUN(16)
    RUN(24)
    RUN(32)
    RUN(40)
    RUN(48)
    RUN(64)
    return 0;
}
