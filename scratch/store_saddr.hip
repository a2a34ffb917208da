#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void rows_vaddr(float *t, uint8_t *h, int64_t R, int64_t T, int rpb) {
    int64_t j0 = ((int64_t)blockIdx.y * 256 + threadIdx.x) * 4; if (j0 >= T) return;
    int64_t r0 = (int64_t)blockIdx.x * rpb;
    f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (int64_t r = r0; r < r0 + rpb && r < R; ++r) {
        __builtin_nontemporal_store(v, (f4 *)(t + r * T + j0));
        __builtin_nontemporal_store(0x01000100u, (uint32_t *)(h + r * T + j0));
    }
}
// buffer stores: V# (SGPRs) describes the block's slab of rows; lane offset 32-bit VGPR, row offset SGPR
__global__ __launch_bounds__(256) void rows_buffer(float *t, uint8_t *h, int64_t R, int64_t T, int rpb) {
    const uint32_t j0 = (blockIdx.y * 256u + threadIdx.x) * 4u; if (j0 >= T) return;
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    __amdgpu_buffer_rsrc_t tr = __builtin_amdgcn_make_buffer_rsrc(t + r0 * T, 0, (int)(rpb * T * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(h + r0 * T, 0, (int)(rpb * T), 0x00020000);
    const int n = (r0 + rpb <= R) ? rpb : (int)(R - r0);
    for (int i = 0; i < n; ++i) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), tr, j0 * 4u, (int)(i * T * 4), 0);
        __builtin_amdgcn_raw_buffer_store_b32(0x01000100u, hr, j0, (int)(i * T), 0);
    }
}
int main() {
    const int64_t R = 65536, T = 10000; size_t bytes = (size_t)R * T * 5;
    char *buf; hipMalloc(&buf, bytes);
    float *t = (float *)buf; uint8_t *h = (uint8_t *)(buf + (size_t)R * T * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto fn) {
        fn(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 10; ++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-20s %.3f ms\n", name, ms / 10);
    };
    for (int rep = 0; rep < 3; ++rep) {
        time("vaddr  rpb=64", [&] { rows_vaddr<<<dim3(R / 64, 10), 256>>>(t, h, R, T, 64); });
        time("buffer rpb=64", [&] { rows_buffer<<<dim3(R / 64, 10), 256>>>(t, h, R, T, 64); });
    }
    // verify the buffer variant wrote what the vaddr variant writes
    hipMemset(buf, 0, bytes);
    rows_buffer<<<dim3(R / 64, 10), 256>>>(t, h, R, T, 64);
    float ht[8]; uint8_t hh[8];
    hipMemcpy(ht, t + 12345 * T + 9996, 16, hipMemcpyDeviceToHost);
    hipMemcpy(hh, h + 12345 * T + 9996, 4, hipMemcpyDeviceToHost);
    printf("check %g %g %g %g  %d %d %d %d\n", ht[0], ht[1], ht[2], ht[3], hh[0], hh[1], hh[2], hh[3]);
    return 0;
}
