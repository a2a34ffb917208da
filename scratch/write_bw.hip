// Write-only HBM bandwidth ceiling on MI355X: float4 fill kernels (plain / nontemporal), the mix of
// 16-B + 4-B stores the dense kernel issues, and hipMemsetAsync; 3.3 GB per pass.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT> __global__ __launch_bounds__(256) void fill(f4 *p, size_t n) {
    f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
}
// rows of T floats + T bytes, lane writes 16 B + 4 B like mt_dense_kernel (R rows)
template <bool NT> __global__ __launch_bounds__(256) void fill_rows(float *t, uint8_t *h, int64_t R, int64_t T, int rpb) {
    int64_t j0 = ((int64_t)blockIdx.y * 256 + threadIdx.x) * 4; if (j0 >= T) return;
    int64_t r0 = (int64_t)blockIdx.x * rpb;
    f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (int64_t r = r0; r < r0 + rpb && r < R; ++r) {
        if (NT) { __builtin_nontemporal_store(v, (f4 *)(t + r * T + j0)); __builtin_nontemporal_store(0x01000100u, (uint32_t *)(h + r * T + j0)); }
        else { *(f4 *)(t + r * T + j0) = v; *(uint32_t *)(h + r * T + j0) = 0x01000100u; }
    }
}
// same stores, but blockIdx.x = triangle column (fast) so the 10 column blocks of one ray chunk run
// together and every output row is completed within a short time window
template <bool NT> __global__ __launch_bounds__(256) void fill_rows_xcol(float *t, uint8_t *h, int64_t R, int64_t T, int rpb) {
    int64_t j0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; if (j0 >= T) return;
    int64_t r0 = (int64_t)blockIdx.y * rpb;
    f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (int64_t r = r0; r < r0 + rpb && r < R; ++r) {
        if (NT) { __builtin_nontemporal_store(v, (f4 *)(t + r * T + j0)); __builtin_nontemporal_store(0x01000100u, (uint32_t *)(h + r * T + j0)); }
        else { *(f4 *)(t + r * T + j0) = v; *(uint32_t *)(h + r * T + j0) = 0x01000100u; }
    }
}
// MODE 0: t only (16 B/lane); 1: hit only (4 B/lane); 2: hit only, 16 B from every 4th lane; 3: t + hit(16B from every 4th lane)
template <int MODE> __global__ __launch_bounds__(256) void fill_parts(float *t, uint8_t *h, int64_t R, int64_t T, int rpb) {
    int64_t j0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; if (j0 >= T) return;
    int64_t r0 = (int64_t)blockIdx.y * rpb;
    f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 hv = {0x01000100u, 0x01000100u, 0x01000100u, 0x01000100u};
    for (int64_t r = r0; r < r0 + rpb && r < R; ++r) {
        if (MODE == 0 || MODE == 3) __builtin_nontemporal_store(v, (f4 *)(t + r * T + j0));
        if (MODE == 1) __builtin_nontemporal_store(0x01000100u, (uint32_t *)(h + r * T + j0));
        if (MODE == 2 || MODE == 3) { if ((threadIdx.x & 3) == 0 && j0 + 16 <= T) __builtin_nontemporal_store(hv, (u4 *)(h + r * T + j0)); }
    }
}
// XCD-aware order: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own L2); here the
// column blocks of one chunk of rows all land on the SAME XCD so that its L2 sees whole rows
template <bool NT> __global__ __launch_bounds__(256) void fill_rows_xcd(float *t, uint8_t *h, int64_t R, int64_t T, int rpb, int ncol) {
    const int64_t bid = blockIdx.x;
    const int64_t xcd = bid & 7, q = bid >> 3;
    const int64_t col = q % ncol, rc = (q / ncol) * 8 + xcd;
    int64_t j0 = (col * 256 + threadIdx.x) * 4; if (j0 >= T) return;
    int64_t r0 = rc * rpb;
    f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (int64_t r = r0; r < r0 + rpb && r < R; ++r) {
        if (NT) { __builtin_nontemporal_store(v, (f4 *)(t + r * T + j0)); __builtin_nontemporal_store(0x01000100u, (uint32_t *)(h + r * T + j0)); }
        else { *(f4 *)(t + r * T + j0) = v; *(uint32_t *)(h + r * T + j0) = 0x01000100u; }
    }
}
template <int THREADS> __global__ __launch_bounds__(THREADS) void fill_rows_bs(float *t, uint8_t *h, int64_t R, int64_t T, int rpb) {
    int64_t j0 = ((int64_t)blockIdx.x * THREADS + threadIdx.x) * 4; if (j0 >= T) return;
    int64_t r0 = (int64_t)blockIdx.y * rpb;
    f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (int64_t r = r0; r < r0 + rpb && r < R; ++r) {
        __builtin_nontemporal_store(v, (f4 *)(t + r * T + j0)); __builtin_nontemporal_store(0x01000100u, (uint32_t *)(h + r * T + j0));
    }
}
int main() {
    const int64_t R = 65536, T = 10000; size_t bytes = (size_t)R * T * 5;
    char *buf; hipMalloc(&buf, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto fn) {
        fn(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 10; ++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10; printf("%-28s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
    };
    size_t n4 = bytes / 16;
    time("fill float4 plain", [&] { fill<false><<<256 * 16, 256>>>((f4 *)buf, n4); });
    time("fill float4 nontemporal", [&] { fill<true><<<256 * 16, 256>>>((f4 *)buf, n4); });
    time("fill float4 plain, big grid", [&] { fill<false><<<(unsigned)((n4 + 255) / 256), 256>>>((f4 *)buf, n4); });
    time("hipMemsetAsync", [&] { hipMemsetAsync(buf, 1, bytes, 0); });
    float *t = (float *)buf; uint8_t *h = (uint8_t *)(buf + (size_t)R * T * 4);
    time("rows 16B+4B plain rpb=64", [&] { fill_rows<false><<<dim3(R / 64, 10), 256>>>(t, h, R, T, 64); });
    time("rows 16B+4B nt rpb=64", [&] { fill_rows<true><<<dim3(R / 64, 10), 256>>>(t, h, R, T, 64); });
    time("rows 16B+4B nt rpb=16", [&] { fill_rows<true><<<dim3(R / 16, 10), 256>>>(t, h, R, T, 16); });
    time("parts t only (80% bytes)", [&] { fill_parts<0><<<dim3(10, R / 16), 256>>>(t, h, R, T, 16); });
    time("parts hit only 4B/lane", [&] { fill_parts<1><<<dim3(10, R / 16), 256>>>(t, h, R, T, 16); });
    time("parts hit only 16B/4th lane", [&] { fill_parts<2><<<dim3(10, R / 16), 256>>>(t, h, R, T, 16); });
    time("parts t + hit 16B/4th lane", [&] { fill_parts<3><<<dim3(10, R / 16), 256>>>(t, h, R, T, 16); });
    time("block 512  rpb=16", [&] { fill_rows_bs<512><<<dim3(5, R / 16), 512>>>(t, h, R, T, 16); });
    time("block 1024 rpb=16", [&] { fill_rows_bs<1024><<<dim3(3, R / 16), 1024>>>(t, h, R, T, 16); });
    time("block 1024 rpb=4", [&] { fill_rows_bs<1024><<<dim3(3, R / 4), 1024>>>(t, h, R, T, 4); });
    time("block 1024 rpb=64", [&] { fill_rows_bs<1024><<<dim3(3, R / 64), 1024>>>(t, h, R, T, 64); });
    for (int rpb : {4, 16, 64}) {
        char name[64];
        snprintf(name, 64, "xcd-aware plain rpb=%d", rpb);
        time(name, [&] { fill_rows_xcd<false><<<dim3(10 * (R / rpb)), 256>>>(t, h, R, T, rpb, 10); });
        snprintf(name, 64, "xcd-aware nt    rpb=%d", rpb);
        time(name, [&] { fill_rows_xcd<true><<<dim3(10 * (R / rpb)), 256>>>(t, h, R, T, rpb, 10); });
    }
    for (int rpb : {16}) {
        char name[64];
        snprintf(name, 64, "xcol plain rpb=%d", rpb);
        time(name, [&] { fill_rows_xcol<false><<<dim3(10, R / rpb), 256>>>(t, h, R, T, rpb); });
        snprintf(name, 64, "xcol nt    rpb=%d", rpb);
        time(name, [&] { fill_rows_xcol<true><<<dim3(10, R / rpb), 256>>>(t, h, R, T, rpb); });
    }
    return 0;
}
