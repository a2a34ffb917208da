// store_lab.hip -- which properties of the dense operator's [R,T] f32 + u8 output pattern cost HBM
// write bandwidth on MI355X?  Write-only kernels, no arithmetic.  3.28 GB per pass at T = 10000.
//   fill16 / fill4   : flat grid-stride fills, 16 B or 4 B per lane (per-instruction cost vs bytes)
//   rows<MODE>       : the dense kernel's pattern; MODE bit0 = t (16 B/lane), bit1 = hit 4 B/lane,
//                      bit2 = hit as 16 B/lane covering 4 consecutive rows (quad transposition)
//   T = 10000 (rows 64-B / 16-B aligned) vs T = 10240 (rows 4-KiB / 1-KiB aligned): alignment effect
// build: hipcc --offload-arch=gfx950 -O3 -o scratch/store_lab scratch/store_lab.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fill16(f4 *p, size_t n) {
    f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        __builtin_nontemporal_store(v, p + i);
}
__global__ __launch_bounds__(256) void fill4(uint32_t *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        __builtin_nontemporal_store((uint32_t)threadIdx.x, p + i);
}
// one store per wave then exit (big grid)
__global__ __launch_bounds__(256) void fill16_once(f4 *p, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    if (i < n) __builtin_nontemporal_store(v, p + i);
}

template <int MODE, bool NT>
__global__ __launch_bounds__(256) void rows(float *t, uint8_t *h, int64_t R, int64_t T, int rpb) {
    const uint32_t j0 = (blockIdx.y * 256u + threadIdx.x) * 4u;
    if (j0 >= T) return;
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int n = (int)((r0 + rpb < R) ? rpb : R - r0);
    const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    const u4 hv = {0x01000100u, 0x01000100u, 0x01000100u, (uint32_t)threadIdx.x};
    char *trow = (char *)(t + r0 * T);
    char *hrow = (char *)(h + r0 * T);
    const uint32_t toff = j0 * 4u, hoff = j0;
    const uint32_t k = threadIdx.x & 3u;
    const uint32_t hq = (j0 - 4u * k) + k * (uint32_t)T;
    for (int i = 0; i < n; ++i) {
        if (MODE & 1) {
            if (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(toff), "v"(v), "s"(trow) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(toff), "v"(v), "s"(trow) : "memory");
        }
        if (MODE & 2) {
            if (NT) asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(hoff), "v"(hv.x), "s"(hrow) : "memory");
            else asm volatile("global_store_dword %0, %1, %2" ::"v"(hoff), "v"(hv.x), "s"(hrow) : "memory");
        }
        if ((MODE & 4) && (i & 3) == 0) {
            if (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(hq), "v"(hv), "s"(hrow) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(hq), "v"(hv), "s"(hrow) : "memory");
        }
        trow += T * 4;
        hrow += T;
    }
}

// whole-row blocks: a block of 256 threads owns `rpb` complete rows and walks along each row in
// 4-KiB (t) / 1-KiB (hit) steps -> every block writes two contiguous regions, like a fill
template <int MODE>
__global__ __launch_bounds__(256) void rows_whole(float *t, uint8_t *h, int64_t R, int64_t T, int rpb) {
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int n = (int)((r0 + rpb < R) ? rpb : R - r0);
    const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    const uint32_t hv = 0x01000100u;
    for (int i = 0; i < n; ++i) {
        char *trow = (char *)(t + (r0 + i) * T);
        char *hrow = (char *)(h + (r0 + i) * T);
        for (uint32_t j0 = threadIdx.x * 4u; j0 < T; j0 += 1024u) {
            const uint32_t toff = j0 * 4u;
            if (MODE & 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(toff), "v"(v), "s"(trow) : "memory");
            if (MODE & 2) asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(j0), "v"(hv), "s"(hrow) : "memory");
        }
    }
}

int main() {
    const int64_t R = 65536, Tmax = 10240;
    const size_t cap = (size_t)R * Tmax * 5;
    char *buf;
    if (hipMalloc(&buf, cap) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](const char *name, double bytes, auto fn) {
        fn();
        fn();
        hipDeviceSynchronize();
        std::vector<float> ms;
        for (int it = 0; it < 7; ++it) {
            hipEventRecord(e0);
            fn();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float m;
            hipEventElapsedTime(&m, e0, e1);
            ms.push_back(m);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-44s min %.3f med %.3f ms  %.2f TB/s\n", name, ms[0], ms[3], bytes / ms[3] / 1e9);
        fflush(stdout);
    };
    {
        const double b = (double)R * 10000 * 5;
        const size_t n16 = (size_t)(b / 16), n4 = (size_t)(b / 4);
        time("fill16 grid 4096", b, [&] { fill16<<<4096, 256>>>((f4 *)buf, n16); });
        time("fill16 grid 2048", b, [&] { fill16<<<2048, 256>>>((f4 *)buf, n16); });
        time("fill16 one store per wave", b, [&] { fill16_once<<<(unsigned)((n16 + 255) / 256), 256>>>((f4 *)buf, n16); });
        time("fill4  grid 4096 (same bytes)", b, [&] { fill4<<<4096, 256>>>((uint32_t *)buf, n4); });
        time("fill4  grid 4096 (1/5 bytes)", b / 5, [&] { fill4<<<4096, 256>>>((uint32_t *)buf, n4 / 5); });
        time("hipMemsetAsync", b, [&] { hipMemsetAsync(buf, 1, (size_t)b, 0); });
    }
    for (int64_t T : {10000, 10240}) {
        float *t = (float *)buf;
        uint8_t *h = (uint8_t *)(buf + (size_t)R * T * 4);
        const double bt = (double)R * T * 4, bh = (double)R * T;
        const unsigned cols = (unsigned)((T + 1023) / 1024);
        char nm[96];
        for (int rpb : {64, 16}) {
            const dim3 g((unsigned)(R / rpb), cols);
#define RUN(MODE, NT, BYTES, LABEL)                                                           \
    snprintf(nm, 96, "T=%lld rpb=%d %s %s", (long long)T, rpb, LABEL, NT ? "nt" : "plain");    \
    time(nm, BYTES, [&] { rows<MODE, NT><<<g, 256>>>(t, h, R, T, rpb); });
            RUN(1, true, bt, "t only");
            RUN(2, true, bh, "hit4 only");
            RUN(4, true, bh, "hit16x4rows only");
            RUN(3, true, bt + bh, "t + hit4");
            RUN(5, true, bt + bh, "t + hit16x4rows");
            RUN(3, false, bt + bh, "t + hit4");
            RUN(5, false, bt + bh, "t + hit16x4rows");
        }
        for (int rpb : {8, 32}) {
            snprintf(nm, 96, "T=%lld whole-row blocks rpb=%d t+hit4", (long long)T, rpb);
            time(nm, bt + bh, [&] { rows_whole<3><<<(unsigned)(R / rpb), 256>>>(t, h, R, T, rpb); });
            snprintf(nm, 96, "T=%lld whole-row blocks rpb=%d t only", (long long)T, rpb);
            time(nm, bt, [&] { rows_whole<1><<<(unsigned)(R / rpb), 256>>>(t, h, R, T, rpb); });
        }
    }
    return 0;
}
