// Lab: the dense tracer's OUTPUT PATTERN without its arithmetic -- is the "packed vs far" placement effect of
// profiles/r04/dense.md a property of the memory system?  Every wave owns 64 consecutive rows and, for each of 64 planes
// (plane stride C rows), writes 3072 B of "vertices", 1024 B of "objects", 512 B of "types" and 64 B of "mask" with
// 16-byte nontemporal stores, exactly the tracer's segments.  Variants: which streams are written, and a DELAY of the
// three small streams by d planes (what a wave could do by holding them in LDS).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/streams_lab scratch/streams_lab.hip && /tmp/streams_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int BYTES>  // one wave stores BYTES contiguous bytes at p (16 B per lane per trip)
__device__ __forceinline__ void wave_store(char *p, int lane, uint32_t tag) {
    const u4 v = {tag, tag + 1, tag + 2, tag + 3};
#pragma unroll
    for (int off = 0; off < BYTES; off += 1024)
        if (off + lane * 16 < BYTES) __builtin_nontemporal_store(v, reinterpret_cast<u4 *>(p + off + lane * 16));
}

__global__ __launch_bounds__(256) void streams_kernel(char *v, char *o, char *t, char *m, int64_t C, int planes, int which,
                                                      int delay, int xcd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nrb = C / 256, per = (nrb + 7) / 8;
    for (int64_t rb = blockIdx.x; rb < 8 * per; rb += gridDim.x) {
        const int64_t rbm = xcd ? (rb % 8) * per + rb / 8 : rb;
        if (rbm >= nrb) continue;
        const int64_t row = rbm * 256 + wave * 64;
        for (int p = 0; p < planes + delay; ++p) {
            if (p < planes && (which & 1)) wave_store<3072>(v + ((int64_t)p * C + row) * 48, lane, (uint32_t)p);
            const int q = p - delay;
            if (q >= 0 && q < planes) {
                const int64_t g = (int64_t)q * C + row;
                if (which & 2) wave_store<1024>(o + g * 16, lane, (uint32_t)q);
                if (which & 4) wave_store<512>(t + g * 8, lane, (uint32_t)q);
                if ((which & 8) && lane < 4) __builtin_nontemporal_store(u4{1, 1, 1, 1}, reinterpret_cast<u4 *>(m + g) + lane);
            }
        }
    }
}

// wave-exit form: one workgroup per (group of `ppb` planes, row block), plane-major launch order -- the chip-wide write
// order is then (almost) the memory order of each array, like a fill
__global__ __launch_bounds__(256) void streams_exit_kernel(char *v, char *o, char *t, char *m, int64_t C, int planes, int which,
                                                           int ppb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nrb = C / 256;
    const int64_t pg = blockIdx.x / nrb, rb = blockIdx.x % nrb;
    const int64_t row = rb * 256 + wave * 64;
    for (int p = (int)pg * ppb; p < (int)(pg + 1) * ppb && p < planes; ++p) {
        const int64_t g = (int64_t)p * C + row;
        if (which & 1) wave_store<3072>(v + g * 48, lane, (uint32_t)p);
        if (which & 2) wave_store<1024>(o + g * 16, lane, (uint32_t)p);
        if (which & 4) wave_store<512>(t + g * 8, lane, (uint32_t)p);
        if ((which & 8) && lane < 4) __builtin_nontemporal_store(u4{1, 1, 1, 1}, reinterpret_cast<u4 *>(m + g) + lane);
    }
}

int main(int argc, char **argv) {
    const int64_t C = argc > 1 ? atoll(argv[1]) : (1 << 20);
    const int planes = 64;
    const int64_t rows = C * planes;
    char *arena = nullptr;
    const size_t G = 1ull << 30;
    CK(hipMalloc(&arena, 48 * G));
    char *base = arena + ((G - (uintptr_t)arena % G) % G);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    auto run = [&](const char *name, size_t gap, int which, int delay, int xcd) {
        auto up = [](size_t x) { return (x + (2u << 20) - 1) / (2u << 20) * (2u << 20); };
        char *v = base, *o = base + up(rows * 48) + gap, *t = o + up(rows * 16), *m = t + up(rows * 8);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(streams_kernel, dim3(2048), dim3(256), 0, 0, v, o, t, m, C, planes, which, delay, xcd);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (rep && ms < best) best = ms;
        }
        double bytes = 0;
        if (which & 1) bytes += rows * 48.0;
        if (which & 2) bytes += rows * 16.0;
        if (which & 4) bytes += rows * 8.0;
        if (which & 8) bytes += rows * 1.0;
        printf("%-44s gap %2zu GiB which %2d delay %d xcd %d : %.3f ms  %.2f TB/s\n", name, gap >> 30, which, delay, xcd, best, bytes / best * 1e-9);
    };
    printf("arena %p base %p (C = %lld)\n", (void *)arena, (void *)base, (long long)C);
    auto run_exit = [&](const char *name, size_t gap, int which, int ppb) {
        auto up = [](size_t x) { return (x + (2u << 20) - 1) / (2u << 20) * (2u << 20); };
        char *v = base, *o = base + up(rows * 48) + gap, *t = o + up(rows * 16), *m = t + up(rows * 8);
        float best = 1e9f;
        const int64_t nblocks = (C / 256) * ((planes + ppb - 1) / ppb);
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(streams_exit_kernel, dim3((unsigned)nblocks), dim3(256), 0, 0, v, o, t, m, C, planes, which, ppb);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (rep && ms < best) best = ms;
        }
        double bytes = 0;
        if (which & 1) bytes += rows * 48.0;
        if (which & 2) bytes += rows * 16.0;
        if (which & 4) bytes += rows * 8.0;
        if (which & 8) bytes += rows * 1.0;
        printf("%-44s gap %2zu GiB which %2d planes/block %2d : %.3f ms  %.2f TB/s\n", name, gap >> 30, which, ppb, best, bytes / best * 1e-9);
    };
    if (argc > 2 && argv[2][0] == 'e') {
        for (int ppb : {1, 2, 4, 8, 16, 64}) {
            run_exit("wave-exit form, all four streams", 0, 15, ppb);
            run_exit("wave-exit form, vertices only", 0, 1, ppb);
        }
        run("persistent, all four, XCD", 0, 15, 0, 1);
        run("persistent, vertices only, XCD", 0, 1, 0, 1);
        return 0;
    }
    if (argc > 2) {  // scan: where does the second region begin?
        for (size_t g : {0, 4, 8, 12, 16, 20, 24, 26, 28, 29, 30, 31, 32, 33, 34, 36, 40})
            run("all four streams, XCD row ranges", g * G, 15, 0, 1);
        return 0;
    }
    for (size_t gap : {(size_t)0, 32 * G}) {
        run("all four streams", gap, 15, 0, 0);
        run("all four streams, XCD row ranges", gap, 15, 0, 1);
        run("vertices only", gap, 1, 0, 0);
        run("objects + types + mask", gap, 14, 0, 0);
        run("vertices + objects", gap, 3, 0, 0);
        run("small streams 1 plane late", gap, 15, 1, 0);
        run("small streams 2 planes late", gap, 15, 2, 0);
        run("small streams 4 planes late", gap, 15, 4, 1);
    }
    return 0;
}
