// literal_lab.hip -- where do the ~9.9 us of the configs[1] launch (256 rays x 10 000 triangles) go?
// Variants of mt_dense_aligned_kernel's structure timed inside a HIP graph of 200 launches.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I differt_amd/csrc -I include \
//        -o scratch/literal_lab scratch/literal_lab.hip
#include "../differt_amd/csrc/ray_ops.hip"
#include "../differt_amd/csrc/core.hip"

#include <algorithm>
#include <cstdio>
#include <vector>
using namespace drt;

__global__ __launch_bounds__(256) void empty_kernel(const float *ro, float *t_out, int n) {
    if (n < 0) t_out[threadIdx.x] = ro[0];
}

// MODE bit0: no arithmetic (t = tri/ray sums); bit1: no t store; bit2: no hit path (LDS + flush);
// bit3: triangles synthesised in registers (no loads)
template <int MODE>
__global__ __launch_bounds__(kDenseThreads) __attribute__((amdgpu_waves_per_eu(7, 7)))
void lab_kernel(const float *__restrict__ ro, const float *__restrict__ rd, int64_t R,
                const float *__restrict__ tv, int64_t T, float eps, float *__restrict__ t_out,
                uint8_t *__restrict__ hit_out, int rays_per_block) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_h[2][kDenseGroup][kDenseThreads];
    const uint32_t col0 = blockIdx.y * (uint32_t)kDenseCols;
    const uint32_t j0 = col0 + threadIdx.x * 4u;
    const bool active = j0 < T;
    const int64_t r0 = (int64_t)blockIdx.x * rays_per_block;
    const int n = (int)((r0 + rays_per_block < R) ? rays_per_block : R - r0);
    const uint32_t W = (uint32_t)((T - col0 < kDenseCols) ? T - col0 : kDenseCols);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    TriE tri[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t j = (j0 + q < T) ? j0 + q : T - 1;
        if (MODE & 8) {
            const float f = (float)j;
            tri[q] = make_tri(V3{f, 1, 2}, V3{f + 1, 2, 3}, V3{f, 4, 1});
        } else {
            tri[q] = load_tri(tv + 9 * j);
        }
        asm volatile("" : "+v"(tri[q].e1.x), "+v"(tri[q].e1.y), "+v"(tri[q].e1.z),
                          "+v"(tri[q].e2.x), "+v"(tri[q].e2.y), "+v"(tri[q].e2.z));
    }
    const float *po = ro + 3 * r0, *pd = rd + 3 * r0;
    char *trow = reinterpret_cast<char *>(t_out + r0 * T);
    const uint32_t toff = j0 * 4u;
    V3 o = ld3(po), d = ld3(pd);
    const int ngroups = (n + kDenseGroup - 1) / kDenseGroup;
    for (int g = 0; g < ngroups; ++g) {
        const int buf = g & 1;
        const int cnt = (n - g * kDenseGroup < kDenseGroup) ? n - g * kDenseGroup : kDenseGroup;
        for (int s = 0; s < cnt; ++s) {
            const int more = (g * kDenseGroup + s + 1 < n) ? 3 : 0;
            po += more;
            pd += more;
            const V3 on = ld3(po), dn = ld3(pd);
            float t[4];
            uint32_t hh;
            if (MODE & 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) t[q] = tri[q].v0.x + o.x + d.y + tri[q].e1.y + tri[q].e2.z;
                hh = __float_as_uint(t[0]) & 0x01010101u;
            } else {
                moller_trumbore_x4(o, d, tri, eps, t, hh);
            }
            if (!(MODE & 2)) {
                if (active) store_nt_b128(trow, toff, f32x4{t[0], t[1], t[2], t[3]});
            } else if (t[0] == 1.2345f && t[1] == 2.5f) {
                store_nt_b128(trow, toff, f32x4{t[0], t[1], t[2], t[3]});
            }
            if (!(MODE & 4)) lds_h[buf][s][threadIdx.x] = hh;
            else if (hh == 0x12345678u) hit_out[j0] = 1;
            trow += T * 4;
            o = on;
            d = dn;
        }
        if (!(MODE & 4)) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kDenseGroup / 4; ++k) {
                const int s = wave + 4 * k;
                if (s < cnt) {
                    const int64_t A = (r0 + (int64_t)g * kDenseGroup + s) * T + col0;
                    const uint32_t head = (128u - ((uint32_t)A & 127u)) & 127u;
                    const uint32_t hd = head < W ? head : W;
                    const uint32_t body = (W - hd) & ~127u;
                    const uint32_t off = line_first_offset((uint32_t)lane * 16u, hd, body);
                    if (off < W) {
                        const u32x4 v = *reinterpret_cast<const u32x4 *>(
                            reinterpret_cast<const char *>(&lds_h[buf][s][0]) + off);
                        store_nt_b128(reinterpret_cast<char *>(hit_out) + A, off, v);
                    }
                }
            }
        }
    }
}

int main(int argc, char **argv) {
    const int64_t R = argc > 1 ? atoll(argv[1]) : 256, T = 10000;
    std::vector<float> ho(R * 3), hd(R * 3), htv(T * 9);
    uint32_t st = 12345;
    auto rnd = [&] { st = st * 1664525u + 1013904223u; return (float)(st >> 8) / 16777216.0f * 100.f - 50.f; };
    for (auto &x : ho) x = rnd();
    for (auto &x : hd) x = rnd();
    for (int64_t j = 0; j < T; ++j) {
        float c[3] = {rnd(), rnd(), rnd()};
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) htv[j * 9 + v * 3 + k] = c[k] + (v ? rnd() * 0.04f : 0.f);
    }
    float *ro, *rd, *tv, *t;
    uint8_t *h;
    hipMalloc(&ro, R * 12); hipMalloc(&rd, R * 12); hipMalloc(&tv, T * 36); hipMalloc(&t, R * T * 4); hipMalloc(&h, R * T);
    hipMemcpy(ro, ho.data(), R * 12, hipMemcpyHostToDevice);
    hipMemcpy(rd, hd.data(), R * 12, hipMemcpyHostToDevice);
    hipMemcpy(tv, htv.data(), T * 36, hipMemcpyHostToDevice);
    hipStream_t s;
    hipStreamCreate(&s);
    const float eps = 1.1920929e-6f;
    auto time = [&](const char *name, auto fn) {
        for (int i = 0; i < 20; ++i) fn();
        hipStreamSynchronize(s);
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 200; ++i) fn();
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        std::vector<float> v;
        for (int it = 0; it < 7; ++it) {
            hipEventRecord(e0, s);
            hipGraphLaunch(ge, s);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            v.push_back(ms * 1e3f / 200);
        }
        std::sort(v.begin(), v.end());
        printf("%-52s min %.2f med %.2f us/launch\n", name, v[0], v[3]);
        fflush(stdout);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    };
    {
        char nm[128];
        snprintf(nm, 128, "product R=%lld blocks=%s stage=%s", (long long)R, getenv("DRT_DENSE_BLOCKS") ? getenv("DRT_DENSE_BLOCKS") : "dflt",
                 getenv("DRT_DENSE_STAGE") ? getenv("DRT_DENSE_STAGE") : "dflt");
        time(nm, [&] { drt_ray_intersect_triangle_dense(ro, rd, R, tv, T, eps, t, h, s); });
        if (argc > 1) return 0;
    }
    for (int rpb : {1, 2, 4, 8}) {
        dim3 grid((unsigned)((R + rpb - 1) / rpb), 10);
        char nm[96];
        snprintf(nm, 96, "empty kernel, grid %ux10", grid.x);
        time(nm, [&] { hipLaunchKernelGGL(empty_kernel, grid, dim3(256), 0, s, ro, t, 0); });
#define RUN(MODE, LABEL)                                                                   \
    snprintf(nm, 96, "rpb=%d %s", rpb, LABEL);                                             \
    time(nm, [&] { hipLaunchKernelGGL((lab_kernel<MODE>), grid, dim3(256), 0, s, ro, rd, R, tv, T, eps, t, h, rpb); });
        RUN(0, "full");
        RUN(1, "no arithmetic");
        RUN(2, "no t store");
        RUN(4, "no hit path");
        RUN(6, "no stores at all");
        RUN(7, "no arithmetic, no stores (loads only)");
        RUN(8, "no triangle loads");
        RUN(15, "nothing (ray loads + loop only)");
    }
    return 0;
}
