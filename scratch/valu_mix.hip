// valu_mix.hip -- relative issue cost of the instruction classes of the dense Moller-Trumbore loop on
// gfx950: 32 waves per CU, 8 independent chains per wave, s_memtime-free (HIP events; numbers are
// meaningful RELATIVE to v_mul_f32 measured in the same run, the clock floats with the load).
// build: hipcc --offload-arch=gfx950 -O3 -o scratch/valu_mix scratch/valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2048
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a, float b) {
    float x0 = threadIdx.x + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float sa = __builtin_amdgcn_readfirstlane(a), sb = __builtin_amdgcn_readfirstlane(b);
    unsigned zero = 0, one = 1;
    asm volatile("v_mov_b32 %0, 0\n v_mov_b32 %1, 1" : "=v"(zero), "=v"(one));
    for (int i = 0; i < ITER; ++i) {
#define OPS "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
        if (MODE == 0) {  // v_mul_f32 vgpr,vgpr
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n" : OPS : "v"(a));
        } else if (MODE == 1) {  // v_mul_f32 sgpr,vgpr
            asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                         "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n" : OPS : "s"(sa));
        } else if (MODE == 2) {  // v_cmp_gt_f32 -> sgpr pair (VOP3)
            asm volatile("v_cmp_gt_f32 s[20:21], %0, %8\n v_cmp_gt_f32 s[22:23], %1, %8\n v_cmp_gt_f32 s[24:25], %2, %8\n v_cmp_gt_f32 s[26:27], %3, %8\n"
                         "v_cmp_gt_f32 s[20:21], %4, %8\n v_cmp_gt_f32 s[22:23], %5, %8\n v_cmp_gt_f32 s[24:25], %6, %8\n v_cmp_gt_f32 s[26:27], %7, %8\n"
                         : OPS : "v"(a) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (MODE == 3) {  // v_cmp_gt_f32 -> vcc (VOPC)
            asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n"
                         "v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n" : OPS : "v"(a) : "vcc");
        } else if (MODE == 4) {  // v_rcp_f32
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n" : OPS);
        } else if (MODE == 5) {  // v_min3_f32 with abs modifiers
            asm volatile("v_min3_f32 %0, |%0|, |%1|, %8\n v_min3_f32 %1, |%1|, |%2|, %8\n v_min3_f32 %2, |%2|, |%3|, %8\n v_min3_f32 %3, |%3|, |%4|, %8\n"
                         "v_min3_f32 %4, |%4|, |%5|, %8\n v_min3_f32 %5, |%5|, |%6|, %8\n v_min3_f32 %6, |%6|, |%7|, %8\n v_min3_f32 %7, |%7|, |%0|, %8\n" : OPS : "v"(a));
        } else if (MODE == 6) {  // v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : OPS : "v"(a), "v"(b));
        } else if (MODE == 7) {  // v_cndmask_b32 sdwa byte select
            asm volatile("v_cndmask_b32_sdwa %0, %8, %9, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                         "v_cndmask_b32_sdwa %1, %8, %9, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                         "v_cndmask_b32_sdwa %2, %8, %9, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                         "v_cndmask_b32_sdwa %3, %8, %9, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                         "v_cndmask_b32_sdwa %4, %8, %9, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                         "v_cndmask_b32_sdwa %5, %8, %9, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                         "v_cndmask_b32_sdwa %6, %8, %9, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                         "v_cndmask_b32_sdwa %7, %8, %9, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                         : OPS : "v"(zero), "v"(one) : "vcc");
        } else if (MODE == 8) {  // mul (vgpr same bank pattern): src regs forced equal -> same bank
            asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n"
                         "v_mul_f32 %4, %4, %4\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_mul_f32 %7, %7, %7\n" : OPS);
        } else if (MODE == 9) {  // 8 v_mul + 2 s_and_b64 + 1 s_add (SALU interleaved like the real loop)
            asm volatile("v_mul_f32 %0, %0, %8\n s_and_b64 s[20:21], s[20:21], s[22:23]\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "s_and_b64 s[22:23], s[20:21], s[22:23]\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n s_add_u32 s24, s24, 1\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : OPS : "v"(a) : "s20", "s21", "s22", "s23", "s24", "scc");
        } else if (MODE == 10) {  // dependent chain of 8 on ONE register (latency-bound per wave, 8 waves/SIMD hide it?)
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n"
                         "v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n" : OPS : "v"(a));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + sb;
}
template <int MODE> double run(const char *name, float *out, int blocks_per_cu, double ref) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * blocks_per_cu;
    k<MODE><<<blocks, 256>>>(out, 1.0001f, 0.5f);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MODE><<<blocks, 256>>>(out, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double inst_per_simd = (double)blocks_per_cu * ITER * 8;  // 4 waves per block, 4 SIMDs per CU
    double ns_per_inst = ms * 1e6 / inst_per_simd;
    printf("%-34s waves/SIMD %d  %8.3f ms  %.3f ns per wave-inst per SIMD (= %.2f cyc @2.4 GHz)  x%.2f of v_mul\n", name, blocks_per_cu,
           ms, ns_per_inst, ns_per_inst * 2.4, ref > 0 ? ns_per_inst / ref : 1.0);
    return ns_per_inst;
}
int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {8, 4, 2, 1}) {
        double ref = run<0>("v_mul_f32 v,v", out, w, 0);
        run<1>("v_mul_f32 s,v", out, w, ref);
        run<6>("v_fma_f32", out, w, ref);
        run<2>("v_cmp_gt_f32 -> sgpr pair (e64)", out, w, ref);
        run<3>("v_cmp_gt_f32 -> vcc (e32)", out, w, ref);
        run<4>("v_rcp_f32", out, w, ref);
        run<5>("v_min3_f32 |a|,|b|,c", out, w, ref);
        run<7>("v_cndmask_b32_sdwa", out, w, ref);
        run<8>("v_mul_f32 x,x,x (same reg)", out, w, ref);
        run<9>("8 v_mul + 3 SALU interleaved (per 8)", out, w, ref);
        run<10>("v_mul dependent chain", out, w, ref);
    }
    return 0;
}
