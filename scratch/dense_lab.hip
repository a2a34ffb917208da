// dense_lab.hip -- A/B harness for variants of the dense Moller-Trumbore kernel (row a1) on the
// bench shape (65 536 rays x 10 000 triangles).  Every variant is checked bit for bit (t bit
// patterns and hit bytes) against the shipped formulation (`K_base`, a copy of round 1's
// mt_dense_kernel<4,true>) and timed with HIP events (median of 7 after 2 warm-ups).
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize \
//              -I differt_amd/csrc -o scratch/dense_lab scratch/dense_lab.hip
// run  : scratch/dense_lab [variant-substring]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "geom.hpp"

#pragma clang fp contract(off)

using namespace drt;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

// ------------------------------------------------------------------------------------------
// K_base: round-1 kernel
// ------------------------------------------------------------------------------------------
template <bool STORE_T, bool STORE_H>
__global__ __launch_bounds__(256) void K_base(const float *__restrict__ ro, const float *__restrict__ rd,
                                              int64_t R, const float *__restrict__ tv, int64_t T, float eps,
                                              float *__restrict__ t_out, uint8_t *__restrict__ hit_out,
                                              int rays_per_block) {
    const int64_t j0 = ((int64_t)blockIdx.y * 256 + threadIdx.x) * 4;
    if (j0 >= T) return;
    const int64_t r0 = (int64_t)blockIdx.x * rays_per_block;
    const int64_t r1 = (r0 + rays_per_block < R) ? r0 + rays_per_block : R;
    TriE tri[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t j = (j0 + q < T) ? j0 + q : T - 1;
        tri[q] = load_tri(tv + 9 * j);
        asm volatile("" : "+v"(tri[q].e1.x), "+v"(tri[q].e1.y), "+v"(tri[q].e1.z), "+v"(tri[q].e2.x),
                     "+v"(tri[q].e2.y), "+v"(tri[q].e2.z));
    }
    float sink = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
        const V3 o = ld3(ro + 3 * r);
        const V3 d = ld3(rd + 3 * r);
        float t[4];
        bool h[4];
        moller_trumbore_n<4>(o, d, tri, eps, t, h);
        const int64_t base = r * T + j0;
        f32x4 tt = {t[0], t[1], t[2], t[3]};
        const uint32_t hh = (uint32_t)h[0] | ((uint32_t)h[1] << 8) | ((uint32_t)h[2] << 16) | ((uint32_t)h[3] << 24);
        if (STORE_T) __builtin_nontemporal_store(tt, reinterpret_cast<f32x4 *>(t_out + base));
        else sink += t[0] + t[1] + t[2] + t[3];
        if (STORE_H) __builtin_nontemporal_store(hh, reinterpret_cast<uint32_t *>(hit_out + base));
        else sink += (float)hh;
    }
    if (!(STORE_T && STORE_H) && sink == 12345.678f) t_out[j0] = sink;
}

// ------------------------------------------------------------------------------------------
// New arithmetic: 4 tests of one ray with a cheaper fast path.
//   * fast path taken when every |a| of the wave is inside [2^-126, 2^126] (no zero, no denormal,
//     no huge, no nan): range check = max3/min3 over the lane's 4 determinants + 2 compares;
//     no `a == 0` handling (cannot occur), reciprocal = v_rcp + 1 Newton step (exhaustively verified).
//   * `u <= 1` is implied by `v >= 0 && u+v <= 1` (rounding is monotone: u <= rn(u+v)); `u >= 0 &&
//     v >= 0` is `min(u,v) >= 0` (a NaN in u or v makes u+v NaN, which fails `u+v <= 1`).
//   * anything else: the generic phased routine (identical results by construction).
// hit bytes are packed with SDWA byte-select v_cndmask (4 instead of 6 instructions).
// ------------------------------------------------------------------------------------------
// the production formulation (geom.hpp); vzero / vone are leftovers of the SDWA packing experiment
// (v_cndmask_b32_sdwa measured 8x the issue cost of a plain VALU instruction, scratch/valu_mix.hip)
__device__ __forceinline__ void mt4_fast(V3 o, V3 d, const TriE (&tr)[4], float eps, float (&t_out)[4],
                                         uint32_t &hits, uint32_t, uint32_t) {
    moller_trumbore_x4(o, d, tr, eps, t_out, hits);
}

// stores with a wave-uniform 64-bit base (SGPR pair) + 32-bit lane offset: no per-row VALU address math
template <bool NT>
__device__ __forceinline__ void store_b128(char *base, uint32_t off, f32x4 v) {
    if (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt" : : "v"(off), "v"(v), "s"(base) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(off), "v"(v), "s"(base) : "memory");
}
template <bool NT>
__device__ __forceinline__ void store_b128u(char *base, uint32_t off, u32x4 v) {
    if (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt" : : "v"(off), "v"(v), "s"(base) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(off), "v"(v), "s"(base) : "memory");
}
template <bool NT>
__device__ __forceinline__ void store_b32(char *base, uint32_t off, uint32_t v) {
    if (NT) asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(off), "v"(v), "s"(base) : "memory");
    else asm volatile("global_store_dword %0, %1, %2" : : "v"(off), "v"(v), "s"(base) : "memory");
}

// K_v1: new arithmetic, 32-bit lane offsets on wave-uniform row pointers (saddr stores), int loop
template <bool STORE_T, bool STORE_H, bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void K_v1(const float *__restrict__ ro, const float *__restrict__ rd,
                                            int64_t R, const float *__restrict__ tv, int64_t T, float eps,
                                            float *__restrict__ t_out, uint8_t *__restrict__ hit_out,
                                            int rays_per_block) {
    const uint32_t j0 = (blockIdx.y * 256u + threadIdx.x) * 4u;
    if (j0 >= T) return;
    const int64_t r0 = (int64_t)blockIdx.x * rays_per_block;
    const int n = (int)((r0 + rays_per_block < R) ? rays_per_block : R - r0);
    TriE tri[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t j = (j0 + q < T) ? j0 + q : T - 1;
        tri[q] = load_tri(tv + 9 * j);
        asm volatile("" : "+v"(tri[q].e1.x), "+v"(tri[q].e1.y), "+v"(tri[q].e1.z), "+v"(tri[q].e2.x),
                     "+v"(tri[q].e2.y), "+v"(tri[q].e2.z));
    }
    uint32_t vzero, vone;  // constants pinned in VGPRs for the SDWA byte selects
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 1" : "=v"(vzero), "=v"(vone));
    const float *po = ro + 3 * r0, *pd = rd + 3 * r0;
    char *trow = reinterpret_cast<char *>(t_out + r0 * T);
    char *hrow = reinterpret_cast<char *>(hit_out + r0 * T);
    const uint32_t toff = j0 * 4u, hoff = j0;
    float sink = 0.f;
    for (int i = 0; i < n; ++i) {
        const V3 o = ld3(po);
        const V3 d = ld3(pd);
        po += 3;
        pd += 3;
        float t[4];
        uint32_t hh;
        mt4_fast(o, d, tri, eps, t, hh, vzero, vone);
        f32x4 tt = {t[0], t[1], t[2], t[3]};
        if (STORE_T) store_b128<NT>(trow, toff, tt);
        else sink += t[0] + t[1] + t[2] + t[3];
        if (STORE_H) store_b32<NT>(hrow, hoff, hh);
        else sink += (float)hh;
        trow += T * 4;
        hrow += T;
    }
    if (!(STORE_T && STORE_H) && sink == 12345.678f) t_out[j0] = sink;
}

// K_v2: like K_v1 but hit bytes of 4 consecutive rays are transposed inside lane quads (DPP) so that
// one 16-B store per lane covers (ray r+k, 16 triangles): a quarter of the hit store instructions.
template <bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void K_v2(const float *__restrict__ ro, const float *__restrict__ rd,
                                            int64_t R, const float *__restrict__ tv, int64_t T, float eps,
                                            float *__restrict__ t_out, uint8_t *__restrict__ hit_out,
                                            int rays_per_block) {
    // requires T % 16 == 0 and rays_per_block % 4 == 0 and R % 4 == 0 (lab shape)
    const uint32_t j0 = (blockIdx.y * 256u + threadIdx.x) * 4u;
    if (j0 >= T) return;
    const int64_t r0 = (int64_t)blockIdx.x * rays_per_block;
    const int n = (int)((r0 + rays_per_block < R) ? rays_per_block : R - r0);
    TriE tri[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t j = (j0 + q < T) ? j0 + q : T - 1;
        tri[q] = load_tri(tv + 9 * j);
        asm volatile("" : "+v"(tri[q].e1.x), "+v"(tri[q].e1.y), "+v"(tri[q].e1.z), "+v"(tri[q].e2.x),
                     "+v"(tri[q].e2.y), "+v"(tri[q].e2.z));
    }
    uint32_t vzero, vone;  // constants pinned in VGPRs for the SDWA byte selects
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 1" : "=v"(vzero), "=v"(vone));
    const float *po = ro + 3 * r0, *pd = rd + 3 * r0;
    char *trow = reinterpret_cast<char *>(t_out + r0 * T);
    char *hrow = reinterpret_cast<char *>(hit_out + r0 * T);
    const uint32_t toff = j0 * 4u;
    const uint32_t k = threadIdx.x & 3u;
    // lane 4q+k stores row (i+k), bytes of triangles [16q', 16q'+16) where 16q' = j0 - 4k
    const uint32_t hoff = (j0 - 4u * k) + k * (uint32_t)T;
    for (int i = 0; i < n; i += 4) {
        uint32_t hh[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const V3 o = ld3(po);
            const V3 d = ld3(pd);
            po += 3;
            pd += 3;
            float t[4];
            mt4_fast(o, d, tri, eps, t, hh[s], vzero, vone);
            f32x4 tt = {t[0], t[1], t[2], t[3]};
            store_b128<NT>(trow, toff, tt);
            trow += T * 4;
        }
        // 4x4 transpose inside every lane quad (two butterfly stages of DPP exchanges):
        // afterwards hh[c] of lane 4q+k holds the packed hits of ray i+k for triangles 16q+4c..+3
        {
            const bool odd = (k & 1u) != 0u, hi = (k & 2u) != 0u;
#pragma unroll
            for (int a = 0; a < 4; a += 2) {  // exchange across lane bit 0: pairs (hh[a], hh[a+1])
                const uint32_t send = odd ? hh[a] : hh[a + 1];
                const uint32_t recv = __builtin_amdgcn_mov_dpp(send, 0xb1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
                hh[a] = odd ? recv : hh[a];
                hh[a + 1] = odd ? hh[a + 1] : recv;
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {  // exchange across lane bit 1: pairs (hh[a], hh[a+2])
                const uint32_t send = hi ? hh[a] : hh[a + 2];
                const uint32_t recv = __builtin_amdgcn_mov_dpp(send, 0x4e, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
                hh[a] = hi ? recv : hh[a];
                hh[a + 2] = hi ? hh[a + 2] : recv;
            }
        }
        const u32x4 out = {hh[0], hh[1], hh[2], hh[3]};
        store_b128u<NT>(hrow, hoff, out);
        hrow += T * 4;
    }
}

// K_v3: deferred stores -- results of ray i are stored in the middle of ray i+1's arithmetic (the
// compiler is free to move them; an asm barrier pins the order)
template <bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) void K_v3(const float *__restrict__ ro, const float *__restrict__ rd,
                                            int64_t R, const float *__restrict__ tv, int64_t T, float eps,
                                            float *__restrict__ t_out, uint8_t *__restrict__ hit_out,
                                            int rays_per_block) {
    const uint32_t j0 = (blockIdx.y * 256u + threadIdx.x) * 4u;
    if (j0 >= T) return;
    const int64_t r0 = (int64_t)blockIdx.x * rays_per_block;
    const int n = (int)((r0 + rays_per_block < R) ? rays_per_block : R - r0);
    TriE tri[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t j = (j0 + q < T) ? j0 + q : T - 1;
        tri[q] = load_tri(tv + 9 * j);
        asm volatile("" : "+v"(tri[q].e1.x), "+v"(tri[q].e1.y), "+v"(tri[q].e1.z), "+v"(tri[q].e2.x),
                     "+v"(tri[q].e2.y), "+v"(tri[q].e2.z));
    }
    uint32_t vzero, vone;  // constants pinned in VGPRs for the SDWA byte selects
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 1" : "=v"(vzero), "=v"(vone));
    const float *po = ro + 3 * r0, *pd = rd + 3 * r0;
    char *trow = reinterpret_cast<char *>(t_out + r0 * T);
    char *hrow = reinterpret_cast<char *>(hit_out + r0 * T);
    const uint32_t toff = j0 * 4u, hoff = j0;
    // two rays per trip: the second ray's tests are independent of the first one's stores
    int i = 0;
    for (; i + 2 <= n; i += 2) {
        const V3 oa = ld3(po), da = ld3(pd), ob = ld3(po + 3), db = ld3(pd + 3);
        po += 6;
        pd += 6;
        float ta[4], tb[4];
        uint32_t ha, hb;
        mt4_fast(oa, da, tri, eps, ta, ha, vzero, vone);
        mt4_fast(ob, db, tri, eps, tb, hb, vzero, vone);
        f32x4 va = {ta[0], ta[1], ta[2], ta[3]}, vb = {tb[0], tb[1], tb[2], tb[3]};
        store_b128<NT>(trow, toff, va);
        store_b128<NT>(trow + T * 4, toff, vb);
        store_b32<NT>(hrow, hoff, ha);
        store_b32<NT>(hrow + T, hoff, hb);
        trow += T * 8;
        hrow += T * 2;
    }
    for (; i < n; ++i) {
        const V3 o = ld3(po), d = ld3(pd);
        po += 3;
        pd += 3;
        float t[4];
        uint32_t hh;
        mt4_fast(o, d, tri, eps, t, hh, vzero, vone);
        f32x4 tt = {t[0], t[1], t[2], t[3]};
        *reinterpret_cast<f32x4 *>(trow + toff) = tt;
        *reinterpret_cast<uint32_t *>(hrow + hoff) = hh;
        trow += T * 4;
        hrow += T;
    }
}

// ------------------------------------------------------------------------------------------
// K_v4: K_v1 + line-aligned hit stores.  Measured (store_lab): a store instruction that covers whole
// 128-B lines costs ~0.64x of one that straddles them, and with T = 10 000 every hit row starts at a
// different 16-B phase, so each 256-B wave segment straddles.  Here the block stages the packed hit
// bytes of GROUP = 8 consecutive rays in LDS (8 x 1 KiB, double buffered, one barrier per group) and
// every wave then flushes two rows with ONE dwordx4 store each: lanes 0.. cover the whole 128-B lines
// of the row segment, the last lanes the partial head and tail.  t still goes out directly.
// PROBE: 0 = real kernel; 1 = no stores at all; 2 = rcp replaced by a move (timing probe, wrong
// results); 3 = compares removed (timing probe)
// ------------------------------------------------------------------------------------------
template <bool NT, int WAVES, int PROBE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void K_v4(
    const float *__restrict__ ro, const float *__restrict__ rd, int64_t R, const float *__restrict__ tv, int64_t T,
    float eps, float *__restrict__ t_out, uint8_t *__restrict__ hit_out, int rays_per_block) {
    constexpr int GROUP = 8;
    __shared__ __attribute__((aligned(16))) uint32_t lds_h[2][GROUP][256];
    const uint32_t col0 = blockIdx.y * 1024u;
    const uint32_t j0 = col0 + threadIdx.x * 4u;
    const bool active = j0 < T;  // T % 4 == 0: a lane is either fully inside or fully outside
    const int64_t r0 = (int64_t)blockIdx.x * rays_per_block;
    const int n = (int)((r0 + rays_per_block < R) ? rays_per_block : R - r0);
    const uint32_t W = (uint32_t)((T - col0 < 1024) ? T - col0 : 1024);  // hit bytes per row of this block
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    TriE tri[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t j = (j0 + q < T) ? j0 + q : T - 1;
        tri[q] = load_tri(tv + 9 * j);
        asm volatile("" : "+v"(tri[q].e1.x), "+v"(tri[q].e1.y), "+v"(tri[q].e1.z), "+v"(tri[q].e2.x),
                     "+v"(tri[q].e2.y), "+v"(tri[q].e2.z));
    }
    uint32_t vzero, vone;
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 1" : "=v"(vzero), "=v"(vone));
    const float *po = ro + 3 * r0, *pd = rd + 3 * r0;
    char *trow = reinterpret_cast<char *>(t_out + r0 * T);
    const uint32_t toff = j0 * 4u;
    float sink = 0.f;
    V3 o = ld3(po), d = ld3(pd);
    const int ngroups = (n + GROUP - 1) / GROUP;
    for (int g = 0; g < ngroups; ++g) {
        const int buf = g & 1;
        const int cnt = (n - g * GROUP < GROUP) ? n - g * GROUP : GROUP;
        for (int s = 0; s < cnt; ++s) {
            // prefetch the next ray (clamped: the last ray is re-read) so that the scalar loads of ray
            // i+1 are in flight during the arithmetic of ray i
            const int more = (g * GROUP + s + 1 < n) ? 3 : 0;
            po += more;
            pd += more;
            const V3 on = ld3(po), dn = ld3(pd);
            float t[4];
            uint32_t hh;
            mt4_fast(o, d, tri, eps, t, hh, vzero, vone);
            if (PROBE == 1) {
                sink += t[0] + t[1] + t[2] + t[3] + (float)hh;
            } else {
                if (active) store_b128<NT>(trow, toff, f32x4{t[0], t[1], t[2], t[3]});
                lds_h[buf][s][threadIdx.x] = hh;
            }
            trow += T * 4;
            o = on;
            d = dn;
        }
        if (PROBE == 1) continue;
        __syncthreads();
        // flush: wave w owns rows w and w + 4 of the group
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int s = wave + 4 * k;
            if (s < cnt) {
                const int64_t A = (r0 + (int64_t)g * GROUP + s) * T + col0;  // first hit byte of the row segment
                const uint32_t phi = (uint32_t)A & 127u;
                const uint32_t head = (128u - phi) & 127u;
                const uint32_t hd = head < W ? head : W;
                const uint32_t body = (W - hd) & ~127u;
                const uint32_t p = (uint32_t)lane * 16u;
                // lanes below body/16: whole lines; then the head pieces, then the tail pieces
                uint32_t off;
                if (p < body) off = hd + p;
                else {
                    const uint32_t e = p - body;
                    off = (e < hd) ? e : body + e;
                }
                if (off < W) {
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(&lds_h[buf][s][0]) + off);
                    store_b128u<NT>(reinterpret_cast<char *>(hit_out) + A, off, v);
                }
            }
        }
    }
    if (PROBE == 1 && sink == 12345.678f) t_out[j0] = sink;
}

// ------------------------------------------------------------------------------------------
// K_v5: both outputs staged in LDS per GROUP rays and flushed as whole 128-B lines (t rows of
// T = 10 000 floats start at a 64-B phase on every other row).  Piece order per row segment: the
// whole lines first, then the partial head, then the partial tail; 64 pieces of 16 B per store.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t aligned_piece(uint32_t p, uint32_t hd, uint32_t body) {
    if (p < body) return hd + p;
    const uint32_t e = p - body;
    return (e < hd) ? e : body + e;
}

template <int GROUP, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void K_v5(
    const float *__restrict__ ro, const float *__restrict__ rd, int64_t R, const float *__restrict__ tv, int64_t T,
    float eps, float *__restrict__ t_out, uint8_t *__restrict__ hit_out, int rays_per_block) {
    __shared__ __attribute__((aligned(16))) float lds_t[2][GROUP][1024];
    __shared__ __attribute__((aligned(16))) uint32_t lds_h[2][GROUP][256];
    const uint32_t col0 = blockIdx.y * 1024u;
    const uint32_t j0 = col0 + threadIdx.x * 4u;
    const int64_t r0 = (int64_t)blockIdx.x * rays_per_block;
    const int n = (int)((r0 + rays_per_block < R) ? rays_per_block : R - r0);
    const uint32_t W = (uint32_t)((T - col0 < 1024) ? T - col0 : 1024);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    TriE tri[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t j = (j0 + q < T) ? j0 + q : T - 1;
        tri[q] = load_tri(tv + 9 * j);
        asm volatile("" : "+v"(tri[q].e1.x), "+v"(tri[q].e1.y), "+v"(tri[q].e1.z), "+v"(tri[q].e2.x),
                     "+v"(tri[q].e2.y), "+v"(tri[q].e2.z));
    }
    uint32_t vzero, vone;
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 1" : "=v"(vzero), "=v"(vone));
    const float *po = ro + 3 * r0, *pd = rd + 3 * r0;
    // t flush: wave w owns pieces [64 w, 64 w + 64) of every row of the group; the byte phase of a row
    // segment is (row * 4T + 4 col0) mod 128: only two values when T % 16 == 0 -> both precomputed
    const uint32_t W4 = W * 4u;
    uint32_t toffs[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const uint32_t phi = (uint32_t)((((r0 + par) * T + col0) * 4) & 127);
        const uint32_t head = (128u - phi) & 127u;
        const uint32_t hd = head < W4 ? head : W4;
        const uint32_t body = (W4 - hd) & ~127u;
        toffs[par] = aligned_piece((uint32_t)(wave * 64 + lane) * 16u, hd, body);
    }
    V3 o = ld3(po), d = ld3(pd);
    const int ngroups = (n + GROUP - 1) / GROUP;
    for (int g = 0; g < ngroups; ++g) {
        const int buf = g & 1;
        const int cnt = (n - g * GROUP < GROUP) ? n - g * GROUP : GROUP;
        for (int s = 0; s < cnt; ++s) {
            const int more = (g * GROUP + s + 1 < n) ? 3 : 0;
            po += more;
            pd += more;
            const V3 on = ld3(po), dn = ld3(pd);
            float t[4];
            uint32_t hh;
            mt4_fast(o, d, tri, eps, t, hh, vzero, vone);
            *reinterpret_cast<f32x4 *>(&lds_t[buf][s][threadIdx.x * 4]) = f32x4{t[0], t[1], t[2], t[3]};
            lds_h[buf][s][threadIdx.x] = hh;
            o = on;
            d = dn;
        }
        __syncthreads();
        const int64_t rg = r0 + (int64_t)g * GROUP;
        for (int s = 0; s < cnt; ++s) {
            const uint32_t off = toffs[(g * GROUP + s) & 1];
            if (off < W4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(&lds_t[buf][s][0]) + off);
                store_b128<true>(reinterpret_cast<char *>(t_out + (rg + s) * T + col0), off, v);
            }
        }
        for (int s = wave; s < cnt; s += 4) {
            const int64_t A = (rg + s) * T + col0;
            const uint32_t phi = (uint32_t)A & 127u;
            const uint32_t head = (128u - phi) & 127u;
            const uint32_t hd = head < W ? head : W;
            const uint32_t body = (W - hd) & ~127u;
            const uint32_t off = aligned_piece((uint32_t)lane * 16u, hd, body);
            if (off < W) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(&lds_h[buf][s][0]) + off);
                store_b128u<true>(reinterpret_cast<char *>(hit_out) + A, off, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// K_ws: wave specialisation.  Block = 4 compute waves + 1 store wave.  Compute waves write the
// t / hit values of GROUP consecutive rays x 1024 triangles into an LDS buffer (double buffered);
// the store wave drains the other buffer with 1-KiB dwordx4 stores (hit bytes as 16 B per lane:
// 4 row segments of 256 B per instruction).  One block barrier per GROUP rays.
// ------------------------------------------------------------------------------------------
template <int GROUP, bool NT>
__global__ __launch_bounds__(320) void K_ws(const float *__restrict__ ro, const float *__restrict__ rd,
                                            int64_t R, const float *__restrict__ tv, int64_t T, float eps,
                                            float *__restrict__ t_out, uint8_t *__restrict__ hit_out,
                                            int rays_per_block) {
    static_assert(GROUP % 4 == 0, "");
    __shared__ __attribute__((aligned(16))) float lds_t[2][GROUP][1024];
    __shared__ __attribute__((aligned(16))) uint32_t lds_h[2][GROUP][256];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const bool is_store = (wave == 4);
    const uint32_t col0 = blockIdx.y * 1024u;
    const int64_t r0 = (int64_t)blockIdx.x * rays_per_block;
    const int n = (int)((r0 + rays_per_block < R) ? rays_per_block : R - r0);
    const int ngroups = (n + GROUP - 1) / GROUP;
    const uint32_t tcols = (uint32_t)((T - col0 < 1024) ? T - col0 : 1024);  // valid columns of this block

    TriE tri[4];
    const uint32_t ct = threadIdx.x & 255u;  // compute thread id
    const uint32_t j0 = col0 + ct * 4u;
    if (!is_store) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t j = (j0 + q < T) ? j0 + q : T - 1;
            tri[q] = load_tri(tv + 9 * j);
            asm volatile("" : "+v"(tri[q].e1.x), "+v"(tri[q].e1.y), "+v"(tri[q].e1.z), "+v"(tri[q].e2.x),
                         "+v"(tri[q].e2.y), "+v"(tri[q].e2.z));
        }
    }
    uint32_t vzero, vone;  // constants pinned in VGPRs for the SDWA byte selects
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 1" : "=v"(vzero), "=v"(vone));
    const float *po = ro + 3 * r0, *pd = rd + 3 * r0;
    for (int g = 0; g <= ngroups; ++g) {
        const int buf = g & 1;
        if (!is_store) {
            if (g < ngroups) {
                const int cnt = (n - g * GROUP < GROUP) ? n - g * GROUP : GROUP;
                for (int s = 0; s < cnt; ++s) {
                    const V3 o = ld3(po), d = ld3(pd);
                    po += 3;
                    pd += 3;
                    float t[4];
                    uint32_t hh;
                    mt4_fast(o, d, tri, eps, t, hh, vzero, vone);
                    *reinterpret_cast<f32x4 *>(&lds_t[buf][s][ct * 4]) = f32x4{t[0], t[1], t[2], t[3]};
                    lds_h[buf][s][ct] = hh;
                }
            }
        } else if (g > 0) {
            const int pb = buf ^ 1;
            const int gg = g - 1;
            const int cnt = (n - gg * GROUP < GROUP) ? n - gg * GROUP : GROUP;
            const int64_t rbase = r0 + (int64_t)gg * GROUP;
            // t: GROUP rows x 4 KiB, one 1-KiB store per (row, quarter)
            for (int s = 0; s < cnt; ++s) {
                char *trow = reinterpret_cast<char *>(t_out + (rbase + s) * T + col0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t e = (uint32_t)(c * 64 + lane) * 4u;  // first column of this lane
                    if (e < tcols) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(&lds_t[pb][s][e]);
                        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(trow + e * 4u));
                        else *reinterpret_cast<f32x4 *>(trow + e * 4u) = v;
                    }
                }
            }
            // hit: rows of 1 KiB = 64 lanes x 16 B
            for (int s = 0; s < cnt; ++s) {
                char *hrow = reinterpret_cast<char *>(hit_out + (rbase + s) * T + col0);
                const uint32_t e = (uint32_t)lane * 16u;
                if (e < tcols) {
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(&lds_h[pb][s][lane * 4]);
                    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(hrow + e));
                    else *reinterpret_cast<u32x4 *>(hrow + e) = v;
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// compare / harness
// ------------------------------------------------------------------------------------------
__global__ void cmp_kernel(const uint32_t *a, const uint32_t *b, size_t n, unsigned long long *bad) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n; i += (size_t)gridDim.x * 256) c += (a[i] != b[i]);
    if (c) atomicAdd(bad, c);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}
static double nrand() {
    double s = 0;
    for (int i = 0; i < 12; ++i) s += urand();
    return s - 6.0;
}

int main(int argc, char **argv) {
    const char *filter = argc > 1 ? argv[1] : "";
    const int64_t R = 65536, T = 10000;
    const float eps = 10.0f * 1.1920929e-07f;
    std::vector<float> o(3 * R), d(3 * R), tv(9 * T);
    for (int64_t i = 0; i < 3 * R; ++i) {
        o[i] = (float)((urand() * 2 - 1) * 50);
        d[i] = (float)((urand() * 2 - 1) * 50) - o[i];
    }
    for (int64_t j = 0; j < T; ++j) {
        float c[3];
        for (int k = 0; k < 3; ++k) c[k] = (float)((urand() * 2 - 1) * 50);
        for (int k = 0; k < 3; ++k) tv[9 * j + k] = c[k];
        for (int k = 0; k < 3; ++k) tv[9 * j + 3 + k] = c[k] + (float)(nrand() * 2);
        for (int k = 0; k < 3; ++k) tv[9 * j + 6 + k] = c[k] + (float)(nrand() * 2);
    }
    // a few adversarial rays: axis-aligned directions (a == 0 lanes), zero direction, huge / tiny scales
    for (int k = 0; k < 64; ++k) {
        const int64_t r = 1000 + 997 * k;
        float *dd = &d[3 * r];
        switch (k % 6) {
            case 0: dd[0] = 1.f; dd[1] = 0.f; dd[2] = 0.f; break;
            case 1: dd[0] = dd[1] = dd[2] = 0.f; break;
            case 2: dd[0] *= 1e18f; dd[1] *= 1e18f; dd[2] *= 1e18f; break;
            case 3: dd[0] *= 1e-30f; dd[1] *= 1e-30f; dd[2] *= 1e-30f; break;
            case 4: dd[0] = __builtin_nanf(""); break;
            default: dd[0] = __builtin_inff(); break;
        }
    }
    // flat triangles in the z = c plane make a == 0 for rays with d.z == 0
    for (int j = 0; j < 32; ++j) {
        float *t9 = &tv[9 * (123 + 311 * j)];
        t9[5] = t9[2];
        t9[8] = t9[2];
    }
    for (int k = 0; k < 16; ++k) d[3 * (5000 + 13 * k) + 2] = 0.f;

    float *d_o, *d_d, *d_tv, *t_ref, *t_new;
    uint8_t *h_ref, *h_new;
    unsigned long long *d_bad;
    CHECK(hipMalloc(&d_o, o.size() * 4));
    CHECK(hipMalloc(&d_d, d.size() * 4));
    CHECK(hipMalloc(&d_tv, tv.size() * 4));
    CHECK(hipMalloc(&t_ref, (size_t)R * T * 4));
    CHECK(hipMalloc(&t_new, (size_t)R * T * 4));
    CHECK(hipMalloc(&h_ref, (size_t)R * T));
    CHECK(hipMalloc(&h_new, (size_t)R * T));
    CHECK(hipMalloc(&d_bad, 16));
    CHECK(hipMemcpy(d_o, o.data(), o.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_d, d.data(), d.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tv, tv.data(), tv.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    const int rpb = 64;
    const dim3 grid((unsigned)(R / rpb), 10);
    K_base<true, true><<<grid, 256>>>(d_o, d_d, R, d_tv, T, eps, t_ref, h_ref, rpb);
    CHECK(hipDeviceSynchronize());

    auto run = [&](const char *name, bool check, auto launch) {
        if (!strstr(name, filter)) return;
        if (check) {
            CHECK(hipMemset(t_new, 0xcd, (size_t)R * T * 4));
            CHECK(hipMemset(h_new, 0xcd, (size_t)R * T));
        }
        launch();
        launch();
        CHECK(hipDeviceSynchronize());
        std::vector<float> ms;
        for (int it = 0; it < 7; ++it) {
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float m;
            CHECK(hipEventElapsedTime(&m, e0, e1));
            ms.push_back(m);
        }
        std::sort(ms.begin(), ms.end());
        unsigned long long bad[2] = {0, 0};
        if (check) {
            CHECK(hipMemset(d_bad, 0, 16));
            cmp_kernel<<<4096, 256>>>((const uint32_t *)t_ref, (const uint32_t *)t_new, (size_t)R * T, d_bad);
            cmp_kernel<<<4096, 256>>>((const uint32_t *)h_ref, (const uint32_t *)h_new, (size_t)R * T / 4, d_bad + 1);
            CHECK(hipMemcpy(bad, d_bad, 16, hipMemcpyDeviceToHost));
        }
        const double gb = 5.0 * R * T + 24.0 * R + 36.0 * T;
        printf("%-34s min %.3f med %.3f max %.3f ms  %.2f TB/s (med)  frac8 %.3f  %s", name, ms.front(), ms[3],
               ms.back(), gb / ms[3] / 1e9, gb / ms[3] / 1e9 / 8.0,
               check ? (bad[0] || bad[1] ? "MISMATCH" : "bit-exact") : "(no check)");
        if (check && (bad[0] || bad[1])) printf(" t:%llu hit-words:%llu", bad[0], bad[1]);
        printf("\n");
        fflush(stdout);
    };

#define ARGS d_o, d_d, R, d_tv, T, eps, t_new, h_new, rpb
    for (int rep = 0; rep < 2; ++rep) {
        run("base", true, [&] { K_base<true, true><<<grid, 256>>>(ARGS); });
        run("base nostore", false, [&] { K_base<false, false><<<grid, 256>>>(ARGS); });
        run("v1 nt", true, [&] { K_v1<true, true, true><<<grid, 256>>>(ARGS); });
        run("v1 nostore", false, [&] { K_v1<false, false, true><<<grid, 256>>>(ARGS); });
        run("v3 2rays nt", true, [&] { K_v3<true><<<grid, 256>>>(ARGS); });
        run("v4 aligned-hit nt w8", true, [&] { K_v4<true, 8, 0><<<grid, 256>>>(ARGS); });
        run("v4 aligned-hit nt w7", true, [&] { K_v4<true, 7, 0><<<grid, 256>>>(ARGS); });
        run("v4 aligned-hit nt w6", true, [&] { K_v4<true, 6, 0><<<grid, 256>>>(ARGS); });
        run("v4 aligned-hit nt w5", true, [&] { K_v4<true, 5, 0><<<grid, 256>>>(ARGS); });
        run("v5 lds t+hit g2 w8", true, [&] { K_v5<2, 8><<<grid, 256>>>(ARGS); });
        run("v5 lds t+hit g2 w7", true, [&] { K_v5<2, 7><<<grid, 256>>>(ARGS); });
        run("v5 lds t+hit g2 w6", true, [&] { K_v5<2, 6><<<grid, 256>>>(ARGS); });
        run("v5 lds t+hit g4 w4", true, [&] { K_v5<4, 4><<<grid, 256>>>(ARGS); });
        run("v5 lds t+hit g4 w6", true, [&] { K_v5<4, 6><<<grid, 256>>>(ARGS); });
        run("v4 nostore w8", false, [&] { K_v4<true, 8, 1><<<grid, 256>>>(ARGS); });
        run("v4 nostore w7", false, [&] { K_v4<true, 7, 1><<<grid, 256>>>(ARGS); });
        run("v4 nostore w6", false, [&] { K_v4<true, 6, 1><<<grid, 256>>>(ARGS); });
        run("v4 nostore w4", false, [&] { K_v4<true, 4, 1><<<grid, 256>>>(ARGS); });
    }
    // rays-per-block sweep of the best simple variant
    for (int rp : {8, 16, 32}) {
        char nm[64];
        snprintf(nm, 64, "v4 nt w8 rpb=%d", rp);
        const dim3 g4((unsigned)(R / rp), 10);
        run(nm, true, [&] { K_v4<true, 8, 0><<<g4, 256>>>(d_o, d_d, R, d_tv, T, eps, t_new, h_new, rp); });
        snprintf(nm, 64, "v4 nt w7 rpb=%d", rp);
        run(nm, true, [&] { K_v4<true, 7, 0><<<g4, 256>>>(d_o, d_d, R, d_tv, T, eps, t_new, h_new, rp); });
        snprintf(nm, 64, "v5 g2 w7 rpb=%d", rp);
        run(nm, true, [&] { K_v5<2, 7><<<g4, 256>>>(d_o, d_d, R, d_tv, T, eps, t_new, h_new, rp); });
    }
    for (int rp : {16, 32, 128, 256}) {
        char nm[64];
        snprintf(nm, 64, "v1 nt rpb=%d", rp);
        const dim3 g2((unsigned)(R / rp), 10);
        run(nm, true, [&] { K_v1<true, true, true><<<g2, 256>>>(d_o, d_d, R, d_tv, T, eps, t_new, h_new, rp); });
    }
    return 0;
}
