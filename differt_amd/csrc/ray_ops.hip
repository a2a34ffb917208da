// ray_ops.hip -- batched ray/triangle operators for gfx950:
//   (a1) dense / paired Moller-Trumbore          reference geometry/_utils.py:1157-1322
//   (a2) any-hit reduction over a triangle set     reference geometry/_utils.py:1353-1537
//   (a4) first-hit (argmin, min) with tile ties    reference geometry/_utils.py:1775-1960
//   (a5) VJP of the first-hit distance             reference geometry/_mesh.py:226-344
//
// Mapping choices (wave = 64 lanes, 256 CUs):
//   dense MT   : HBM-write bound (5 B out per test).  One lane owns 4 consecutive triangles in
//                registers and walks a chunk of rays whose origin/direction are wave-uniform
//                (scalar loads); every lane stores 16 B of t per ray (each wave writes 1 KiB of an
//                output row); the u8 hit rows are staged in LDS and flushed as whole 128-B lines.
//   any/first  : FP32-VALU bound.  One lane owns one ray; the block stages triangle tiles in LDS
//                as (v0, e1, e2, active) records read back with broadcast ds_read_b128; triangle
//                ranges are split over blockIdx.y so that small ray batches still fill the chip.
//   per-ray triangle sets: one wavefront per ray, lanes stride over the ray's triangles,
//                ballot / shuffle reduction.
#include <cstdlib>

#include "common.hpp"
#include "geom.hpp"
#include "tri_tile.hpp"

#pragma clang fp contract(off)

namespace drt {

// ------------------------------------------------------------------------------------------
// (a1) dense
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kDenseThreads = 256;

__device__ __forceinline__ void atomic_add3f(float *p, V3 v) {
    atomicAdd(p + 0, v.x);
    atomicAdd(p + 1, v.y);
    atomicAdd(p + 2, v.z);
}
constexpr int kDenseCols = kDenseThreads * 4;  // triangles per block
constexpr int kDenseGroup = 8;                 // rays whose hit bytes are staged in LDS together
// batched launches (drt_ray_intersect_triangle_dense_batched): element strides between two problems
struct DenseBatch {
    int64_t ray_stride;  // floats between the ray arrays of consecutive problems (3 R, or 0 = shared rays)
    int64_t tv_stride;   // floats between their triangle sets (9 T, or 0 = shared triangles)
};
constexpr int kDenseStageDwords = 5 * 64 * 4;  // LDS triangle staging per wave: 5 x b128 per lane (32 lanes x 4 triangles x 9 floats = 1152 dwords used)

// Stores with a wave-uniform 64-bit base (SGPR pair) + 32-bit lane offset: the per-row address math
// stays on the scalar unit.  Nontemporal: measured 0.83 ms vs 0.95 ms for plain stores on the bench
// shape (profiles/r02/dense_lab.md).
// gfx9-family hazard: a VMEM store of more than 64 bits followed by a VALU write of its data VGPRs needs
// one wait state, and LLVM's hazard recogniser does not look inside inline asm -- the `s_nop 0` covers it
// whatever the scheduler places after the store (one issue cycle per 1-KiB wave store).
__device__ __forceinline__ void store_nt_b128(char *base, uint32_t off, f32x4 v) {
#if defined(DRT_LAB) && defined(DRT_LAB_AGPR_STORE)
    // experiment: the same store sourced from accumulation registers -- separates "the store's
    // data movement blocks the SIMD's VALU issue" from "VGPR read-port contention"
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0" : : "v"(off), "a"(v), "s"(base) : "memory");
#elif defined(DRT_LAB) && defined(DRT_LAB_SETPRIO)
    asm volatile("s_setprio 3\n\tglobal_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0\n\ts_setprio 0" : : "v"(off), "v"(v), "s"(base) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0" : : "v"(off), "v"(v), "s"(base) : "memory");
#endif
}
__device__ __forceinline__ void store_nt_b128(char *base, uint32_t off, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0" : : "v"(off), "v"(v), "s"(base) : "memory");
}

// Byte offset, inside a row segment of `hd` head bytes + `body` bytes of whole 128-B lines + a
// tail, of the p-th stored byte when the whole lines go first, then the head, then the tail.
__device__ __forceinline__ uint32_t line_first_offset(uint32_t p, uint32_t hd, uint32_t body) {
    if (p < body) return hd + p;
    const uint32_t e = p - body;
    return (e < hd) ? e : body + e;
}

// Main dense kernel (requires T % 16 == 0, t_out / hit_out 16-B aligned).
//   * lane = 4 consecutive triangles kept in VGPRs (v0, e1, e2); block = 1024 triangles x
//     `rays_per_block` rays; ray origin / direction are wave-uniform scalar loads, prefetched one
//     ray ahead; arithmetic = moller_trumbore_x4 (geom.hpp).
//   * `t`: every lane stores 16 B per ray -> each wave writes 1 KiB of an output row.
//   * `hit`: measured on MI355X (scratch/store_lab.hip), a store instruction that covers whole
//     128-B lines costs 0.64x of one that straddles them, and a u8 row of T = 10 000 starts at a
//     different 16-B phase on every ray, so a wave's 256-B segment always straddles (hit stores
//     alone: 0.233 ms straddling vs 0.113 ms aligned per 655 MB).  The block therefore stages the
//     packed hit dwords of kDenseGroup consecutive rays in LDS (8 x 1 KiB, double buffered, ONE
//     barrier per group) and each wave flushes two rows with one dwordx4 store per row: the first
//     lanes cover the whole lines of the 1-KiB row segment, the last lanes its partial head and
//     tail.  A quarter of the hit store instructions, 7 of 9 lines written whole.
__global__ __launch_bounds__(kDenseThreads) __attribute__((amdgpu_waves_per_eu(7, 7)))
void mt_dense_aligned_kernel(const float *__restrict__ ro, const float *__restrict__ rd, int64_t R,
                             const float *__restrict__ tv, int64_t T, float eps,
                             float *__restrict__ t_out, uint8_t *__restrict__ hit_out,
                             int rays_per_block, int stage_tris, DenseBatch nb) {
    // blockIdx.z = problem of a batched launch (R rays x T triangles each; outputs [B,R,T]): wave-uniform
    // pointer offsets on the scalar unit, nothing else changes (nb = {0, 0} for the plain operator)
    ro += (int64_t)blockIdx.z * nb.ray_stride;
    rd += (int64_t)blockIdx.z * nb.ray_stride;
    tv += (int64_t)blockIdx.z * nb.tv_stride;
    t_out += (int64_t)blockIdx.z * R * T;
    hit_out += (int64_t)blockIdx.z * R * T;
    // one LDS allocation, two uses: triangle staging (20 KiB, start of the block) and then the hit rows
    // of kDenseGroup rays x 2 buffers (16 KiB)
    __shared__ __attribute__((aligned(16))) uint32_t lds_raw[4 * kDenseStageDwords];
    static_assert(sizeof(lds_raw) >= 2 * kDenseGroup * kDenseThreads * sizeof(uint32_t), "hit rows fit");
    uint32_t (*lds_h)[kDenseGroup][kDenseThreads] =
        reinterpret_cast<uint32_t (*)[kDenseGroup][kDenseThreads]>(lds_raw);
    const uint32_t col0 = blockIdx.y * (uint32_t)kDenseCols;
    const uint32_t j0 = col0 + threadIdx.x * 4u;
    const bool active = j0 < T;  // T % 4 == 0: a lane is entirely inside or outside the row
    const int64_t r0 = (int64_t)blockIdx.x * rays_per_block;
    const int n = (int)((r0 + rays_per_block < R) ? rays_per_block : R - r0);
    const uint32_t W = (uint32_t)((T - col0 < kDenseCols) ? T - col0 : kDenseCols);  // hit bytes per row
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;

    // Triangles.  A lane's 4 triangles are 144 contiguous bytes, so a direct per-lane load touches 64
    // different 128-B lines per instruction (9 x 64 tag look-ups per wave: measured 10 us of the 14.8-us
    // launch at 1 ray per block, scratch/literal_lab.hip).  Instead the wave reads its 9216-B segment
    // coalesced (16 B per lane, 8 lines per instruction) into LDS, half a wave's worth at a time, and every
    // lane picks its 36 dwords back with 9 conflict-free ds_read_b128 (stride 36 dwords).
    TriE tri[4];
    if (stage_tris) {
        uint32_t *stg = lds_raw + wave * kDenseStageDwords;
        const int64_t jw = (int64_t)col0 + wave * 256;                  // first triangle of the wave
        const int64_t nv = (T - jw < 256) ? ((T - jw > 0) ? T - jw : 0) : 256;  // valid triangles of the wave
        const char *src = reinterpret_cast<const char *>(tv + 9 * jw);
        f32x4 raw[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) raw[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (nv > 0) {  // wave-uniform
            const uint32_t last16 = (uint32_t)nv * 36u - 16u;  // chunks past the row's end re-read its last 16 B
            f32x4 ld[2][5];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < 5; ++k) {  // all ten loads in flight before the first LDS write
                    const uint32_t off = (uint32_t)h * (32u * 144u) + (uint32_t)(k * 64 + lane) * 16u;
                    ld[h][k] = *reinterpret_cast<const f32x4 *>(src + (off < last16 ? off : last16));  // re-read by every row block: cached
                }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int k = 0; k < 5; ++k) *reinterpret_cast<f32x4 *>(stg + 4 * (k * 64 + lane)) = ld[h][k];
                if ((lane >> 5) == h) {
#pragma unroll
                    for (int q = 0; q < 9; ++q)
                        raw[q] = *reinterpret_cast<const f32x4 *>(stg + 36 * (lane & 31) + 4 * q);
                }
            }
        }
        if (!active) {  // lanes past the end of the row: any finite triangle (results never stored)
            const f32x4 *last = reinterpret_cast<const f32x4 *>(tv + 9 * (T - 4));
#pragma unroll
            for (int q = 0; q < 9; ++q) raw[q] = last[q];
        }
        const float *rf = reinterpret_cast<const float *>(raw);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            tri[q] = make_tri(V3{rf[9 * q], rf[9 * q + 1], rf[9 * q + 2]}, V3{rf[9 * q + 3], rf[9 * q + 4], rf[9 * q + 5]},
                              V3{rf[9 * q + 6], rf[9 * q + 7], rf[9 * q + 8]});
        __syncthreads();  // the staging area becomes the hit rows
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // lanes past the end re-read the last triangle; their results are never stored
            const int64_t j = (j0 + q < T) ? j0 + q : T - 1;
            tri[q] = load_tri(tv + 9 * j);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // keep the edges live in VGPRs (otherwise they may be re-derived from the vertices per ray)
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
            // next ray's scalars in flight during this ray's arithmetic (the last ray is re-read)
            const int more = (g * kDenseGroup + s + 1 < n) ? 3 : 0;
            po += more;
            pd += more;
            const V3 on = ld3(po), dn = ld3(pd);
            float t[4];
            uint32_t hh;
            moller_trumbore_x4(o, d, tri, eps, t, hh);
            if (active) store_nt_b128(trow, toff, f32x4{t[0], t[1], t[2], t[3]});
            lds_h[buf][s][threadIdx.x] = hh;
            trow += T * 4;
            o = on;
            d = dn;
        }
        // buffer `buf` is rewritten two groups later, i.e. after the NEXT barrier, which every wave
        // reaches only after its flush below: one barrier per group is enough
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kDenseGroup / 4; ++k) {
            const int s = wave + 4 * k;  // wave w flushes rows w and w + 4 of the group
            if (s < cnt) {
                // first hit byte, relative to the problem's rows; the line phase is that of the ADDRESS
                const int64_t A = (r0 + (int64_t)g * kDenseGroup + s) * T + col0;
                const uint32_t phase = (uint32_t)(reinterpret_cast<uintptr_t>(hit_out) + (uintptr_t)A) & 127u;
                const uint32_t head = (128u - phase) & 127u;
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

// Fallback for rows that are not 16-B tileable (T % 16 != 0 or unaligned outputs): same arithmetic,
// direct stores (VEC: T % 4 == 0 and 16-B / 4-B aligned outputs; otherwise scalar stores).
template <bool VEC>
__global__ __launch_bounds__(kDenseThreads) void mt_dense_kernel(
    const float *__restrict__ ro, const float *__restrict__ rd, int64_t R,
    const float *__restrict__ tv, int64_t T, float eps, float *__restrict__ t_out,
    uint8_t *__restrict__ hit_out, int rays_per_block, DenseBatch nb) {
    ro += (int64_t)blockIdx.z * nb.ray_stride;
    rd += (int64_t)blockIdx.z * nb.ray_stride;
    tv += (int64_t)blockIdx.z * nb.tv_stride;
    t_out += (int64_t)blockIdx.z * R * T;
    hit_out += (int64_t)blockIdx.z * R * T;
    const int64_t j0 = ((int64_t)blockIdx.y * kDenseThreads + threadIdx.x) * 4;
    if (j0 >= T) return;
    const int64_t r0 = (int64_t)blockIdx.x * rays_per_block;
    const int64_t r1 = (r0 + rays_per_block < R) ? r0 + rays_per_block : R;

    TriE tri[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t j = (j0 + q < T) ? j0 + q : T - 1;
        tri[q] = load_tri(tv + 9 * j);
        asm volatile("" : "+v"(tri[q].e1.x), "+v"(tri[q].e1.y), "+v"(tri[q].e1.z),
                          "+v"(tri[q].e2.x), "+v"(tri[q].e2.y), "+v"(tri[q].e2.z));
    }
    for (int64_t r = r0; r < r1; ++r) {
        const V3 o = ld3(ro + 3 * r);  // wave-uniform -> scalar loads
        const V3 d = ld3(rd + 3 * r);
        float t[4];
        uint32_t hh;
        moller_trumbore_x4(o, d, tri, eps, t, hh);
        const int64_t base = r * T + j0;
        if (VEC) {
            f32x4 tt = {t[0], t[1], t[2], t[3]};
            __builtin_nontemporal_store(tt, reinterpret_cast<f32x4 *>(t_out + base));
            __builtin_nontemporal_store(hh, reinterpret_cast<uint32_t *>(hit_out + base));
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (j0 + q < T) {
                    t_out[base + q] = t[q];
                    hit_out[base + q] = (uint8_t)((hh >> (8 * q)) & 1u);
                }
        }
    }
}

// (a1) paired: one lane per element
__global__ __launch_bounds__(256) void mt_paired_kernel(const float *__restrict__ ro,
                                                        const float *__restrict__ rd,
                                                        const float *__restrict__ tv, int64_t n,
                                                        float eps, float *__restrict__ t_out,
                                                        uint8_t *__restrict__ hit_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float t;
    bool h = moller_trumbore(ld3(ro + 3 * i), ld3(rd + 3 * i), load_tri(tv + 9 * i), eps, t);
    t_out[i] = t;
    hit_out[i] = (uint8_t)h;
}

// ------------------------------------------------------------------------------------------
// LDS triangle tile shared by the any-hit / first-hit kernels
// ------------------------------------------------------------------------------------------
constexpr int kQueryThreads = 256;
// (a2) shared triangle set, lane = ray.  out must be zero-initialised.
__global__ __launch_bounds__(kQueryThreads) void any_hit_shared_kernel(
    const float *__restrict__ ro, const float *__restrict__ rd, int64_t R,
    const float *__restrict__ tv, int64_t T, const uint8_t *__restrict__ active, float eps,
    float thr, uint8_t *__restrict__ out, int64_t tri_per_split) {
    __shared__ TriRec lds[kTile];
    const int64_t r = (int64_t)blockIdx.x * kQueryThreads + threadIdx.x;
    const bool valid = r < R;
    const int lane = threadIdx.x & 63;
    // idle lanes of the last wave repeat the last ray (a zero ray would push the whole wave off
    // Moller-Trumbore's fast path); they are never in `want`
    const int64_t rr = valid ? r : R - 1;
    const V3 o = ld3(ro + 3 * rr);
    const V3 d = ld3(rd + 3 * rr);
    const int64_t begin = (int64_t)blockIdx.y * tri_per_split;
    const int64_t end = (begin + tri_per_split < T) ? begin + tri_per_split : T;
    bool any = false;
    uint64_t want = __builtin_amdgcn_ballot_w64(valid);  // rays of this wave still unblocked
    for (int64_t base = begin; base < end; base += kTile) {
        __syncthreads();
        stage_tile(lds, tv, active, base, end);
        __syncthreads();
        if (want == 0) continue;  // wave-uniform: every ray of this wave is already blocked
        const int n = (int)((end - base < kTile) ? end - base : kTile);
        for (int j = 0; j < n; ++j) {
            const TriRec rec = lds[j];  // broadcast read
            float t;
            const uint64_t m = moller_trumbore_wave<true>(o, d, rec_tri(rec), eps, want, &t);
            if (m != 0) {  // rare, wave-uniform: the distance / activity tests and the update of `want`
                asm volatile("" ::: "memory");  // keep this a scalar branch (not folded into a lane mask)
                any = any || (((m >> lane) & 1ull) && (t < thr) && rec.active);
                want &= ~__builtin_amdgcn_ballot_w64(any);
            }
        }
    }
    if (valid && any) out[r] = 1;
}

// (a4) key = (ordered(t) << 32) | tie, tie = (num_tiles-1-tile)*bs + idx_in_tile:
// the smallest key is the smallest t; among equal t the LATEST tile, then the lowest index.
struct TileTie {
    int64_t bs, nb, ntiles;
};
__device__ __forceinline__ uint64_t first_hit_key(float t, int64_t j, const TileTie &tt) {
    int64_t tile = (j < tt.nb * tt.bs) ? j / tt.bs : tt.nb;
    int64_t in_tile = j - tile * tt.bs;
    uint64_t tie = (uint64_t)((tt.ntiles - 1 - tile) * tt.bs + in_tile);
    return ((uint64_t)float_to_ordered(t) << 32) | tie;
}

__global__ __launch_bounds__(kQueryThreads) void first_hit_shared_kernel(
    const float *__restrict__ ro, const float *__restrict__ rd, int64_t R,
    const float *__restrict__ tv, int64_t T, const uint8_t *__restrict__ active, float eps,
    TileTie tt, unsigned long long *__restrict__ keys, int64_t tri_per_split, int64_t index_offset) {
    __shared__ TriRec lds[kTile];
    const int64_t r = (int64_t)blockIdx.x * kQueryThreads + threadIdx.x;
    const bool valid = r < R;
    const int lane = threadIdx.x & 63;
    const int64_t rr = valid ? r : R - 1;  // idle lanes repeat the last ray (see any_hit_shared_kernel)
    const V3 o = ld3(ro + 3 * rr);
    const V3 d = ld3(rd + 3 * rr);
    const int64_t begin = (int64_t)blockIdx.y * tri_per_split;
    const int64_t end = (begin + tri_per_split < T) ? begin + tri_per_split : T;
    const uint64_t want = __builtin_amdgcn_ballot_w64(valid);
    uint64_t best = ~0ull;
    for (int64_t base = begin; base < end; base += kTile) {
        __syncthreads();
        stage_tile(lds, tv, active, base, end);
        __syncthreads();
        const int n = (int)((end - base < kTile) ? end - base : kTile);
        for (int j = 0; j < n; ++j) {
            const TriRec rec = lds[j];
            float t;
            const uint64_t m = moller_trumbore_wave<true>(o, d, rec_tri(rec), eps, want, &t);
            // rare, wave-uniform: key construction only when some ray of the wave hits this triangle.
            // A hit with t == +inf is treated as a miss by the reference (isinf/isfinite fix-ups)
            if (m != 0) {
                asm volatile("" ::: "memory");  // keep this a scalar branch (not folded into a lane mask)
                if (((m >> lane) & 1ull) && rec.active && is_finite(t)) {
                    uint64_t k = first_hit_key(t, index_offset + base + j, tt);
                    best = (k < best) ? k : best;
                }
            }
        }
    }
    if (valid && best != ~0ull) atomicMin(keys + r, (unsigned long long)best);
}

__global__ __launch_bounds__(256) void first_hit_finalize_kernel(
    const unsigned long long *__restrict__ keys, int64_t R, TileTie tt, int32_t *__restrict__ idx,
    float *__restrict__ t_out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const uint64_t k = keys[r];
    if (k == ~0ull) {
        idx[r] = -1;
        t_out[r] = kInf;
        return;
    }
    const uint64_t tie = k & 0xffffffffull;
    const int64_t tile = tt.ntiles - 1 - (int64_t)(tie / (uint64_t)tt.bs);
    const int64_t in_tile = (int64_t)(tie % (uint64_t)tt.bs);
    idx[r] = (int32_t)(tile * tt.bs + in_tile);
    t_out[r] = ordered_to_float((uint32_t)(k >> 32));
}

// per-ray triangle sets and/or per-ray active masks: one wavefront per ray
template <bool FIRST>
__global__ __launch_bounds__(256) void per_ray_kernel(
    const float *__restrict__ ro, const float *__restrict__ rd, int64_t R,
    const float *__restrict__ tv, int64_t T, int64_t tv_stride, const uint8_t *__restrict__ active,
    int64_t act_stride, float eps, float thr, TileTie tt, uint8_t *__restrict__ any_out,
    int32_t *__restrict__ idx_out, float *__restrict__ t_out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (r >= R) return;  // wave-uniform
    const V3 o = ld3(ro + 3 * r), d = ld3(rd + 3 * r);
    const float *tvr = tv + r * tv_stride;
    const uint8_t *act = active ? active + r * act_stride : nullptr;
    bool any = false;
    uint64_t best = ~0ull;
    for (int64_t j = lane; j < T; j += 64) {
        float t;
        bool h = moller_trumbore(o, d, load_tri(tvr + 9 * j), eps, t);
        h = h && (!act || act[j]);
        if (FIRST) {
            if (h && is_finite(t)) {
                uint64_t k = first_hit_key(t, j, tt);
                best = (k < best) ? k : best;
            }
        } else {
            any = any || (h && (t < thr));
        }
    }
    if (FIRST) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            uint32_t lo = __shfl_xor((uint32_t)best, off, 64);
            uint32_t hi = __shfl_xor((uint32_t)(best >> 32), off, 64);
            uint64_t other = ((uint64_t)hi << 32) | lo;
            best = (other < best) ? other : best;
        }
        if (lane == 0) {
            if (best == ~0ull) {
                idx_out[r] = -1;
                t_out[r] = kInf;
            } else {
                const uint64_t tie = best & 0xffffffffull;
                const int64_t tile = tt.ntiles - 1 - (int64_t)(tie / (uint64_t)tt.bs);
                idx_out[r] = (int32_t)(tile * tt.bs + (int64_t)(tie % (uint64_t)tt.bs));
                t_out[r] = ordered_to_float((uint32_t)(best >> 32));
            }
        }
    } else {
        const bool w = __ballot(any) != 0ull;
        if (lane == 0) any_out[r] = (uint8_t)w;
    }
}

// ------------------------------------------------------------------------------------------
// (a5) reverse mode of t = f * <q, e2> on the hit face (geometry/_mesh.py:226-255, 327-338)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void first_hit_vjp_kernel(
    const float *__restrict__ vertices, const int32_t *__restrict__ triangles,
    const float *__restrict__ ro, const float *__restrict__ rd, const int32_t *__restrict__ face,
    const float *__restrict__ tbar_in, int64_t R, float *__restrict__ gv, float *__restrict__ go,
    float *__restrict__ gd) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const int32_t f_id = face[r];
    V3 obar{0, 0, 0}, dbar{0, 0, 0};
    if (f_id >= 0) {
        const int32_t i0 = triangles[3 * (int64_t)f_id], i1 = triangles[3 * (int64_t)f_id + 1],
                      i2 = triangles[3 * (int64_t)f_id + 2];
        const V3 v0 = ld3(vertices + 3 * (int64_t)i0), v1 = ld3(vertices + 3 * (int64_t)i1),
                 v2 = ld3(vertices + 3 * (int64_t)i2);
        const V3 o = ld3(ro + 3 * r), d = ld3(rd + 3 * r);
        const MtHardBar g = mt_t_vjp(o, d, v0, v1, v2, tbar_in[r]);
        obar = g.o;
        dbar = g.d;
        const V3 e1bar = g.v1, e2bar = g.v2;
        if (gv) {
            const V3 v0bar = g.v0;
            atomicAdd(gv + 3 * (int64_t)i0 + 0, v0bar.x);
            atomicAdd(gv + 3 * (int64_t)i0 + 1, v0bar.y);
            atomicAdd(gv + 3 * (int64_t)i0 + 2, v0bar.z);
            atomicAdd(gv + 3 * (int64_t)i1 + 0, e1bar.x);
            atomicAdd(gv + 3 * (int64_t)i1 + 1, e1bar.y);
            atomicAdd(gv + 3 * (int64_t)i1 + 2, e1bar.z);
            atomicAdd(gv + 3 * (int64_t)i2 + 0, e2bar.x);
            atomicAdd(gv + 3 * (int64_t)i2 + 1, e2bar.y);
            atomicAdd(gv + 3 * (int64_t)i2 + 2, e2bar.z);
        }
    }
    if (go) st3(go + 3 * r, obar);
    if (gd) st3(gd + 3 * r, dbar);
}

// ------------------------------------------------------------------------------------------
// (a1) reverse mode of the hard-mode `t` of ray_intersect_triangle (_utils.py:1316; also the free
// first_triangle_hit_by_ray, whose t is the paired operator on the hit triangle)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// dense: lane = triangle (its 9 cotangents accumulate in registers over the block's rays, one set
// of atomics at the end), rays wave-uniform (their 6 cotangents: wave reduction, one atomic per wave)
__global__ __launch_bounds__(256) void mt_dense_vjp_kernel(
    const float *__restrict__ ro, const float *__restrict__ rd, int64_t R, const float *__restrict__ tv,
    int64_t T, const float *__restrict__ tbar, float *__restrict__ go, float *__restrict__ gd,
    float *__restrict__ gtv, int rays_per_block) {
    const int64_t j = (int64_t)blockIdx.y * 256 + threadIdx.x;
    const bool in = j < T;
    const int64_t jj = in ? j : T - 1;
    const V3 v0 = ld3(tv + 9 * jj), v1 = ld3(tv + 9 * jj + 3), v2 = ld3(tv + 9 * jj + 6);
    const int64_t r0 = (int64_t)blockIdx.x * rays_per_block;
    const int64_t r1 = (r0 + rays_per_block < R) ? r0 + rays_per_block : R;
    V3 a0{0, 0, 0}, a1{0, 0, 0}, a2{0, 0, 0};
    const int lane = threadIdx.x & 63;
    for (int64_t r = r0; r < r1; ++r) {
        const V3 o = ld3(ro + 3 * r), d = ld3(rd + 3 * r);
        const float tb = in ? tbar[r * T + j] : 0.0f;
        MtHardBar g = mt_t_vjp(o, d, v0, v1, v2, tb);
        if (!in || tb == 0.0f) g = MtHardBar{};  // a zero cotangent contributes nothing (also kills 0 * inf)
        a0 = a0 + g.v0;
        a1 = a1 + g.v1;
        a2 = a2 + g.v2;
        if (go) {
            const float x = wave_sum(g.o.x), y = wave_sum(g.o.y), z = wave_sum(g.o.z);
            if (lane == 0) atomic_add3f(go + 3 * r, V3{x, y, z});
        }
        if (gd) {
            const float x = wave_sum(g.d.x), y = wave_sum(g.d.y), z = wave_sum(g.d.z);
            if (lane == 0) atomic_add3f(gd + 3 * r, V3{x, y, z});
        }
    }
    if (gtv && in) {
        atomic_add3f(gtv + 9 * j, a0);
        atomic_add3f(gtv + 9 * j + 3, a1);
        atomic_add3f(gtv + 9 * j + 6, a2);
    }
}

// paired: lane = element, every gradient has its own slot (plain stores)
__global__ __launch_bounds__(256) void mt_paired_vjp_kernel(
    const float *__restrict__ ro, const float *__restrict__ rd, const float *__restrict__ tv, int64_t n,
    const float *__restrict__ tbar, float *__restrict__ go, float *__restrict__ gd, float *__restrict__ gtv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float tb = tbar[i];
    MtHardBar g = mt_t_vjp(ld3(ro + 3 * i), ld3(rd + 3 * i), ld3(tv + 9 * i), ld3(tv + 9 * i + 3), ld3(tv + 9 * i + 6), tb);
    if (tb == 0.0f) g = MtHardBar{};
    if (go) st3(go + 3 * i, g.o);
    if (gd) st3(gd + 3 * i, g.d);
    if (gtv) {
        st3(gtv + 9 * i, g.v0);
        st3(gtv + 9 * i + 3, g.v1);
        st3(gtv + 9 * i + 6, g.v2);
    }
}

static TileTie make_tie(int64_t T, int64_t batch_size) {
    int64_t bs = batch_size <= 0 ? T : batch_size;  // _utils.py:1832-1834
    if (bs > T) bs = T;
    if (bs < 1) bs = 1;
    TileTie tt;
    tt.bs = bs;
    tt.nb = T / bs;
    tt.ntiles = tt.nb + ((T % bs) ? 1 : 0);
    return tt;
}

// triangles per blockIdx.y split so that small ray batches still launch >= ~2048 blocks
static int64_t choose_split(int64_t ray_blocks, int64_t T, int64_t *nsplit_out) {
    int64_t tiles = ceil_div(T, kTile);
    int64_t want = ceil_div(2048, ray_blocks);
    if (want < 1) want = 1;
    if (want > tiles) want = tiles;
    if (want > 65535) want = 65535;
    int64_t tiles_per_split = ceil_div(tiles, want);
    *nsplit_out = ceil_div(tiles, tiles_per_split);
    return tiles_per_split * kTile;
}

}  // namespace drt

using namespace drt;

extern "C" {

static int32_t dense_launch(const float *ro, const float *rd, int64_t R, const float *tv, int64_t T, float eps,
                            float *t_out, uint8_t *hit_out, int64_t B, DenseBatch nb, void *stream) {
    // with B > 1 every problem's output block starts R*T elements after the previous one: the aligned
    // kernel needs each of them 16-B aligned (T % 16 == 0 makes R*T a multiple of 16)
    const bool al16 = (T % 16 == 0) && ((reinterpret_cast<uintptr_t>(t_out) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(hit_out) & 15) == 0);
    const bool al4 = (T % 4 == 0) && ((reinterpret_cast<uintptr_t>(t_out) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(hit_out) & 3) == 0);
    const int64_t cols = ceil_div(T, (int64_t)kDenseCols);
    DRT_REQUIRE(cols <= 65535, "too many triangles for one launch (%lld)", (long long)T);
    // rays per block: as many as possible (amortises the triangle loads) while keeping >= ~1280 blocks.
    // Measured on the literal configs[1] launch (256 rays, inside a HIP graph, with the LDS-staged triangle
    // loads): 7.5 / 7.4 / 9.2 us at 640 / 1280 / 2560 blocks; 1024 rays: 20.1 / 17.9 / 18.5 us
    // (profiles/r02/literal_lab.txt).  Upper bound 32: 0.806 ms vs 0.83 ms at 64 on the bench shape.
#ifdef DRT_LAB  // lab builds only (-DDRT_LAB): DRT_DENSE_BLOCKS=<n>
    static const int64_t target_blocks = [] {
        const char *e = getenv("DRT_DENSE_BLOCKS");
        const long v = e ? atol(e) : 0;
        return (int64_t)(v > 0 ? v : 1280);
    }();
#else
    constexpr int64_t target_blocks = 1280;
#endif
    int64_t rpb = (R * cols * B) / target_blocks;
    if (rpb < 1) rpb = 1;
    if (rpb > 32) rpb = 32;
    if (al16 && rpb > kDenseGroup) rpb -= rpb % kDenseGroup;  // whole groups: no short trailing group
    const int64_t rows = ceil_div(R, rpb);
    DRT_REQUIRE(rows < (1ll << 31), "too many rays for one launch");
    // 32-bit byte offsets inside a block's rows: rpb * T * 4 bytes must fit
    DRT_REQUIRE(T <= (1ll << 24), "too many triangles per row for one launch (%lld)", (long long)T);
    dim3 grid((unsigned)rows, (unsigned)cols, (unsigned)B);
    hipStream_t s = as_stream(stream);
    // coalesced LDS staging of the triangles needs 16-B aligned rows of 4 triangles
#ifdef DRT_LAB  // lab builds only: DRT_DENSE_STAGE=0 keeps the direct per-lane loads
    static const bool stage_ok = [] {
        const char *e = getenv("DRT_DENSE_STAGE");
        return !(e && e[0] == '0');
    }();
#else
    constexpr bool stage_ok = true;
#endif
    // worth it only when a block walks few rays (configs[1]: 8.0 -> 7.4 us; at 32 rays per block the direct
    // loads hide behind the other blocks' arithmetic and staging costs 1 %: profiles/r02/literal_lab.txt)
    const int stage = (stage_ok && rpb <= kDenseGroup && (reinterpret_cast<uintptr_t>(tv) & 15) == 0 &&
                       (nb.tv_stride % 4) == 0) ? 1 : 0;
    if (al16)
        hipLaunchKernelGGL(mt_dense_aligned_kernel, grid, dim3(kDenseThreads), 0, s, ro, rd, R, tv, T, eps,
                           t_out, hit_out, (int)rpb, stage, nb);
    else if (al4)
        hipLaunchKernelGGL((mt_dense_kernel<true>), grid, dim3(kDenseThreads), 0, s, ro, rd, R, tv, T,
                           eps, t_out, hit_out, (int)rpb, nb);
    else
        hipLaunchKernelGGL((mt_dense_kernel<false>), grid, dim3(kDenseThreads), 0, s, ro, rd, R, tv, T,
                           eps, t_out, hit_out, (int)rpb, nb);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_ray_intersect_triangle_dense(const float *ro, const float *rd, int64_t R,
                                         const float *tv, int64_t T, float eps, float *t_out,
                                         uint8_t *hit_out, void *stream) {
    DRT_REQUIRE(R >= 0 && T >= 0, "negative size");
    if (R == 0 || T == 0) return DRT_OK;
    DRT_REQUIRE(ro && rd && tv && t_out && hit_out, "null pointer");
    return dense_launch(ro, rd, R, tv, T, eps, t_out, hit_out, 1, DenseBatch{0, 0}, stream);
}

int32_t drt_ray_intersect_triangle_dense_batched(const float *ro, const float *rd, int64_t ray_batch_stride,
                                                 int64_t R, const float *tv, int64_t tv_batch_stride, int64_t T,
                                                 int64_t B, float eps, float *t_out, uint8_t *hit_out,
                                                 void *stream) {
    DRT_REQUIRE(R >= 0 && T >= 0 && B >= 0, "negative size");
    DRT_REQUIRE(ray_batch_stride == 0 || ray_batch_stride == 3 * R, "ray_batch_stride is 0 (shared) or 3 * num_rays");
    DRT_REQUIRE(tv_batch_stride == 0 || tv_batch_stride == 9 * T, "tv_batch_stride is 0 (shared) or 9 * num_triangles");
    DRT_REQUIRE(B <= 65535, "at most 65535 problems per launch");
    if (R == 0 || T == 0 || B == 0) return DRT_OK;
    DRT_REQUIRE(ro && rd && tv && t_out && hit_out, "null pointer");
    return dense_launch(ro, rd, R, tv, T, eps, t_out, hit_out, B, DenseBatch{ray_batch_stride, tv_batch_stride},
                        stream);
}

int32_t drt_ray_intersect_triangle_paired(const float *ro, const float *rd, const float *tv,
                                          int64_t n, float eps, float *t_out, uint8_t *hit_out,
                                          void *stream) {
    DRT_REQUIRE(n >= 0, "negative size");
    if (n == 0) return DRT_OK;
    DRT_REQUIRE(ro && rd && tv && t_out && hit_out, "null pointer");
    hipLaunchKernelGGL(mt_paired_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       as_stream(stream), ro, rd, tv, n, eps, t_out, hit_out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_ray_intersect_triangle_vjp(const float *ro, const float *rd, int64_t R, const float *tv,
                                       int64_t T, int32_t dense, const float *t_cot, float *go,
                                       float *gd, float *gtv, void *stream) {
    DRT_REQUIRE(R >= 0 && T >= 0, "negative size");
    DRT_REQUIRE(dense || R == T, "paired layout needs one triangle per ray");
    if (R == 0 || T == 0) return DRT_OK;
    DRT_REQUIRE(ro && rd && tv && t_cot, "null pointer");
    hipStream_t s = as_stream(stream);
    if (dense) {
        const int64_t cols = ceil_div(T, 256);
        DRT_REQUIRE(cols <= 65535, "too many triangles for one launch (%lld)", (long long)T);
        int64_t rpb = (R * cols) / 2048;
        if (rpb < 1) rpb = 1;
        if (rpb > 64) rpb = 64;
        hipLaunchKernelGGL(mt_dense_vjp_kernel, dim3((unsigned)ceil_div(R, rpb), (unsigned)cols), dim3(256), 0, s,
                           ro, rd, R, tv, T, t_cot, go, gd, gtv, (int)rpb);
    } else {
        hipLaunchKernelGGL(mt_paired_vjp_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, s, ro, rd, tv, R,
                           t_cot, go, gd, gtv);
    }
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_ray_intersect_any_triangle(const float *ro, const float *rd, int64_t R,
                                       const float *tv, int64_t T, int64_t tv_ray_stride,
                                       const uint8_t *active, int64_t active_ray_stride, float eps,
                                       float hit_tol, uint8_t *out, void *stream) {
    DRT_REQUIRE(R >= 0 && T >= 0, "negative size");
    if (R == 0) return DRT_OK;
    DRT_REQUIRE(out, "null output");
    DRT_HIP(fill_bytes_async(out, 0, (size_t)R, as_stream(stream)));
    if (T == 0) return DRT_OK;  // _utils.py:1441-1450
    DRT_REQUIRE(ro && rd && tv, "null pointer");
    DRT_REQUIRE(tv_ray_stride == 0 || tv_ray_stride == 9 * T, "tv_ray_stride must be 0 or 9*T");
    DRT_REQUIRE(active_ray_stride == 0 || active_ray_stride == T, "active_ray_stride must be 0 or T");
    const float thr = 1.0f - hit_tol;  // _utils.py:1422
    if (tv_ray_stride == 0 && active_ray_stride == 0) {
        const int64_t ray_blocks = ceil_div(R, kQueryThreads);
        int64_t nsplit;
        const int64_t tps = choose_split(ray_blocks, T, &nsplit);
        hipLaunchKernelGGL(any_hit_shared_kernel, dim3((unsigned)ray_blocks, (unsigned)nsplit),
                           dim3(kQueryThreads), 0, as_stream(stream), ro, rd, R, tv, T, active, eps,
                           thr, out, tps);
    } else {
        TileTie tt = make_tie(T, 0);
        hipLaunchKernelGGL(per_ray_kernel<false>, dim3((unsigned)ceil_div(R * 64, 256)), dim3(256),
                           0, as_stream(stream), ro, rd, R, tv, T, tv_ray_stride, active,
                           active_ray_stride, eps, thr, tt, out, (int32_t *)nullptr,
                           (float *)nullptr);
    }
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

size_t drt_first_triangle_hit_by_ray_workspace_size(int64_t num_rays) {
    return num_rays > 0 ? (size_t)num_rays * 8 : 0;
}

int32_t drt_first_triangle_hit_by_ray(const float *ro, const float *rd, int64_t R,
                                      const float *tv, int64_t T, int64_t tv_ray_stride,
                                      const uint8_t *active, int64_t active_ray_stride, float eps,
                                      int64_t batch_size, int32_t *idx, float *t_out, void *ws,
                                      size_t ws_bytes, void *stream) {
    DRT_REQUIRE(R >= 0 && T >= 0, "negative size");
    if (R == 0) return DRT_OK;
    DRT_REQUIRE(idx && t_out, "null output");
    DRT_REQUIRE(T < (1ll << 31), "too many triangles");
    DRT_REQUIRE(tv_ray_stride == 0 || tv_ray_stride == 9 * T, "tv_ray_stride must be 0 or 9*T");
    DRT_REQUIRE(active_ray_stride == 0 || active_ray_stride == T, "active_ray_stride must be 0 or T");
    hipStream_t s = as_stream(stream);
    TileTie tt = make_tie(T > 0 ? T : 1, batch_size);
    if (T == 0 || (tv_ray_stride == 0 && active_ray_stride == 0)) {
        if (ws_bytes < (size_t)R * 8 || !ws)
            return fail(DRT_E_CAPACITY, "workspace too small: need %zu bytes", (size_t)R * 8);
        auto *keys = reinterpret_cast<unsigned long long *>(ws);
        DRT_HIP(fill_bytes_async(keys, 0xff, (size_t)R * 8, s));
        if (T > 0) {
            DRT_REQUIRE(ro && rd && tv, "null pointer");
            const int64_t ray_blocks = ceil_div(R, kQueryThreads);
            int64_t nsplit;
            const int64_t tps = choose_split(ray_blocks, T, &nsplit);
            hipLaunchKernelGGL(first_hit_shared_kernel,
                               dim3((unsigned)ray_blocks, (unsigned)nsplit), dim3(kQueryThreads), 0,
                               s, ro, rd, R, tv, T, active, eps, tt, keys, tps, (int64_t)0);
            DRT_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(first_hit_finalize_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256),
                           0, s, keys, R, tt, idx, t_out);
    } else {
        DRT_REQUIRE(ro && rd && tv, "null pointer");
        hipLaunchKernelGGL(per_ray_kernel<true>, dim3((unsigned)ceil_div(R * 64, 256)), dim3(256), 0,
                           s, ro, rd, R, tv, T, tv_ray_stride, active, active_ray_stride, eps, 0.0f,
                           tt, (uint8_t *)nullptr, idx, t_out);
    }
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_first_hit_keys(const float *ro, const float *rd, int64_t R, const float *tv_block,
                           int64_t T_block, int64_t index_offset, int64_t total_triangles,
                           const uint8_t *active_block, float eps, int64_t batch_size,
                           uint64_t *keys, int32_t init, void *stream) {
    DRT_REQUIRE(R >= 0 && T_block >= 0 && index_offset >= 0, "negative size");
    DRT_REQUIRE(index_offset + T_block <= total_triangles, "block exceeds the mesh");
    DRT_REQUIRE(total_triangles < (1ll << 31), "too many triangles");
    if (R == 0) return DRT_OK;
    DRT_REQUIRE(keys, "null keys");
    hipStream_t s = as_stream(stream);
    if (init) DRT_HIP(fill_bytes_async(keys, 0xff, (size_t)R * 8, s));
    if (T_block == 0) return DRT_OK;
    DRT_REQUIRE(ro && rd && tv_block, "null pointer");
    const TileTie tt = make_tie(total_triangles, batch_size);
    const int64_t ray_blocks = ceil_div(R, kQueryThreads);
    int64_t nsplit;
    const int64_t tps = choose_split(ray_blocks, T_block, &nsplit);
    hipLaunchKernelGGL(first_hit_shared_kernel, dim3((unsigned)ray_blocks, (unsigned)nsplit),
                       dim3(kQueryThreads), 0, s, ro, rd, R, tv_block, T_block, active_block, eps, tt,
                       reinterpret_cast<unsigned long long *>(keys), tps, index_offset);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_first_hit_finalize(const uint64_t *keys, int64_t R, int64_t total_triangles,
                               int64_t batch_size, int32_t *idx, float *t_out, void *stream) {
    DRT_REQUIRE(R >= 0, "negative size");
    if (R == 0) return DRT_OK;
    DRT_REQUIRE(keys && idx && t_out, "null pointer");
    const TileTie tt = make_tie(total_triangles > 0 ? total_triangles : 1, batch_size);
    hipLaunchKernelGGL(first_hit_finalize_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0,
                       as_stream(stream), reinterpret_cast<const unsigned long long *>(keys), R, tt, idx,
                       t_out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_first_hit_vjp(const float *vertices, const int32_t *triangles, const float *ro,
                          const float *rd, const int32_t *hit_index, const float *t_cotangent,
                          int64_t R, float *gv, float *go, float *gd, void *stream) {
    DRT_REQUIRE(R >= 0, "negative size");
    if (R == 0) return DRT_OK;
    DRT_REQUIRE(vertices && triangles && ro && rd && hit_index && t_cotangent, "null pointer");
    hipLaunchKernelGGL(first_hit_vjp_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0,
                       as_stream(stream), vertices, triangles, ro, rd, hit_index, t_cotangent, R, gv,
                       go, gd);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

}  // extern "C"
