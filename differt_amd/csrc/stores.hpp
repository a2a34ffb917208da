// stores.hpp -- wave-staged, coalesced output of array-of-struct rows (dense tracer, image method).
//
// A lane that owns one output ROW of R dwords (R = 12 for the vertices of an order-2 path) and stores it directly
// issues R dword stores at a lane stride of 4R bytes: every wave store instruction touches 64 * 4R / 128 = R / 2
// ... 2R different 128-B lines, and the R instructions of the row touch the same lines again.  Here the wave first
// puts its 64 rows into a wave-private LDS region (the rows of a wave are contiguous in the output), then stores
// the region with one 16-B `global_store_dwordx4 nt` per lane per KiB whose chunks start on 128-B lines of the
// OUTPUT: a store instruction covers eight whole lines; a region that does not start on a line boundary gets one
// partial store at its head and one at its tail.  (Measured for the dense Moeller-Trumbore operator,
// profiles/r02/store_lab.txt: whole-line stores cost 0.64x of straddling ones; nontemporal 0.87x of plain.)
//
// No block barrier: LDS operations of one wave execute in program order, a wave-level fence keeps the compiler
// from reordering the staging writes and the flush reads.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace drt {

typedef uint32_t st_u32x4 __attribute__((ext_vector_type(4)));

// wave-uniform pointer -> SGPR pair (the "s" constraint of the store below needs a value the compiler KNOWS to be
// uniform; a pointer derived from threadIdx.x >> 6 is uniform in fact but not by analysis)
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<T *>(((uint64_t)hi << 32) | lo);
}

// 16-B nontemporal store at (wave-uniform base) + (32-bit lane offset).  gfx9-family hazard: a VMEM store of more
// than 64 bits followed by a VALU write of its data VGPRs needs one wait state, and LLVM's hazard recogniser does
// not look inside inline asm -- the `s_nop 0` covers it.
__device__ __forceinline__ void nt_store_b128(char *base, uint32_t off, st_u32x4 v) {
#if defined(DRT_LAB) && defined(DRT_STORE_LAB_PLAIN)  // lab: the same store without the nontemporal hint
    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 0" : : "v"(off), "v"(v), "s"(base) : "memory");
#elif defined(DRT_LAB) && defined(DRT_STORE_LAB_SC)   // lab: system-coherent write-through hints
    asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 0" : : "v"(off), "v"(v), "s"(base) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0" : : "v"(off), "v"(v), "s"(base) : "memory");
#endif
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Flush `nbytes` (multiple of 16, wave-uniform) of the wave's LDS region to `g` (16-B aligned, wave-uniform).
// REGION_BYTES = size of a full region (compile-time bound of the pass count).
template <int REGION_BYTES>
__device__ __forceinline__ void flush_region_b128(const uint32_t *lds_w, char *g, uint32_t nbytes, int lane) {
    const uint32_t head = (uint32_t)reinterpret_cast<uint64_t>(g) & 127u;  // multiple of 16
    char *a0 = uniform_ptr(g - head);
    const uint32_t end = head + nbytes;
    constexpr int kPasses = (REGION_BYTES + 112 + 1023) / 1024;
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
        if ((uint32_t)p * 1024u < end) {  // wave-uniform
            const uint32_t off = (uint32_t)p * 1024u + (uint32_t)lane * 16u;
            if (off >= head && off < end) {
                const st_u32x4 v =
                    *reinterpret_cast<const st_u32x4 *>(reinterpret_cast<const char *>(lds_w) + (off - head));
                nt_store_b128(a0, off, v);
            }
        }
    }
}

// Any alignment: dword q of the region goes to g[q]; a wave store instruction covers 256 contiguous bytes.
__device__ __forceinline__ void flush_region_b32(const uint32_t *lds_w, uint32_t *g, uint32_t ndw, int lane) {
    for (uint32_t q = (uint32_t)lane; q < ndw; q += 64u) __builtin_nontemporal_store(lds_w[q], g + q);
}

// ---- the mirror image for INPUT rows: a wave reads its 64 contiguous rows with 16-B loads (1 KiB per instruction),
// parks them in its LDS region and every lane picks its own row back (a lane that loads its row directly issues
// R dword loads at a 4R-byte stride: 2R lines per instruction, the same lines R times) ----
template <int REGION_BYTES>
struct RegionRegs {
    st_u32x4 v[(REGION_BYTES + 1023) / 1024];
};

// g: wave-uniform, 16-B aligned; nbytes: wave-uniform, multiple of 16
template <int REGION_BYTES>
__device__ __forceinline__ void region_load_b128(const char *g, uint32_t nbytes, int lane, RegionRegs<REGION_BYTES> &r) {
#pragma unroll
    for (int p = 0; p < (REGION_BYTES + 1023) / 1024; ++p) {
        const uint32_t off = (uint32_t)p * 1024u + (uint32_t)lane * 16u;
        r.v[p] = st_u32x4{0u, 0u, 0u, 0u};
        if (off < nbytes) r.v[p] = __builtin_nontemporal_load(reinterpret_cast<const st_u32x4 *>(g + off));
    }
}

template <int REGION_BYTES>
__device__ __forceinline__ void region_to_lds(uint32_t *lds_w, int lane, const RegionRegs<REGION_BYTES> &r) {
#pragma unroll
    for (int p = 0; p < (REGION_BYTES + 1023) / 1024; ++p) {
        const uint32_t off = (uint32_t)p * 1024u + (uint32_t)lane * 16u;
        if (off < (uint32_t)REGION_BYTES) *reinterpret_cast<st_u32x4 *>(reinterpret_cast<char *>(lds_w) + off) = r.v[p];
    }
}

// any alignment / row count: dword q of the rows goes to lds_w[q]
__device__ __forceinline__ void region_load_b32(uint32_t *lds_w, const uint32_t *g, uint32_t ndw, int lane) {
    for (uint32_t q = (uint32_t)lane; q < ndw; q += 64u) lds_w[q] = g[q];
}

// bits 4d .. 4d+3 of `bits16` -> the four 0/1 bytes of dword d (a mask row written from a wave ballot)
__device__ __forceinline__ uint32_t nibble_to_bytes(uint32_t nib) {
    return (nib * 0x00204081u) & 0x01010101u;  // the 16 partial products are distinct powers of two: no carries
}

}  // namespace drt
