// trace_common.hpp -- pieces shared by the hard-mask tracer (trace.hip) and the smoothed tracer
// (smooth.hip): candidate sources (table or GPU unranking), mirror gathers, path reconstruction
// from a flat (tx, rx, candidate) index, and the host-side argument builders.
// Reference: geometry/_solvers.py:499-586 (`_trace_path_candidates`, gathers + image method).
#pragma once

#include "common.hpp"
#include "geom.hpp"
#include "image_chain.hpp"
#include "mesh.hpp"

#pragma clang fp contract(off)

namespace drt {

template <int K>
struct KA {
    static constexpr int n = K > 0 ? K : 1;  // array extent that is legal for K == 0
};

struct CandSrc {
    const int32_t *table;
    int64_t count;    // number of candidate rows
    int64_t rank_lo;
    int64_t num_nodes;
    const int32_t *node_map;
    int32_t id_scale;
    int64_t pw[DRT_MAX_ORDER];  // pw[j] = (num_nodes-1)^(K-1-j); product mode: sizes of the positions after j
    // product mode (visibility-pruned candidate space of the hybrid tracer): position 0 draws from
    // first_map, position K-1 from last_map, the others from node_map / all nodes
    int32_t product;
    int32_t small;  // rank_lo + count and every pw[] fit 32 bits: unrank with 32-bit divisions
    const int32_t *first_map, *last_map;
    // ragged mode: one row space per (tx, rx) pair, concatenated (pair_offsets = prefix sums)
    int32_t ragged;
    int32_t prefix_kernel;   // stage A: lane = prefix (first K-1 interactions), see trace_filter_prefix_kernel
    int64_t max_prefixes;    // max over transmitters of F_tx * num_nodes^(K-2) (grid sizing)
    int64_t npairs, mid_pw;  // mid_pw = num_nodes^(K-2)
    // packed keys (DRT_CAND_PACKED_KEYS): a flat key IS the candidate -- (tx * nrx + rx) * n^K + sum_j m_j n^(K-1-j),
    // the keys drt_trace_paths_beam returns; pw[j] = n^(K-1-j), count = n^K
    int32_t packed;
    int32_t pair_blocks;  // per-pair table made of coplanar-pair blocks of 2^K rows (DRT_CAND_PAIR_BLOCKS)
    const int64_t *pair_offsets, *first_off, *last_off;
};

struct TraceArgs {
    const float *tri_verts;  // [T,3,3]
    const float *normals;    // [T,3]
    const uint8_t *mask;     // [T] or null
    int64_t T;
    int64_t T_occ;  // triangles the occlusion stage tests: T, or 0 with DRT_TRACE_SKIP_OCCLUSION
    const float *tx;
    int64_t ntx;
    const float *rx;
    int64_t nrx;
    float eps, thr, min_len;
};

// candidate row -> K primitive ids (table value or GPU unranking of rank_lo + row):
// c_0 = r / (n-1)^(K-1);  d_j = (r / (n-1)^(K-1-j)) mod (n-1);  c_j = d_j + (d_j >= c_{j-1})
// which enumerates "no two equal neighbours" tuples in lexicographic order (graph.rs:400-470).
// UInt = uint32_t when the whole window fits 32 bits (CandSrc::small): a 64-bit division costs ~100
// instructions on gfx950, which matters when a launch holds a single (tx, rx) pair.
template <int K, typename UInt>
__device__ __forceinline__ void unrank_candidate(const CandSrc &s, int64_t row, int32_t (&id)[KA<K>::n]) {
    UInt r = (UInt)(s.rank_lo + row);
    if (s.product) {
        // plain mixed-radix product, lexicographic; a tuple with two equal neighbours is not a path of
        // the graph (no self loops, graph.rs): it becomes a padding row (id -1 -> invalid, never emitted)
        int32_t prev = -1;
        bool bad = false;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const UInt q = r / (UInt)s.pw[j];
            r -= q * (UInt)s.pw[j];
            const int32_t *map = (j == 0) ? s.first_map : ((j == K - 1) ? s.last_map : s.node_map);
            const int32_t v = map ? map[q] : (int32_t)q;
            bad = bad || (v == prev);
            prev = v;
            id[j] = v * s.id_scale;
        }
        if (bad) {
#pragma unroll
            for (int j = 0; j < K; ++j) id[j] = -1;
        }
    } else {
        int64_t prev = -1;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const UInt q = r / (UInt)s.pw[j];
            r -= q * (UInt)s.pw[j];
            int64_t c = (int64_t)q;
            if (j > 0) c += (c >= prev) ? 1 : 0;
            prev = c;
            int32_t v = (int32_t)c;
            if (s.node_map) v = s.node_map[c];
            id[j] = v * s.id_scale;
        }
    }
}

template <int K>
__device__ __forceinline__ void load_candidate(const CandSrc &s, int64_t row,
                                               int32_t (&id)[KA<K>::n]) {
    if (K == 0) return;
    if (s.table) {
#pragma unroll
        for (int j = 0; j < K; ++j) id[j] = s.table[row * K + j];
    } else if (s.small) {
        unrank_candidate<K, uint32_t>(s, row, id);
    } else {
        unrank_candidate<K, uint64_t>(s, row, id);
    }
}

template <int K, bool QUADS>
struct Mirrors {
    V3 p[KA<K>::n], n[KA<K>::n];
    TriE tri[KA<K>::n];
    TriE tri2[QUADS ? KA<K>::n : 1];
    bool ok;      // ids in range (negative / out-of-range ids = padding rows: invalid)
    bool active;  // every touched triangle unmasked
};

template <int K, bool QUADS>
__device__ __forceinline__ void load_mirrors(const TraceArgs &a, const int32_t (&id)[KA<K>::n],
                                             Mirrors<K, QUADS> &m) {
    m.ok = true;
    m.active = true;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int64_t i = id[j];
        const bool ok = (i >= 0) && (i + (QUADS ? 1 : 0) < a.T);
        m.ok = m.ok && ok;
        const int64_t s = ok ? i : 0;
        m.tri[j] = load_tri(a.tri_verts + 9 * s);
        m.p[j] = m.tri[j].v0;                 // SV:552-557: first vertex of the (even) triangle
        m.n[j] = ld3(a.normals + 3 * s);      // SV:560-562
        if (QUADS) m.tri2[j] = load_tri(a.tri_verts + 9 * (s + 1));
        if (a.mask) {
            m.active = m.active && (a.mask[s] != 0);
            if (QUADS) m.active = m.active && (a.mask[s + 1] != 0);
        }
    }
}

// one mirror of a candidate (slot j); returns whether its id is in range, updates the active flag
template <int K, bool QUADS>
__device__ __forceinline__ bool load_one_mirror(const TraceArgs &a, int64_t i, int j, Mirrors<K, QUADS> &m,
                                                bool &active) {
    const bool ok = (i >= 0) && (i + (QUADS ? 1 : 0) < a.T);
    const int64_t s = ok ? i : 0;
    m.tri[j] = load_tri(a.tri_verts + 9 * s);
    m.p[j] = m.tri[j].v0;
    m.n[j] = ld3(a.normals + 3 * s);
    if (QUADS) m.tri2[j] = load_tri(a.tri_verts + 9 * (s + 1));
    if (a.mask) {
        active = active && (a.mask[s] != 0);
        if (QUADS) active = active && (a.mask[s + 1] != 0);
    }
    return ok;
}

template <int K>
__device__ __forceinline__ bool path_finite(const V3 (&full)[K + 2]) {
    bool fin = true;
#pragma unroll
    for (int j = 0; j < K + 2; ++j)
        fin = fin && is_finite(full[j].x) && is_finite(full[j].y) && is_finite(full[j].z);
    return fin;
}

// ragged mode: global row g -> (tx, rx) by binary search in the prefix sums, then the mixed-radix digits
// of the local rank over F_tx x N^(K-2) x L_rx (last digit fastest = lexicographic order)
template <int K, typename UInt>
__device__ __forceinline__ void ragged_digits(const CandSrc &s, int64_t local, int64_t it, int64_t ir,
                                              int32_t (&id)[KA<K>::n]) {
    const int32_t *F = s.first_map + s.first_off[it];
    const int32_t *L = s.last_map + s.last_off[ir];
    const UInt nL = (UInt)(s.last_off[ir + 1] - s.last_off[ir]);
    const UInt nN = (UInt)s.num_nodes;
    UInt r = (UInt)local;
    int32_t dig[KA<K>::n];
    {
        const UInt q = r / nL;
        dig[K - 1] = L[r - q * nL];
        r = q;
    }
#pragma unroll
    for (int j = K - 2; j >= 1; --j) {
        const UInt q = r / nN;
        const UInt d = r - q * nN;
        dig[j] = s.node_map ? s.node_map[d] : (int32_t)d;
        r = q;
    }
    dig[0] = F[r];
    bool bad = false;
#pragma unroll
    for (int j = 1; j < K; ++j) bad = bad || (dig[j] == dig[j - 1]);
#pragma unroll
    for (int j = 0; j < K; ++j) id[j] = bad ? -1 : dig[j] * s.id_scale;
}

template <int K>
__device__ __forceinline__ void ragged_decode(const CandSrc &s, int64_t nrx, int64_t g, int64_t &it,
                                              int64_t &ir, int32_t (&id)[KA<K>::n]) {
    int64_t lo = 0, hi = s.npairs;  // largest lo with pair_offsets[lo] <= g
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (s.pair_offsets[mid] <= g) lo = mid; else hi = mid;
    }
    it = lo / nrx;
    ir = lo - it * nrx;
    const int64_t local = g - s.pair_offsets[lo];
    if (s.table) {  // per-pair table (beam-pruned rows): ids are stored, nothing to unrank
        if constexpr (K >= 1) {
#pragma unroll
            for (int j = 0; j < K; ++j) id[j] = s.table[g * K + j];
        }
        return;
    }
    if constexpr (K >= 2) {
        if (s.small) ragged_digits<K, uint32_t>(s, local, it, ir, id);
        else ragged_digits<K, uint64_t>(s, local, it, ir, id);
    }
}

template <int K>
__device__ __forceinline__ bool key_to_path(const TraceArgs &a, const CandSrc &cs, int64_t flat,
                                            int64_t &it, int64_t &ir, int32_t (&id)[KA<K>::n],
                                            V3 (&p)[KA<K>::n], V3 (&n)[KA<K>::n], V3 (&full)[K + 2]) {
    if (cs.packed) {
        const int64_t pair = flat / cs.count;
        uint64_t rest = (uint64_t)(flat - pair * cs.count);
        it = pair / a.nrx;
        ir = pair - it * a.nrx;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint64_t q = rest / (uint64_t)cs.pw[j];
            rest -= q * (uint64_t)cs.pw[j];
            id[j] = (int32_t)q * cs.id_scale;
        }
    } else if (cs.ragged) {
        ragged_decode<K>(cs, a.nrx, flat, it, ir, id);
    } else {
        const int64_t pair = flat / cs.count;
        const int64_t row = flat - pair * cs.count;
        it = pair / a.nrx;
        ir = pair - it * a.nrx;
        load_candidate<K>(cs, row, id);
    }
    bool ok = (flat >= 0) && (it < a.ntx);
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const bool in = (id[j] >= 0) && ((int64_t)id[j] < a.T);
        ok = ok && in;
        const int64_t s = in ? id[j] : 0;
        p[j] = ld3(a.tri_verts + 9 * s);
        n[j] = ld3(a.normals + 3 * s);
    }
    if (!ok) it = 0;
    full[0] = ld3(a.tx + 3 * it);
    full[K + 1] = ld3(a.rx + 3 * ir);
    if constexpr (K > 0) {
        V3 path[KA<K>::n];
        image_chain<KA<K>::n>(full[0], full[K + 1], p, n, path);
#pragma unroll
        for (int j = 0; j < K; ++j) full[j + 1] = path[j];
    }
    return ok;
}

__device__ __forceinline__ void atomic_add3(float *p, V3 v) {
    atomicAdd(p + 0, v.x);
    atomicAdd(p + 1, v.y);
    atomicAdd(p + 2, v.z);
}

// Reverse of "mirror point = v0 of triangle `tri`, mirror normal = normalize((v1-v0) x (v2-v1))"
// (_mesh.py:950-956): scatters the cotangents (p_bar, n_bar) of one mirror into the mesh vertices.
// the three (vertex index, cotangent) contributions of one mirror
__device__ __forceinline__ void mirror_vjp_contrib(const float *__restrict__ mesh_vertices,
                                                   const int32_t *__restrict__ mesh_triangles, int64_t tri, V3 p_bar,
                                                   V3 n_bar, int32_t (&idx)[3], V3 (&vec)[3]) {
    const int32_t i0 = mesh_triangles[3 * tri], i1 = mesh_triangles[3 * tri + 1],
                  i2 = mesh_triangles[3 * tri + 2];
    const V3 v0 = ld3(mesh_vertices + 3 * (int64_t)i0), v1 = ld3(mesh_vertices + 3 * (int64_t)i1),
             v2 = ld3(mesh_vertices + 3 * (int64_t)i2);
    const V3 ea = v1 - v0, eb = v2 - v1;
    const V3 c = cross(ea, eb);
    const float len = __builtin_sqrtf(dot(c, c));
    V3 cbar;
    if (len == 0.0f) {
        cbar = n_bar;  // normalize divides by 1 for zero-length vectors
    } else {
        const float inv = 1.0f / len;
        const float proj = dot(n_bar, c) * inv * inv * inv;
        cbar = n_bar * inv - c * proj;
    }
    const V3 ea_bar = cross(eb, cbar);  // c = ea x eb
    const V3 eb_bar = cross(cbar, ea);
    idx[0] = i0; idx[1] = i1; idx[2] = i2;
    vec[0] = p_bar - ea_bar;
    vec[1] = ea_bar - eb_bar;
    vec[2] = eb_bar;
}

__device__ __forceinline__ void mirror_vjp_to_mesh(const float *__restrict__ mesh_vertices,
                                                   const int32_t *__restrict__ mesh_triangles,
                                                   int64_t tri, V3 p_bar, V3 n_bar,
                                                   float *__restrict__ g_vertices) {
    int32_t idx[3];
    V3 vec[3];
    mirror_vjp_contrib(mesh_vertices, mesh_triangles, tri, p_bar, n_bar, idx, vec);
#pragma unroll
    for (int k = 0; k < 3; ++k) atomic_add3(g_vertices + 3 * (int64_t)idx[k], vec[k]);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int32_t cand_fits_32(const CandSrc &s, int order) {
    const uint64_t lim = 0xffffffffull;
    if ((uint64_t)s.rank_lo + (uint64_t)s.count > lim) return 0;
    for (int j = 0; j < order; ++j)
        if ((uint64_t)s.pw[j] > lim) return 0;
    return 1;
}

static int32_t make_cand_src(const drt_candidates *c, int32_t id_scale, CandSrc *out) {
    DRT_REQUIRE(c, "candidates is null");
    DRT_REQUIRE(c->order >= 0 && c->order <= DRT_MAX_ORDER, "order %d out of range [0, %d]",
                (int)c->order, DRT_MAX_ORDER);
    DRT_REQUIRE(c->num_candidates >= 0, "negative candidate count");
    CandSrc s{};
    s.table = c->table;
    s.count = c->num_candidates;
    s.rank_lo = c->rank_lo;
    s.num_nodes = c->num_nodes;
    s.node_map = c->node_map;
    s.id_scale = id_scale;
    for (int j = 0; j < DRT_MAX_ORDER; ++j) s.pw[j] = 1;
    if (!c->table && (c->reserved & DRT_CAND_PACKED_KEYS)) {
        DRT_REQUIRE(c->order >= 1 && c->num_nodes >= 1, "packed keys need order >= 1 and num_nodes >= 1");
        unsigned __int128 pw = 1;
        for (int j = c->order - 1; j >= 0; --j) {
            DRT_REQUIRE(pw < ((unsigned __int128)1 << 62), "packed key space too large");
            s.pw[j] = (int64_t)pw;
            pw *= (unsigned __int128)c->num_nodes;
        }
        DRT_REQUIRE(pw < ((unsigned __int128)1 << 62), "packed key space too large");
        s.count = (int64_t)pw;
        s.packed = 1;
        s.node_map = nullptr;
        *out = s;
        return DRT_OK;
    }
    if (c->table && c->pair_offsets) {  // per-pair table: rows grouped by pair, CSR offsets
        DRT_REQUIRE(c->order >= 1, "a per-pair table needs order >= 1");
        s.ragged = 1;
        s.pair_offsets = c->pair_offsets;
        s.small = 1;
        if (c->reserved & DRT_CAND_PAIR_BLOCKS) {
            DRT_REQUIRE(id_scale == 1, "coplanar-pair blocks are rows of a TRIANGLE mesh (no assume_quads)");
            DRT_REQUIRE(c->order <= 8 && (c->num_candidates & (((int64_t)1 << c->order) - 1)) == 0,
                        "coplanar-pair blocks: the table must hold whole blocks of 2^order rows");
            s.pair_blocks = 1;
        }
        *out = s;
        return DRT_OK;
    }
    if (!c->table && c->pair_offsets) {
        DRT_REQUIRE(c->order >= 2, "ragged pair spaces need order >= 2");
        DRT_REQUIRE(c->first_offsets && c->last_offsets && c->rank_lo == 0 && c->num_nodes >= 0,
                    "bad ragged candidate space");
        DRT_REQUIRE((c->first_map && c->last_map) || c->num_candidates == 0, "null id arrays");
        s.ragged = 1;
        s.product = 1;
        s.first_map = c->first_map;
        s.last_map = c->last_map;
        s.pair_offsets = c->pair_offsets;
        s.first_off = c->first_offsets;
        s.last_off = c->last_offsets;
        s.small = c->reserved & 1;
        s.prefix_kernel = (c->reserved & 2) ? 1 : 0;
        s.mid_pw = 1;
        for (int j = 0; j < c->order - 2; ++j) s.mid_pw *= c->num_nodes;
        s.max_prefixes = c->num_first * s.mid_pw;  // ragged mode: num_first = the largest per-transmitter set
        *out = s;
        return DRT_OK;
    }
    if (!c->table && (c->first_map || c->last_map || c->num_first > 0 || c->num_last > 0)) {
        // product mode: F x N^(order-2) x L
        DRT_REQUIRE(c->order >= 2, "the pruned product space needs order >= 2 (order 1: pass the intersection as node_map)");
        DRT_REQUIRE(c->num_first >= 0 && c->num_last >= 0 && c->num_nodes >= 0 && c->rank_lo >= 0, "bad product space");
        DRT_REQUIRE((c->first_map || c->num_first == 0) && (c->last_map || c->num_last == 0), "null position map");
        s.product = 1;
        s.first_map = c->first_map;
        s.last_map = c->last_map;
        unsigned __int128 pw = 1;
        for (int j = c->order - 1; j >= 0; --j) {
            DRT_REQUIRE(pw < ((unsigned __int128)1 << 62), "candidate space too large for 64-bit ranks");
            s.pw[j] = (int64_t)(pw == 0 ? 1 : pw);
            const int64_t size = (j == 0) ? c->num_first : ((j == c->order - 1) ? c->num_last : c->num_nodes);
            pw *= (unsigned __int128)size;
        }
        DRT_REQUIRE((unsigned __int128)c->rank_lo + (unsigned __int128)c->num_candidates <= pw,
                    "rank window [%lld, %lld) exceeds the product space", (long long)c->rank_lo,
                    (long long)(c->rank_lo + c->num_candidates));
        s.small = cand_fits_32(s, c->order);
        *out = s;
        return DRT_OK;
    }
    if (!c->table && c->order > 0 && c->num_candidates > 0) {
        DRT_REQUIRE(c->num_nodes >= 1 && c->rank_lo >= 0, "bad rank window");
        // total = n * (n-1)^(order-1) must fit and contain the window
        unsigned __int128 total = (unsigned __int128)c->num_nodes;
        unsigned __int128 pw = 1;
        for (int j = c->order - 1; j >= 0; --j) {
            DRT_REQUIRE(pw < ((unsigned __int128)1 << 62), "candidate space too large for 64-bit ranks");
            s.pw[j] = (int64_t)pw;
            if (j > 0) pw *= (unsigned __int128)(c->num_nodes - 1);
        }
        total = (unsigned __int128)c->num_nodes * (unsigned __int128)s.pw[0];
        DRT_REQUIRE((unsigned __int128)c->rank_lo + (unsigned __int128)c->num_candidates <= total,
                    "rank window [%lld, %lld) exceeds the %s candidates",
                    (long long)c->rank_lo, (long long)(c->rank_lo + c->num_candidates), "available");
        for (int j = 0; j < c->order; ++j)
            if (s.pw[j] == 0) s.pw[j] = 1;  // num_nodes == 1: only order 1 has a (single) candidate
        s.small = cand_fits_32(s, c->order);
    }
    *out = s;
    return DRT_OK;
}

static TraceArgs make_args(drt_mesh_t mesh, const drt_trace_params *pr, const float *tx, int64_t ntx,
                           const float *rx, int64_t nrx) {
    TraceArgs a{};
    a.tri_verts = mesh->tri_verts;
    a.normals = mesh->normals;
    a.mask = mesh->has_mask ? mesh->mask : nullptr;
    a.T = mesh->num_triangles;
    a.T_occ = (pr && (pr->flags & DRT_TRACE_SKIP_OCCLUSION)) ? 0 : a.T;
    a.tx = tx;
    a.ntx = ntx;
    a.rx = rx;
    a.nrx = nrx;
    if (pr) {
        a.eps = pr->epsilon;
        a.thr = 1.0f - pr->hit_tol;
        a.min_len = pr->min_len;
    }
    return a;
}

#define DRT_ORDER_SWITCH(k, CALL)                                                     \
    switch (k) {                                                                      \
        case 0: CALL(0); break;                                                       \
        case 1: CALL(1); break;                                                       \
        case 2: CALL(2); break;                                                       \
        case 3: CALL(3); break;                                                       \
        case 4: CALL(4); break;                                                       \
        case 5: CALL(5); break;                                                       \
        case 6: CALL(6); break;                                                       \
        case 7: CALL(7); break;                                                       \
        case 8: CALL(8); break;                                                       \
        default: return fail(DRT_E_UNSUPPORTED, "order %d not supported", (int)(k)); \
    }

}  // namespace drt
