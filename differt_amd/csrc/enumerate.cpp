// enumerate.cpp -- host-side path-candidate enumeration (no GPU needed).
//
// Replaces the reference's Rust generator (differt-core/src/geometry/graph.rs):
//   CompleteGraph paths  graph.rs:127-491 -- here: exact count (:314-377) + O(depth) lexicographic
//                        UNRANKING of any row, so tables are filled for arbitrary rank windows and in
//                        parallel instead of by a sequential odometer (:400-470);
//   DiGraph              graph.rs:594-1120 -- adjacency lists + depth-first path iterator.
// Row order is the reference's: lexicographic (graph.rs:1488-1513, 1536-1549).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/differt_amd.h"

// common.hpp pulls in the HIP runtime; this translation unit stays host-only, so it reports errors
// through a tiny shim implemented in core.hip.
extern "C" int32_t drt_internal_set_error(int32_t code, const char *msg);
#define EFAIL(code, msg) drt_internal_set_error((code), (msg))

namespace {

typedef unsigned __int128 u128;
const u128 kSat = ((u128)1) << 126;

inline u128 sat_add(u128 a, u128 b) { return (a + b >= kSat) ? kSat : a + b; }
inline u128 sat_mul(u128 a, u128 b) {
    if (a == 0 || b == 0) return 0;
    if (a >= kSat / b) return kSat;
    return a * b;
}

bool checked_pow(uint64_t base, uint64_t exp, uint64_t *out) {
    uint64_t r = 1;
    for (uint64_t i = 0; i < exp; ++i)
        if (__builtin_mul_overflow(r, base, &r)) return false;
    *out = r;
    return true;
}

// graph.rs:314-377 -- the count as the reference computes it, including WHERE it overflows.
bool reference_count(uint64_t n, uint64_t from, uint64_t to, uint64_t depth, uint64_t *count) {
    if (depth < 2) { *count = 0; return true; }
    if (depth == 2) { *count = (from == to) ? 0 : 1; return true; }
    const uint64_t inter = depth - 2;
    const bool fin = from < n, tin = to < n;
    const uint64_t nm1 = n > 0 ? n - 1 : 0;
    if (fin && tin) {
        const uint64_t e = (from != to) ? depth - 1 : depth - 2;
        uint64_t p;
        if (!checked_pow(nm1, e, &p)) return false;
        if (e % 2 == 0) {
            if (p == 0) return false;  // checked_add_signed(-1) on 0
            p -= 1;
        } else if (__builtin_add_overflow(p, (uint64_t)1, &p)) {
            return false;
        }
        *count = (from != to) ? p / n : (p / n) * nm1;
        return true;
    }
    if (!fin && !tin) {
        uint64_t p;
        if (!checked_pow(nm1, inter > 0 ? inter - 1 : 0, &p)) return false;
        return !__builtin_mul_overflow(n, p, count);
    }
    return checked_pow(nm1, inter, count);
}

// Completion counts: A[l] = number of ways to place l more nodes when the previous node IS `to`,
// B[l] = when it is a graph node different from `to` (see the derivation in DESIGN.md).
struct Completion {
    std::vector<u128> A, B;
    uint64_t n, from, to, L;
    bool fin, tin;
    Completion(uint64_t n_, uint64_t from_, uint64_t to_, uint64_t L_)
        : A(L_ + 1), B(L_ + 1), n(n_), from(from_), to(to_), L(L_), fin(from_ < n_), tin(to_ < n_) {
        A[0] = 0;
        B[0] = 1;
        for (uint64_t l = 1; l <= L; ++l) {
            const u128 nm1 = n > 0 ? n - 1 : 0, nm2 = n > 1 ? n - 2 : 0;
            if (tin) {
                A[l] = sat_mul(nm1, B[l - 1]);
                B[l] = sat_add(A[l - 1], sat_mul(nm2, B[l - 1]));
            } else {
                A[l] = 0;
                B[l] = sat_mul(nm1, B[l - 1]);
            }
        }
    }
    // weight of choosing node c with l nodes still to place afterwards
    u128 w(uint64_t c, uint64_t l) const { return (tin && c == to) ? A[l] : B[l]; }

    // rank -> intermediate nodes m[0..L-1]; returns false if rank is out of range
    bool unrank(u128 r, uint64_t *m) const {
        uint64_t prev = from;
        for (uint64_t i = 0; i < L; ++i) {
            const uint64_t l = L - 1 - i;
            const bool prev_in = prev < n;
            const uint64_t nchoices = prev_in ? (n > 0 ? n - 1 : 0) : n;
            if (nchoices == 0) return false;
            const u128 Bc = B[l], Ac = A[l];
            // list index of `to` among the choices (ascending nodes without prev), if it is a choice
            const bool to_is_choice = tin && to != prev;
            const uint64_t ti = to_is_choice ? to - ((prev_in && to > prev) ? 1 : 0) : nchoices;
            uint64_t idx;
            const u128 before_to = sat_mul((u128)ti, Bc);
            if (r < before_to) {
                idx = (uint64_t)(r / Bc);
                r -= (u128)idx * Bc;
            } else if (to_is_choice && r < sat_add(before_to, Ac)) {
                idx = ti;
                r -= before_to;
            } else {
                if (!to_is_choice) return false;  // r >= nchoices * Bc
                r -= before_to + Ac;
                if (Bc == 0) return false;
                const u128 k = r / Bc;
                if (k >= (u128)(nchoices - ti - 1)) return false;
                idx = ti + 1 + (uint64_t)k;
                r -= k * Bc;
            }
            if (idx >= nchoices) return false;
            const uint64_t node = idx + ((prev_in && idx >= prev) ? 1 : 0);
            m[i] = node;
            prev = node;
        }
        return r == 0;
    }
};

}  // namespace

// ---------------------------------------------------------------------------------------------
struct drt_digraph {
    std::vector<std::vector<uint64_t>> edges;
};

struct drt_digraph_iter {
    const drt_digraph *g;
    uint64_t to, depth;
    bool include;
    // explicit DFS state: level i iterates the adjacency list of visited[i] from cursor[i]
    std::vector<uint64_t> visited;
    std::vector<size_t> cursor;
    bool done;
};

extern "C" {

int32_t drt_complete_graph_count(uint64_t num_nodes, uint64_t from_, uint64_t to, uint64_t depth,
                                 uint64_t *count_out, int32_t *overflow_out) {
    if (!count_out) return EFAIL(DRT_E_INVALID, "count_out is null");
    uint64_t c = 0;
    const bool ok = reference_count(num_nodes, from_, to, depth, &c);
    *count_out = ok ? c : UINT64_MAX;  // graph.rs:368-375
    if (overflow_out) *overflow_out = ok ? 0 : 1;
    return DRT_OK;
}

int32_t drt_complete_graph_count_exact(uint64_t n, uint64_t from_, uint64_t to, uint64_t depth,
                                       uint64_t *count_out, int32_t *exceeds_out) {
    if (!count_out) return EFAIL(DRT_E_INVALID, "count_out is null");
    u128 c = 0;
    if (depth == 2) {
        c = (from_ == to) ? 0 : 1;
    } else if (depth > 2) {
        const uint64_t L = depth - 2;
        const Completion comp(n, from_, to, L);
        if (from_ < n) {
            c = (to < n && from_ == to) ? comp.A[L] : comp.B[L];
        } else {  // first node: n choices, one of which may be `to`
            c = (to < n) ? sat_add(comp.A[L - 1], sat_mul(n > 0 ? n - 1 : 0, comp.B[L - 1]))
                         : sat_mul(n, comp.B[L - 1]);
        }
    }
    const bool exceeds = c > (u128)UINT64_MAX;
    *count_out = exceeds ? UINT64_MAX : (uint64_t)c;
    if (exceeds_out) *exceeds_out = exceeds ? 1 : 0;
    return DRT_OK;
}

int32_t drt_complete_graph_fill_host(uint64_t n, uint64_t from_, uint64_t to, uint64_t depth,
                                     int32_t include_from_and_to, uint64_t rank_lo,
                                     uint64_t rank_hi, uint64_t *out) {
    if (rank_hi < rank_lo) return EFAIL(DRT_E_INVALID, "rank_hi < rank_lo");
    if (rank_hi == rank_lo) return DRT_OK;
    if (!out) return EFAIL(DRT_E_INVALID, "out_host is null");
    if (depth < 2) return EFAIL(DRT_E_INVALID, "rank out of range (no path of depth < 2)");
    const uint64_t L = depth - 2;
    const uint64_t width = include_from_and_to ? depth : L;
    if (L == 0) {
        if (from_ == to || rank_lo != 0 || rank_hi != 1)
            return EFAIL(DRT_E_INVALID, "rank out of range");
        if (include_from_and_to) {
            out[0] = from_;
            out[1] = to;
        }
        return DRT_OK;
    }
    const Completion comp(n, from_, to, L);
    const int64_t rows = (int64_t)(rank_hi - rank_lo);
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t i = 0; i < rows; ++i) {
        uint64_t *row = out + (uint64_t)i * width;
        uint64_t *m = include_from_and_to ? row + 1 : row;
        if (!comp.unrank((u128)(rank_lo + (uint64_t)i), m)) bad |= 1;
        if (include_from_and_to) {
            row[0] = from_;
            row[depth - 1] = to;
        }
    }
    if (bad) return EFAIL(DRT_E_INVALID, "rank out of range");
    return DRT_OK;
}

// ---------------------------------------------------------------------------------------------
int32_t drt_digraph_from_complete_graph(uint64_t n, drt_digraph_t *out) {
    if (!out) return EFAIL(DRT_E_INVALID, "out is null");
    drt_digraph *g = new (std::nothrow) drt_digraph();
    if (!g) return EFAIL(DRT_E_INVALID, "out of memory");
    g->edges.resize(n);
    for (uint64_t i = 0; i < n; ++i) {
        g->edges[i].reserve(n ? n - 1 : 0);
        for (uint64_t j = 0; j < n; ++j)
            if (j != i) g->edges[i].push_back(j);
    }
    *out = g;
    return DRT_OK;
}

int32_t drt_digraph_from_adjacency_matrix(const uint8_t *m, uint64_t n, drt_digraph_t *out) {
    if (!out || (!m && n)) return EFAIL(DRT_E_INVALID, "null pointer");
    drt_digraph *g = new (std::nothrow) drt_digraph();
    if (!g) return EFAIL(DRT_E_INVALID, "out of memory");
    g->edges.resize(n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i)
        for (uint64_t j = 0; j < n; ++j)
            if (m[(uint64_t)i * n + j]) g->edges[i].push_back(j);
    *out = g;
    return DRT_OK;
}

int32_t drt_digraph_destroy(drt_digraph_t g) {
    delete g;
    return DRT_OK;
}

uint64_t drt_digraph_num_nodes(drt_digraph_t g) { return g ? g->edges.size() : 0; }

int32_t drt_digraph_insert_from_and_to_nodes(drt_digraph_t g, int32_t direct_path,
                                             const uint8_t *from_adj, const uint8_t *to_adj,
                                             uint64_t *from_out, uint64_t *to_out) {
    if (!g) return EFAIL(DRT_E_INVALID, "graph is null");
    const uint64_t from = g->edges.size(), to = from + 1;
    for (uint64_t i = 0; i < from; ++i)
        if (!to_adj || to_adj[i]) g->edges[i].push_back(to);
    std::vector<uint64_t> fe;
    for (uint64_t i = 0; i < from; ++i)
        if (!from_adj || from_adj[i]) fe.push_back(i);
    if (direct_path) fe.push_back(to);
    g->edges.push_back(std::move(fe));
    g->edges.emplace_back();
    if (from_out) *from_out = from;
    if (to_out) *to_out = to;
    return DRT_OK;
}

int32_t drt_digraph_filter_by_mask(drt_digraph_t g, const uint8_t *mask, uint64_t mask_len,
                                   int32_t fast_mode) {
    if (!g) return EFAIL(DRT_E_INVALID, "graph is null");
    if (mask_len > g->edges.size())
        return EFAIL(DRT_E_INVALID,
                     "'mask' length must be smaller than or equal to the number of nodes in the graph");
    for (uint64_t i = 0; i < mask_len; ++i)
        if (!mask[i]) g->edges[i].clear();
    if (!fast_mode)
        for (auto &e : g->edges)
            e.erase(std::remove_if(e.begin(), e.end(),
                                   [&](uint64_t node) { return node < mask_len && !mask[node]; }),
                    e.end());
    return DRT_OK;
}

int32_t drt_digraph_disconnect_nodes(drt_digraph_t g, const uint64_t *nodes, uint64_t n,
                                     int32_t fast_mode) {
    if (!g) return EFAIL(DRT_E_INVALID, "graph is null");
    for (uint64_t i = 0; i < n; ++i)
        if (nodes[i] >= g->edges.size()) return EFAIL(DRT_E_INVALID, "node is out-of-bounds");
    for (uint64_t i = 0; i < n; ++i) g->edges[nodes[i]].clear();
    if (!fast_mode) {
        std::vector<uint64_t> sorted(nodes, nodes + n);
        std::sort(sorted.begin(), sorted.end());
        for (auto &e : g->edges)
            e.erase(std::remove_if(e.begin(), e.end(),
                                   [&](uint64_t node) {
                                       return std::binary_search(sorted.begin(), sorted.end(), node);
                                   }),
                    e.end());
    }
    return DRT_OK;
}

int32_t drt_digraph_iter_create(drt_digraph_t g, uint64_t from_, uint64_t to, uint64_t depth,
                                int32_t include_from_and_to, drt_digraph_iter_t *out) {
    if (!g || !out) return EFAIL(DRT_E_INVALID, "null pointer");
    if (from_ >= g->edges.size()) return EFAIL(DRT_E_INVALID, "'from_' is not a node of the graph");
    drt_digraph_iter *it = new (std::nothrow) drt_digraph_iter();
    if (!it) return EFAIL(DRT_E_INVALID, "out of memory");
    it->g = g;
    it->to = to;
    it->depth = depth;
    it->include = include_from_and_to != 0;
    it->visited.push_back(from_);
    it->cursor.push_back(0);
    it->done = false;
    *out = it;
    return DRT_OK;
}

int32_t drt_digraph_iter_destroy(drt_digraph_iter_t it) {
    delete it;
    return DRT_OK;
}

int32_t drt_digraph_iter_next_chunk(drt_digraph_iter_t it, uint64_t max_rows, uint64_t *out,
                                    uint64_t *rows_out) {
    if (!it || !rows_out) return EFAIL(DRT_E_INVALID, "null pointer");
    const uint64_t width = it->include ? it->depth : (it->depth >= 2 ? it->depth - 2 : 0);
    uint64_t rows = 0;
    // depth-first walk; a path is complete when depth-1 nodes are fixed and `to` is adjacent to the
    // last one (adjacency lists are sorted, so membership is a binary search: graph.rs:1077)
    while (rows < max_rows && !it->visited.empty()) {
        const uint64_t node = it->visited.back();
        const std::vector<uint64_t> &adj = it->g->edges[node];
        if (it->visited.size() + 1 == it->depth) {
            const bool reach = std::binary_search(adj.begin(), adj.end(), it->to);
            if (reach) {
                if (out && width) {
                    uint64_t *row = out + rows * width;
                    if (it->include) {
                        std::memcpy(row, it->visited.data(), it->visited.size() * 8);
                        row[it->depth - 1] = it->to;
                    } else {
                        std::memcpy(row, it->visited.data() + 1, (it->visited.size() - 1) * 8);
                    }
                }
                ++rows;
            }
            it->visited.pop_back();
            it->cursor.pop_back();
        } else if (it->visited.size() + 1 < it->depth && it->cursor.back() < adj.size()) {
            const uint64_t child = adj[it->cursor.back()++];
            if (child >= it->g->edges.size()) continue;  // defensive: dangling edge
            it->visited.push_back(child);
            it->cursor.push_back(0);
        } else {
            it->visited.pop_back();
            it->cursor.pop_back();
        }
    }
    *rows_out = rows;
    return DRT_OK;
}

}  // extern "C"
