// BASELINE configs[4] END TO END behind the C ABI -- no torch, no Python, no MPI in the process: the prefix-sharded pruned
// search (the Python form is differt_amd/distributed.py: trace_beam_pruned_sharded) with its whole
// epilogue on the library's own RCCL communicator.
//
//   abi_beam_sharded <rank> <world> <id-file> <scene.bin> [device]
//
// scene.bin: int64 {Nv, T, quads, ntx, nrx, order, max_paths}, then V f32[Nv,3], Tr i32[T,3], tx f32[ntx,3], rx f32[nrx,3]
// (the format of abi_beam_example.cpp).  Every rank:
//   1. drt_trace_paths_beam with prefix_shard = (rank, world): the level-1 prefixes (tx, m) with (tx n + m) % world == rank
//      -- every valid path has exactly one, so the shards partition the result; no collective during the search;
//   2. drt_allgather_bytes of the 8-byte path counts (their maximum sizes the record buffer: the one host read);
//   3. drt_allgather_bytes of the padded packed records  key i64 | vertices f32[(k+2) 3] | objects i32[k+2];
//   4. key sort of the union (= the single-GPU masked_vertices order on every rank);
//   5. drt_trace_paths_vjp of its OWN paths (cotangent = ones) and drt_allreduce_sum_f32 of grad(TX).
// Rank 0 then runs the UNSHARDED call and compares: keys, objects, vertex bits equal; grad(TX) within 1e-5 of its largest
// entry (float sums in another order).  Prints "OK <paths> <paths of this rank>".
//
// Runs as a world of one on 1-GPU boxes (RCCL still carries both collectives) and with one rank per GPU where several are
// visible (RCCL refuses two ranks on one device).  tests/test_abi_native_caller.py compiles it on the CPU and runs it.
// Build: hipcc -O2 -std=c++17 -I include tests/abi/abi_beam_sharded.cpp -L differt_amd/lib -ldiffert_amd
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "differt_amd.h"

static int g_rank = 0;
#define CHECK(x)                                                                                  \
    do {                                                                                          \
        int32_t rc_ = (x);                                                                        \
        if (rc_ != DRT_OK) {                                                                      \
            std::fprintf(stderr, "[rank %d] %s -> %d: %s\n", g_rank, #x, rc_, drt_last_error()); \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)
#define HIPCHECK(x)                                                                                           \
    do {                                                                                                      \
        if ((x) != hipSuccess) { std::fprintf(stderr, "[rank %d] hip error at %s\n", g_rank, #x); return 1; } \
    } while (0)

template <typename T>
static bool read_vec(FILE *f, std::vector<T> &v, size_t n) {
    v.resize(n);
    return n == 0 || std::fread(v.data(), sizeof(T), n, f) == n;
}
template <typename T>
static T *to_device(const std::vector<T> &h) {
    T *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T) + 16) != hipSuccess) std::abort();
    if (!h.empty() && hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) std::abort();
    return d;
}

struct Paths {
    std::vector<int64_t> keys;
    std::vector<float> verts;
    std::vector<int32_t> objs;
};

int main(int argc, char **argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s <rank> <world> <id-file> <scene.bin> [device]\n", argv[0]);
        return 2;
    }
    const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
    const std::string id_file = argv[3];
    g_rank = rank;
    int ndev = 0;
    HIPCHECK(hipGetDeviceCount(&ndev));
    HIPCHECK(hipSetDevice((argc > 5) ? std::atoi(argv[5]) : rank % (ndev > 0 ? ndev : 1)));
    if (drt_abi_version() != DRT_ABI_VERSION) return 2;
    CHECK(drt_device_check());

    FILE *f = std::fopen(argv[4], "rb");
    if (!f) return 2;
    int64_t hd[7];
    if (std::fread(hd, 8, 7, f) != 7) return 2;
    const int64_t Nv = hd[0], T = hd[1], quads = hd[2], ntx = hd[3], nrx = hd[4], order = hd[5], max_paths = hd[6];
    std::vector<float> V, tx, rx;
    std::vector<int32_t> Tr;
    if (!read_vec(f, V, Nv * 3) || !read_vec(f, Tr, T * 3) || !read_vec(f, tx, ntx * 3) || !read_vec(f, rx, nrx * 3)) return 2;
    std::fclose(f);

    // ---- rendezvous: the 128-byte RCCL id through a file ----
    uint8_t id[DRT_COMM_ID_BYTES];
    if (rank == 0) {
        CHECK(drt_comm_unique_id(id));
        const std::string tmp = id_file + ".tmp";
        FILE *o = std::fopen(tmp.c_str(), "wb");
        if (!o || std::fwrite(id, 1, sizeof(id), o) != sizeof(id)) return 3;
        std::fclose(o);
        if (std::rename(tmp.c_str(), id_file.c_str()) != 0) return 3;
    } else {
        bool got = false;
        for (int tries = 0; tries < 600 && !got; ++tries) {
            if (FILE *i = std::fopen(id_file.c_str(), "rb")) {
                got = std::fread(id, 1, sizeof(id), i) == sizeof(id);
                std::fclose(i);
            }
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        if (!got) return 3;
    }
    drt_comm_t comm = nullptr;
    CHECK(drt_comm_init(id, rank, world, &comm));

    hipStream_t stream;
    HIPCHECK(hipStreamCreate(&stream));
    float *dV = to_device(V), *dtx = to_device(tx), *drx = to_device(rx);
    int32_t *dT = to_device(Tr);
    drt_mesh_t mesh = nullptr;
    CHECK(drt_mesh_create(dV, Nv, dT, T, nullptr, (int32_t)quads, stream, &mesh));  // the mesh is REPLICATED (10 MB at 200k triangles)
    CHECK(drt_mesh_build_bvh(mesh, stream));
    const float e = 1.1920929e-7f;
    drt_trace_params pr = {10.0f * e, 100.0f * e, 10.0f * e, DRT_TRACE_USE_BVH, nullptr};
    const int64_t nprim = T / (quads ? 2 : 1), k2 = order + 2;
    const int64_t rec_bytes = 8 + 12 * k2 + 4 * k2;

    // one pruned search (a shard, or everything) -> host copies + device keys for the VJP
    int64_t *keys = nullptr;
    float *pv = nullptr;
    int32_t *po = nullptr;
    HIPCHECK(hipMalloc(&keys, max_paths * 8 + 16));
    HIPCHECK(hipMalloc(&pv, max_paths * k2 * 12 + 16));
    HIPCHECK(hipMalloc(&po, max_paths * k2 * 4 + 16));
    auto search = [&](int64_t srank, int64_t sworld, Paths &out, int64_t &nvalid) -> int {
        drt_beam_params bp = {};
        bp.shard_rank = srank;
        bp.shard_world = sworld;
        const size_t wbytes = drt_trace_beam_workspace_size(ntx, nrx, nprim, (int32_t)order, &bp, max_paths);
        void *ws = nullptr;
        HIPCHECK(hipMalloc(&ws, wbytes));
        nvalid = -1;
        CHECK(drt_trace_paths_beam(mesh, &pr, &bp, dtx, ntx, drx, nrx, (int32_t)order, max_paths, keys, pv, po, &nvalid, ws, wbytes,
                                   stream));
        out.keys.resize((size_t)nvalid);
        out.verts.resize((size_t)nvalid * k2 * 3);
        out.objs.resize((size_t)nvalid * k2);
        if (nvalid) {
            HIPCHECK(hipMemcpyAsync(out.keys.data(), keys, out.keys.size() * 8, hipMemcpyDeviceToHost, stream));
            HIPCHECK(hipMemcpyAsync(out.verts.data(), pv, out.verts.size() * 4, hipMemcpyDeviceToHost, stream));
            HIPCHECK(hipMemcpyAsync(out.objs.data(), po, out.objs.size() * 4, hipMemcpyDeviceToHost, stream));
        }
        HIPCHECK(hipStreamSynchronize(stream));
        HIPCHECK(hipFree(ws));
        return 0;
    };
    auto grad_tx = [&](int64_t nvalid, float *gtx) -> int {  // d sum(vertices) / d tx of the paths whose keys sit in `keys`
        HIPCHECK(hipMemsetAsync(gtx, 0, ntx * 12, stream));
        std::vector<float> cot((size_t)nvalid * k2 * 3, 1.0f);
        float *dcot = to_device(cot);
        drt_candidates cand = {};
        cand.order = (int32_t)order;
        cand.num_nodes = nprim > 0 ? nprim : 1;
        if (order == 0) cand.num_candidates = 1;
        else cand.reserved = DRT_CAND_PACKED_KEYS;
        CHECK(drt_trace_paths_vjp(mesh, dtx, ntx, drx, nrx, &cand, keys, dcot, nvalid, gtx, nullptr, nullptr, stream));
        HIPCHECK(hipStreamSynchronize(stream));
        HIPCHECK(hipFree(dcot));
        return 0;
    };

    // ---- 1. this rank's shard ----
    Paths mine;
    int64_t nmine = 0;
    if (search(rank, world, mine, nmine)) return 1;
    float *gtx = nullptr;
    HIPCHECK(hipMalloc(&gtx, ntx * 12 + 16));
    if (grad_tx(nmine, gtx)) return 1;  // (keys still holds this rank's paths)

    // ---- 2. counts ----
    int64_t *d_count = nullptr, *d_counts = nullptr;
    HIPCHECK(hipMalloc(&d_count, 8));
    HIPCHECK(hipMalloc(&d_counts, 8 * world));
    HIPCHECK(hipMemcpyAsync(d_count, &nmine, 8, hipMemcpyHostToDevice, stream));
    CHECK(drt_allgather_bytes(comm, d_count, d_counts, 8, stream));
    std::vector<int64_t> counts((size_t)world);
    HIPCHECK(hipMemcpyAsync(counts.data(), d_counts, 8 * world, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    const int64_t cmax = std::max<int64_t>(1, *std::max_element(counts.begin(), counts.end()));
    const int64_t total = std::accumulate(counts.begin(), counts.end(), (int64_t)0);

    // ---- 3. padded packed records ----
    std::vector<uint8_t> block((size_t)(cmax * rec_bytes), 0);
    for (int64_t i = 0; i < nmine; ++i) {
        uint8_t *r = block.data() + i * rec_bytes;
        std::memcpy(r, &mine.keys[i], 8);
        std::memcpy(r + 8, &mine.verts[i * k2 * 3], 12 * k2);
        std::memcpy(r + 8 + 12 * k2, &mine.objs[i * k2], 4 * k2);
    }
    uint8_t *d_block = to_device(block), *d_all = nullptr;
    HIPCHECK(hipMalloc(&d_all, (size_t)(world * cmax * rec_bytes)));
    CHECK(drt_allgather_bytes(comm, d_block, d_all, cmax * rec_bytes, stream));
    std::vector<uint8_t> all((size_t)(world * cmax * rec_bytes));
    HIPCHECK(hipMemcpyAsync(all.data(), d_all, all.size(), hipMemcpyDeviceToHost, stream));
    // ---- 5. gradients: SUM over the ranks (every path is on exactly one) ----
    CHECK(drt_allreduce_sum_f32(comm, gtx, ntx * 3, stream));
    std::vector<float> g_sharded((size_t)ntx * 3);
    HIPCHECK(hipMemcpyAsync(g_sharded.data(), gtx, g_sharded.size() * 4, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));

    // ---- 4. union, sorted by key ----
    std::vector<const uint8_t *> recs;
    recs.reserve((size_t)total);
    for (int r = 0; r < world; ++r)
        for (int64_t i = 0; i < counts[r]; ++i) recs.push_back(all.data() + ((int64_t)r * cmax + i) * rec_bytes);
    std::sort(recs.begin(), recs.end(), [](const uint8_t *a, const uint8_t *b) {
        int64_t ka, kb;
        std::memcpy(&ka, a, 8);
        std::memcpy(&kb, b, 8);
        return ka < kb;
    });

    int rc = 0;
    if (rank == 0) {  // the unsharded search: the same rows, bit for bit
        Paths full;
        int64_t nfull = 0;
        if (search(0, 1, full, nfull)) return 1;
        if (nfull != total) {
            std::fprintf(stderr, "sharded union has %lld paths, the unsharded search %lld\n", (long long)total, (long long)nfull);
            rc = 5;
        }
        for (int64_t i = 0; i < nfull && rc == 0; ++i) {
            const uint8_t *r = recs[(size_t)i];
            if (std::memcmp(r, &full.keys[i], 8) != 0 || std::memcmp(r + 8, &full.verts[i * k2 * 3], 12 * k2) != 0 ||
                std::memcmp(r + 8 + 12 * k2, &full.objs[i * k2], 4 * k2) != 0) {
                std::fprintf(stderr, "row %lld of the sorted union differs from the unsharded search\n", (long long)i);
                rc = 6;
            }
        }
        if (rc == 0) {
            float *g1 = nullptr;
            HIPCHECK(hipMalloc(&g1, ntx * 12 + 16));
            if (grad_tx(nfull, g1)) return 1;
            std::vector<float> g_full((size_t)ntx * 3);
            HIPCHECK(hipMemcpy(g_full.data(), g1, g_full.size() * 4, hipMemcpyDeviceToHost));
            float gmax = 0.0f, dmax = 0.0f;
            for (size_t i = 0; i < g_full.size(); ++i) {
                gmax = std::max(gmax, std::fabs(g_full[i]));
                dmax = std::max(dmax, std::fabs(g_full[i] - g_sharded[i]));
            }
            if (!(dmax <= 1e-5f * std::max(gmax, 1e-30f))) {
                std::fprintf(stderr, "grad(TX): sharded differs from unsharded by %g of %g\n", (double)dmax, (double)gmax);
                rc = 7;
            }
        }
    }
    CHECK(drt_comm_destroy(comm));
    CHECK(drt_mesh_destroy(mesh));
    if (rc == 0) std::printf("OK %lld %lld\n", (long long)total, (long long)nmine);
    return rc;
}
