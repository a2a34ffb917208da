// Two PROCESSES, two GPUs, no torch, no Python, no MPI: the triangle-block first-hit reduce of BASELINE configs[4]
//  on the library's own RCCL communicator (drt_comm_*).
//
//   abi_comm_two_rank <rank> <world> <id-file> [device]
//
// Rank 0 draws the 128-byte id (drt_comm_unique_id) and publishes it through <id-file> (written to a temporary name,
// then renamed: readers never see half a file); the others poll for it.  Every rank builds the SAME seeded scene and
// rays, keeps the triangle block [rank T / world, (rank + 1) T / world), computes packed (t, tie) keys for all rays
// against its block (drt_first_hit_keys), MIN-all-reduces them (drt_allreduce_min_u64: 8 B per ray over xGMI) and
// decodes (drt_first_hit_finalize).  Rank 0 also runs the unsharded operator (drt_first_triangle_hit_by_ray) and
// compares indices and t bit for bit.  Prints "OK <rays> <hits>" on success.
//
// Build: hipcc -O2 -std=c++17 -I include tests/abi/abi_comm_two_rank.cpp -L differt_amd/lib -ldiffert_amd
// The CPU test suite compiles it; it RUNS only where two GPUs are visible (tests/test_abi_native_caller.py).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "differt_amd.h"

#define CHECK(x)                                                                           \
    do {                                                                                   \
        int32_t rc_ = (x);                                                                 \
        if (rc_ != DRT_OK) {                                                               \
            std::fprintf(stderr, "[rank %d] %s -> %d: %s\n", g_rank, #x, rc_, drt_last_error()); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)
#define HIPCHECK(x)                                                                                       \
    do {                                                                                                  \
        if ((x) != hipSuccess) { std::fprintf(stderr, "[rank %d] hip error at %s\n", g_rank, #x); return 1; } \
    } while (0)

static int g_rank = 0;

template <typename T>
static T *to_device(const std::vector<T> &h) {
    T *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T) + 16) != hipSuccess) std::abort();
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) std::abort();
    return d;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <rank> <world> <id-file> [device]\n", argv[0]);
        return 2;
    }
    const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
    const std::string id_file = argv[3];
    g_rank = rank;
    int ndev = 0;
    HIPCHECK(hipGetDeviceCount(&ndev));
    const int dev = (argc > 4) ? std::atoi(argv[4]) : rank % (ndev > 0 ? ndev : 1);
    HIPCHECK(hipSetDevice(dev));
    if (drt_abi_version() != DRT_ABI_VERSION) return 2;
    CHECK(drt_device_check());

    // ---- rendezvous: 128 bytes through a file ----
    uint8_t id[DRT_COMM_ID_BYTES];
    if (rank == 0) {
        CHECK(drt_comm_unique_id(id));
        const std::string tmp = id_file + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id)) return 3;
        std::fclose(f);
        if (std::rename(tmp.c_str(), id_file.c_str()) != 0) return 3;
    } else {
        bool got = false;
        for (int tries = 0; tries < 600 && !got; ++tries) {
            if (FILE *f = std::fopen(id_file.c_str(), "rb")) {
                got = std::fread(id, 1, sizeof(id), f) == sizeof(id);
                std::fclose(f);
            }
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        if (!got) {
            std::fprintf(stderr, "[rank %d] no id in %s after 60 s\n", rank, id_file.c_str());
            return 3;
        }
    }
    drt_comm_t comm = nullptr;
    CHECK(drt_comm_init(id, rank, world, &comm));
    if (drt_comm_rank(comm) != rank || drt_comm_world(comm) != world) return 4;

    // ---- the same scene and rays on every rank ----
    const int64_t T = 20000, R = 1 << 16;
    std::mt19937 gen(1234);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    std::normal_distribution<float> N(0.0f, 1.0f);
    std::vector<float> tv(T * 9), o(R * 3), d(R * 3);
    for (int64_t t = 0; t < T; ++t) {
        const float c[3] = {U(gen) * 50, U(gen) * 50, U(gen) * 50};
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) tv[t * 9 + v * 3 + k] = c[k] + (v ? N(gen) * 2 : 0.0f);
    }
    for (int64_t r = 0; r < R; ++r)
        for (int k = 0; k < 3; ++k) {
            o[r * 3 + k] = U(gen) * 50;
            d[r * 3 + k] = U(gen) * 50 - o[r * 3 + k];
        }
    hipStream_t stream;
    HIPCHECK(hipStreamCreate(&stream));
    float *dtv = to_device(tv), *dO = to_device(o), *dD = to_device(d);
    const int64_t lo = rank * T / world, hi = (rank + 1) * T / world;
    uint64_t *keys = nullptr;
    int32_t *idx = nullptr;
    float *tt = nullptr;
    HIPCHECK(hipMalloc(&keys, R * 8));
    HIPCHECK(hipMalloc(&idx, R * 4));
    HIPCHECK(hipMalloc(&tt, R * 4));
    const float eps = 10.0f * 1.1920929e-7f;
    CHECK(drt_first_hit_keys(dO, dD, R, dtv + 9 * lo, hi - lo, lo, T, nullptr, eps, 512, keys, 1, stream));
    CHECK(drt_allreduce_min_u64(comm, keys, R, stream));  // the path's one exchange step
    CHECK(drt_first_hit_finalize(keys, R, T, 512, idx, tt, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    std::vector<int32_t> h_idx(R);
    std::vector<float> h_t(R);
    HIPCHECK(hipMemcpy(h_idx.data(), idx, R * 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(h_t.data(), tt, R * 4, hipMemcpyDeviceToHost));
    int64_t hits = 0;
    for (int64_t r = 0; r < R; ++r) hits += h_idx[r] >= 0;

    int rc = 0;
    if (rank == 0) {  // the unsharded operator on the whole mesh: indices and t must agree bit for bit
        int32_t *idx1 = nullptr;
        float *t1 = nullptr;
        void *ws = nullptr;
        HIPCHECK(hipMalloc(&idx1, R * 4));
        HIPCHECK(hipMalloc(&t1, R * 4));
        const size_t wsb = drt_first_triangle_hit_by_ray_workspace_size(R);
        HIPCHECK(hipMalloc(&ws, wsb));
        CHECK(drt_first_triangle_hit_by_ray(dO, dD, R, dtv, T, 0, nullptr, 0, eps, 512, idx1, t1, ws, wsb, stream));
        HIPCHECK(hipStreamSynchronize(stream));
        std::vector<int32_t> r_idx(R);
        std::vector<float> r_t(R);
        HIPCHECK(hipMemcpy(r_idx.data(), idx1, R * 4, hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(r_t.data(), t1, R * 4, hipMemcpyDeviceToHost));
        for (int64_t r = 0; r < R && rc == 0; ++r)
            if (r_idx[r] != h_idx[r] || std::memcmp(&r_t[r], &h_t[r], 4) != 0) {
                std::fprintf(stderr, "ray %lld: sharded (%d, %g) != unsharded (%d, %g)\n", (long long)r, h_idx[r], h_t[r],
                             r_idx[r], r_t[r]);
                rc = 5;
            }
    }
    CHECK(drt_comm_destroy(comm));
    if (rc == 0) std::printf("OK %lld %lld\n", (long long)R, (long long)hits);
    return rc;
}
