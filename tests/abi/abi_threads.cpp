// Re-entrancy of the C ABI ("safe to call from multiple host threads on distinct streams"):
// no torch, no Python.  A small city of rotated boxes is traced
//   (1) serially on one stream: dense Moller-Trumbore, the compact trace of a rank window, the beam-pruned trace;
//   (2) by TWO host threads at once, each with its OWN drt_mesh_t, stream, buffers and workspace (the handles build
//       their LBVH / primitive clusters lazily INSIDE the concurrent calls);
//   (3) by two host threads that SHARE one handle whose LBVH and clusters were built beforehand -- the only calls
//       that write to a handle (include/differt_amd.h, "THREADS").
// Every result must equal the serial one bit for bit, over several rounds.
//   abi_threads [rounds]
// Build: hipcc -O2 -pthread -I include tests/abi/abi_threads.cpp -L differt_amd/lib -ldiffert_amd
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "differt_amd.h"

namespace {

struct Scene {
    std::vector<float> V, tx, rx, rays_o, rays_d;
    std::vector<int32_t> Tr;
};

// boxes (walls + roof, 10 triangles each) rotated about z by arbitrary angles: nothing axis-aligned
Scene make_scene(int nboxes, unsigned seed) {
    Scene s;
    auto rnd = [&seed]() {
        seed = seed * 1664525u + 1013904223u;
        return (float)((seed >> 8) & 0xffffff) / 16777216.0f;
    };
    static const int tri[10][3] = {{0, 1, 5}, {0, 5, 4}, {1, 2, 6}, {1, 6, 5}, {2, 3, 7}, {2, 7, 6}, {3, 0, 4}, {3, 4, 7}, {4, 5, 6}, {4, 6, 7}};
    const int side = (int)std::ceil(std::sqrt((double)nboxes));
    for (int b = 0; b < nboxes; ++b) {
        const float cx = 30.0f * (float)(b % side) + 6.0f * rnd(), cy = 30.0f * (float)(b / side) + 6.0f * rnd();
        const float l = 8.0f + 8.0f * rnd(), w = 6.0f + 6.0f * rnd(), h = 10.0f + 25.0f * rnd(), a = 3.1f * rnd();
        const float ca = std::cos(a), sa = std::sin(a);
        const float xs[4] = {-l, l, l, -l}, ys[4] = {-w, -w, w, w};
        for (int z = 0; z < 2; ++z)
            for (int k = 0; k < 4; ++k) {
                s.V.push_back(cx + 0.5f * (ca * xs[k] - sa * ys[k]));
                s.V.push_back(cy + 0.5f * (sa * xs[k] + ca * ys[k]));
                s.V.push_back(z ? h : 0.0f);
            }
        for (auto &t : tri)
            for (int k = 0; k < 3; ++k) s.Tr.push_back(8 * b + t[k]);
    }
    const float ext = 30.0f * (float)side;
    for (int i = 0; i < 6; ++i) {
        s.tx.push_back(ext * rnd());
        s.tx.push_back(ext * rnd());
        s.tx.push_back(20.0f + 30.0f * rnd());
    }
    for (int i = 0; i < 24; ++i) {
        s.rx.push_back(ext * rnd());
        s.rx.push_back(ext * rnd());
        s.rx.push_back(1.5f);
    }
    for (int i = 0; i < 512; ++i) {
        for (int k = 0; k < 3; ++k) s.rays_o.push_back(ext * rnd());
        for (int k = 0; k < 3; ++k) s.rays_d.push_back(ext * (rnd() - 0.5f));
    }
    return s;
}

template <typename T>
T *dev_copy(const std::vector<T> &h) {
    T *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T) + 16) != hipSuccess) std::abort();
    if (!h.empty() && hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) std::abort();
    return d;
}

struct Result {
    std::vector<uint8_t> hit;
    std::vector<uint32_t> t_bits;
    std::vector<int64_t> ckeys, bkeys;
    std::vector<uint32_t> cverts, bverts;
    std::string error;
    bool operator==(const Result &o) const {
        return hit == o.hit && t_bits == o.t_bits && ckeys == o.ckeys && bkeys == o.bkeys && cverts == o.cverts && bverts == o.bverts;
    }
};

// one full pass of the three operators on `mesh` / `stream`, transmitters [tx0, tx0 + ntx)
Result run_all(drt_mesh_t mesh, hipStream_t stream, const Scene &s, int64_t tx0, int64_t ntx) {
    Result r;
#define TRY(x)                                                                                  \
    do {                                                                                        \
        const int32_t rc_ = (x);                                                                \
        if (rc_ != DRT_OK) {                                                                    \
            r.error = std::string(#x) + " -> " + std::to_string(rc_) + ": " + drt_last_error(); \
            return r;                                                                           \
        }                                                                                       \
    } while (0)
#define HIPTRY(x)                                       \
    do {                                                \
        if ((x) != hipSuccess) {                        \
            r.error = std::string("hip error: ") + #x;  \
            return r;                                   \
        }                                               \
    } while (0)
    const int64_t T = (int64_t)s.Tr.size() / 3, R = (int64_t)s.rays_o.size() / 3, nrx = (int64_t)s.rx.size() / 3;
    const float e = 1.1920929e-7f;
    float *dtx = dev_copy(std::vector<float>(s.tx.begin() + 3 * tx0, s.tx.begin() + 3 * (tx0 + ntx))), *drx = dev_copy(s.rx);
    float *dro = dev_copy(s.rays_o), *drd = dev_copy(s.rays_d);
    // dense Moller-Trumbore against the handle's own gathered triangles
    float *dt = nullptr;
    uint8_t *dh = nullptr;
    HIPTRY(hipMalloc(&dt, (size_t)R * T * 4));
    HIPTRY(hipMalloc(&dh, (size_t)R * T));
    TRY(drt_ray_intersect_triangle_dense(dro, drd, R, drt_mesh_triangle_vertices(mesh), T, 10.0f * e, dt, dh, stream));
    r.hit.resize((size_t)R * T);
    r.t_bits.resize((size_t)R * T);
    HIPTRY(hipMemcpyAsync(r.hit.data(), dh, r.hit.size(), hipMemcpyDeviceToHost, stream));
    HIPTRY(hipMemcpyAsync(r.t_bits.data(), dt, r.t_bits.size() * 4, hipMemcpyDeviceToHost, stream));
    // compact trace of all order-2 candidates, occlusion on the LBVH (built lazily by the first call on a handle)
    drt_trace_params pr = {10.0f * e, 100.0f * e, 10.0f * e, DRT_TRACE_USE_BVH, nullptr};
    TRY(drt_mesh_build_bvh(mesh, stream));  // (a no-op once built)
    const int64_t max_paths = 4096, max_surv = 1 << 20;
    int64_t *keys = nullptr;
    float *pv = nullptr;
    int32_t *po = nullptr;
    HIPTRY(hipMalloc(&keys, max_paths * 8));
    HIPTRY(hipMalloc(&pv, max_paths * 4 * 12));
    HIPTRY(hipMalloc(&po, max_paths * 4 * 4));
    {
        drt_candidates c = {};
        c.num_candidates = T * (T - 1);
        c.num_nodes = T;
        c.order = 2;
        const size_t wb = drt_trace_compact_workspace_size(max_surv, max_paths);
        void *ws = nullptr;
        HIPTRY(hipMalloc(&ws, wb));
        int64_t nv = 0;
        TRY(drt_trace_paths_compact(mesh, &pr, dtx, ntx, drx, nrx, &c, max_surv, max_paths, keys, pv, po, &nv, ws, wb, stream));
        r.ckeys.resize((size_t)nv);
        r.cverts.resize((size_t)nv * 12);
        HIPTRY(hipMemcpyAsync(r.ckeys.data(), keys, (size_t)nv * 8, hipMemcpyDeviceToHost, stream));
        HIPTRY(hipMemcpyAsync(r.cverts.data(), pv, (size_t)nv * 48, hipMemcpyDeviceToHost, stream));
        HIPTRY(hipStreamSynchronize(stream));
        HIPTRY(hipFree(ws));
    }
    {  // the beam-pruned trace of the same space (clusters built lazily by the first call on a handle)
        drt_beam_params bp = {};
        const size_t wb = drt_trace_beam_workspace_size(ntx, nrx, T, 2, &bp, max_paths);
        void *ws = nullptr;
        HIPTRY(hipMalloc(&ws, wb));
        int64_t nv = 0;
        TRY(drt_trace_paths_beam(mesh, &pr, &bp, dtx, ntx, drx, nrx, 2, max_paths, keys, pv, po, &nv, ws, wb, stream));
        r.bkeys.resize((size_t)nv);
        r.bverts.resize((size_t)nv * 12);
        HIPTRY(hipMemcpyAsync(r.bkeys.data(), keys, (size_t)nv * 8, hipMemcpyDeviceToHost, stream));
        HIPTRY(hipMemcpyAsync(r.bverts.data(), pv, (size_t)nv * 48, hipMemcpyDeviceToHost, stream));
        HIPTRY(hipStreamSynchronize(stream));
        HIPTRY(hipFree(ws));
    }
    for (void *p : {(void *)dtx, (void *)drx, (void *)dro, (void *)drd, (void *)dt, (void *)dh, (void *)keys, (void *)pv, (void *)po}) (void)hipFree(p);
#undef TRY
#undef HIPTRY
    return r;
}

drt_mesh_t make_mesh(const Scene &s, hipStream_t stream) {
    float *dV = dev_copy(s.V);
    int32_t *dT = dev_copy(s.Tr);
    drt_mesh_t m = nullptr;
    if (drt_mesh_create(dV, (int64_t)s.V.size() / 3, dT, (int64_t)s.Tr.size() / 3, nullptr, 0, stream, &m) != DRT_OK) {
        std::fprintf(stderr, "drt_mesh_create: %s\n", drt_last_error());
        std::abort();
    }
    (void)hipFree(dV);  // the handle copies vertices and triangles
    (void)hipFree(dT);
    return m;
}

}  // namespace

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 3;
    if (drt_abi_version() != DRT_ABI_VERSION || drt_device_check() != DRT_OK) {
        std::fprintf(stderr, "%s\n", drt_last_error());
        return 2;
    }
    const Scene s = make_scene(40, 12345u);  // 400 triangles: 159 600 order-2 candidates per (tx, rx) pair
    hipStream_t s0, sa, sb;
    if (hipStreamCreate(&s0) != hipSuccess || hipStreamCreate(&sa) != hipSuccess || hipStreamCreate(&sb) != hipSuccess) return 2;
    // (1) serial reference: transmitters [0, 3) and [3, 6)
    drt_mesh_t m0 = make_mesh(s, s0);
    const Result ref_a = run_all(m0, s0, s, 0, 3), ref_b = run_all(m0, s0, s, 3, 3);
    if (!ref_a.error.empty() || !ref_b.error.empty()) {
        std::fprintf(stderr, "serial: %s %s\n", ref_a.error.c_str(), ref_b.error.c_str());
        return 1;
    }
    // (the two key formats differ -- rank within the window vs packed mirror ids -- the order and the vertices do not)
    if (ref_a.cverts != ref_a.bverts || ref_b.cverts != ref_b.bverts) {
        std::fprintf(stderr, "serial: pruned and exhaustive traces differ\n");
        return 1;
    }
    long long paths = (long long)(ref_a.ckeys.size() + ref_b.ckeys.size());
    int bad = 0;
    for (int round = 0; round < rounds; ++round) {
        // (2) own handles: LBVH and clusters are built inside the concurrent calls
        drt_mesh_t ma = make_mesh(s, sa), mb = make_mesh(s, sb);
        Result ra, rb;
        {
            std::thread ta([&] { ra = run_all(ma, sa, s, 0, 3); });
            std::thread tb([&] { rb = run_all(mb, sb, s, 3, 3); });
            ta.join();
            tb.join();
        }
        if (!ra.error.empty() || !rb.error.empty() || !(ra == ref_a) || !(rb == ref_b)) {
            std::fprintf(stderr, "round %d, own handles: %s %s mismatch\n", round, ra.error.c_str(), rb.error.c_str());
            ++bad;
        }
        (void)drt_mesh_destroy(ma);
        (void)drt_mesh_destroy(mb);
        // (3) one shared handle, everything that writes to it done beforehand
        drt_mesh_t ms = make_mesh(s, s0);
        if (drt_mesh_build_bvh(ms, s0) != DRT_OK || drt_mesh_build_beam_clusters(ms, s0) != DRT_OK) return 1;
        (void)hipStreamSynchronize(s0);
        {
            std::thread ta([&] { ra = run_all(ms, sa, s, 0, 3); });
            std::thread tb([&] { rb = run_all(ms, sb, s, 3, 3); });
            ta.join();
            tb.join();
        }
        if (!ra.error.empty() || !rb.error.empty() || !(ra == ref_a) || !(rb == ref_b)) {
            std::fprintf(stderr, "round %d, shared handle: %s %s mismatch\n", round, ra.error.c_str(), rb.error.c_str());
            ++bad;
        }
        (void)drt_mesh_destroy(ms);
    }
    (void)drt_mesh_destroy(m0);
    if (bad) return 1;
    std::printf("OK %d rounds, %lld valid paths per round, two threads x two streams bit-equal to serial\n", rounds, paths);
    return 0;
}
