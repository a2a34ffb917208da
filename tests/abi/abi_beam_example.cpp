// A torch-free, Python-free caller of the beam-pruned tracer: what a DiffeRT host (an XLA FFI handler) does
// with drt_trace_paths_beam + drt_trace_paths_vjp.
//   abi_beam_example <scene.bin> <out.bin> [kappa]
// scene.bin: int64 {Nv, T, quads, ntx, nrx, order, max_paths}, then V f32[Nv,3], Tr i32[T,3], tx f32[ntx,3],
//            rx f32[nrx,3]
// out.bin  : int64 {nvalid, rows, levels[0..2], grazing}, keys i64[nvalid], objects i32[nvalid,order+2],
//            vertices f32[nvalid,order+2,3], grad_tx f32[ntx,3] (cotangent = ones)
// Build: hipcc -O2 -I include tests/abi/abi_beam_example.cpp -L differt_amd/lib -ldiffert_amd
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "differt_amd.h"

#define CHECK(x)                                                               \
    do {                                                                       \
        int32_t rc_ = (x);                                                     \
        if (rc_ != DRT_OK) {                                                   \
            std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, drt_last_error()); \
            return 1;                                                          \
        }                                                                      \
    } while (0)
#define HIPCHECK(x)                                                                         \
    do {                                                                                    \
        if ((x) != hipSuccess) { std::fprintf(stderr, "hip error at %s\n", #x); return 1; } \
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

static_assert(sizeof(drt_beam_params) == 72 && sizeof(drt_beam_stats) == 96, "layouts the ctypes binding relies on");

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int64_t hd[7];
    if (std::fread(hd, 8, 7, f) != 7) return 2;
    const int64_t Nv = hd[0], T = hd[1], quads = hd[2], ntx = hd[3], nrx = hd[4], order = hd[5], max_paths = hd[6];
    std::vector<float> V, tx, rx;
    std::vector<int32_t> Tr;
    if (!read_vec(f, V, Nv * 3) || !read_vec(f, Tr, T * 3) || !read_vec(f, tx, ntx * 3) || !read_vec(f, rx, nrx * 3)) return 2;
    std::fclose(f);
    CHECK(drt_device_check());
    hipStream_t stream;
    HIPCHECK(hipStreamCreate(&stream));
    float *dV = to_device(V), *dtx = to_device(tx), *drx = to_device(rx);
    int32_t *dT = to_device(Tr);
    drt_mesh_t mesh = nullptr;
    CHECK(drt_mesh_create(dV, Nv, dT, T, nullptr, (int32_t)quads, stream, &mesh));
    CHECK(drt_mesh_build_bvh(mesh, stream));
    const float e = 1.1920929e-7f;
    drt_trace_params pr = {10.0f * e, 100.0f * e, 10.0f * e, DRT_TRACE_USE_BVH, nullptr};
    drt_beam_stats st;
    drt_beam_params bp = {};
    bp.kappa = argc > 3 ? (float)std::atof(argv[3]) : 0.0f;  // 0: the library's default
    bp.stats = &st;
    const int64_t nprim = T / (quads ? 2 : 1);
    const size_t wbytes = drt_trace_beam_workspace_size(ntx, nrx, nprim, (int32_t)order, &bp, max_paths);
    void *ws;
    int64_t *keys;
    float *pv, *gtx;
    int32_t *po;
    const int64_t k2 = order + 2;
    HIPCHECK(hipMalloc(&ws, wbytes));
    HIPCHECK(hipMalloc(&keys, max_paths * 8 + 16));
    HIPCHECK(hipMalloc(&pv, max_paths * k2 * 12 + 16));
    HIPCHECK(hipMalloc(&po, max_paths * k2 * 4 + 16));
    HIPCHECK(hipMalloc(&gtx, ntx * 12 + 16));
    int64_t nvalid = -1;
    CHECK(drt_trace_paths_beam(mesh, &pr, &bp, dtx, ntx, drx, nrx, (int32_t)order, max_paths, keys, pv, po, &nvalid, ws,
                               wbytes, stream));
    // gradient of sum(vertices) w.r.t. the transmitters through the self-describing keys
    HIPCHECK(hipMemsetAsync(gtx, 0, ntx * 12, stream));
    std::vector<float> cot((size_t)nvalid * k2 * 3, 1.0f);
    float *dcot = to_device(cot);
    drt_candidates cand = {};
    cand.order = (int32_t)order;
    if (order == 0) {
        cand.num_candidates = 1;
        cand.num_nodes = nprim > 0 ? nprim : 1;
    } else {
        cand.num_nodes = nprim;
        cand.reserved = DRT_CAND_PACKED_KEYS;
    }
    CHECK(drt_trace_paths_vjp(mesh, dtx, ntx, drx, nrx, &cand, keys, dcot, nvalid, gtx, nullptr, nullptr, stream));
    std::vector<int64_t> hk((size_t)nvalid);
    std::vector<int32_t> ho((size_t)nvalid * k2);
    std::vector<float> hv((size_t)nvalid * k2 * 3), hg((size_t)ntx * 3);
    if (nvalid) {
        HIPCHECK(hipMemcpyAsync(hk.data(), keys, hk.size() * 8, hipMemcpyDeviceToHost, stream));
        HIPCHECK(hipMemcpyAsync(ho.data(), po, ho.size() * 4, hipMemcpyDeviceToHost, stream));
        HIPCHECK(hipMemcpyAsync(hv.data(), pv, hv.size() * 4, hipMemcpyDeviceToHost, stream));
    }
    HIPCHECK(hipMemcpyAsync(hg.data(), gtx, hg.size() * 4, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    FILE *o = std::fopen(argv[2], "wb");
    if (!o) return 2;
    const int64_t head[6] = {nvalid, st.rows, st.levels[0], st.levels[1], st.levels[2], st.grazing_prefixes};
    std::fwrite(head, 8, 6, o);
    std::fwrite(hk.data(), 8, hk.size(), o);
    std::fwrite(ho.data(), 4, ho.size(), o);
    std::fwrite(hv.data(), 4, hv.size(), o);
    std::fwrite(hg.data(), 4, hg.size(), o);
    std::fclose(o);
    std::printf("OK %lld valid paths, %lld rows, unit %g m\n", (long long)nvalid, (long long)st.rows, (double)st.unit_m);
    CHECK(drt_mesh_destroy(mesh));
    return 0;
}
