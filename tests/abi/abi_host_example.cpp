// A torch-free, Python-free caller of the C ABI: what a host application (or an XLA FFI handler)
// does.  Build: hipcc -O2 -I include tests/abi/abi_host_example.cpp -L differt_amd/lib -ldiffert_amd
// Prints "OK <hits> <first_index> <valid_paths>" and exits 0 when every call succeeds.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "differt_amd.h"

#define CHECK(x)                                                                 \
    do {                                                                         \
        int32_t rc_ = (x);                                                       \
        if (rc_ != DRT_OK) {                                                     \
            std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, drt_last_error());   \
            return 1;                                                            \
        }                                                                        \
    } while (0)
#define HIPCHECK(x)                                                                          \
    do {                                                                                     \
        if ((x) != hipSuccess) { std::fprintf(stderr, "hip error at %s\n", #x); return 1; }  \
    } while (0)

template <typename T>
static T *to_device(const std::vector<T> &h) {
    T *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T) + 16) != hipSuccess) std::abort();
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) std::abort();
    return d;
}

static_assert(sizeof(drt_trace_params) == 24 && sizeof(drt_trace_stats) == 40 && sizeof(drt_candidates) == 104 && sizeof(drt_em_params) == 40,
              "struct layouts the ctypes binding (differt_amd/_lib.py) relies on");

int main() {
    if (drt_abi_version() != DRT_ABI_VERSION) return 2;
    CHECK(drt_device_check());
    // a unit cube (reference Mesh.box(with_top=True): 8 vertices, 12 triangles)
    std::vector<float> V = {.5f, .5f, .5f, .5f, .5f, -.5f, -.5f, .5f, -.5f, -.5f, .5f, .5f,
                            -.5f, -.5f, -.5f, -.5f, -.5f, .5f, .5f, -.5f, -.5f, .5f, -.5f, .5f};
    std::vector<int32_t> Tr = {0, 1, 2, 0, 2, 3, 3, 2, 4, 3, 4, 5, 5, 4, 6, 5, 6, 7, 7, 6, 1, 7, 1, 0,
                               1, 4, 2, 1, 6, 4, 0, 3, 5, 0, 5, 7};
    float *dV = to_device(V);
    int32_t *dT = to_device(Tr);
    hipStream_t stream;
    HIPCHECK(hipStreamCreate(&stream));
    drt_mesh_t mesh = nullptr;
    CHECK(drt_mesh_create(dV, 8, dT, 12, nullptr, 0, stream, &mesh));

    // 1. dense Moller-Trumbore: 2 rays x 12 triangles
    std::vector<float> o = {0, 0, 3, 3, 0.1f, 0.2f}, d = {0, 0, -1, -1, 0, 0};
    float *dO = to_device(o), *dD = to_device(d);
    float *dt;
    uint8_t *dh;
    HIPCHECK(hipMalloc(&dt, 24 * 4));
    HIPCHECK(hipMalloc(&dh, 24));
    const float eps = 10.0f * 1.1920929e-7f;
    CHECK(drt_ray_intersect_triangle_dense(dO, dD, 2, drt_mesh_triangle_vertices(mesh), 12, eps, dt, dh, stream));
    std::vector<uint8_t> hit(24);
    HIPCHECK(hipMemcpyAsync(hit.data(), dh, 24, hipMemcpyDeviceToHost, stream));
    // 2. first hit (brute force + BVH must agree)
    int32_t *di, *di2;
    float *dft, *dft2;
    void *ws;
    HIPCHECK(hipMalloc(&di, 8)); HIPCHECK(hipMalloc(&di2, 8)); HIPCHECK(hipMalloc(&dft, 8)); HIPCHECK(hipMalloc(&dft2, 8));
    HIPCHECK(hipMalloc(&ws, drt_first_triangle_hit_by_ray_workspace_size(2)));
    CHECK(drt_first_triangle_hit_by_ray(dO, dD, 2, drt_mesh_triangle_vertices(mesh), 12, 0, nullptr, 0, eps, 512,
                                        di, dft, ws, 16, stream));
    CHECK(drt_mesh_first_triangle_hit_by_ray(mesh, dO, dD, 2, eps, 512, di2, dft2, stream));
    int32_t idx[2], idx2[2];
    float tt[2], tt2[2];
    HIPCHECK(hipMemcpyAsync(idx, di, 8, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipMemcpyAsync(idx2, di2, 8, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipMemcpyAsync(tt, dft, 8, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipMemcpyAsync(tt2, dft2, 8, hipMemcpyDeviceToHost, stream));
    // 3. image-method trace inside the cube: order 1, every candidate rank unranked on the GPU
    std::vector<float> tx = {0.1f, -0.2f, 0.05f}, rx = {-0.3f, 0.25f, -0.1f};
    float *dtx = to_device(tx), *drx = to_device(rx);
    drt_trace_params pr = {eps, 100.0f * 1.1920929e-7f, eps, 0, nullptr};
    drt_candidates cand = {nullptr, 12, 0, 12, nullptr, 1, 0};
    const int64_t cap = 64;
    size_t wbytes = drt_trace_compact_workspace_size(cap, cap);
    void *tws;
    int64_t *keys;
    float *pv;
    int32_t *po;
    HIPCHECK(hipMalloc(&tws, wbytes)); HIPCHECK(hipMalloc(&keys, cap * 8));
    HIPCHECK(hipMalloc(&pv, cap * 3 * 3 * 4)); HIPCHECK(hipMalloc(&po, cap * 3 * 4));
    int64_t nvalid = -1;
    CHECK(drt_trace_paths_compact(mesh, &pr, dtx, 1, drx, 1, &cand, cap, cap, keys, pv, po, &nvalid, tws, wbytes, stream));
    // 4. gradient of the traced vertices w.r.t. tx (cotangent = ones)
    std::vector<float> cot((size_t)nvalid * 9, 1.0f);
    float *dcot = to_device(cot), *gtx;
    HIPCHECK(hipMalloc(&gtx, 12));
    HIPCHECK(hipMemsetAsync(gtx, 0, 12, stream));
    CHECK(drt_trace_paths_vjp(mesh, dtx, 1, drx, 1, &cand, keys, dcot, nvalid, gtx, nullptr, nullptr, stream));
    float g[3];
    HIPCHECK(hipMemcpyAsync(g, gtx, 12, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    // 5. smoothed (soft mask) tracer, dense layout: confidences in [0,1] and their gradient w.r.t. tx
    drt_candidates cand2 = {nullptr, 12, 0, 12, nullptr, 1, 0};
    std::vector<int32_t> table(12);
    for (int i = 0; i < 12; ++i) table[i] = i;
    int32_t *dtab = to_device(table);
    cand2.table = dtab;
    float *sv, *sm, *gtx2;
    int32_t *so;
    HIPCHECK(hipMalloc(&sv, 12 * 3 * 3 * 4)); HIPCHECK(hipMalloc(&so, 12 * 3 * 4)); HIPCHECK(hipMalloc(&sm, 12 * 4));
    HIPCHECK(hipMalloc(&gtx2, 12));
    HIPCHECK(hipMemsetAsync(gtx2, 0, 12, stream));
    const float alpha = 20.0f;
    CHECK(drt_trace_paths_dense_smooth(mesh, &pr, alpha, 512, dtx, 1, drx, 1, &cand2, sv, so, sm, stream));
    std::vector<float> mcot(12, 1.0f);
    float *dmcot = to_device(mcot);
    CHECK(drt_trace_paths_dense_smooth_vjp(mesh, &pr, alpha, 512, dtx, 1, drx, 1, &cand2, nullptr, dmcot, gtx2, nullptr,
                                           nullptr, stream));
    float conf[12], g2[3];
    HIPCHECK(hipMemcpyAsync(conf, sm, 48, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipMemcpyAsync(g2, gtx2, 12, hipMemcpyDeviceToHost, stream));
    // 6. EM post-processing of those 12 paths: complex channel coefficient, power, delay, angles
    std::vector<int32_t> fmat(12, 0);
    int32_t *dfm = to_device(fmat);
    const float eta_r = 5.24f, sigma = 0.09f;
    float n_host[2];
    CHECK(drt_complex_refractive_index(&eta_r, &sigma, 1, 2.4e9, n_host));
    std::vector<float> nc = {n_host[0], n_host[1]}, thick = {-1.0f};
    float *dnc = to_device(nc), *dth = to_device(thick), *em_out;
    HIPCHECK(hipMalloc(&em_out, 12 * 10 * 4));
    drt_em_params ep = {2.4e9, 0, {0, 0, 0}, 0, {0, 0, 0}};
    CHECK(drt_paths_channel(sv, so, 12, 1, drt_mesh_normals(mesh), dfm, 12, dnc, dth, 1, &ep, em_out, em_out + 24,
                            em_out + 36, em_out + 48, em_out + 60, em_out + 72, em_out + 84, em_out + 96, em_out + 108,
                            stream));
    float delay[12], power[12];
    HIPCHECK(hipMemcpyAsync(power, em_out + 24, 48, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipMemcpyAsync(delay, em_out + 60, 48, hipMemcpyDeviceToHost, stream));
    // 7. cell ids of equal rows (TracedPaths.group_by_objects)
    std::vector<int32_t> rows = {4, 1, 0, 3, 2, 7, 0, 3, 4, 1};
    int32_t *drows = to_device(rows), *dids;
    void *gws;
    const size_t gbytes = drt_row_cell_ids_workspace_size(5);
    HIPCHECK(hipMalloc(&dids, 20)); HIPCHECK(hipMalloc(&gws, gbytes));
    CHECK(drt_row_cell_ids(drows, 5, 2, dids, gws, gbytes, stream));
    int32_t ids[5];
    HIPCHECK(hipMemcpyAsync(ids, dids, 20, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    bool ok2 = std::isfinite(g2[0]) && ids[0] == 0 && ids[1] == 1 && ids[2] == 2 && ids[3] == 1 && ids[4] == 0;
    for (int i = 0; i < 12; ++i)
        ok2 = ok2 && conf[i] >= 0.0f && conf[i] <= 1.0f && delay[i] > 0.0f && delay[i] < 1e-7f && power[i] < 0.0f;
    // 8. host-side candidate count (no GPU involved)
    uint64_t count = 0;
    int32_t ovf = 0;
    CHECK(drt_complete_graph_count(10000, 10000, 10001, 4, &count, &ovf));
    int nhit = 0;
    for (uint8_t h : hit) nhit += h;
    const bool ok = nhit >= 2 && idx[0] == idx2[0] && idx[1] == idx2[1] && tt[0] == tt2[0] && idx[0] >= 0 &&
                    std::fabs(tt[0] - 2.5f) < 1e-6f && nvalid >= 6 && std::isfinite(g[0]) &&
                    count == 10000ull * 9999ull && !ovf && ok2;
    std::printf("%s %d %d %lld\n", ok ? "OK" : "MISMATCH", nhit, idx[0], (long long)nvalid);
    CHECK(drt_mesh_destroy(mesh));
    return ok ? 0 : 3;
}
