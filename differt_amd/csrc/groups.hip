// groups.hip -- "cell ids" of equal rows: out[r] = the smallest index i with rows[i] == rows[r].
//
// Reference: geometry/_paths.py:21-38 (`_cell_ids`, a reverse lax.scan that compares every row with
// every other one, O(n^2 width)), used by TracedPaths.group_by_objects :378-421, multipath_cells
// :331-376, merge_cell_ids :41-74 and (as "first occurrence") mask_duplicate_objects :196-252.
//
// Here: merge sort of the row INDICES with a lexicographic row comparator (ties by index, so the head
// of every run of equal rows is its smallest index), head flags, an inclusive max-scan that carries
// the head position over its run, and a scatter.  O(n log n) row comparisons that stop at the first
// differing column; integer work, bit-exact by construction.
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace drt {

struct RowLess {
    const int32_t *rows;
    int32_t width;
    __device__ bool operator()(const int32_t &a, const int32_t &b) const {
        const int32_t *ra = rows + (int64_t)a * width, *rb = rows + (int64_t)b * width;
        for (int32_t j = 0; j < width; ++j) {
            const int32_t x = ra[j], y = rb[j];
            if (x != y) return x < y;
        }
        return a < b;
    }
};

__global__ __launch_bounds__(256) void iota_kernel(int32_t *p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (int32_t)i;
}

__global__ __launch_bounds__(256) void head_kernel(const int32_t *__restrict__ rows, int32_t width,
                                                   const int32_t *__restrict__ order, int64_t n,
                                                   int32_t *__restrict__ headpos) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    bool head = (p == 0);
    if (!head) {
        const int32_t *a = rows + (int64_t)order[p] * width, *b = rows + (int64_t)order[p - 1] * width;
        for (int32_t j = 0; j < width && !head; ++j) head = a[j] != b[j];
    }
    headpos[p] = head ? (int32_t)p : 0;
}

__global__ __launch_bounds__(256) void scatter_ids_kernel(const int32_t *__restrict__ order,
                                                          const int32_t *__restrict__ headpos, int64_t n,
                                                          int32_t *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p < n) out[order[p]] = order[headpos[p]];
}

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

static int32_t temp_sizes(int64_t n, size_t *sort_bytes, size_t *scan_bytes) {
    *sort_bytes = 0;
    *scan_bytes = 0;
    RowLess cmp{nullptr, 0};
    DRT_HIP(rocprim::merge_sort(nullptr, *sort_bytes, (int32_t *)nullptr, (int32_t *)nullptr, (size_t)n, cmp));
    DRT_HIP(rocprim::inclusive_scan(nullptr, *scan_bytes, (int32_t *)nullptr, (int32_t *)nullptr, (size_t)n,
                                    rocprim::maximum<int32_t>()));
    return DRT_OK;
}

}  // namespace drt

using namespace drt;

extern "C" {

size_t drt_row_cell_ids_workspace_size(int64_t num_rows) {
    if (num_rows <= 0) return 256;
    size_t a = 0, b = 0;
    if (temp_sizes(num_rows, &a, &b) != DRT_OK) return 0;
    return 3 * align256((size_t)num_rows * 4) + align256(a > b ? a : b) + 256;
}

int32_t drt_row_cell_ids(const int32_t *rows, int64_t num_rows, int32_t width, int32_t *ids_out, void *workspace,
                         size_t workspace_bytes, void *stream) {
    DRT_REQUIRE(num_rows >= 0 && width >= 0, "negative size");
    DRT_REQUIRE(num_rows < ((int64_t)1 << 31), "more than 2^31 - 1 rows");
    if (num_rows == 0) return DRT_OK;
    DRT_REQUIRE(ids_out && (rows || width == 0), "null pointer");
    const size_t need = drt_row_cell_ids_workspace_size(num_rows);
    if (!workspace || workspace_bytes < need) return fail(DRT_E_CAPACITY, "workspace too small: need %zu bytes", need);
    hipStream_t s = as_stream(stream);
    size_t sort_bytes = 0, scan_bytes = 0;
    int32_t rc = temp_sizes(num_rows, &sort_bytes, &scan_bytes);
    if (rc != DRT_OK) return rc;
    char *w = reinterpret_cast<char *>(workspace);
    const size_t col = align256((size_t)num_rows * 4);
    int32_t *iota = reinterpret_cast<int32_t *>(w);
    int32_t *order = reinterpret_cast<int32_t *>(w + col);
    int32_t *headpos = reinterpret_cast<int32_t *>(w + 2 * col);
    void *temp = w + 3 * col;
    const dim3 grid((unsigned)ceil_div(num_rows, 256));
    hipLaunchKernelGGL(iota_kernel, grid, dim3(256), 0, s, iota, num_rows);
    RowLess cmp{rows, width};
    DRT_HIP(rocprim::merge_sort(temp, sort_bytes, iota, order, (size_t)num_rows, cmp, s));
    hipLaunchKernelGGL(head_kernel, grid, dim3(256), 0, s, rows, width, order, num_rows, headpos);
    DRT_HIP(rocprim::inclusive_scan(temp, scan_bytes, headpos, headpos, (size_t)num_rows, rocprim::maximum<int32_t>(),
                                    s));
    hipLaunchKernelGGL(scatter_ids_kernel, grid, dim3(256), 0, s, order, headpos, num_rows, ids_out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

}  // extern "C"
