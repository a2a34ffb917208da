// core.hip -- library-level entry points: ABI version, thread-local error text, device probe.
#include "common.hpp"

namespace drt {
std::string &last_error_ref() {
    static thread_local std::string msg;
    return msg;
}

// Byte fill as a KERNEL.  hipMemsetAsync becomes a memset node under stream capture, and on ROCm 7.2
// a graph that holds such nodes replays correctly once and then fills with garbage (observed:
// tests/test_hipgraph_gpu.py, second replay of the trace graph); every initialisation that can sit
// inside a captured call therefore goes through this kernel.
__global__ __launch_bounds__(256) void fill_bytes_kernel(uint8_t *__restrict__ p, uint32_t word, size_t n) {
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;
    const size_t h = head < n ? head : n;
    const size_t nvec = (n - h) / 16;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    uint4 *v = reinterpret_cast<uint4 *>(p + h);
    const uint4 w = {word, word, word, word};
    for (size_t i = tid; i < nvec; i += step) v[i] = w;
    const size_t tail0 = h + nvec * 16;
    if (tid < h) p[tid] = (uint8_t)word;
    if (tid < n - tail0) p[tail0 + tid] = (uint8_t)word;
}

hipError_t fill_bytes_async(void *p, int byte, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t b = (uint32_t)(byte & 0xff), word = b | (b << 8) | (b << 16) | (b << 24);
    size_t blocks = (n / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                       reinterpret_cast<uint8_t *>(p), word, n);
    return hipGetLastError();
}
}  // namespace drt

extern "C" {

// error shim for the host-only translation units (enumerate.cpp); not part of the public ABI
int32_t drt_internal_set_error(int32_t code, const char *msg) { return drt::fail(code, "%s", msg); }

int32_t drt_abi_version(void) { return DRT_ABI_VERSION; }

const char *drt_last_error(void) { return drt::last_error_ref().c_str(); }

int32_t drt_device_check(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return drt::fail(DRT_E_NO_DEVICE, "no HIP device visible (%s)",
                         e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return drt::fail(DRT_E_NO_DEVICE, "cannot query the current HIP device");
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return drt::fail(DRT_E_NO_DEVICE, "device is %s, this library is built for gfx950 only",
                         prop.gcnArchName);
    return DRT_OK;
}

}  // extern "C"
