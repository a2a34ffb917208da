// core.hip -- library-level entry points: ABI version, thread-local error text, device probe.
#include "common.hpp"

namespace drt {
std::string &last_error_ref() {
    static thread_local std::string msg;
    return msg;
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
