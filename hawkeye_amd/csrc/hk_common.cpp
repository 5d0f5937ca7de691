#include "hk_common.h"

namespace hk {

char *error_buffer() {
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int status, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return status;
}

}  // namespace hk

extern "C" const char *hk_last_error(void) { return hk::error_buffer(); }

extern "C" int hk_version(void) { return 1; }

extern "C" int hk_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess)
        return hk::fail(HK_ERR_HIP, "hipGetDeviceCount failed: %s",
                        hipGetErrorString(e));
    return n;
}
