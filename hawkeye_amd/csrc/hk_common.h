// Shared host-side helpers of libhawkeye_hip.so: status codes, the
// thread-local error string and HIP call checking.  Nothing in here is
// visible across the C ABI except through hk_last_error().
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/hk.h"

namespace hk {

char *error_buffer();                       // thread-local, 512 bytes
int fail(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// RAII "make this device current, restore the previous one on exit".
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != device && hipSetDevice(device) != hipSuccess) ok = false;
        if (prev == device) prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace hk

#define HK_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess)                                                 \
            return hk::fail(HK_ERR_HIP, "%s failed: %s (%s:%d)", #call,       \
                            hipGetErrorString(e_), __FILE__, __LINE__);       \
    } while (0)

#define HK_LAUNCH_CHECK(name)                                                 \
    do {                                                                      \
        hipError_t e_ = hipGetLastError();                                    \
        if (e_ != hipSuccess)                                                 \
            return hk::fail(HK_ERR_HIP, "launch of %s failed: %s", name,      \
                            hipGetErrorString(e_));                           \
    } while (0)
