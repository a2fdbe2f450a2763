// common.h -- shared host-side helpers for libmeganerf_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mnr_api.h"

namespace mnr {

// thread-local last-error string (mnr_last_error)
char *err_buf();
int set_err(int code, const char *fmt, ...);

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(MNR_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return MNR_OK;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// index of the calling thread's current device for per-device one-time set-up flags (function attributes are per device)
constexpr int MAX_DEVICES = 64;
inline int device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
    return d < 0 ? 0 : (d >= MAX_DEVICES ? MAX_DEVICES - 1 : d);
}

struct F3 { float x, y, z; };

}  // namespace mnr

#define MNR_REQUIRE(cond, ...) \
    do { if (!(cond)) return mnr::set_err(MNR_E_INVALID, __VA_ARGS__); } while (0)
