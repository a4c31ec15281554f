// Shared host-side helpers of libick.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/ick.h"

namespace ick {

char* err_buf();  // thread-local 512-byte buffer (ick_api.hip)

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
  return 0;
}

// hipGetLastError() is sticky per host thread: a benign failure recorded by ANOTHER library's earlier runtime call
// (e.g. a device probe during framework start-up) must not be reported as this launch's status, so clear it first.
#define ICK_LAUNCH(...)            \
  do {                             \
    (void)hipGetLastError();       \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace ick

#define ICK_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) return ick::fail(-1, __VA_ARGS__); \
  } while (0)
