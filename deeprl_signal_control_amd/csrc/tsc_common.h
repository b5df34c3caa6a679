// Shared host-side helpers for the tsc C-ABI library (error plumbing only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace tsc {

inline char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return 1;
}

#define TSC_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return tsc::fail("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
    } while (0)

template <typename T>
inline hipError_t upload(T **dst, const T *src, size_t count) {
    hipError_t e = hipMalloc((void **)dst, sizeof(T) * (count ? count : 1));
    if (e != hipSuccess) return e;
    if (count) e = hipMemcpy(*dst, src, sizeof(T) * count, hipMemcpyHostToDevice);
    return e;
}

}  // namespace tsc
