// Shared host-side helpers for the tsc C-ABI library (error plumbing only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

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

// A create function's handle until it is handed out: an error return on the way destroys it (and the device buffers it owns);
// the error text stays in the thread's buffer (the destroy functions do not write it).
template <class H, int (*Destroy)(H *)>
struct CreateGuard {
    H *h;
    explicit CreateGuard(H *h_) : h(h_) {}
    CreateGuard(const CreateGuard &) = delete;
    CreateGuard &operator=(const CreateGuard &) = delete;
    ~CreateGuard() { if (h) (void)Destroy(h); }
    H *release() { H *t = h; h = nullptr; return t; }
};

template <typename T>
inline hipError_t upload(T **dst, const T *src, size_t count) {
    hipError_t e = hipMalloc((void **)dst, sizeof(T) * (count ? count : 1));
    if (e != hipSuccess) return e;
    if (count) e = hipMemcpy(*dst, src, sizeof(T) * count, hipMemcpyHostToDevice);
    return e;
}


// ---- per-kernel timing with HIP events (bench.py's live roofline figure) -------------------------
// Off by default.  When on, launch sites bracket their kernel with two events on the launch stream;
// tsc_profile_read() synchronises and folds the elapsed times per kernel id.  An event pair costs a few
// microseconds of serialisation between dependent kernels (5 % of the rollout when every launch is bracketed), so
// the per-control-step kernels are SAMPLED: every `stride`-th launch of a kernel id is timed, all are counted, and
// the reported total is average x launches.
enum KernelId : int {
    KID_ENV_STEP = 0, KID_FC_GEMM, KID_ZX_GEMM, KID_LSTM_FWD, KID_HEAD_FWD, KID_SAMPLE, KID_ADD_TRANS,
    KID_RETURNS, KID_HEAD_BWD, KID_LSTM_BWD, KID_DWO_GEMM, KID_DWH_GEMM, KID_DWX_GEMM, KID_DX1_GEMM,
    KID_DW1_GEMM, KID_GRADNORM, KID_RMSPROP, KID_TRANSPOSE, KID_FINGERPRINT, KID_FUSED_FWD,
    KID_IQL_ACT, KID_IQL_GRAD, KID_IQL_REDUCE, KID_IQL_SAMPLE, KID_IQL_ADD, KID_IQL_ADAM, KID_COUNT
};

struct ProfState {
    bool on = false;
    unsigned long long only = ~0ull;              // bit k: kernel id k is timed (tsc_profile_select); the others are only counted
    int stride = 1;                               // time every stride-th launch of the per-control-step kernels
    long long seq[KID_COUNT] = {0};               // all launches
    struct Rec { int id; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    double total_ms[KID_COUNT] = {0};
    long long count[KID_COUNT] = {0};
};
ProfState &prof();          // defined in tsc_env.hip

struct ProfScope {
    int id; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(int id_, hipStream_t st_) : id(id_), st(st_) {
        ProfState &p = prof();
        if (!p.on) return;
        const bool per_step = id == KID_ENV_STEP || id == KID_FUSED_FWD || id == KID_ADD_TRANS || id == KID_FINGERPRINT ||
                              id == KID_SAMPLE || id == KID_IQL_ACT || id == KID_IQL_ADD;
        if (p.seq[id]++ % (per_step ? p.stride : 1) != 0) return;
        if (!((p.only >> id) & 1ull)) return;
        auto get = [&]() { hipEvent_t e; if (!p.pool.empty()) { e = p.pool.back(); p.pool.pop_back(); } else { (void)hipEventCreate(&e); } return e; };
        a = get(); b = get();
        (void)hipEventRecord(a, st);
    }
    void stop() {
        if (!a) return;
        (void)hipEventRecord(b, st);
        prof().recs.push_back({id, a, b});
        a = nullptr;
    }
    ~ProfScope() { stop(); }
};

}  // namespace tsc

// base + 32-bit byte offset: with a workgroup-uniform base the compiler uses the scalar-base addressing mode
// (one VGPR per address instead of a 64-bit pair that tends to be spilled when many addresses are live)
template <class T> __device__ __forceinline__ T ldg(const T *base, unsigned byte_off) {
    return *(const T *)((const char *)base + byte_off);
}
template <class T> __device__ __forceinline__ void stg(T *base, unsigned byte_off, T v) {
    *(T *)((char *)base + byte_off) = v;
}

