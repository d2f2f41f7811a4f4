// Internal helpers shared by the libpvhip translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/pvhip.h"

namespace pvhip {

constexpr int kMaxStreams = 8;

struct State {
    bool        ready  = false;
    int         device = -1;
    hipStream_t stream = nullptr;                 // the CURRENT stream: every launch and copy goes here
    hipStream_t streams[kMaxStreams] = {};        // streams[0] is created by pvhip_init, the others on first select
    int         current = 0;
    bool        forked  = false;                  // a stream other than 0 has been used since the last full sync
};
State& state();

// Records a formatted message for pvhip_last_error() and returns `code`.
int fail(int code, const char* fmt, ...);

constexpr int kWave      = 64;    // CDNA wavefront
constexpr int kNumCU     = 256;   // MI355X
constexpr int kBlock     = 256;   // default workgroup: 4 waves, one per SIMD
constexpr int kMaxBlocks = kNumCU * 8;

inline int grid_for(size_t work_items, int per_block = kBlock) {
    size_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > (size_t)kMaxBlocks) b = kMaxBlocks;
    return (int)b;
}

// Division by a run-time-uniform divisor as multiply-high + shift (exact for n < 2^31).
struct FastDiv {
    unsigned mul, shift, d;
};
inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f{0u, 0u, d};
    if (d <= 1) return f;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;                       // 2^(s-1) < d <= 2^s
    const unsigned long long k = 31ull + s;
    f.mul   = (unsigned)(((1ull << k) + d - 1) / d);   // ceil(2^k / d) in [2^31, 2^32)
    f.shift = s - 1;
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
    return f.d <= 1 ? n : (__umulhi(n, f.mul) >> f.shift);
}

// (bias + alpha * sum)^beta of the LRN kernels.  beta_mode: 1 -> d^0.75 as sqrt(d)*sqrt(sqrt(d)) (two correctly
// rounded roots), 2 -> d^0.5, 3 -> d, 0 -> powf.
__device__ __forceinline__ float lrn_pow_f(float d, float beta, int beta_mode) {
    if (beta_mode == 1) {
        const float s = sqrtf(d);
        return s * sqrtf(s);
    }
    if (beta_mode == 2) return sqrtf(d);
    if (beta_mode == 3) return d;
    return powf(d, beta);
}

// x / (bias + alpha * sum)^beta.  Mode 4 is d^-0.75 = rsq(d) * rsq(sqrt(d)) on the hardware's 1-ulp square-root
// instructions (3 transcendental issues instead of two IEEE square roots and an IEEE division, ~1/3 of the VALU work of
// an LRN element; a few ulp from the exact quotient, far inside the 1e-4 of the path).  It is chosen on the host only
// when bias keeps d in the normal range, where those instructions are accurate.
__device__ __forceinline__ float lrn_div(float x, float d, float beta, int beta_mode) {
    if (beta_mode == 4) return x * (__builtin_amdgcn_rsqf(d) * __builtin_amdgcn_rsqf(__builtin_amdgcn_sqrtf(d)));
    return x / lrn_pow_f(d, beta, beta_mode);
}

inline int lrn_beta_mode(float beta, float bias) {
    if (beta == 0.75f) return bias >= 1e-20f ? 4 : 1;
    if (beta == 0.5f) return 2;
    if (beta == 1.0f) return 3;
    return 0;
}

}  // namespace pvhip

#define PVHIP_REQUIRE_INIT()                                                          \
    do {                                                                              \
        if (!pvhip::state().ready)                                                    \
            return pvhip::fail(PVHIP_ENOTINIT, "%s: pvhip_init() not called", __func__); \
    } while (0)

#define PVHIP_HIP(call)                                                               \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess)                                                         \
            return pvhip::fail(PVHIP_EHIP, "%s: %s -> %s", __func__, #call,           \
                               hipGetErrorString(e_));                                \
    } while (0)

#define PVHIP_CHECK_ARG(cond)                                                         \
    do {                                                                              \
        if (!(cond))                                                                  \
            return pvhip::fail(PVHIP_EINVAL, "%s: argument check failed: %s",         \
                               __func__, #cond);                                      \
    } while (0)

// After a kernel launch: surface launch-configuration errors immediately.
#define PVHIP_LAUNCH_CHECK()                                                          \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess)                                                         \
            return pvhip::fail(PVHIP_EHIP, "%s: kernel launch -> %s", __func__,       \
                               hipGetErrorString(e_));                                \
    } while (0)
