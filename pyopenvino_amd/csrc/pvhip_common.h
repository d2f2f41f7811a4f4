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
