// DIAGNOSTIC BUILD ONLY (make diag -> libpvhip_diag.so): a family of float4 streaming kernels for scripts/sweep_stream.py, which
// settles what a copy / ReLU stream can reach on the box the library runs on -- the ceiling the memory-bound kernels (ReLU.py:9-12,
// Add.py:9-14, MaxPool.py:41-72 replacements) are judged against besides the 8 TB/s spec (MI355X guide: 6.29 TB/s float4 copy).
// Not part of the product library.
#include "pvhip_common.h"

using namespace pvhip;

namespace {

typedef float float4v __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ float4v ld(const float4v* p) {
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__device__ __forceinline__ void st(float4v* p, float4v v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// U 16-byte loads per lane issued before the first store.
//   layout 0: grid-stride, the U loads of a lane are a whole grid apart (U far-apart streams per lane)
//   layout 1: a workgroup owns U adjacent pieces of blockDim * 16 bytes per iteration (one contiguous run of U * 4 KiB at 256 threads)
//   layout 2: a lane owns U adjacent float4 (64 * U contiguous bytes per lane; a wave-instruction strides by U * 16 bytes)
template <int U, bool NT, bool RELU>
__global__ void diag_stream_kernel(const float4v* __restrict__ x, float4v* __restrict__ y, size_t n4, int layout) {
    const size_t tid = threadIdx.x, nb = gridDim.x, bd = blockDim.x;
    const size_t per_iter = nb * bd * U;
    for (size_t base = 0; base < n4; base += per_iter) {
        float4v v[U];
        size_t  idx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (layout == 0) idx[u] = base + (size_t)u * nb * bd + blockIdx.x * bd + tid;
            else if (layout == 1) idx[u] = base + ((size_t)blockIdx.x * U + u) * bd + tid;
            else idx[u] = base + ((size_t)blockIdx.x * bd + tid) * U + u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (idx[u] < n4) v[u] = ld<NT>(x + idx[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (idx[u] < n4) {
                float4v a = v[u];
                if (RELU) {
                    a.x = a.x < 0.0f ? 0.0f : a.x; a.y = a.y < 0.0f ? 0.0f : a.y;
                    a.z = a.z < 0.0f ? 0.0f : a.z; a.w = a.w < 0.0f ? 0.0f : a.w;
                }
                st<NT>(y + idx[u], a);
            }
    }
}

template <int U, bool NT>
void launch(const float* x, float* y, size_t n4, int relu, int layout, int blocks, int threads) {
    const float4v* x4 = reinterpret_cast<const float4v*>(x);
    float4v*       y4 = reinterpret_cast<float4v*>(y);
    if (relu) hipLaunchKernelGGL((diag_stream_kernel<U, NT, true>), dim3(blocks), dim3(threads), 0, state().stream, x4, y4, n4, layout);
    else      hipLaunchKernelGGL((diag_stream_kernel<U, NT, false>), dim3(blocks), dim3(threads), 0, state().stream, x4, y4, n4, layout);
}

}  // namespace

extern "C" int pvhip_diag_stream_f32(const float* x, float* y, unsigned long long n, int relu, int unroll, int nt, int layout, int blocks,
                                     int threads) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr && n % 4 == 0 && blocks >= 1 && threads >= 64 && threads <= 1024 && threads % 64 == 0);
    PVHIP_CHECK_ARG(layout >= 0 && layout <= 2);
    const size_t n4 = (size_t)(n / 4);
#define PV_DS(U_) { if (nt) launch<U_, true>(x, y, n4, relu, layout, blocks, threads); else launch<U_, false>(x, y, n4, relu, layout, blocks, threads); }
    switch (unroll) {
        case 1: PV_DS(1) break;
        case 2: PV_DS(2) break;
        case 4: PV_DS(4) break;
        case 8: PV_DS(8) break;
        default: return fail(PVHIP_EINVAL, "pvhip_diag_stream_f32: unroll must be 1, 2, 4 or 8");
    }
#undef PV_DS
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}
