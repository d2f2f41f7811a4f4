// SoftMax (row-wise, wave-shuffle reductions) and cross-channel LRN (register sliding window).
#include "pvhip_common.h"

using namespace pvhip;

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// One workgroup per row: exp once into registers (cols <= kBlock * kPerThread) or recompute.
// y = exp(x) / sum(exp(x)) with no max shift, exactly the reference expression (SoftMax.py:12-13).
constexpr int kSoftmaxPerThread = 8;

__global__ __launch_bounds__(kBlock) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               int rows, int cols) {
    __shared__ float wave_part[kBlock / kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wid  = threadIdx.x / kWave;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const float* __restrict__ xr = x + (size_t)r * cols;
        float* __restrict__       yr = y + (size_t)r * cols;
        float e[kSoftmaxPerThread];
        float part = 0.0f;
        const bool in_regs = cols <= kBlock * kSoftmaxPerThread;
        if (in_regs) {
#pragma unroll
            for (int j = 0; j < kSoftmaxPerThread; ++j) {
                const int c = threadIdx.x + j * kBlock;
                e[j]        = (c < cols) ? expf(xr[c]) : 0.0f;
                part += e[j];
            }
        } else {
            for (int c = threadIdx.x; c < cols; c += kBlock) part += expf(xr[c]);
        }
        part = wave_sum(part);
        __syncthreads();  // wave_part free from the previous row
        if (lane == 0) wave_part[wid] = part;
        __syncthreads();
        float total = 0.0f;
#pragma unroll
        for (int wv = 0; wv < kBlock / kWave; ++wv) total += wave_part[wv];
        if (in_regs) {
#pragma unroll
            for (int j = 0; j < kSoftmaxPerThread; ++j) {
                const int c = threadIdx.x + j * kBlock;
                if (c < cols) yr[c] = e[j] / total;
            }
        } else {
            for (int c = threadIdx.x; c < cols; c += kBlock) yr[c] = expf(xr[c]) / total;
        }
    }
}

// One lane owns VEC adjacent pixels of one image and walks the channel axis in chunks of T = 8 channels
// (C must be a multiple of 8; other channel counts take the generic kernel).  The T loads of chunk k+1 are
// issued before the outputs that chunk k completes are computed, so T independent 16-byte loads per lane are
// in flight; every element is read once and written once.  After chunk k has landed the computable outputs
// are channels [T*k - HALF, T*k + T - HALF).  The loop body is branch-free: the first chunk, the last chunk and
// the trailing HALF outputs are peeled.  Window for channel c is [c - SIZE/2, c + SIZE/2] clipped to [0, C);
// squares are summed in ascending channel order, the order np.sum(axis=1) uses (LRN.py:19); channels outside
// the tensor contribute an exact 0.
template <int SIZE, int VEC, int BETA_MODE>
__global__ __launch_bounds__(kBlock) void lrn_window_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                                             int c, int hw, float alpha, float beta, float bias) {
    constexpr int beta_mode = BETA_MODE;   // compile-time: keeps the per-element path branch-free
    constexpr int HALF = SIZE / 2;
    constexpr int T    = 8;
    constexpr int E    = 2 * HALF + T;      // ext[]: channels [T*k - 2*HALF, T*k + T)
    static_assert(2 * HALF <= T, "window halo must fit in one chunk");
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int      cols_per_img = hw / VEC;
    const unsigned total        = (unsigned)n * (unsigned)cols_per_img;
    const unsigned stride       = gridDim.x * blockDim.x;
    const int      n_chunks     = c / T;
    const size_t   cstride      = (size_t)hw / VEC;   // channel stride in vec_t units

    // one output: centre ext[j + HALF], window ext[j .. j + SIZE)
#define PV_LRN_OUT(j_, ch_)                                                                    \
    {                                                                                          \
        vec_t s_ = ext[(j_)] * ext[(j_)];                                                      \
        _Pragma("unroll") for (int q = 1; q < SIZE; ++q) s_ = s_ + ext[(j_) + q] * ext[(j_) + q]; \
        vec_t o_;                                                                              \
        _Pragma("unroll") for (int v = 0; v < VEC; ++v) {                                      \
            const float d_ = bias + alpha * s_[v];                                             \
            o_[v]          = lrn_div(ext[(j_) + HALF][v], d_, beta, beta_mode);                \
        }                                                                                      \
        yv[(size_t)(ch_) * cstride] = o_;                                                      \
    }

    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const unsigned img  = t / (unsigned)cols_per_img;
        const unsigned col  = t - img * (unsigned)cols_per_img;
        const size_t   base = (size_t)img * c * hw + (size_t)col * VEC;
        const vec_t* __restrict__ xv = reinterpret_cast<const vec_t*>(x + base);
        vec_t* __restrict__       yv = reinterpret_cast<vec_t*>(y + base);
        vec_t ext[E], nxt[T];
#pragma unroll
        for (int j = 0; j < 2 * HALF; ++j) ext[j] = (vec_t)(0.0f);
#pragma unroll
        for (int j = 0; j < T; ++j) ext[2 * HALF + j] = xv[(size_t)j * cstride];
        // chunks 0 .. n_chunks-2: prefetch the next chunk, emit what this one completes
        for (int k = 0; k + 1 < n_chunks; ++k) {
            const vec_t* __restrict__ xn = xv + (size_t)(k + 1) * T * cstride;
#pragma unroll
            for (int j = 0; j < T; ++j) nxt[j] = xn[(size_t)j * cstride];
            if (k == 0) {
#pragma unroll
                for (int j = HALF; j < T; ++j) PV_LRN_OUT(j, j - HALF)
            } else {
                const int ch0 = k * T - HALF;
#pragma unroll
                for (int j = 0; j < T; ++j) PV_LRN_OUT(j, ch0 + j)
            }
#pragma unroll
            for (int j = 0; j < 2 * HALF; ++j) ext[j] = ext[T + j];
#pragma unroll
            for (int j = 0; j < T; ++j) ext[2 * HALF + j] = nxt[j];
        }
        // last chunk (no prefetch), then the trailing HALF channels whose windows run past the tensor
        {
            const int k = n_chunks - 1, ch0 = k * T - HALF;
            if (k == 0) {
#pragma unroll
                for (int j = HALF; j < T; ++j) PV_LRN_OUT(j, j - HALF)
            } else {
#pragma unroll
                for (int j = 0; j < T; ++j) PV_LRN_OUT(j, ch0 + j)
            }
#pragma unroll
            for (int j = 0; j < 2 * HALF; ++j) ext[j] = ext[T + j];
#pragma unroll
            for (int j = 0; j < T; ++j) ext[2 * HALF + j] = (vec_t)(0.0f);
#pragma unroll
            for (int j = 0; j < HALF; ++j) PV_LRN_OUT(j, c - HALF + j)
        }
    }
#undef PV_LRN_OUT
}

// Fallback for window sizes without a register-window instantiation: every lane re-reads its window.
__global__ __launch_bounds__(kBlock) void lrn_generic_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                                              int c, int hw, int size, float alpha, float beta,
                                                              float bias, int beta_mode) {
    const size_t total  = (size_t)n * c * hw;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int    half   = size / 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t p   = e % hw;
        const size_t q   = e / hw;
        const int    ch  = (int)(q % c);
        const size_t img = q / c;
        int lo = ch - half, hi = ch + half + 1;
        if (lo < 0) lo = 0;
        if (hi > c) hi = c;
        float s = 0.0f;
        for (int k = lo; k < hi; ++k) {
            const float v = x[(img * c + k) * hw + p];
            s             = (k == lo) ? v * v : s + v * v;
        }
        const float d = bias + alpha * s;
        y[e]          = lrn_div(x[e], d, beta, beta_mode);
    }
}

template <int SIZE, int BETA_MODE>
void launch_lrn_window_b(const float* x, float* y, int n, int c, int hw, float alpha, float beta, float bias) {
    if (hw % 4 == 0) {
        const size_t cols = (size_t)n * (hw / 4);
        hipLaunchKernelGGL((lrn_window_kernel<SIZE, 4, BETA_MODE>), dim3(grid_for(cols)), dim3(kBlock), 0, state().stream, x, y,
                           n, c, hw, alpha, beta, bias);
    } else {
        const size_t cols = (size_t)n * hw;
        hipLaunchKernelGGL((lrn_window_kernel<SIZE, 1, BETA_MODE>), dim3(grid_for(cols)), dim3(kBlock), 0, state().stream, x, y,
                           n, c, hw, alpha, beta, bias);
    }
}

template <int SIZE>
void launch_lrn_window(const float* x, float* y, int n, int c, int hw, float alpha, float beta, float bias, int bm) {
    switch (bm) {
        case 1: launch_lrn_window_b<SIZE, 1>(x, y, n, c, hw, alpha, beta, bias); break;
        case 2: launch_lrn_window_b<SIZE, 2>(x, y, n, c, hw, alpha, beta, bias); break;
        case 3: launch_lrn_window_b<SIZE, 3>(x, y, n, c, hw, alpha, beta, bias); break;
        case 4: launch_lrn_window_b<SIZE, 4>(x, y, n, c, hw, alpha, beta, bias); break;
        default: launch_lrn_window_b<SIZE, 0>(x, y, n, c, hw, alpha, beta, bias); break;
    }
}

}  // namespace

extern "C" {

int pvhip_softmax_rows_f32(const float* x, float* y, int rows, int cols) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(rows >= 0 && cols >= 0);
    if (rows == 0 || cols == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr);
    const int g = rows < kMaxBlocks ? rows : kMaxBlocks;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(g), dim3(kBlock), 0, state().stream, x, y, rows, cols);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_lrn_f32(const float* x, float* y, int n, int c, int hw, int size, float alpha, float beta, float bias) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && c >= 0 && hw >= 0 && size >= 1);
    if ((size_t)n * c * hw == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr);
    if ((unsigned long long)n * c * hw >= (1ull << 31))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_lrn_f32: tensor exceeds 2^31 elements");
    const int bm = lrn_beta_mode(beta, bias);
    switch ((c % 8 == 0 && c >= 8) ? size : 0) {
        case 3: launch_lrn_window<3>(x, y, n, c, hw, alpha, beta, bias, bm); break;
        case 5: launch_lrn_window<5>(x, y, n, c, hw, alpha, beta, bias, bm); break;
        case 7: launch_lrn_window<7>(x, y, n, c, hw, alpha, beta, bias, bm); break;
        default:
            hipLaunchKernelGGL(lrn_generic_kernel, dim3(grid_for((size_t)n * c * hw)), dim3(kBlock), 0, state().stream, x, y,
                               n, c, hw, size, alpha, beta, bias, bm);
    }
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // extern "C"
