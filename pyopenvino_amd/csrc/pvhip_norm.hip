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

__device__ __forceinline__ float lrn_pow(float d, float beta, int beta_mode) {
    // beta_mode: 1 -> d^0.75 as sqrt(d)*sqrt(sqrt(d)) (two correctly rounded roots), 2 -> d^0.5,
    // 3 -> d, 0 -> powf
    if (beta_mode == 1) {
        const float s = sqrtf(d);
        return s * sqrtf(s);
    }
    if (beta_mode == 2) return sqrtf(d);
    if (beta_mode == 3) return d;
    return powf(d, beta);
}

// One lane owns VEC adjacent pixels of one image and walks the channel axis in chunks of T channels: the
// loads of chunk k+1 are issued before the outputs of chunk k are computed (T independent 16-byte loads in
// flight per lane), every element is read once and written once.  The outputs computed after chunk k has
// landed are those whose whole window lies in chunks <= k, i.e. channels [T*k - HALF, T*k + T - HALF).
// Window for channel c is [c - SIZE/2, c + SIZE/2] clipped to [0, C); squares are summed in ascending
// channel order, the order np.sum(axis=1) uses (LRN.py:19); out-of-range channels contribute an exact 0.
template <int SIZE, int VEC>
__global__ __launch_bounds__(kBlock) void lrn_window_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                                             int c, int hw, float alpha, float beta, float bias,
                                                             int beta_mode) {
    constexpr int HALF = SIZE / 2;
    constexpr int T    = 8;                 // channels per chunk
    static_assert(2 * HALF <= T, "window halo must fit in one chunk");
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int      cols_per_img = hw / VEC;
    const unsigned total        = (unsigned)n * (unsigned)cols_per_img;
    const unsigned stride       = gridDim.x * blockDim.x;
    const int      n_chunks     = (c + T - 1) / T;
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const unsigned img  = t / (unsigned)cols_per_img;
        const unsigned col  = t - img * (unsigned)cols_per_img;
        const size_t   base = (size_t)img * c * hw + (size_t)col * VEC;
        const vec_t* __restrict__ xv = reinterpret_cast<const vec_t*>(x + base);
        vec_t* __restrict__       yv = reinterpret_cast<vec_t*>(y + base);
        const size_t cstride = (size_t)hw / VEC;   // channel stride in vec_t units
        // win[0 .. 2*HALF-1]: last 2*HALF channels of the previous chunks; cur[]: chunk k; nxt[]: chunk k+1
        vec_t win[2 * HALF], cur[T], nxt[T];
#pragma unroll
        for (int j = 0; j < 2 * HALF; ++j) win[j] = (vec_t)(0.0f);
#pragma unroll
        for (int j = 0; j < T; ++j) cur[j] = (j < c) ? xv[(size_t)j * cstride] : (vec_t)(0.0f);
        for (int k = 0; k <= n_chunks; ++k) {
            // issue chunk k+1 (zeros past the last channel)
            const int c1 = (k + 1) * T;
#pragma unroll
            for (int j = 0; j < T; ++j) nxt[j] = (c1 + j < c) ? xv[(size_t)(c1 + j) * cstride] : (vec_t)(0.0f);
            // outputs [T*k - HALF, T*k + T - HALF): inputs win[] ++ cur[] form channels [T*k - 2*HALF, T*k + T)
            vec_t ext[2 * HALF + T];
#pragma unroll
            for (int j = 0; j < 2 * HALF; ++j) ext[j] = win[j];
#pragma unroll
            for (int j = 0; j < T; ++j) ext[2 * HALF + j] = cur[j];
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const int ch = k * T - HALF + j;       // centre channel = ext[j + HALF]
                if (ch >= 0 && ch < c) {
                    vec_t s = ext[j] * ext[j];
#pragma unroll
                    for (int q = 1; q < SIZE; ++q) s = s + ext[j + q] * ext[j + q];
                    vec_t o;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const float d = bias + alpha * s[v];
                        o[v]          = ext[j + HALF][v] / lrn_pow(d, beta, beta_mode);
                    }
                    yv[(size_t)ch * cstride] = o;
                }
            }
#pragma unroll
            for (int j = 0; j < 2 * HALF; ++j) win[j] = cur[T - 2 * HALF + j];
#pragma unroll
            for (int j = 0; j < T; ++j) cur[j] = nxt[j];
        }
    }
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
        y[e]          = x[e] / lrn_pow(d, beta, beta_mode);
    }
}

template <int SIZE>
void launch_lrn_window(const float* x, float* y, int n, int c, int hw, float alpha, float beta, float bias, int bm) {
    const bool vec4 = (hw % 4 == 0);
    if (vec4) {
        const size_t cols = (size_t)n * (hw / 4);
        hipLaunchKernelGGL((lrn_window_kernel<SIZE, 4>), dim3(grid_for(cols)), dim3(kBlock), 0, state().stream, x, y, n, c,
                           hw, alpha, beta, bias, bm);
    } else {
        const size_t cols = (size_t)n * hw;
        hipLaunchKernelGGL((lrn_window_kernel<SIZE, 1>), dim3(grid_for(cols)), dim3(kBlock), 0, state().stream, x, y, n, c,
                           hw, alpha, beta, bias, bm);
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
    int bm = 0;
    if (beta == 0.75f) bm = 1;
    else if (beta == 0.5f) bm = 2;
    else if (beta == 1.0f) bm = 3;
    switch (size) {
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
