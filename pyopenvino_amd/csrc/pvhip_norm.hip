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

// One lane owns VEC adjacent pixels of one image and walks the channel axis keeping the last SIZE
// inputs in registers: every element is read once and written once.  Window for channel c is
// [c - SIZE/2, c + SIZE/2] clipped to [0, C); squares are summed in ascending channel order, the
// order np.sum(axis=1) uses (LRN.py:19).
template <int SIZE, int VEC>
__global__ __launch_bounds__(kBlock) void lrn_window_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                                             int c, int hw, float alpha, float beta, float bias,
                                                             int beta_mode) {
    constexpr int HALF = SIZE / 2;
    const int     cols_per_img = hw / VEC;
    const unsigned total       = (unsigned)n * (unsigned)cols_per_img;
    const unsigned stride      = gridDim.x * blockDim.x;
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const unsigned img = t / (unsigned)cols_per_img;
        const unsigned col = t - img * (unsigned)cols_per_img;
        const size_t   base = (size_t)img * c * hw + (size_t)col * VEC;
        float win[SIZE][VEC];
#pragma unroll
        for (int j = 0; j < SIZE; ++j)
#pragma unroll
            for (int v = 0; v < VEC; ++v) win[j][v] = 0.0f;
        // preload channels 0 .. HALF-1 into the upper part of the window
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            if (j < c) {
                if (VEC == 4) {
                    const float4 q = *reinterpret_cast<const float4*>(x + base + (size_t)j * hw);
                    win[HALF + 1 + j][0] = q.x; win[HALF + 1 + j][1] = q.y;
                    win[HALF + 1 + j][2] = q.z; win[HALF + 1 + j][3] = q.w;
                } else {
                    win[HALF + 1 + j][0] = x[base + (size_t)j * hw];
                }
            }
        }
        for (int ch = 0; ch < c; ++ch) {
            // shift: win[j] <- win[j+1]; then bring in channel ch + HALF
#pragma unroll
            for (int j = 0; j < SIZE - 1; ++j)
#pragma unroll
                for (int v = 0; v < VEC; ++v) win[j][v] = win[j + 1][v];
            const int cin = ch + HALF;
            if (cin < c) {
                if (VEC == 4) {
                    const float4 q = *reinterpret_cast<const float4*>(x + base + (size_t)cin * hw);
                    win[SIZE - 1][0] = q.x; win[SIZE - 1][1] = q.y; win[SIZE - 1][2] = q.z; win[SIZE - 1][3] = q.w;
                } else {
                    win[SIZE - 1][0] = x[base + (size_t)cin * hw];
                }
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) win[SIZE - 1][v] = 0.0f;
            }
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                // out-of-range channels hold 0 and add exactly 0 to the sum
                float s = 0.0f;
#pragma unroll
                for (int j = 0; j < SIZE; ++j) {
                    const float sq = win[j][v] * win[j][v];
                    s              = (j == 0) ? sq : s + sq;
                }
                const float d = bias + alpha * s;
                o[v]          = win[HALF][v] / lrn_pow(d, beta, beta_mode);
            }
            if (VEC == 4)
                *reinterpret_cast<float4*>(y + base + (size_t)ch * hw) = make_float4(o[0], o[1], o[2], o[3]);
            else
                y[base + (size_t)ch * hw] = o[0];
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
