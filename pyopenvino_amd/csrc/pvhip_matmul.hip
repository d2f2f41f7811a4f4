// Dense MatMul on the fp32 matrix cores: C[M,N] = op(A) . op(B), any of the four transpose
// combinations through element strides.  64x64 output tile per workgroup, 4 waves as 2x2, each wave
// one 32x32 accumulator fed by v_mfma_f32_32x32x2_f32; the output column is on the lane, so C rows
// are written as 128-byte runs.  Operand tiles are staged K-major in LDS ([16][64+1]).
#include "pvhip_common.h"

using namespace pvhip;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kTM = 64, kTN = 64, kTK = 16;

struct MatMulArgs {
    const float* a;
    const float* b;
    float*       c;         // output, or the partial-sum workspace [splits][M][N] when splits > 1
    int          M, N, K;
    int          k_chunk;   // reduction elements per split (multiple of kTK); blockIdx.z selects the split
    long         sam, sak;  // A(m,k) = a[m*sam + k*sak]
    long         sbk, sbn;  // B(k,n) = b[k*sbk + n*sbn]
};

// kF16 (the MatMul of an FP16 IR, pvhip_matmul_f16): both operands are rounded to fp16 (round to nearest even, v_cvt_f16_f32) while they
// are staged, then multiplied on the fp32 matrix cores: a product of two fp16 values is exact in fp32, so this is the arithmetic of
// v_mfma_f32_32x32x16_f16 (fp16 operands, fp32 accumulation) in another summation order -- on the kernel that splits the reduction
// over workgroups, which is what the launch-size-bound FC layers need (0.027 ms against 0.175 ms for the f16 tile kernel).
template <bool kF16>
__global__ __launch_bounds__(kBlock) void matmul_kernel(MatMulArgs p) {
    __shared__ float As[kTK][kTM + 1];
    __shared__ float Bs[kTK][kTN + 1];
    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid  = tid / kWave;
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * kTM, n0 = blockIdx.x * kTN;
    const bool a_k_fast = (p.sak == 1);  // consecutive lanes walk the contiguous global axis
    const bool b_n_fast = (p.sbn == 1);

    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    const int k_begin = blockIdx.z * p.k_chunk;
    const int k_end   = min(p.K, k_begin + p.k_chunk);
    for (int k0 = k_begin; k0 < k_end; k0 += kTK) {
#pragma unroll
        for (int j = 0; j < (kTM * kTK) / kBlock; ++j) {
            const int e = tid + j * kBlock;
            int       m, k;
            if (a_k_fast) { m = e / kTK; k = e % kTK; } else { k = e / kTM; m = e % kTM; }
            const int gm = m0 + m, gk = k0 + k;
            float av = (gm < p.M && gk < k_end) ? p.a[(long)gm * p.sam + (long)gk * p.sak] : 0.0f;
            if (kF16) av = (float)(_Float16)av;
            As[k][m] = av;
        }
#pragma unroll
        for (int j = 0; j < (kTN * kTK) / kBlock; ++j) {
            const int e = tid + j * kBlock;
            int       n, k;
            if (b_n_fast) { k = e / kTN; n = e % kTN; } else { n = e / kTK; k = e % kTK; }
            const int gn = n0 + n, gk2 = k0 + k;
            float bv = (gn < p.N && gk2 < k_end) ? p.b[(long)gk2 * p.sbk + (long)gn * p.sbn] : 0.0f;
            if (kF16) bv = (float)(_Float16)bv;
            Bs[k][n] = bv;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kTK / 2; ++kk) {
            const float af = As[2 * kk + lh][wm * 32 + l31];
            const float bf = Bs[2 * kk + lh][wn * 32 + l31];
            acc            = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int gn = n0 + wn * 32 + l31;
    float* __restrict__ cz = p.c + (size_t)blockIdx.z * p.M * p.N;
    if (gn < p.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (gm < p.M) cz[(size_t)gm * p.N + gn] = acc[r];
        }
    }
}

// Sum the split-K partial results in split order (fixed order: the result is reproducible bit for bit).
__global__ __launch_bounds__(kBlock) void matmul_reduce_kernel(const float* __restrict__ part, float* __restrict__ c,
                                                                size_t mn, int splits) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < mn; i += stride) {
        float s = part[i];
        for (int z = 1; z < splits; ++z) s += part[(size_t)z * mn + i];
        c[i] = s;
    }
}

}  // namespace

namespace pvhip {

int matmul_impl(const float* a, const float* b, float* c, int m, int n, int k, int trans_a, int trans_b, int round_f16) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(m >= 0 && n >= 0 && k >= 0);
    if ((size_t)m * n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(c != nullptr && (k == 0 || (a != nullptr && b != nullptr)));
    if (k == 0) return pvhip_memset(c, 0, (size_t)m * n * sizeof(float));
    MatMulArgs p;
    p.a = a; p.b = b; p.c = c; p.M = m; p.N = n; p.K = k;
    if (trans_a) { p.sam = 1; p.sak = m; } else { p.sam = k; p.sak = 1; }   // stored [K,M] / [M,K]
    if (trans_b) { p.sbk = 1; p.sbn = k; } else { p.sbk = n; p.sbn = 1; }   // stored [N,K] / [K,N]
    dim3 grid((n + kTN - 1) / kTN, (m + kTM - 1) / kTM, 1);
    if (grid.y > 65535) return fail(PVHIP_EUNSUPPORTED, "pvhip_matmul: M=%d too large for the tile grid", m);
    // Split the reduction over workgroups when the output has too few tiles to fill the chip (the FC layers
    // of the IRs: 64 tiles at batch 256); partial tiles go to a workspace and are summed in split order.
    const long tiles  = (long)grid.x * grid.y;
    int        splits = 1;
    if (tiles < 2 * kNumCU) {
        splits = (int)((4 * kNumCU + tiles - 1) / tiles);
        const int max_splits = (k + 4 * kTK - 1) / (4 * kTK);      // at least 4 MFMA steps of 16 per split
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    p.k_chunk = ((k + splits - 1) / splits + kTK - 1) / kTK * kTK;
    splits    = (k + p.k_chunk - 1) / p.k_chunk;
    if (splits <= 1) {
        p.k_chunk = (k + kTK - 1) / kTK * kTK;
        if (round_f16) hipLaunchKernelGGL(matmul_kernel<true>, grid, dim3(kBlock), 0, state().stream, p);
        else           hipLaunchKernelGGL(matmul_kernel<false>, grid, dim3(kBlock), 0, state().stream, p);
        PVHIP_LAUNCH_CHECK();
        return PVHIP_OK;
    }
    void* ws = nullptr;
    int   rc = pvhip_malloc(&ws, (size_t)splits * m * n * sizeof(float));
    if (rc) return rc;
    p.c    = static_cast<float*>(ws);
    grid.z = splits;
    if (round_f16) hipLaunchKernelGGL(matmul_kernel<true>, grid, dim3(kBlock), 0, state().stream, p);
    else           hipLaunchKernelGGL(matmul_kernel<false>, grid, dim3(kBlock), 0, state().stream, p);
    hipLaunchKernelGGL(matmul_reduce_kernel, dim3(grid_for((size_t)m * n)), dim3(kBlock), 0, state().stream,
                       static_cast<const float*>(ws), c, (size_t)m * n, splits);
    (void)pvhip_free(ws);   // stream-ordered: the pool hands the block out again only to later work on this stream
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // namespace pvhip

extern "C" {

int pvhip_matmul_f32(const float* a, const float* b, float* c, int m, int n, int k, int trans_a, int trans_b) {
    return pvhip::matmul_impl(a, b, c, m, n, k, trans_a, trans_b, 0);
}

}  // extern "C"

#ifdef PVHIP_DIAG   // diagnostic build only (include/pvhip_diag.h): not in the product library
// ---------------------------------------------------------------------------------------------------------------------------
// Measurement utility (bench.py's roofline.sustained): what THIS device sustains on v_mfma_f32_32x32x2_f32 with nothing else in
// the instruction stream -- operands in registers, random data (all-zero operands let the chip hold a higher clock), every SIMD
// of every CU issuing.  The chip lowers its clock under such a load, so this, not 157.3 TFLOP/s (the rate at 2.4 GHz), is the
// ceiling a convolution kernel can be held against on the box it ran on.  mode 1 adds a second wave per SIMD that issues only
// v_fma_f32: the fp32 matrix and vector instructions of one SIMD do not overlap (the MFMA rate drops by the VALU wave's share).
namespace {
__global__ __launch_bounds__(512) void mfma_ceiling_kernel(float* sink, unsigned long long* clocks, int iters, int mode) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float a = 1.0f + 0.001f * (float)((lane * 37 + wid * 11 + blockIdx.x) & 255), b = 0.5f + 0.002f * (float)((lane * 17 + blockIdx.x * 3) & 127);
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    if (mode == 1 && wid >= 4) {                      // waves 4-7 share the SIMDs of waves 0-3: VALU only
        float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                v0 = __builtin_fmaf(v0, 0.999f, b); v1 = __builtin_fmaf(v1, 0.998f, a);
                v2 = __builtin_fmaf(v2, 0.997f, b); v3 = __builtin_fmaf(v3, 0.996f, a);
            }
        }
        if (v0 + v1 + v2 + v3 == 123.456f) sink[0] = v0;
    } else if (mode == 2) {                           // the 16x16x4 shape: the same flops per cycle, half the operand reuse
        typedef float floatx4 __attribute__((ext_vector_type(4)));
        floatx4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, c7, 0, 0, 0);
            a = a * 0.99999f;
        }
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += c0[r] + c1[r] + c2[r] + c3[r] + c4[r] + c5[r] + c6[r] + c7[r];
        if (sum == 123.456f) sink[1] = sum;
    } else {
        floatx16 c0, c1, c2, c3;
#pragma unroll
        for (int r = 0; r < 16; ++r) { c0[r] = 0.0f; c1[r] = 0.0f; c2[r] = 0.0f; c3[r] = 0.0f; }
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
            a = a * 0.99999f;                         // keeps the operands from being loop invariants folded away; stays finite
        }
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += c0[r] + c1[r] + c2[r] + c3[r];
        if (sum == 123.456f) sink[1] = sum;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clocks[0] = __builtin_readcyclecounter() - t0;
        clocks[1] = __builtin_amdgcn_s_memrealtime() - r0;       // 100 MHz
    }
}
}  // namespace

extern "C" int pvhip_mfma_ceiling_f32(int mode, int iters, double* tflops, double* clock_ghz) {
    PVHIP_REQUIRE_INIT();
    if (iters <= 0 || (mode < 0 || mode > 2) || tflops == nullptr || clock_ghz == nullptr)
        return pvhip::fail(PVHIP_EINVAL, "pvhip_mfma_ceiling_f32: mode 0 | 1 | 2, iters > 0");
    float*              sink = nullptr;
    unsigned long long* clocks = nullptr;
    PVHIP_HIP(hipMalloc(&sink, 16));
    PVHIP_HIP(hipMalloc(&clocks, 16));
    hipEvent_t e0, e1;
    PVHIP_HIP(hipEventCreate(&e0));
    PVHIP_HIP(hipEventCreate(&e1));
    hipStream_t st = pvhip::state().stream;
    const int   threads = mode == 1 ? 512 : 256;     // one MFMA wave per SIMD (+ one VALU wave per SIMD in mode 1)
    const dim3  grid(pvhip::kNumCU);
    hipLaunchKernelGGL(mfma_ceiling_kernel, grid, dim3(threads), 0, st, sink, clocks, iters / 8 + 1, mode);      // clocks ramp up
    PVHIP_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(mfma_ceiling_kernel, grid, dim3(threads), 0, st, sink, clocks, iters, mode);
    PVHIP_HIP(hipEventRecord(e1, st));
    PVHIP_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    PVHIP_HIP(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2] = {0, 0};
    PVHIP_HIP(hipMemcpy(h, clocks, sizeof(h), hipMemcpyDeviceToHost));
    *tflops    = (double)pvhip::kNumCU * 4.0 * (double)iters * 4.0 * 4096.0 / ((double)ms * 1e-3) / 1e12;
    *clock_ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    (void)hipFree(clocks);
    return PVHIP_OK;
}
#endif  // PVHIP_DIAG
