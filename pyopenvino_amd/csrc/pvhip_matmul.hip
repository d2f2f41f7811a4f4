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
            As[k][m] = (gm < p.M && gk < k_end) ? p.a[(long)gm * p.sam + (long)gk * p.sak] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < (kTN * kTK) / kBlock; ++j) {
            const int e = tid + j * kBlock;
            int       n, k;
            if (b_n_fast) { k = e / kTN; n = e % kTN; } else { n = e / kTK; k = e % kTK; }
            const int gn = n0 + n, gk = k0 + k;
            Bs[k][n] = (gn < p.N && gk < k_end) ? p.b[(long)gk * p.sbk + (long)gn * p.sbn] : 0.0f;
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

extern "C" {

int pvhip_matmul_f32(const float* a, const float* b, float* c, int m, int n, int k, int trans_a, int trans_b) {
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
    if (grid.y > 65535) return fail(PVHIP_EUNSUPPORTED, "pvhip_matmul_f32: M=%d too large for the tile grid", m);
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
        hipLaunchKernelGGL(matmul_kernel, grid, dim3(kBlock), 0, state().stream, p);
        PVHIP_LAUNCH_CHECK();
        return PVHIP_OK;
    }
    void* ws = nullptr;
    int   rc = pvhip_malloc(&ws, (size_t)splits * m * n * sizeof(float));
    if (rc) return rc;
    p.c    = static_cast<float*>(ws);
    grid.z = splits;
    hipLaunchKernelGGL(matmul_kernel, grid, dim3(kBlock), 0, state().stream, p);
    hipLaunchKernelGGL(matmul_reduce_kernel, dim3(grid_for((size_t)m * n)), dim3(kBlock), 0, state().stream,
                       static_cast<const float*>(ws), c, (size_t)m * n, splits);
    (void)pvhip_free(ws);   // stream-ordered: the pool hands the block out again only to later work on this stream
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // extern "C"
