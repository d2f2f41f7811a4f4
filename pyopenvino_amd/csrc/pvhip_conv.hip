// Convolution as an implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), plus the
// depthwise (GroupConvolution) VALU kernel.
//
//   D[k_out][pixel] = sum_kred  Wt[kred][k_out] * Col[kred][pixel]
//
//   * MFMA A operand = weights (rows = output channels), B operand = im2col tile (cols = output
//     pixels).  The 32x32 accumulator then has the pixel on the lane (col = lane & 31), so the NCHW
//     store of one accumulator register is two 128-byte runs of consecutive pixels.
//   * The im2col matrix is never materialised: each lane owns one output pixel of the tile, keeps its
//     (n, ih0, iw0) in registers and gathers x[n, c, ih0 + r, iw0 + s] for the tile's 16 reduction
//     rows; (c, r, s) of a row is wave-uniform and comes from a small table built with the weights,
//     so it is decoded on the scalar unit.  Zero padding is a predicate, never memory.
//   * Weights are repacked once per tensor into the K-major panel [kred_pad][kout_pad] (zero padded),
//     so the A tile is a run of aligned float4 loads with no bounds checks.
//   * Both tiles are staged K-major in LDS ([BK][BM] / [BK][BN]); the MFMA operand read of a wave is
//     then two 128-byte rows per ds_read_b32 (lanes 0-31 -> k, lanes 32-63 -> k+1): conflict-free,
//     no padding.  Register-staged double buffering: global loads of step t+1 are issued before the
//     MFMAs of step t and written to the other LDS buffer after them; one barrier per step.
#include <climits>

#include "pvhip_common.h"

using namespace pvhip;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kBK       = 16;   // reduction rows per LDS stage
constexpr int kKoutAlign = 128;  // packed panel width is a multiple of this

struct ConvArgs {
    const float* x;
    const int*   ktab;  // [kred_pad] : (c << 16) | (r << 8) | s, or negative for a padding row
    const float* wp;    // [kred_pad][kout_pad]
    float*       y;
    const float* bias;  // optional [K]
    int N, C, H, W, K, OH, OW;
    int sh, sw, pt, pl;
    int kred_pad, kout_pad;
    int P;              // N*OH*OW
    int n_mtiles;
    int relu;
};

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(kBlock) void conv_igemm_kernel(ConvArgs a) {
    static_assert(WAVES_M * WAVES_N == kBlock / kWave, "4 waves per workgroup");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1 && WM % 32 == 0 && WN % 32 == 0, "wave tile is a multiple of 32x32");
    static_assert(BN % kWave == 0 && kBlock % BN == 0, "a wave gathers one reduction row");
    constexpr int B_ROWS_PER_PASS = kBlock / BN;
    constexpr int B_LOADS         = kBK / B_ROWS_PER_PASS;
    constexpr int A_F4_TOTAL      = kBK * BM / 4;
    constexpr int A_F4            = (A_F4_TOTAL + kBlock - 1) / kBlock;

    __shared__ __attribute__((aligned(16))) float As[2][kBK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][kBK][BN];

    // ---- tile assignment: XCD-aware remap so that workgroups sharing an L2 work on neighbouring
    // pixel tiles and all output-channel tiles of one pixel tile run back to back on one XCD.
    const int nwg = gridDim.x;
    int       lid;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt    = lid % a.n_mtiles;
    const int ptile = lid / a.n_mtiles;
    const int m0    = mt * BM;

    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid  = __builtin_amdgcn_readfirstlane(tid / kWave);

    // ---- this lane's output pixel for the gather
    const int OHW = a.OH * a.OW;
    const int HW  = a.H * a.W;
    const int pc  = tid % BN;
    const int prow0 = __builtin_amdgcn_readfirstlane(tid / BN);
    int       ih0, iw0, xbase;
    {
        const int gp = ptile * BN + pc;
        if (gp < a.P) {
            const int n   = gp / OHW;
            const int rem = gp - n * OHW;
            const int oy  = rem / a.OW;
            const int ox  = rem - oy * a.OW;
            ih0           = oy * a.sh - a.pt;
            iw0           = ox * a.sw - a.pl;
            xbase         = n * a.C * HW + ih0 * a.W + iw0;
        } else {
            ih0 = INT_MIN / 2;  // every bounds test fails
            iw0 = 0;
            xbase = 0;
        }
    }

    float  breg[B_LOADS];
    float4 areg[A_F4];

    auto load_tiles = [&](int kt) {
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            const int kg  = kt * kBK + prow0 + j * B_ROWS_PER_PASS;  // wave-uniform
            const int ent = a.ktab[kg];
            const int c   = (ent >> 16) & 0x7fff;
            const int r   = (ent >> 8) & 0xff;
            const int s   = ent & 0xff;
            const bool ok = (ent >= 0) && ((unsigned)(ih0 + r) < (unsigned)a.H) && ((unsigned)(iw0 + s) < (unsigned)a.W);
            const int idx = ok ? (xbase + c * HW + r * a.W + s) : 0;
            const float v = a.x[idx];
            breg[j]       = ok ? v : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            const int f = tid + j * kBlock;
            if (A_F4_TOTAL % kBlock == 0 || f < A_F4_TOTAL) {
                const int arow = f / (BM / 4);
                const int ac4  = f % (BM / 4);
                areg[j] = *reinterpret_cast<const float4*>(a.wp + (size_t)(kt * kBK + arow) * a.kout_pad + m0 + ac4 * 4);
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) Bs[buf][prow0 + j * B_ROWS_PER_PASS][pc] = breg[j];
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            const int f = tid + j * kBlock;
            if (A_F4_TOTAL % kBlock == 0 || f < A_F4_TOTAL) {
                const int arow = f / (BM / 4);
                const int ac4  = f % (BM / 4);
                *reinterpret_cast<float4*>(&As[buf][arow][ac4 * 4]) = areg[j];
            }
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int wm  = wid / WAVES_N, wn = wid % WAVES_N;
    const int l31 = lane & 31, lh = lane >> 5;
    const int a_col = wm * WM + l31;
    const int b_col = wn * WN + l31;

    const int nk = a.kred_pad / kBK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int  buf  = kt & 1;
        const bool more = (kt + 1 < nk);
        if (more) load_tiles(kt + 1);
#pragma unroll
        for (int kk = 0; kk < kBK / 2; ++kk) {
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = As[buf][2 * kk + lh][a_col + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bs[buf][2 * kk + lh][b_col + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: accumulator register r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gp = ptile * BN + wn * WN + j * 32 + l31;
        if (gp >= a.P) continue;
        const int    n    = gp / OHW;
        const int    rem  = gp - n * OHW;
        float* __restrict__ yp = a.y + (size_t)n * a.K * OHW + rem;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ko = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ko < a.K) {
                    float v = acc[i][j][r];
                    if (a.bias != nullptr) v = v + a.bias[ko];
                    if (a.relu) v = (v < 0.0f) ? 0.0f : v;
                    yp[(size_t)ko * OHW] = v;
                }
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void conv_pack_kernel(const float* __restrict__ w, int* __restrict__ ktab,
                                                            float* __restrict__ wp, int K, int C, int kh, int kw,
                                                            int kred, int kred_pad, int kout_pad) {
    const size_t total  = (size_t)kred_pad * kout_pad;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int kr = (int)(e / kout_pad);
        const int ko = (int)(e % kout_pad);
        wp[e]        = (kr < kred && ko < K) ? w[(size_t)ko * kred + kr] : 0.0f;
        if (ko == 0) {
            int ent = INT_MIN;
            if (kr < kred) {
                const int s = kr % kw;
                const int t = kr / kw;
                const int r = t % kh;
                const int c = t / kh;
                ent         = (c << 16) | (r << 8) | s;
            }
            ktab[kr] = ent;
        }
    }
}

struct DwArgs {
    int G, H, W, OH, OW, kh, kw, sh, sw, pt, pl;
};

// Depthwise 3x3-style convolution: one lane per output, lanes along the output row.
__global__ __launch_bounds__(kBlock) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ y, DwArgs a, unsigned total) {
    const unsigned stride = gridDim.x * blockDim.x;
    const unsigned ohw    = (unsigned)(a.OH * a.OW);
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const unsigned plane = e / ohw;  // n*G + g
        const unsigned rem   = e - plane * ohw;
        const int      oy    = (int)(rem / (unsigned)a.OW);
        const int      ox    = (int)(rem - (unsigned)oy * (unsigned)a.OW);
        const int      g     = (int)(plane % (unsigned)a.G);
        const float* __restrict__ xp = x + (size_t)plane * (size_t)(a.H * a.W);
        const float* __restrict__ wg = w + (size_t)g * (a.kh * a.kw);
        const int iy0 = oy * a.sh - a.pt, ix0 = ox * a.sw - a.pl;
        float     sum = 0.0f;
        for (int r = 0; r < a.kh; ++r) {
            const int iy = iy0 + r;
            if ((unsigned)iy >= (unsigned)a.H) continue;
            for (int s = 0; s < a.kw; ++s) {
                const int ix = ix0 + s;
                if ((unsigned)ix >= (unsigned)a.W) continue;
                sum += xp[iy * a.W + ix] * wg[r * a.kw + s];
            }
        }
        y[e] = sum;
    }
}

inline int round_up_int(int v, int q) { return (v + q - 1) / q * q; }

template <int BM, int BN, int WAVES_M, int WAVES_N>
void launch_conv(const ConvArgs& a, int n_ptiles) {
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), 0,
                       state().stream, a);
}

}  // namespace

extern "C" {

size_t pvhip_conv2d_pack_elems(int k_out, int c, int kh, int kw) {
    if (k_out <= 0 || c <= 0 || kh <= 0 || kw <= 0) return 0;
    const size_t kred_pad = (size_t)round_up_int(c * kh * kw, kBK);
    const size_t kout_pad = (size_t)round_up_int(k_out, kKoutAlign);
    return kred_pad + kred_pad * kout_pad;
}

int pvhip_conv2d_pack_f32(const float* w_oihw, float* wpack, int k_out, int c, int kh, int kw) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(w_oihw != nullptr && wpack != nullptr);
    PVHIP_CHECK_ARG(k_out > 0 && c > 0 && kh > 0 && kw > 0);
    if (c >= 32768 || kh >= 256 || kw >= 256)
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_pack_f32: C=%d kh=%d kw=%d outside table encoding", c, kh, kw);
    const int kred = c * kh * kw, kred_pad = round_up_int(kred, kBK), kout_pad = round_up_int(k_out, kKoutAlign);
    int*   ktab = reinterpret_cast<int*>(wpack);
    float* wp   = wpack + kred_pad;
    hipLaunchKernelGGL(conv_pack_kernel, dim3(grid_for((size_t)kred_pad * kout_pad)), dim3(kBlock), 0, state().stream,
                       w_oihw, ktab, wp, k_out, c, kh, kw, kred, kred_pad, kout_pad);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_conv2d_f32(const float* x, const float* wpack, float* y, int n, int c, int h, int w, int k_out, int kh, int kw,
                     int oh, int ow, int sh, int sw, int pad_top, int pad_left, const float* bias, int relu) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && c > 0 && h > 0 && w > 0 && k_out > 0 && kh > 0 && kw > 0 && oh >= 0 && ow >= 0);
    PVHIP_CHECK_ARG(sh > 0 && sw > 0 && pad_top >= 0 && pad_left >= 0);
    if (c >= 32768 || kh >= 256 || kw >= 256)
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f32: C=%d kh=%d kw=%d outside table encoding", c, kh, kw);
    const unsigned long long in_e = (unsigned long long)n * c * h * w, out_e = (unsigned long long)n * k_out * oh * ow;
    if (in_e >= (1ull << 31) || out_e >= (1ull << 31))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f32: tensor exceeds 2^31 elements");
    if (out_e == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && wpack != nullptr && y != nullptr);

    ConvArgs a;
    a.kred_pad = round_up_int(c * kh * kw, kBK);
    a.kout_pad = round_up_int(k_out, kKoutAlign);
    a.x        = x;
    a.ktab     = reinterpret_cast<const int*>(wpack);
    a.wp       = wpack + a.kred_pad;
    a.y        = y;
    a.bias     = bias;
    a.N = n; a.C = c; a.H = h; a.W = w; a.K = k_out; a.OH = oh; a.OW = ow;
    a.sh = sh; a.sw = sw; a.pt = pad_top; a.pl = pad_left;
    a.P    = n * oh * ow;
    a.relu = relu;

    // ---- tile selection: output-channel tile with the least padding (ties -> larger), pixel tile
    // 256 unless that leaves fewer than two workgroups per CU.
    int best_bm = 32, best_pad = round_up_int(k_out, 32);
    for (int bm : {64, 128}) {
        const int padded = round_up_int(k_out, bm);
        if (padded <= best_pad) { best_pad = padded; best_bm = bm; }
    }
    int       bm = best_bm, bn = 256;
    const char* env = getenv("PVHIP_CONV_TILE");  // "BMxBN" override for tuning experiments
    if (env != nullptr) {
        int ebm = 0, ebn = 0;
        if (sscanf(env, "%dx%d", &ebm, &ebn) == 2 && (ebm == 32 || ebm == 64 || ebm == 128) && (ebn == 128 || ebn == 256)) {
            bm = ebm; bn = ebn;
        }
    } else {
        const long blocks256 = (long)((a.P + 255) / 256) * (best_pad / bm);
        if (blocks256 < 2 * kNumCU) bn = 128;
    }
    a.n_mtiles       = (k_out + bm - 1) / bm;
    const int n_ptiles = (a.P + bn - 1) / bn;

    if (bm == 128 && bn == 256) launch_conv<128, 256, 2, 2>(a, n_ptiles);
    else if (bm == 128 && bn == 128) launch_conv<128, 128, 2, 2>(a, n_ptiles);
    else if (bm == 64 && bn == 256) launch_conv<64, 256, 1, 4>(a, n_ptiles);
    else if (bm == 64 && bn == 128) launch_conv<64, 128, 1, 4>(a, n_ptiles);
    else if (bm == 32 && bn == 256) launch_conv<32, 256, 1, 4>(a, n_ptiles);
    else launch_conv<32, 128, 1, 4>(a, n_ptiles);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_dwconv2d_f32(const float* x, const float* w, float* y, int n, int g, int h, int wdt, int kh, int kw, int oh,
                       int ow, int sh, int sw, int pad_top, int pad_left) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && g > 0 && h > 0 && wdt > 0 && kh > 0 && kw > 0 && oh >= 0 && ow >= 0 && sh > 0 && sw > 0);
    PVHIP_CHECK_ARG(pad_top >= 0 && pad_left >= 0);
    const unsigned long long in_e = (unsigned long long)n * g * h * wdt, out_e = (unsigned long long)n * g * oh * ow;
    if (in_e >= (1ull << 31) || out_e >= (1ull << 31))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_dwconv2d_f32: tensor exceeds 2^31 elements");
    if (out_e == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && w != nullptr && y != nullptr);
    DwArgs a{g, h, wdt, oh, ow, kh, kw, sh, sw, pad_top, pad_left};
    hipLaunchKernelGGL(dwconv_kernel, dim3(grid_for((size_t)out_e)), dim3(kBlock), 0, state().stream, x, w, y, a,
                       (unsigned)out_e);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // extern "C"
