// f16-MFMA Convolution and MatMul for FP16 IRs: fp16 operands, fp32 accumulation (v_mfma_f32_32x32x16_f16,
// 16x the fp32 MFMA rate), fp32 tensors in HBM.
//
// The reference runs an FP16 IR in numpy float16 (common_def.py:13-17 maps FP16 -> np.float16; Convolution.py:57-87 and
// MatMul.py:9-17 then multiply and ACCUMULATE in float16).  Here the constants of such an IR (exactly representable in
// fp16) and the activations (rounded to fp16 where a Convolution / MatMul reads them, round-to-nearest-even like numpy's
// astype) are the operands of the matrix cores, and the sum is kept in fp32: at least as close to exact arithmetic as
// the reference's own float16 run, against which it is tested at a stated fp16 tolerance.
//
//   conv_f16_kernel   implicit GEMM D[k_out][pixel], any window / stride / padding: a workgroup (4 waves) owns 64 output
//                     channels x 128 pixels; per stage of 32 reduction rows (c-major: row = (c*kh + r)*kw + s) a lane
//                     gathers 16 rows of ITS pixel with range-checked buffer loads (padding = out-of-range offset -> 0.0,
//                     the window bit and byte offset of every row come from a small table through the scalar unit),
//                     rounds them to fp16 and writes them as two 16-byte LDS stores into the pixel-major image
//                     [pixel][32 + 8] halves -- so that an MFMA operand (8 consecutive reduction rows of one pixel) is ONE
//                     conflict-free ds_read_b128; the weight tile [64][32 + 8] comes from a panel packed once in that
//                     layout.  Register-staged double buffering: the gathers of stage t+1 are in flight under the MFMAs
//                     of stage t.
//   pvhip_matmul_f16  MatMul of an FP16 IR: fp16-rounded operands on the split-K fp32-MFMA kernel of pvhip_matmul.hip (exact products,
//                     fp32 accumulation: the arithmetic of the f16 instruction; the FC layers are launch-size bound).
#include <hip/hip_fp16.h>

#include "pvhip_common.h"

using namespace pvhip;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) int* const_int_p;

constexpr int kBK = 32;            // reduction rows per stage = two MFMA steps of 16
constexpr int kLd = kBK + 8;       // LDS row of a pixel / an output channel in halves (80 bytes: b128 reads conflict-free)
constexpr int kCBM = 64, kCBN = 128;
constexpr unsigned kOob = 0x80000000u;

struct ConvF16Args {
    const float*    x;
    const int2*     tab;   // [kred_pad + kBK] {byte offset (c*H*W + r*W + s)*4, window bit r*kw + s}; padding rows: bit 63
    const _Float16* wp;    // [n_mtiles][kred_pad / 32][64][32]
    float*          y;
    const float*    bias;
    int N, C, H, W, K, OH, OW, sh, sw, pt, pl, kh, kw;
    unsigned x_bytes;
    int kred_pad, n_mtiles, P;
    int   act;
    float lo, hi;
    int y_ctotal, y_coff;
};

__device__ __forceinline__ unsigned pack_half2(float a, float b) {
    const half2v h = {(_Float16)a, (_Float16)b};        // v_cvt_f16_f32: round to nearest even, as numpy's astype(float16)
    return __builtin_bit_cast(unsigned, h);
}

__global__ __launch_bounds__(kBlock) void conv_f16_kernel(ConvF16Args a) {
    __shared__ __attribute__((aligned(16))) _Float16 As[2][kCBM][kLd];
    __shared__ __attribute__((aligned(16))) _Float16 Bs[2][kCBN][kLd];

    const int nwg = gridDim.x;
    int       lid;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt    = lid % a.n_mtiles;
    const int ptile = lid / a.n_mtiles;
    const int tid   = threadIdx.x;
    const int lane  = tid & (kWave - 1);
    const int wid   = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- gather role: lane <-> pixel tid % 128, reduction rows 16*(tid / 128) .. +15 of every stage (wave-uniform half)
    const int gpix  = tid & (kCBN - 1);
    const int khalf = __builtin_amdgcn_readfirstlane(tid / kCBN);
    const int OHW = a.OH * a.OW, HW = a.H * a.W;
    unsigned           xoff = 0;
    unsigned long long inb  = 0;
    {
        const int gp = ptile * kCBN + gpix;
        if (gp < a.P) {
            const int n = gp / OHW, rem = gp - n * OHW;
            const int oy = rem / a.OW, ox = rem - oy * a.OW;
            const int ih0 = oy * a.sh - a.pt, iw0 = ox * a.sw - a.pl;
            xoff = (unsigned)(n * a.C * HW + ih0 * a.W + iw0) * 4u;
            for (int r = 0; r < a.kh; ++r)
                for (int s = 0; s < a.kw; ++s)
                    if ((unsigned)(ih0 + r) < (unsigned)a.H && (unsigned)(iw0 + s) < (unsigned)a.W) inb |= 1ull << (r * a.kw + s);
        }
    }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const const_int_p tab = (const_int_p)(unsigned long)a.tab;         // constant address space: scalar loads; {offset, bit} pairs
    const int nk = a.kred_pad / kBK;
    // weight tile of a stage: 64 x 32 halves = 4 KB = one 16-byte load per thread
    const uint4* __restrict__ wsrc = reinterpret_cast<const uint4*>(a.wp + (size_t)mt * nk * (kCBM * kBK)) + tid;
    const int am = tid >> 2, ak = (tid & 3) * 8;                         // its place in the tile: row am, halves ak .. ak+7

    float breg[16];
    uint4 areg;
#define F16_GATHER(kt_)                                                                                      \
    {                                                                                                        \
        const const_int_p e_ = tab + 2 * ((kt_) * kBK + khalf * 16);                                         \
        _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                     \
            const int tx_ = e_[2 * j], ty_ = e_[2 * j + 1];                                                  \
            const unsigned off_ = ((unsigned)(inb >> ty_) & 1u) ? xoff + (unsigned)tx_ : kOob;               \
            breg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off_, 0, 0));       \
        }                                                                                                    \
        areg = wsrc[(size_t)((kt_) < nk ? (kt_) : nk - 1) * (kCBM * kBK / 8)];                               \
    }
#define F16_STORE(buf_)                                                                                      \
    {                                                                                                        \
        uint4 lo_, hi_;                                                                                      \
        lo_.x = pack_half2(breg[0], breg[1]);   lo_.y = pack_half2(breg[2], breg[3]);                        \
        lo_.z = pack_half2(breg[4], breg[5]);   lo_.w = pack_half2(breg[6], breg[7]);                        \
        hi_.x = pack_half2(breg[8], breg[9]);   hi_.y = pack_half2(breg[10], breg[11]);                      \
        hi_.z = pack_half2(breg[12], breg[13]); hi_.w = pack_half2(breg[14], breg[15]);                      \
        *reinterpret_cast<uint4*>(&Bs[buf_][gpix][khalf * 16]) = lo_;                                        \
        *reinterpret_cast<uint4*>(&Bs[buf_][gpix][khalf * 16 + 8]) = hi_;                                    \
        *reinterpret_cast<uint4*>(&As[buf_][am][ak]) = areg;                                                 \
    }

    floatx16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    F16_GATHER(0);
    F16_STORE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        F16_GATHER(kt + 1);            // past the end: the table's spare stage of padding rows (-> 0.0), the last weight tile again
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const half8 bf = *reinterpret_cast<const half8*>(&Bs[buf][wid * 32 + l31][st * 16 + lh * 8]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const half8 af = *reinterpret_cast<const half8*>(&As[buf][i * 32 + l31][st * 16 + lh * 8]);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[i], 0, 0, 0);
            }
        }
        F16_STORE(buf ^ 1);
        __syncthreads();
    }
#undef F16_GATHER
#undef F16_STORE

    // ---- epilogue: accumulator register r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
    const int gp = ptile * kCBN + wid * 32 + l31;
    if (gp >= a.P) return;
    const int n = gp / OHW, rem = gp - n * OHW;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row0 = mt * kCBM + i * 32 + 4 * lh;
        float* __restrict__ yp = a.y + ((size_t)n * a.y_ctotal + a.y_coff + row0) * OHW + rem;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (row0 + dr < a.K) {
                float v = acc[i][r];
                if (a.bias != nullptr) v = v + a.bias[row0 + dr];
                if (a.act == 1) v = (v < 0.0f) ? 0.0f : v;
                else if (a.act == 2) { v = (v < a.lo) ? a.lo : v; v = (v > a.hi) ? a.hi : v; }
                yp[(size_t)dr * OHW] = v;
            }
        }
    }
}

// wpack layout: [tab: (kred_pad + 32) int2] [panel: n_mtiles * (kred_pad / 32) * 64 * 32 halves]
__global__ __launch_bounds__(kBlock) void conv_f16_pack_kernel(const float* __restrict__ w, int2* __restrict__ tab, _Float16* __restrict__ wp,
                                                                int K, int C, int kh, int kw, int H, int W, int kred, int kred_pad,
                                                                int n_mtiles) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int    nk     = kred_pad / kBK;
    const size_t total  = (size_t)n_mtiles * nk * kCBM * kBK;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int kk = (int)(e % kBK), m = (int)((e / kBK) % kCBM);
        const size_t ts = e / (kBK * kCBM);
        const int st = (int)(ts % nk), mt = (int)(ts / nk);
        const int ko = mt * kCBM + m, kr = st * kBK + kk;
        wp[e] = (ko < K && kr < kred) ? (_Float16)w[(size_t)ko * kred + kr] : (_Float16)0.0f;     // OIHW: (c, r, s) flat == kr
    }
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)(kred_pad + kBK); e += stride) {
        const int i = (int)e;
        int2 v = make_int2(0, 63);                      // padding row: a window bit that is never set
        if (i < kred) {
            const int s = i % kw, t = i / kw, r = t % kh, c = t / kh;
            v = make_int2((c * H * W + r * W + s) * 4, r * kw + s);
        }
        tab[i] = v;
    }
}

inline int round_up_int(int v, int q) { return (v + q - 1) / q * q; }

}  // namespace

extern "C" {

size_t pvhip_conv2d_f16_pack_elems(int k_out, int c, int kh, int kw) {
    if (k_out <= 0 || c <= 0 || kh <= 0 || kw <= 0) return 0;
    const size_t kred_pad = (size_t)round_up_int(c * kh * kw, kBK), n_mtiles = (size_t)(k_out + kCBM - 1) / kCBM;
    return 2 * (kred_pad + kBK) + n_mtiles * kred_pad * kCBM / 2;       // in FLOATS: the int2 table, then the half panel
}

int pvhip_conv2d_f16_pack(const float* w_oihw, float* wpack, int k_out, int c, int kh, int kw, int h, int w) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(w_oihw != nullptr && wpack != nullptr && k_out > 0 && c > 0 && kh > 0 && kw > 0 && h > 0 && w > 0);
    if (kh * kw >= 63 || (unsigned long long)c * h * w >= (1ull << 29))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_pack: window %dx%d / plane C=%d H=%d W=%d outside the table encoding", kh, kw, c, h, w);
    const int kred = c * kh * kw, kred_pad = round_up_int(kred, kBK), n_mtiles = (k_out + kCBM - 1) / kCBM;
    int2*     tab = reinterpret_cast<int2*>(wpack);
    _Float16* wp  = reinterpret_cast<_Float16*>(wpack + 2 * (kred_pad + kBK));
    hipLaunchKernelGGL(conv_f16_pack_kernel, dim3(grid_for((size_t)n_mtiles * kred_pad * kCBM)), dim3(kBlock), 0, state().stream, w_oihw, tab,
                       wp, k_out, c, kh, kw, h, w, kred, kred_pad, n_mtiles);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_conv2d_f16(const float* x, const float* wpack, float* y, int n, int c, int h, int w, int k_out, int kh, int kw, int oh, int ow,
                     int sh, int sw, int pad_top, int pad_left, const float* bias, int act, int out_channel_offset, int out_channels_total,
                     float act_lo, float act_hi) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && c > 0 && h > 0 && w > 0 && k_out > 0 && kh > 0 && kw > 0 && oh >= 0 && ow >= 0);
    PVHIP_CHECK_ARG(sh > 0 && sw > 0 && pad_top >= 0 && pad_left >= 0);
    PVHIP_CHECK_ARG(out_channels_total == 0 || (out_channel_offset >= 0 && out_channel_offset + k_out <= out_channels_total));
    const unsigned long long in_e  = (unsigned long long)n * c * h * w,
                             out_e = (unsigned long long)n * (out_channels_total > 0 ? out_channels_total : k_out) * oh * ow;
    if (kh * kw >= 63 || in_e >= (1ull << 29) || out_e >= (1ull << 31))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16: window of 63+ taps, input of 2^29+ elements or output of 2^31+");
    if (out_e == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && wpack != nullptr && y != nullptr);
    ConvF16Args a;
    a.kred_pad = round_up_int(c * kh * kw, kBK);
    a.n_mtiles = (k_out + kCBM - 1) / kCBM;
    a.x = x; a.tab = reinterpret_cast<const int2*>(wpack);
    a.wp = reinterpret_cast<const _Float16*>(wpack + 2 * (a.kred_pad + kBK));
    a.y = y; a.bias = bias;
    a.N = n; a.C = c; a.H = h; a.W = w; a.K = k_out; a.OH = oh; a.OW = ow;
    a.sh = sh; a.sw = sw; a.pt = pad_top; a.pl = pad_left; a.kh = kh; a.kw = kw;
    a.x_bytes = (unsigned)(in_e * 4ull);
    a.P = n * oh * ow;
    a.act = act; a.lo = act_lo; a.hi = act_hi;
    a.y_ctotal = out_channels_total > 0 ? out_channels_total : k_out;
    a.y_coff   = out_channels_total > 0 ? out_channel_offset : 0;
    const long grid = (long)((a.P + kCBN - 1) / kCBN) * a.n_mtiles;
    hipLaunchKernelGGL(conv_f16_kernel, dim3((unsigned)grid), dim3(kBlock), 0, state().stream, a);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_matmul_f16(const float* a, const float* b, float* c, int m, int n, int k, int trans_a, int trans_b) {
    // fp16-rounded operands on the split-K kernel of pvhip_matmul.hip (see matmul_kernel<true>): the FC layers of the IRs are launch-size
    // bound, and a product of two fp16 values is exact in fp32 -- the arithmetic of the f16 MFMA, six times faster than the tile kernel
    // of round 2 (v_mfma_f32_32x32x16_f16, 64x64 tiles, no split) at (256, 1024) x (1000, 1024)^T: 0.027 against 0.175 ms.
    return pvhip::matmul_impl(a, b, c, m, n, k, trans_a, trans_b, 1);
}

}  // extern "C"
