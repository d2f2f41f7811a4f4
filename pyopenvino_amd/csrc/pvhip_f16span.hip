// The third f16 convolution kernel for FP16 IRs (Convolution.py:57-87 computed in numpy float16 by the reference,
// common_def.py:13-17): "same" convolutions of stride 1 (1x1, 3x3 / pad 1, 5x5 / pad 2 -- every GoogLeNet layer but conv1) with
// C % 16 == 0, fp16 operands on v_mfma_f32_32x32x16_f16, fp32 accumulation, fp32 tensors in HBM.
//
// With the matrix work 16x cheaper than in fp32, what an implicit-GEMM kernel is left with is its im2col tiles: one copy of the
// activation tile per TAP and per channel tile (pvhip_conv2d_f16_dma moves 8-15 TB/s from L2 into LDS on the 3x3 layers).  Here a
// workgroup (4 waves) owns 128 consecutive pixels of ONE image and ALL output channels of a channel group (up to 128), and per stage
// of 16 input channels:
//   1. ONE span of each channel plane comes into LDS by LDS-DMA -- for a stride-1 "same" window the input pixel of tap (r, s) is the
//      output pixel's plane index + (r - pad) * W + (s - pad), so the 128 + 2 * pad * (W + 1) floats around the tile serve every
//      tap: one dense 1-KiB piece per channel;
//   2. the workgroup turns it into the fp16 image the matrix cores want: pixel-major ([pixel][16 channels], 48-byte pixels: a
//      16-byte read per lane is conflict-free), rounded to nearest even ONCE per element (not once per tap), and laid out WITH the
//      zero padding -- rows of W + 2 * pad pixels, the border never written -- so that a tap is a constant shift and needs no test;
//   3. every tap is then one ds_read_b128 per 32 pixels (the eight channels of a lane's MFMA operand) and one MFMA per channel tile.
// The waves split the OUTPUT CHANNELS (wave w: tile w of the group), so no two waves load the same weights; the weights
// never touch LDS: fp16 MFMA fragments packed once ([channel group][stage][tap][tile][lane][8 halves]: one 16-byte load per lane and
// MFMA, the next tap's in flight).  The first version split the PIXELS between the waves, read eight floats and converted them per tap,
// and every wave loaded every weight fragment: 216 KB of weights per workgroup and stage through L1 -- slower than the kernel it replaces.
#include "pvhip_common.h"

using namespace pvhip;

namespace {

typedef float    floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kSC    = 16;           // input channels per stage = the K of one MFMA
constexpr int kSpan  = 256;          // floats of a channel's span = one 1-KiB LDS-DMA piece
constexpr int kRow   = kSpan + 8;    // LDS row of the fp32 span (1056 bytes: column reads of consecutive lanes are conflict-free)
constexpr int kTile  = 128;          // pixels per workgroup
constexpr int kImgPx = 384;          // pixels of the padded fp16 image (rows of W + 2 * pad)
constexpr int kPxH   = 24;           // halves per pixel of that image: 16 channels + 8 of padding (48 bytes)

struct SpanArgs {
    const float*    x;
    const _Float16* wf;      // [n_mtiles][C / 16][taps][tm][64 lanes][8 halves]
    float*          y;
    const float*    bias;
    int N, C, H, W, K;
    int tm;                  // 32-channel tiles per channel group (1..4)
    int tiles_per_image, n_mtiles;
    unsigned x_bytes, wf_bytes;
    int   act;
    float lo, hi;
    int   y_ctotal, y_coff;
    int   abl;               // diagnostic build (PVHIP_CONV_ABLATE bits; results wrong on purpose): 1 no copies, 2 no conversion, 4 no MFMAs, 8 no stores, 16 no weight loads
};

__device__ __forceinline__ void span_dma_b128(__amdgpu_buffer_rsrc_t r, float* dst, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lds = (unsigned)(unsigned long)(lds_ptr_t)dst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds), "v"(voff), "s"(r) : "memory");
#endif
}
__device__ __forceinline__ void span_dma_b32(__amdgpu_buffer_rsrc_t r, float* dst, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lds = (unsigned)(unsigned long)(lds_ptr_t)dst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" :: "s"(lds), "v"(voff), "s"(r) : "memory");
#endif
}
template <int N>
__device__ __forceinline__ void span_dma_wait() {          // until at most N of this wave's copies are in flight
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory");
#endif
}

// TPW: channel tiles per wave (1: groups of up to 4 tiles, 2: up to 8); KS: window (1, 3, 5); VEC: H*W % 4 == 0 (16-byte pieces; else dwords)
// Measured and not kept (scripts/time_f16_span.py, conv2/3x3 at batch 256, 0.53 ms in this form): a fifth wave that only issues the span
// copies, two stages ahead into three buffers (0.65: 69 KB of LDS leave two workgroups per CU); the weight fragments of a whole stage
// loaded a stage ahead instead of a tap ahead (0.68: 72 more registers).  With everything but the loop switched off the launch still
// takes 0.13 ms, without the MFMAs -- and the waits in front of them -- 0.31: what is left to hide is latency, and four waves of a
// workgroup that all wait at the same two barriers per stage do not hide it.
template <int TPW, int KS, bool VEC>
__global__ __launch_bounds__(kBlock) void conv_f16_span_kernel(SpanArgs a) {
    constexpr int TAPS = KS * KS, PAD = (KS - 1) / 2;
    __shared__ __attribute__((aligned(1024))) float    Bs[kSC][kRow];
    __shared__ __attribute__((aligned(16)))   _Float16 Hs[kImgPx][kPxH];

#ifdef PVHIP_DIAG
    const int abl = a.abl;      // scripts/time_f16_span.py: parts switched off (wrong results on purpose)
#else
    constexpr int abl = 0;
#endif
    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid  = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int l31 = lane & 31, lh = lane >> 5;
    const int HW  = a.H * a.W;
    const int ncs = a.C / kSC;
    const int W2  = a.W + 2 * PAD;

    // workgroup -> (channel group, image, tile of the image): the channel groups of a tile back to back
    const int mt  = blockIdx.x % a.n_mtiles;
    const int pt  = blockIdx.x / a.n_mtiles;
    const int img = pt / a.tiles_per_image;
    const int p0  = (pt - img * a.tiles_per_image) * kTile;           // first pixel (plane index) of the tile
    int       s0  = p0 - PAD * a.W - PAD;                             // plane index of span column 0
    if (VEC) s0 &= ~3;                                                 // 16-byte pieces: aligned (H*W % 4 == 0, p0 % 128 == 0)
    const int oy0 = p0 / a.W;
    const int iy0 = oy0 - PAD;                                         // image row of row 0 of the padded fp16 image
    const int nr  = min(a.H - 1, (p0 + kTile - 1) / a.W) - oy0 + 1 + 2 * PAD;      // its rows

    // the fp16 image starts as zeros: the padding and the rows outside the image are never written
    for (int e = tid; e < kImgPx * kPxH / 8; e += kBlock) reinterpret_cast<float4*>(&Hs[0][0])[e] = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- conversion: thread j owns span column j (one pixel, 16 channels); where it goes in the padded image, once
    int dst_px = -1;
    {
        const int q = s0 + tid;
        if (q >= 0 && q < HW) {
            const int iy = q / a.W, ix = q - iy * a.W;
            const int rr = iy - iy0;
            if (rr >= 0 && rr < nr) dst_px = rr * W2 + ix + PAD;
        }
    }
    // ---- this lane's four pixels (one per 32-pixel column block) and the byte address of their top-left tap in the fp16 image
    unsigned pixaddr[4];
    bool     live[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int P = p0 + 32 * n + l31;
        live[n] = P < HW;
        const int oy = live[n] ? P / a.W : oy0, ox = live[n] ? P - oy * a.W : 0;
        pixaddr[n] = (unsigned)(((oy - oy0) * W2 + ox) * kPxH + 8 * lh) * 2u;
    }
    const char* const hs0 = reinterpret_cast<const char*>(&Hs[0][0]);

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.wf), 0, a.wf_bytes, 0x00020000);

    // ---- the span copies of a stage: wave w copies channel rows 4 w .. 4 w + 3, each ONE 1-KiB piece (VEC).  The whole byte offset is
    // in the vector offset (the scalar offset is not range-checked): a start before the tensor wraps to a huge offset and an end behind
    // it is out of range too -- zeros either way; what lies in a neighbouring plane lands in columns the conversion skips.
    const long plane0 = ((long)img * a.C) * HW + s0;                  // float index of column 0 of channel 0 (may be negative)
    const unsigned voff0 = (unsigned)(plane0 * 4) + (unsigned)lane * (VEC ? 16u : 4u);
#define PVS_ISSUE(cs_)                                                                                             \
    {                                                                                                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
            const int      row = 4 * wid + j;                                                                    \
            const unsigned vo  = voff0 + (unsigned)(((cs_) * kSC + row) * HW) * 4u;                              \
            if (abl & 1) continue;                                                                             \
            if (VEC) span_dma_b128(xr, &Bs[row][0], vo);                                                         \
            else {                                                                                               \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) span_dma_b32(xr, &Bs[row][64 * q], vo + 256u * q); \
            }                                                                                                    \
        }                                                                                                        \
    }

    floatx16 acc[TPW][4];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.0f;

    // weight fragments [mt][cs][tap][i][lane][8 halves]; this wave's tiles are i = wid and wid + 4
    const unsigned wlane = (unsigned)lane * 16u;
    bool have[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) have[j] = wid + 4 * j < a.tm && (mt * a.tm + wid + 4 * j) * 32 < a.K;
#define PVS_LOAD_A(dst_, cs_, tap_)                                                                                \
    {                                                                                                            \
        const unsigned so = (unsigned)((((mt * ncs + (cs_)) * TAPS + (tap_)) * a.tm) * 1024);                    \
        _Pragma("unroll") for (int j = 0; j < TPW; ++j)                                                          \
            if (have[j] && !(abl & 16)) dst_[j] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wr, wlane, so + (unsigned)(wid + 4 * j) * 1024u, 0)); \
    }

    PVS_ISSUE(0);
    for (int cs = 0; cs < ncs; ++cs) {
        span_dma_wait<0>();
        __syncthreads();                       // the span of stage cs has landed; every read of the previous fp16 image is done
        if (dst_px >= 0 && !(abl & 2)) {     // fp32 span column -> sixteen fp16 channels of one pixel
            half8 lo8, hi8;
#pragma unroll
            for (int q = 0; q < 8; ++q) { lo8[q] = (_Float16)Bs[q][tid]; hi8[q] = (_Float16)Bs[8 + q][tid]; }
            *reinterpret_cast<half8*>(&Hs[dst_px][0]) = lo8;
            *reinterpret_cast<half8*>(&Hs[dst_px][8]) = hi8;
        }
        __syncthreads();                       // the fp16 image of stage cs is complete, the span buffer is free
        if (cs + 1 < ncs) PVS_ISSUE(cs + 1);
        half8 af[2][TPW];
        PVS_LOAD_A(af[0], cs, 0);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            if (t + 1 < TAPS) PVS_LOAD_A(af[(t + 1) & 1], cs, t + 1);
            const unsigned shift = (unsigned)(((t / KS) * W2 + (t % KS)) * kPxH) * 2u;        // wave-uniform
            half8 b8[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) b8[n] = *reinterpret_cast<const half8*>(hs0 + pixaddr[n] + shift);
#pragma unroll
            for (int j = 0; j < TPW; ++j)
                if (have[j] && !(abl & 4)) {
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t & 1][j], b8[n], acc[j][n], 0, 0, 0);
                }
        }
    }
#undef PVS_ISSUE
#undef PVS_LOAD_A

    // ---- epilogue: register r of accumulator (j, n) is channel (mt * tm + wid + 4 j) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh of pixel 32 n + l31
    const ActBounds ab = act_bounds(a.act, a.lo, a.hi);
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        if (!have[j]) continue;
        const int row0 = (mt * a.tm + wid + 4 * j) * 32 + 4 * lh;
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = row0 + (r & 3) + 8 * (r >> 2);
            bv[r] = (a.bias != nullptr && k < a.K) ? a.bias[k] : -0.0f;
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (!live[n]) continue;
            float* __restrict__ yp = a.y + ((size_t)img * a.y_ctotal + a.y_coff) * HW + p0 + 32 * n + l31;
            float vv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) vv[r] = acc[j][n][r];
            bias_act_n<16>(vv, bv, true, a.act, ab);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = row0 + (r & 3) + 8 * (r >> 2);
                if (k < a.K && !(abl & 8)) conv_store1(yp + (size_t)k * HW, vv[r]);
            }
        }
    }
}

// w (K, C, ks, ks) fp32 -> fp16 fragments [mt][cs][tap][i][lane][q]: channel (mt * TM + i) * 32 + lane % 32, input channel
// cs * 16 + 8 * (lane / 32) + q; channels past K are zero
__global__ __launch_bounds__(kBlock) void conv_f16_span_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ wf, int K, int C,
                                                                    int taps, int tm, size_t total) {
    const int ncs = C / kSC;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(e & 7), lane = (int)((e >> 3) & 63);
        size_t    f = e >> 9;                       // ((mt * ncs + cs) * taps + tap) * tm + i
        const int i = (int)(f % tm);   f /= tm;
        const int tap = (int)(f % taps); f /= taps;
        const int cs = (int)(f % ncs);
        const int mt = (int)(f / ncs);
        const int k = (mt * tm + i) * 32 + (lane & 31), c = cs * kSC + 8 * (lane >> 5) + q;
        wf[e] = k < K ? (_Float16)w[((size_t)k * C + c) * taps + tap] : (_Float16)0.0f;
    }
}

inline int span_mtiles(int k) { return (k + 127) / 128; }      // channel groups of up to 128: one tile per wave (two: 300 registers, one wave per SIMD)
inline int span_tm(int k) { const int t32 = (k + 31) / 32, nm = span_mtiles(k); return (t32 + nm - 1) / nm; }

template <int KS, bool VEC>
void launch_span(const SpanArgs& a, int grid) {
    hipLaunchKernelGGL((conv_f16_span_kernel<1, KS, VEC>), dim3(grid), dim3(kBlock), 0, state().stream, a);
}

}  // namespace

extern "C" {

int pvhip_conv2d_f16_span_supported(int c, int h, int w, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int oh, int ow) {
    if (c <= 0 || c % kSC != 0 || kh != kw || (kh != 1 && kh != 3 && kh != 5) || sh != 1 || sw != 1) return 0;
    const int pad = (kh - 1) / 2;
    if (pad_top != pad || pad_left != pad || oh != h || ow != w || h <= 0 || w <= 0) return 0;
    if (kTile + 2 * pad * (w + 1) + 3 > kSpan) return 0;          // the span of a tile (+ alignment) is one 1-KiB piece
    const int rows = (kTile + w - 2) / w + 1 + 2 * pad;           // rows a 128-pixel tile can touch, + the padding rows
    return rows * (w + 2 * pad) <= kImgPx ? 1 : 0;                // the padded fp16 image of a tile fits its LDS array
}

size_t pvhip_conv2d_f16_span_pack_elems(int k_out, int c, int kh, int kw) {       // FLOATS of the fragment panel
    if (k_out <= 0 || c <= 0 || c % kSC != 0 || kh <= 0 || kw <= 0) return 0;
    const size_t halves = (size_t)span_mtiles(k_out) * (c / kSC) * kh * kw * span_tm(k_out) * 512;
    return (halves + 1) / 2;
}

int pvhip_conv2d_f16_span_pack(const float* w_oihw, float* wf, int k_out, int c, int kh, int kw) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(w_oihw != nullptr && wf != nullptr && k_out > 0 && c > 0 && c % kSC == 0 && kh > 0 && kh == kw && kh <= 5);
    const size_t total = (size_t)span_mtiles(k_out) * (c / kSC) * kh * kw * span_tm(k_out) * 512;
    hipLaunchKernelGGL(conv_f16_span_pack_kernel, dim3(grid_for(total)), dim3(kBlock), 0, state().stream, w_oihw,
                       reinterpret_cast<_Float16*>(wf), k_out, c, kh * kw, span_tm(k_out), total);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_conv2d_f16_span(const float* x, const float* wf, float* y, int n, int c, int h, int w, int k_out, int kh, int kw, int oh, int ow,
                          int sh, int sw, int pad_top, int pad_left, const float* bias, int act, int out_channel_offset,
                          int out_channels_total, float act_lo, float act_hi) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && k_out > 0);
    if (!pvhip_conv2d_f16_span_supported(c, h, w, kh, kw, sh, sw, pad_top, pad_left, oh, ow))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_span: stride-1 \"same\" 1x1 / 3x3 / 5x5 windows with C %% 16 == 0 and rows of at most %d floats", 61);
    PVHIP_CHECK_ARG(out_channels_total == 0 || (out_channel_offset >= 0 && out_channel_offset + k_out <= out_channels_total));
    const unsigned long long in_e = (unsigned long long)n * c * h * w,
                             out_e = (unsigned long long)n * (out_channels_total > 0 ? out_channels_total : k_out) * h * w;
    if (in_e >= (1ull << 29) || out_e >= (1ull << 31)) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_span: input exceeds 2^29 elements or output 2^31");
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && wf != nullptr && y != nullptr);
    SpanArgs a;
    a.x = x; a.wf = reinterpret_cast<const _Float16*>(wf); a.y = y; a.bias = bias;
    a.N = n; a.C = c; a.H = h; a.W = w; a.K = k_out;
    a.tm = span_tm(k_out);
    a.tiles_per_image = (h * w + kTile - 1) / kTile;
    a.n_mtiles = span_mtiles(k_out);
    a.x_bytes  = (unsigned)(in_e * 4ull);
    a.wf_bytes = (unsigned)(pvhip_conv2d_f16_span_pack_elems(k_out, c, kh, kw) * 4);
    a.act = act; a.lo = act_lo; a.hi = act_hi;
    a.y_ctotal = out_channels_total > 0 ? out_channels_total : k_out;
    a.y_coff   = out_channels_total > 0 ? out_channel_offset : 0;
    a.abl = 0;
#ifdef PVHIP_DIAG
    a.abl = settings().conv_ablate;
#endif
    const long grid = (long)n * a.tiles_per_image * a.n_mtiles;
    if (grid > 0x7fffffffL) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_span: grid too large");
    const bool vec = (h * w) % 4 == 0;
    if (kh == 1)      { if (vec) launch_span<1, true>(a, (int)grid); else launch_span<1, false>(a, (int)grid); }
    else if (kh == 3) { if (vec) launch_span<3, true>(a, (int)grid); else launch_span<3, false>(a, (int)grid); }
    else              { if (vec) launch_span<5, true>(a, (int)grid); else launch_span<5, false>(a, (int)grid); }
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // extern "C"
