// FP16 IRs: the first convolution of an image network as an f16 layer -- 7x7 / stride 2 over THREE channels (GoogLeNet's
// conv1; Convolution.py:57-87 computed in numpy float16 by the reference, common_def.py:13-17) -- from ROW SPANS instead of an im2col gather.
// The implicit-GEMM forms gather 4-byte pieces: every input value 12 times over, 320 copy instructions per 128 output pixels, and in an f16
// kernel the copy instructions are the time (LESSONS.md lesson 53).  Here a workgroup owns two output rows of one image, i.e. nine rows of the
// zero-padded input per channel (the padding pass of the plugin has written them, data/mean added, rows of WP = W + 8 floats: 16-byte
// pieces): 27 one-KiB LDS-DMA instructions bring the tile in, once.  The reduction axis is laid out for the reads, not for the tensor:
// a k-slot group of eight = one FILTER ROW (seven taps + one slot of zero weight), a 16-wide MFMA step = two filter rows -- the lane half
// selects the row -- so the operand of a lane is eight consecutive floats of one LDS row: three ds_read_b64 and one ds_read_b32 (stride 2
// keeps the first tap 8-byte aligned), rounded to fp16 (nearest even) on the way.  21 filter rows = 11 steps (10.5: the last half step
// multiplies zeros).  Weights: fp16 fragments in that k order straight from L2, one step ahead; no copy is in flight then.  Output: fp16
// blocked by eight channels, as the MaxPool + LRN launch behind it reads it.
#include "pvhip_common.h"

using namespace pvhip;

namespace {

typedef float    floatx16 __attribute__((ext_vector_type(16)));
typedef float    float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int kC = 3, kKH = 7, kKW = 7, kST = 2;
constexpr int kR = 2;                               // output rows per workgroup
constexpr int kRows = kST * (kR - 1) + kKH;         // input rows per channel: 9
constexpr int kLdsRow = 256;                        // floats per LDS row: one copy instruction (64 lanes x 16 bytes) never reaches into the next row
constexpr int kFRows = kC * kKH;                    // filter rows: 21
constexpr int kSteps = (kFRows + 1) / 2;            // MFMA steps: 11
constexpr unsigned kOob = 0x80000000u;

struct StemArgs {
    const float*    xp;      // zero-padded input [N][3][HP][WP], WP % 4 == 0
    const _Float16* wf;      // [steps][tm][64 lanes][8 halves]
    _Float16*       yb;      // fp16 c8 [N][ceil16(K) / 8][OH * OW][8]
    const float*    bias;
    const float*    pre_add; // DIRECT: one constant per input channel, added to the image (not its padding) in LDS; or null
    int N, HP, WP, OH, OW, K, tm;      // DIRECT: HP, WP = the extents of the UNPADDED image
    int tiles_per_image;
    unsigned x_bytes, wf_bytes;
    int act;
};

// DIRECT (round 5, as conv_stem_f32_kernel of pvhip_stem.hip): the image is read as it is -- no padding pass.  A row's copy lands FOUR floats into its
// LDS row (position p = image column p - 4): positions 1 .. 3 are the left padding, zeroed by the workgroup before the copies are issued (the only
// other writer is lane 63 of the row above, whose piece lies past the image: zeros); lanes past the image write the right padding, rows outside
// the image are out-of-range copies.  The Add in front of the layer (data/mean) happens in LDS, on the image's own positions, by the wave that copied.
template <bool DIRECT>
__global__ __launch_bounds__(kBlock, 4) void conv_f16_stem_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(1024))) float stem_lds[];          // [3 * 9][256] (+ 4 floats: the last row's lane 63 in the DIRECT form)
    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid  = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int l31 = lane & 31, lh = lane >> 5;
    const int img = blockIdx.x / a.tiles_per_image;
    const int oy0 = (blockIdx.x - img * a.tiles_per_image) * kR;
    const int npx = min(kR, a.OH - oy0) * a.OW;
    const int ohw = a.OH * a.OW;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xp), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.wf), 0, a.wf_bytes, 0x00020000);

    // ---- this wave's part: channel tile ct, pixel blocks (wid >> 1) + 2 j
    const int  ct   = wid & 1;
    const bool have = ct < a.tm;
    const unsigned wlane = (unsigned)lane * 16u + (unsigned)ct * 1024u;
#define PVST_LOAD_A(dst_, t_) dst_ = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wr, have ? wlane : kOob, (unsigned)((t_) * a.tm) * 1024u, 0))
    half8 af_cur, af_nxt;
    PVST_LOAD_A(af_cur, 0);                            // (issued in front of the copies: older in the queue, landed when they are)

    // ---- the copies: instruction i = (channel, row), wave w takes i = w, w + 4, ...; lane l: floats 4 l .. 4 l + 3 of the padded row
    {
        const int  iy0   = oy0 * kST - (DIRECT ? 3 : 0);
        const bool colok = lane * 4 < a.WP;
        if (DIRECT) {                   // the left padding of every row (and row 0's position 0 .. 3, which no lane 63 writes)
            if (tid < kC * kRows) *reinterpret_cast<float4*>(stem_lds + tid * kLdsRow) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            __syncthreads();
        }
        for (int i = wid; i < kC * kRows; i += kBlock / kWave) {
            const int  c = i / kRows, rr = i - c * kRows;
            const int  iy = iy0 + rr;
            const bool ok = colok && (unsigned)iy < (unsigned)a.HP;
            const unsigned vo = ok ? (unsigned)((((img * kC + c) * a.HP + iy) * a.WP + lane * 4) * 4) : kOob;
            lds_dma_b128(xr, stem_lds + i * kLdsRow + (DIRECT ? 4 : 0), vo, 0u);
        }
        lds_dma_wait_all();
        if (DIRECT && a.pre_add != nullptr) {      // the rows this wave copied have landed: the Add, image positions only
            for (int i = wid; i < kC * kRows; i += kBlock / kWave) {
                const int c = i / kRows, rr = i - c * kRows;
                if ((unsigned)(iy0 + rr) < (unsigned)a.HP && colok) {
                    const float mc = a.pre_add[c];
                    float4* const p4 = reinterpret_cast<float4*>(stem_lds + i * kLdsRow + 4 + lane * 4);
                    float4 v = *p4;
                    v.x += mc; v.y += mc; v.z += mc; v.w += mc;
                    *p4 = v;
                }
            }
        }
    }
    __syncthreads();

    // ---- pixel geometry: tile pixel p = (row p / OW, column p % OW) reads LDS row 2 (p / OW) + r, columns 2 (p % OW) + s
    unsigned pixoff[4];
    bool     live[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = 32 * ((wid >> 1) + 2 * j) + l31;
        live[j] = p < npx;
        const int pc = live[j] ? p : 0;
        const int pr = pc / a.OW, px = pc - pr * a.OW;
        pixoff[j] = (unsigned)((pr * kST * kLdsRow + px * kST) * 4);     // DIRECT: position 2 px = image column 2 px - 4, one IN FRONT of the first tap: slot 0 carries a zero weight there
    }
    floatx16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    const char* const lds_b = reinterpret_cast<const char*>(stem_lds);

#pragma unroll
    for (int t = 0; t < kSteps; ++t) {
        if (t + 1 < kSteps) PVST_LOAD_A(af_nxt, t + 1);
        // filter rows 2 t (lane half 0) and 2 t + 1 (half 1): LDS row (channel * 9 + r); the slot past the last filter row reads row 20 again (zero weights)
        const int j0 = 2 * t, j1 = (2 * t + 1 < kFRows) ? 2 * t + 1 : kFRows - 1;
        const unsigned off0 = (unsigned)(((j0 / kKH) * kRows + (j0 % kKH)) * kLdsRow * 4), off1 = (unsigned)(((j1 / kKH) * kRows + (j1 % kKH)) * kLdsRow * 4);
        const unsigned roff = lh ? off1 : off0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char* const p0 = lds_b + pixoff[j] + roff;
            const float2v v01 = *reinterpret_cast<const float2v*>(p0), v23 = *reinterpret_cast<const float2v*>(p0 + 8), v45 = *reinterpret_cast<const float2v*>(p0 + 16);
            half8 b8;
            b8[0] = (_Float16)v01.x; b8[1] = (_Float16)v01.y; b8[2] = (_Float16)v23.x; b8[3] = (_Float16)v23.y;
            b8[4] = (_Float16)v45.x; b8[5] = (_Float16)v45.y;
            if (DIRECT) {          // eight floats from an 8-byte boundary: [the column in front of the window (zero weight), taps 0 .. 6]
                const float2v v67 = *reinterpret_cast<const float2v*>(p0 + 24);
                b8[6] = (_Float16)v67.x; b8[7] = (_Float16)v67.y;
            } else {               // [taps 0 .. 6, tap 6 again (zero weight)]
                const float v6 = *reinterpret_cast<const float*>(p0 + 24);
                b8[6] = (_Float16)v6; b8[7] = (_Float16)v6;
            }
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af_cur, b8, acc[j], 0, 0, 0);
        }
        af_cur = af_nxt;
    }
#undef PVST_LOAD_A
    if (!have) return;

    // ---- epilogue: register 4 g + j of accumulator jb is channel 32 ct + 8 g + 4 lh + j of the block's pixel l31: half a blocked piece
    typedef const __attribute__((address_space(4))) float* const_float_p;
    const const_float_p bias_c = (const_float_p)(unsigned long)a.bias;
    const int cbt = (a.K + 15) / 16 * 2;
    const int P0  = oy0 * a.OW + l31;
    // two register groups at a time, v_permlane32_swap between the lane halves: a lane of the lower half ends up with all eight channels of
    // block 4 ct + 2 gp, its partner in the upper half with those of the next block -- one 16-byte store per lane (see conv_f16_c8m_kernel)
    typedef unsigned uint4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
        float bs0[2][4], bs1[2][4];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bs0[h2][j] = bs1[h2][j] = -0.0f;
                if (a.bias != nullptr) {
                    const int k0 = 32 * ct + 8 * (2 * gp + h2) + j;
                    bs0[h2][j] = k0 < a.K ? bias_c[k0] : 0.0f;
                    bs1[h2][j] = k0 + 4 < a.K ? bias_c[min(k0 + 4, a.K - 1)] : 0.0f;
                }
            }
        const int  blk  = 4 * ct + 2 * gp + lh;
        const bool b_ok = blk < cbt;
        _Float16* const yb = a.yb + (((size_t)img * cbt + (b_ok ? blk : 0)) * ohw + P0) * 8;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            unsigned w0[2], w1[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                half4 hv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[jb][4 * (2 * gp + h2) + j] + (lh ? bs1[h2][j] : bs0[h2][j]);
                    if (a.act != 0) v = (v < 0.0f) ? 0.0f : v;
                    hv[j] = (_Float16)v;
                }
                const uint2 u = __builtin_bit_cast(uint2, hv);
                if (h2 == 0) { w0[0] = u.x; w0[1] = u.y; } else { w1[0] = u.x; w1[1] = u.y; }
            }
            const auto s0 = __builtin_amdgcn_permlane32_swap(w0[0], w1[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(w0[1], w1[1], false, false);
            uint4v piece;
            piece[0] = s0[0]; piece[1] = s1[0]; piece[2] = s0[1]; piece[3] = s1[1];
            if (b_ok && live[jb]) *reinterpret_cast<uint4v*>(yb + (size_t)(32 * ((wid >> 1) + 2 * jb)) * 8) = piece;
        }
    }
}

// w (K, 3, 7, 7) fp32 -> fp16 fragments [step t][tile i][lane][q]: channel 32 i + lane % 32; k slot (t, lane / 32, q) = filter row
// j = 2 t + lane / 32 = (input channel j / 7, window row j % 7), tap q; q = 7 and j = 21: zero
// direct: k slot q = tap q - 1 (slot 0: zero) -- the operand of the DIRECT form starts one column in front of the window
__global__ __launch_bounds__(kBlock) void conv_f16_stem_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ wf, int K, int tm, size_t total, int direct) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(e & 7), lane = (int)((e >> 3) & 63);
        const size_t f = e >> 9;
        const int i = (int)(f % tm), t = (int)(f / tm);
        const int k = 32 * i + (lane & 31), j = 2 * t + (lane >> 5);
        float v = 0.0f;
        const int tap = direct ? q - 1 : q;
        if (k < K && j < kFRows && tap >= 0 && tap < kKW) v = w[(((size_t)k * kC + j / kKH) * kKH + j % kKH) * kKW + tap];
        wf[e] = (_Float16)v;
    }
}

}  // namespace

extern "C" {

/* GoogLeNet's conv1 as an FP16 layer: 7x7 / stride 2 / pad 3 over 3 channels, at most 64 output channels, rows of at most 128 output pixels. */
int pvhip_conv2d_f16_stem_supported(int c, int h, int w, int k_out, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int oh, int ow) {
    if (c != kC || kh != kKH || kw != kKW || sh != kST || sw != kST || pad_top != 3 || pad_left != 3 || k_out <= 0 || k_out > 64) return 0;
    if (h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || kR * ow > 256) return 0;
    const int wp = (kST * (ow - 1) + kKW + 3) / 4 * 4;          // floats of a padded row: every tap of the last output column, whole 16-byte pieces
    if (wp > kLdsRow || wp < w + 3) return 0;
    return wp;
}

size_t pvhip_conv2d_f16_stem_pack_elems(int k_out) {            // FLOATS of the fragment panel
    if (k_out <= 0 || k_out > 64) return 0;
    return (size_t)kSteps * ((k_out + 31) / 32) * 512 / 2;
}

static int stem_pack(const float* w_oihw, float* wf, int k_out, int direct) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(w_oihw != nullptr && wf != nullptr && k_out > 0 && k_out <= 64);
    const int tm = (k_out + 31) / 32;
    const size_t total = (size_t)kSteps * tm * 512;
    hipLaunchKernelGGL(conv_f16_stem_pack_kernel, dim3(grid_for(total)), dim3(kBlock), 0, state().stream, w_oihw, reinterpret_cast<_Float16*>(wf), k_out, tm, total, direct);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_conv2d_f16_stem_pack(const float* w_oihw, float* wf, int k_out) { return stem_pack(w_oihw, wf, k_out, 0); }
int pvhip_conv2d_f16_stem_direct_pack(const float* w_oihw, float* wf, int k_out) { return stem_pack(w_oihw, wf, k_out, 1); }

/* xp: the zero-padded input (n, 3, hp, wp) fp32 with hp >= 2 (oh - 1) + 7 rows and wp = _supported()'s answer floats per row (pvhip_pad2d_f32
 * with pad_top = pad_left = 3 and the bottom / right padding that makes those extents); yb: fp16 c8 output; act: none or ReLU.           */
int pvhip_conv2d_f16_stem(const float* xp, const float* wf, void* yb, int n, int hp, int wp, int k_out, int oh, int ow, const float* bias, int act) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && k_out > 0 && k_out <= 64 && oh > 0 && ow > 0 && (act == 0 || act == 1));
    PVHIP_CHECK_ARG(wp % 4 == 0 && wp <= kLdsRow && wp >= kST * (ow - 1) + kKW && hp >= kST * (oh - 1) + kKH && kR * ow <= 256);
    const unsigned long long in_b = (unsigned long long)n * kC * hp * wp * 4ull;
    if (in_b >= (1ull << 31) || (unsigned long long)n * 64 * oh * ow >= (1ull << 31)) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_stem: tensor too large");
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(xp != nullptr && wf != nullptr && yb != nullptr);
    StemArgs a;
    a.xp = xp; a.wf = reinterpret_cast<const _Float16*>(wf); a.yb = static_cast<_Float16*>(yb); a.bias = bias; a.pre_add = nullptr;
    a.N = n; a.HP = hp; a.WP = wp; a.OH = oh; a.OW = ow; a.K = k_out; a.tm = (k_out + 31) / 32;
    a.tiles_per_image = (oh + kR - 1) / kR;
    a.x_bytes  = (unsigned)in_b;
    a.wf_bytes = (unsigned)(pvhip_conv2d_f16_stem_pack_elems(k_out) * 4);
    a.act = act;
    const long grid = (long)n * a.tiles_per_image;
    if (grid > 0x7fffffffL) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_stem: grid too large");
    hipLaunchKernelGGL(conv_f16_stem_kernel<false>, dim3((unsigned)grid), dim3(kBlock), (size_t)kC * kRows * kLdsRow * sizeof(float), state().stream, a);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

/* The same layer straight from the UNPADDED image x (n, 3, h, w), w % 4 == 0 and w <= 248 (ABI v15, as pvhip_conv2d_stem_direct_f32): no padding
 * pass; wf from pvhip_conv2d_f16_stem_direct_pack (the k slots of a filter row start one column in front of the window); pre_add: one constant per
 * input channel added to the image on the way, or NULL.                                                                                       */
int pvhip_conv2d_f16_stem_direct_supported(int c, int h, int w, int k_out, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int oh, int ow) {
    if (pvhip_conv2d_f16_stem_supported(c, h, w, k_out, kh, kw, sh, sw, pad_top, pad_left, oh, ow) <= 0) return 0;
    if (w % 4 != 0 || w + 8 > kLdsRow) return 0;
    if (kST * (oh - 1) + kKH > h + 6 || kST * (ow - 1) + kKW > w + 6) return 0;        // (pads_end = 3 as well)
    return 1;
}

int pvhip_conv2d_f16_stem_direct(const float* x, const float* wf, void* yb, int n, int h, int w, int k_out, int oh, int ow, const float* pre_add,
                                 const float* bias, int act) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && k_out > 0 && k_out <= 64 && h > 0 && w > 0 && oh > 0 && ow > 0 && (act == 0 || act == 1));
    if (!pvhip_conv2d_f16_stem_direct_supported(kC, h, w, k_out, kKH, kKW, kST, kST, 3, 3, oh, ow))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_stem_direct: 7x7 / 2 / pad 3 over three channels, rows of a multiple of four and at most 248 pixels");
    const unsigned long long in_b = (unsigned long long)n * kC * h * w * 4ull;
    if (in_b >= (1ull << 31) || (unsigned long long)n * 64 * oh * ow >= (1ull << 31)) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_stem_direct: tensor too large");
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && wf != nullptr && yb != nullptr);
    StemArgs a;
    a.xp = x; a.wf = reinterpret_cast<const _Float16*>(wf); a.yb = static_cast<_Float16*>(yb); a.bias = bias; a.pre_add = pre_add;
    a.N = n; a.HP = h; a.WP = w; a.OH = oh; a.OW = ow; a.K = k_out; a.tm = (k_out + 31) / 32;
    a.tiles_per_image = (oh + kR - 1) / kR;
    a.x_bytes  = (unsigned)in_b;
    a.wf_bytes = (unsigned)(pvhip_conv2d_f16_stem_pack_elems(k_out) * 4);
    a.act = act;
    const long grid = (long)n * a.tiles_per_image;
    if (grid > 0x7fffffffL) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_stem_direct: grid too large");
    hipLaunchKernelGGL(conv_f16_stem_kernel<true>, dim3((unsigned)grid), dim3(kBlock), (size_t)kC * kRows * kLdsRow * sizeof(float) + 16, state().stream, a);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // extern "C"
