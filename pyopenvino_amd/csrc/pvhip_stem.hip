// The first convolution of an image network in fp32 -- 7x7 / stride 2 / pad 3 over THREE channels, at most 64 output channels (GoogLeNet's
// conv1; Convolution.py:57-87) -- from ROW SPANS of the zero-padded image instead of an im2col gather, with the WHOLE weight tensor resident in
// registers and NO vector-ALU instruction in the reduction loop.
//
// Why (round 5; profiles/r03_issue_mix.md, scripts/time_conv1.py): the fp32 matrix instruction is the vector FMA datapath, so every vector
// instruction of a convolution kernel is matrix time lost, and in the general LDS-DMA kernel conv1's gathers (320 four-byte copy
// instructions per 128 output pixels), weight-image copies and two barriers per 16-row stage each cost 0.07-0.10 ms that did not hide behind
// its 0.44 ms of MFMAs (0.64 ms, 0.60 of the peak).  Here:
//   * a workgroup (8 waves, ONE per CU, persistent) owns tiles of four output rows of one image: 13 rows x 3 channels of the padded input
//     (the plugin's padding pass has written them, data/mean added, rows of WP <= 256 floats) = 39 one-KiB LDS-DMA instructions per tile --
//     issued for tile t + 1 while tile t is computed (two 39 KB buffers), waited for in front of the epilogue: ONE barrier per tile.
//     (Two workgroups of four waves and two-row tiles per CU, half a tile apart so that one's stores run beside the other's MFMAs, were
//     built and measured: 0.70 ms against 0.52 -- the copies of a starting tile queue behind the other workgroup's stores.)
//   * wave (row wr, channel half hf) computes output row oy0 + wr x 32 channels as D[pixel][channel] on v_mfma_f32_16x16x4_f32: seven
//     16-pixel groups x two 16-channel tiles = 56 accumulator registers;
//   * reduction axis = the 147 taps (c, r, s) in the reference's own order (c-major: the bits of the general kernel's ascending chain),
//     four taps per MFMA step, 37 steps (one slot of zero weight).  Lane (pixel ox = lane & 15, slot kq = lane >> 4) reads tap 4 m + kq of
//     its pixel: ONE ds_read_b32 at [per-step lane offset] + [immediate: pixel group, buffer] serves the two MFMAs of a step and group --
//     the 37 lane offsets (which row, which tap column: a step may straddle a filter row or a channel) are made once per kernel;
//   * the weights of the wave's 32 channels x 148 slots live in 74 registers for the whole launch (fragment order, packed once);
//   * epilogue: a lane holds four consecutive pixels of one channel per accumulator tile: bias, activation, one 16-byte store.
#include "pvhip_common.h"

using namespace pvhip;

namespace {

typedef float    floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

constexpr int kC = 3, kKH = 7, kKW = 7, kST = 2, kPad = 3;
constexpr int kTR    = 4;                            // output rows per tile: one per pair of waves
constexpr int kRows  = kST * (kTR - 1) + kKH;        // input rows per channel and tile: 13
constexpr int kLdsRow = 256;                         // floats per LDS row = one copy instruction (64 lanes x 16 bytes)
constexpr int kCopies = kC * kRows;                  // 39 per tile
constexpr int kBufBytes = kCopies * kLdsRow * 4;     // 39936
constexpr int kTaps  = kC * kKH * kKW;               // 147
constexpr int kSteps = (kTaps + 3) / 4;              // 37
constexpr int kNG    = 7;                            // 16-pixel groups per output row: rows of at most 112 pixels
constexpr int kWaves = 2 * kTR;                      // 8: two per SIMD
constexpr int kThreads = kWaves * kWave;             // 512; ONE workgroup per CU
constexpr unsigned kOob = 0x80000000u;

struct StemArgs {
    const float* xp;       // zero-padded input [N][3][HP][WP], WP % 4 == 0
    const float* wf;       // [2 halves][37 steps][2 tiles][64 lanes]
    float*       y;        // [N][K][OH][OW]
    const float* bias;
    const float* pre_add;  // DIRECT: one constant per input channel, added to the image (not to its padding) in LDS; or null
    int N, HP, WP, OH, OW, K;      // DIRECT: HP, WP = the extents of the UNPADDED image (H, W; W % 4 == 0)
    int tiles_per_image, tiles;
    unsigned x_bytes, y_bytes;
    int act;
    float act_lo, act_hi;
};

__device__ __forceinline__ float lds_read_f32(unsigned addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) float*>(addr);
}

// ABL (diagnostic build only; 1-3 wrong on purpose): 1 = no bias / activation / stores (one store per wave keeps the sums alive), 2 = no MFMAs, 3 = no copies after the first tile,
// 4 = the stores of an instruction go to consecutive 16-byte pieces (lane l: piece l of a KiB; the WRONG places): what coalesced stores would cost
// DIRECT: the image is read as it is -- no padding pass.  A row's copy lands FOUR floats into its LDS row (position p = image column p - 4): the
// three positions in front of it are the left padding and stay zero (the kernel zeroes LDS once; the only other writer is lane 63 of the row
// above, whose piece lies past the image: out of range, zeros), lanes past the image write the right padding's zeros, rows above / below
// the image are out-of-range copies: zeros.  The per-channel constant of an Add in front of the layer (data/mean) is added IN LDS, by the wave
// that copied the row, to the image's own positions only -- ten packed adds per wave and tile, outside the reduction loop.
template <int ABL, bool DIRECT>
__global__ __launch_bounds__(kThreads, 1) void conv_stem_f32_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(1024))) float stem_lds[];          // [2][39][256] (+ 4 floats: the last row's lane 63 in the DIRECT form)
    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid  = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int hf = wid & 1, wr = wid >> 1;
    const int l15 = lane & 15, kq = lane >> 4;

    // this workgroup's tiles: a contiguous range (the rows of one image follow each other: their halo rows come out of this CU's L2)
    const int g_ = (int)gridDim.x, b_ = (int)blockIdx.x;
    const int per = a.tiles / g_, extra = a.tiles - per * g_;
    int       tile = b_ * per + min(b_, extra);
    const int tile_end = tile + per + (b_ < extra ? 1 : 0);
    if (tile >= tile_end) return;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xp), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);

    // ---- the weights of this wave's 32 channels: 74 registers, loaded once
    float w[kSteps][2];
    {
        const float* const wp = a.wf + (size_t)hf * kSteps * 2 * kWave + lane;
#pragma unroll
        for (int m = 0; m < kSteps; ++m) {
            w[m][0] = wp[(m * 2 + 0) * kWave];
            w[m][1] = wp[(m * 2 + 1) * kWave];
        }
    }
    // ---- LDS byte offset of this lane's operand at step m (buffer 0, pixel group 0): tap t = 4 m + kq = (channel c, window row r, column s)
    // lies in LDS row c * 13 + 2 wr + r at column 2 ox + s.  The slot past the last tap re-reads tap 146 (its weight is zero).
    const unsigned lds0 = (unsigned)(unsigned long)(lds_void_p)stem_lds;
    unsigned toff[kSteps];
#pragma unroll
    for (int m = 0; m < kSteps; ++m) {
        const int t = min(4 * m + kq, kTaps - 1);
        const int j = t / kKW, s = t - j * kKW;
        const int c = j / kKH, r = j - c * kKH;
        toff[m] = lds0 + (unsigned)(((c * kRows + kST * wr + r) * kLdsRow + kST * l15 + s + (DIRECT ? 1 : 0)) * 4);
    }
    float bias_l[2] = {0.0f, 0.0f};
    const int ch0 = 32 * hf + l15;
    if (a.bias != nullptr) {
        bias_l[0] = ch0 < a.K ? a.bias[ch0] : 0.0f;
        bias_l[1] = ch0 + 16 < a.K ? a.bias[ch0 + 16] : 0.0f;
    }
    const ActBounds ab = act_bounds(a.act, a.act_lo, a.act_hi);
    const bool colok = lane * 4 < a.WP;

    // ---- the copies of one tile: instruction i = (channel, row), wave w takes i = w, w + 8, ...; lane l: floats 4 l .. 4 l + 3 of the (padded) row
    auto issue = [&](int tl, int buf) {
        const int img = tl / a.tiles_per_image;
        const int iy0 = (tl - img * a.tiles_per_image) * (kTR * kST) - (DIRECT ? kPad : 0);
#pragma unroll
        for (int i0 = 0; i0 < kCopies; i0 += kWaves) {
            const int i = i0 + wid;
            if (i < kCopies) {
                const int  c = i / kRows, rr = i - c * kRows;
                const int  iy = iy0 + rr;
                const bool ok = colok && (unsigned)iy < (unsigned)a.HP;
                const unsigned vo = ok ? (unsigned)((((img * kC + c) * a.HP + iy) * a.WP + lane * 4) * 4) : kOob;
                lds_dma_b128(xr, stem_lds + (buf * kCopies + i) * kLdsRow + (DIRECT ? 4 : 0), vo, 0u);
            }
        }
    };
    // DIRECT: the Add in front of the layer, on the rows this wave copied (they have landed: the caller has waited), image positions only;
    // all reads first, then the adds, then the writes: one LDS round trip per tile, not one per row
    float* const lds_mean = stem_lds + 2 * kCopies * kLdsRow + 8;       // the three constants live in LDS (registers: none to spare; scalar ones spilled to scratch)
    auto add_mean = [&](int tl, int buf) {
        if (!DIRECT || a.pre_add == nullptr) return;
        const int img = tl / a.tiles_per_image;
        const int iy0 = (tl - img * a.tiles_per_image) * (kTR * kST) - kPad;
        constexpr int NR = (kCopies + kWaves - 1) / kWaves;
        floatx4 v[NR];
        bool    live[NR];
        float mcs[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int i = q * kWaves + wid;
            const int c = i / kRows, rr = i - c * kRows;
            live[q] = i < kCopies && (unsigned)(iy0 + rr) < (unsigned)a.HP && colok;
            mcs[q] = lds_mean[i < kCopies ? c : 0];
            if (live[q]) v[q] = *reinterpret_cast<const floatx4*>(stem_lds + (buf * kCopies + i) * kLdsRow + 4 + lane * 4);
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int i = q * kWaves + wid;
            const float mc = mcs[q];
            if (live[q]) {
                v[q][0] = v[q][0] + mc; v[q][1] = v[q][1] + mc; v[q][2] = v[q][2] + mc; v[q][3] = v[q][3] + mc;
                *reinterpret_cast<floatx4*>(stem_lds + (buf * kCopies + i) * kLdsRow + 4 + lane * 4) = v[q];
            }
        }
    };

    // (the offsets are opaque from here on: hipcc must KEEP the 37 registers, not re-derive an offset with vector adds inside the loop)
#pragma unroll
    for (int m = 0; m < kSteps; ++m) asm volatile("" : "+v"(toff[m]));

    if (DIRECT) {          // the left padding of every row (and whatever no copy ever writes) is zero for the whole launch
        for (int e = tid; e < (2 * kCopies * kLdsRow + 4) / 4; e += kThreads) reinterpret_cast<floatx4*>(stem_lds)[e] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        if (tid < kC) lds_mean[tid] = a.pre_add != nullptr ? a.pre_add[tid] : 0.0f;
        __syncthreads();
    }
    issue(tile, 0);
    lds_dma_wait_all();
    add_mean(tile, 0);
    __syncthreads();
    int buf = 0;
    for (;;) {
        if (ABL != 3 && tile + 1 < tile_end) issue(tile + 1, buf ^ 1);
        floatx4 acc[kNG][2];
#pragma unroll
        for (int g = 0; g < kNG; ++g) {
            acc[g][0] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
            acc[g][1] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        // ---- the reduction: the seven operands of step m + 1 are read (address = the step's lane offset, pixel group in the immediate) while
        // the fourteen MFMAs of step m run; no vector-ALU instruction in here
        float p[2][kNG];
#pragma unroll
        for (int g = 0; g < kNG; ++g) p[0][g] = lds_read_f32(toff[0] + (unsigned)(g * 16 * kST * 4));
#pragma unroll
        for (int m = 0; m < kSteps; ++m) {
            if (m + 1 < kSteps) {
#pragma unroll
                for (int g = 0; g < kNG; ++g) p[(m + 1) & 1][g] = lds_read_f32(toff[m + 1] + (unsigned)(g * 16 * kST * 4));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (DIRECT && m == kSteps / 2 && ABL != 3 && tile + 1 < tile_end) {
                // half a tile after they were issued this wave's copies of the NEXT tile have landed: the Add in front of the layer on those rows
                // here, between two steps -- ten packed adds and an LDS round trip that the other waves' MFMAs cover -- not in the epilogue
                lds_dma_wait_all();
                add_mean(tile + 1, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g = 0; g < kNG; ++g) {
                if (ABL == 2) {
                    acc[g][0][0] += p[m & 1][g];
                    continue;
                }
                acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[m & 1][g], w[m][0], acc[g][0], 0, 0, 0);
                acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[m & 1][g], w[m][1], acc[g][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // this wave's copies for the NEXT tile were issued a whole tile ago: no wait to speak of, and the barrier below then publishes all of them
        lds_dma_wait_all();
        // ---- epilogue: register r of acc[g][t] = pixel 16 g + 4 kq + r of channel 32 hf + 16 t + l15 of output row oy0 + wr
        if (ABL == 1) {
            floatx4 sum = acc[0][0];
#pragma unroll
            for (int g = 0; g < kNG; ++g) sum += acc[g][1] + (g ? acc[g][0] : acc[g][1]);
            if (sum[0] == 12345.678f) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, sum), yr, (unsigned)(tid * 16), 0, 0);
        } else {
            const int img = tile / a.tiles_per_image;
            const int oy  = (tile - img * a.tiles_per_image) * kTR + wr;
            const unsigned plane = (unsigned)(a.OH * a.OW * 4);
            const unsigned rowb  = (unsigned)(((img * a.K + ch0) * a.OH + oy) * a.OW * 4 + 16 * kq);
            const bool rowok = oy < a.OH;
            // bias and activation behind THREE wave-uniform branches for all 56 values (launch constants; per value it would be a branch each)
            if (a.bias != nullptr) {
#pragma unroll
                for (int g = 0; g < kNG; ++g)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[g][t][r] = acc[g][t][r] + bias_l[t];
            }
            // (each activation stores by itself: merging three register assignments of the 56 sums behind the branches cost ~200 copies and spills)
            auto store_all = [&]() {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bool chok = rowok && ch0 + 16 * t < a.K;
#pragma unroll
                    for (int g = 0; g < kNG; ++g) {
                        const bool ok = chok && 16 * g + 4 * kq < a.OW;
                        unsigned vo = ok ? rowb + (unsigned)t * 16u * plane + (unsigned)(g * 64) : kOob;
                        if (ABL == 4) vo = (unsigned)((((tile * kWaves + wid) * 2 + t) * kNG + g) * 1024 + lane * 16);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, acc[g][t]), yr, vo, 0, 0);
                    }
                }
            };
            if (a.act == 1) {
                // ReLU as ONE instruction per value (v_maximum3_f32: see bias_act_n in pvhip_common.h -- a sum that starts from +0.0 is never -0.0)
                const floatx4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int g = 0; g < kNG; ++g)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[g][t] = __builtin_elementwise_maximum(acc[g][t], zero4);
                store_all();
            } else if (a.act == 2) {
#pragma unroll
                for (int g = 0; g < kNG; ++g)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = acc[g][t][r];
                            v = (v < ab.lo) ? ab.lo : v;
                            acc[g][t][r] = (v > ab.hi) ? ab.hi : v;
                        }
                store_all();
            } else {
                store_all();
            }
        }
        if (++tile >= tile_end) break;
        // the other buffer: 37 additions per tile (a whole tile of MFMAs between them)
        const unsigned delta = buf ? (unsigned)(-kBufBytes) : (unsigned)kBufBytes;
#pragma unroll
        for (int m = 0; m < kSteps; ++m) toff[m] += delta;
        buf ^= 1;
        __syncthreads();
    }
}

// w (K, 3, 7, 7) fp32 -> fragments [half][step m][tile t][lane]: channel 32 half + 16 t + (lane & 15), tap 4 m + (lane >> 4) (zero past tap 146 / channel K)
__global__ __launch_bounds__(kBlock) void conv_stem_f32_pack_kernel(const float* __restrict__ w, float* __restrict__ wf, int K) {
    const int total = 2 * kSteps * 2 * kWave;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int lane = e & 63, t = (e >> 6) & 1, m = (e >> 7) % kSteps, hf = (e >> 7) / kSteps;
        const int ch = 32 * hf + 16 * t + (lane & 15), tap = 4 * m + (lane >> 4);
        wf[e] = (ch < K && tap < kTaps) ? w[(size_t)ch * kTaps + tap] : 0.0f;
    }
}

template <bool DIRECT>
static int stem_launch(StemArgs& a, long tiles) {
    const int grid = (int)(tiles < kNumCU ? tiles : kNumCU);
    const size_t lds = (size_t)2 * kBufBytes + (DIRECT ? 64 : 0);          // DIRECT: + the last row's lane 63, + the three constants of the folded Add
    static bool attr_set = false;
    if (!attr_set) {
        PVHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_f32_kernel<0, DIRECT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
#ifdef PVHIP_DIAG
#define PVS_ABL(N_)                                                                                                                             \
    case N_: PVHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_f32_kernel<N_, DIRECT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
             hipLaunchKernelGGL((conv_stem_f32_kernel<N_, DIRECT>), dim3((unsigned)grid), dim3(kThreads), lds, state().stream, a); PVHIP_LAUNCH_CHECK(); return PVHIP_OK;
    switch (settings().stem_ablate) {     // diagnostic build only: wrong on purpose (scripts/time_stem_abl.py)
    PVS_ABL(1) PVS_ABL(2) PVS_ABL(3) PVS_ABL(4)
    default: break;
    }
#undef PVS_ABL
#endif
    hipLaunchKernelGGL((conv_stem_f32_kernel<0, DIRECT>), dim3((unsigned)grid), dim3(kThreads), lds, state().stream, a);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // namespace

extern "C" {

/* GoogLeNet's conv1 in fp32 from row spans: 7x7 / stride 2 / pad 3 over 3 channels, at most 64 output channels, output rows of at most 112
 * pixels and a multiple of four.  Returns the row length (floats) of the padded image the kernel wants, 0 when it does not cover the layer. */
int pvhip_conv2d_stem_f32_supported(int c, int h, int w, int k_out, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int oh, int ow) {
    if (c != kC || kh != kKH || kw != kKW || sh != kST || sw != kST || pad_top != kPad || pad_left != kPad || k_out <= 0 || k_out > 64) return 0;
    if (h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || ow > 16 * kNG || ow % 4 != 0) return 0;
    const int wp = (kST * (ow - 1) + kKW + 3) / 4 * 4;          // floats of a padded row: every tap of the last output column, whole 16-byte pieces
    if (wp > kLdsRow || wp < w + kPad) return 0;
    return wp;
}

size_t pvhip_conv2d_stem_f32_pack_elems(int k_out) {
    if (k_out <= 0 || k_out > 64) return 0;
    return (size_t)2 * kSteps * 2 * kWave;
}

int pvhip_conv2d_stem_f32_pack(const float* w_oihw, float* wf, int k_out) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(w_oihw != nullptr && wf != nullptr && k_out > 0 && k_out <= 64);
    hipLaunchKernelGGL(conv_stem_f32_pack_kernel, dim3(grid_for((size_t)2 * kSteps * 2 * kWave)), dim3(kBlock), 0, state().stream, w_oihw, wf, k_out);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

/* xp: the zero-padded input (n, 3, hp, wp) fp32 with hp >= 2 (oh - 1) + 7 rows and wp = _supported()'s answer floats per row (pvhip_pad2d_f32
 * with pad_top = pad_left = 3 and the bottom / right padding that makes those extents); y: (n, k_out, oh, ow) fp32; act as pvhip_conv2d_f32. */
int pvhip_conv2d_stem_f32(const float* xp, const float* wf, float* y, int n, int hp, int wp, int k_out, int oh, int ow, const float* bias,
                          int act, float act_lo, float act_hi) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && k_out > 0 && k_out <= 64 && oh > 0 && ow > 0 && ow <= 16 * kNG && ow % 4 == 0 && act >= 0 && act <= 2);
    PVHIP_CHECK_ARG(wp % 4 == 0 && wp <= kLdsRow && wp >= kST * (ow - 1) + kKW && hp >= kST * (oh - 1) + kKH);
    const unsigned long long in_b = (unsigned long long)n * kC * hp * wp * 4ull, out_b = (unsigned long long)n * k_out * oh * ow * 4ull;
    if (in_b >= (1ull << 31) || out_b >= (1ull << 31)) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_stem_f32: tensor too large");
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(xp != nullptr && wf != nullptr && y != nullptr);
    StemArgs a;
    a.xp = xp; a.wf = wf; a.y = y; a.bias = bias; a.pre_add = nullptr;
    a.N = n; a.HP = hp; a.WP = wp; a.OH = oh; a.OW = ow; a.K = k_out;
    a.tiles_per_image = (oh + kTR - 1) / kTR;
    const long tiles = (long)n * a.tiles_per_image;
    if (tiles > 0x3fffffffL) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_stem_f32: too many tiles");
    a.tiles = (int)tiles;
    a.x_bytes = (unsigned)in_b; a.y_bytes = (unsigned)out_b;
    a.act = act; a.act_lo = act_lo; a.act_hi = act_hi;
    return stem_launch<false>(a, tiles);
}

/* The same layer straight from the UNPADDED image x (n, 3, h, w), w % 4 == 0 and w <= 248: no padding pass -- the zero padding is made by where
 * the copies land in LDS and by out-of-range lanes and rows.  pre_add: one fp32 constant per input channel added to the image (not to its
 * padding) on the way, or null -- the Add of a per-channel Const in front of the layer (GoogLeNet's data/mean; Add.py:9-14), the same fp32 add.  */
int pvhip_conv2d_stem_direct_supported(int c, int h, int w, int k_out, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int oh, int ow) {
    if (pvhip_conv2d_stem_f32_supported(c, h, w, k_out, kh, kw, sh, sw, pad_top, pad_left, oh, ow) <= 0) return 0;
    if (w % 4 != 0 || w + 8 > kLdsRow) return 0;
    if (kST * (oh - 1) + kKH > h + 2 * kPad || kST * (ow - 1) + kKW > w + 2 * kPad) return 0;     // (pads_end = 3 as well: every tap inside the padded image)
    return 1;
}

int pvhip_conv2d_stem_direct_f32(const float* x, const float* wf, float* y, int n, int h, int w, int k_out, int oh, int ow, const float* pre_add,
                                 const float* bias, int act, float act_lo, float act_hi) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && k_out > 0 && k_out <= 64 && h > 0 && w > 0 && oh > 0 && ow > 0 && act >= 0 && act <= 2);
    if (!pvhip_conv2d_stem_direct_supported(kC, h, w, k_out, kKH, kKW, kST, kST, kPad, kPad, oh, ow))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_stem_direct_f32: 7x7 / 2 / pad 3 over three channels, rows of a multiple of four and at most 248 pixels");
    const unsigned long long in_b = (unsigned long long)n * kC * h * w * 4ull, out_b = (unsigned long long)n * k_out * oh * ow * 4ull;
    if (in_b >= (1ull << 31) || out_b >= (1ull << 31)) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_stem_direct_f32: tensor too large");
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && wf != nullptr && y != nullptr);
    StemArgs a;
    a.xp = x; a.wf = wf; a.y = y; a.bias = bias; a.pre_add = pre_add;
    a.N = n; a.HP = h; a.WP = w; a.OH = oh; a.OW = ow; a.K = k_out;
    a.tiles_per_image = (oh + kTR - 1) / kTR;
    const long tiles = (long)n * a.tiles_per_image;
    if (tiles > 0x3fffffffL) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_stem_direct_f32: too many tiles");
    a.tiles = (int)tiles;
    a.x_bytes = (unsigned)in_b; a.y_bytes = (unsigned)out_b;
    a.act = act; a.act_lo = act_lo; a.act_hi = act_hi;
    return stem_launch<true>(a, tiles);
}

}  // extern "C"
