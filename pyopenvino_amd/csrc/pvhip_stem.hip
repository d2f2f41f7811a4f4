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
//   * epilogue: a lane holds four consecutive pixels of one channel per accumulator tile: bias, activation -- and (the form that reads the image itself) the
//     16 channels x 112 pixels of a tile pass through the wave's own 7 KB of LDS and leave as whole 448-byte rows, two channels per store instruction
//     (0.584 -> 0.556 ms on one box: 16 x 64-byte pieces per instruction cost the address unit more than the same bytes as two rows).
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
constexpr bool kTransposeStores = true;              // the epilogue's stores as whole output rows, through LDS (a.tstores; PVHIP_TUNE4=1 switches it off)
constexpr int kTsPitch = 116;                        // floats per channel row of a wave's transposition buffer (112 pixels + 4)
constexpr int kTsWave  = 16 * kTsPitch;              // one accumulator tile: 16 channels
constexpr int kTsBase  = 2 * kCopies * kLdsRow + 16; // floats: behind the two tile buffers, the spilled piece and the Add's constants
constexpr int kTsBytes = kWaves * kTsWave * 4;       // 59392

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
    int tstores;           // 1: the epilogue's stores as whole rows through LDS (DIRECT form)
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
                if (kTransposeStores && a.tstores != 0) {
                    // Whole rows instead of 16 x 64 bytes per instruction (round 5): the 16 channels x 112 pixels of an accumulator tile pass through THIS
                    // wave's 7 KB of LDS (row pitch 116 floats; no barrier: LDS executes a wave's instructions in order) and leave as 448-byte rows,
                    // two channels per store instruction (lanes 0-27 / 28-55).  The address pattern is what the stores cost (LESSONS 55, 59).
                    float* const tl = stem_lds + kTsBase + wid * kTsWave;
                    const int c2 = lane / 28, q = lane - 28 * c2;               // reader: channel c2 of a pair (lanes 56-63: none), pixel quad q
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
#pragma unroll
                        for (int g = 0; g < kNG; ++g) *reinterpret_cast<floatx4*>(tl + l15 * kTsPitch + 16 * g + 4 * kq) = acc[g][t];
#pragma unroll
                        for (int p = 0; p < 8; ++p) {
                            const int ch = 32 * hf + 16 * t + 2 * p + c2;
                            const floatx4 v = *reinterpret_cast<const floatx4*>(tl + (2 * p + (c2 & 1)) * kTsPitch + 4 * q);
                            const bool ok = rowok && c2 < 2 && ch < a.K && 4 * q < a.OW;
                            const unsigned vo = ok ? (unsigned)(((img * a.K + ch) * a.OH + oy) * a.OW * 4 + 16 * q) : kOob;
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), yr, vo, 0, 0);
                        }
                    }
                    return;
                }
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
    a.tstores = (DIRECT && settings().tune[4] != 1) ? 1 : 0;
    const size_t lds = (size_t)2 * kBufBytes + (DIRECT ? 64 + kTsBytes : 0);          // DIRECT: + the last row's lane 63, + the three constants of the folded Add, + the waves' transposition buffers
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

// =========================================================================================================================================
// conv1 as WINOGRAD F(3x3, 4x4) on the space-to-depth image (round 5).  A 7x7 / stride 2 convolution over 3 channels IS a 4x4 / stride 1
// convolution over the 12 channels c' = (c, py, px) of x'(c', i, j) = xpad(c, 2 i + py, 2 j + px) with the weights w'(k, c', a, b) =
// w(k, c, 2 a + py, 2 b + px) (zero past the seventh tap); F(3x3, 4x4) -- 6x6 input tiles, 3x3 outputs, the six interpolation points of the
// F(4x4, 3x3) and F(2x2, 5x5) kernels (0, +-1, +-2, inf: B^T is theirs, A^T and G are F(3, 4)'s) -- executes 36 x 12 x K multiplies per
// nine outputs where the row-span kernel above executes 148 x K per output: 0.34 of the multiplies (padding of 112 -> 114 included).
// One WAVE = 16 tiles x 16 output channels on v_mfma_f32_16x16x4_f32, and the operand layouts make the data flow lane-local:
//   * lane (t = lane & 15, g = lane >> 4) is tile t and the phase (py, px) = (g >> 1, g & 1); it reads its 6x6 phase patch of raw channel
//     s = 0, 1, 2 out of the LDS copy of the raw rows (every second float of every second row), transforms it (V = B^T d B, channels 0 and 1 as
//     a packed pair) and the 36 values ARE the B operands of step s (reduction index g = the phase, column t = the tile): no LDS round trip;
//   * the A operands -- transformed weights of the wave's 16 channels: 3 steps x 36 points = 108 registers -- stay resident for the launch;
//   * accumulator register r of point p = output channel 16 kg + 4 g + r of tile t: all 36 points of a (tile, channel) meet in ONE lane, the
//     output transform Y = A^T M A (two channels as a packed pair) runs in registers and nine values leave as three 12-byte stores.
// A workgroup = 4 waves = the four 16-channel groups (K <= 64) over the SAME tiles (each repeats the input transform: 3 of ~11 k cycles per
// 16 tiles), persistent (one per CU), walking bands of two tile rows of one image: 18 raw rows x 3 channels as 54 one-KiB LDS-DMA copies per
// band (left padding = LDS offset, right / top / bottom = out-of-range lanes and rows: zeros; the Add in front of the layer in LDS, as above),
// issued for band b + 1 while band b is computed.
// MEASURED (batch 256, same box, alternating; scripts/time_stem_wino.py): 0.495-0.50 ms against 0.564-0.568 for the row-span kernel, element-wise
// 0.07 of the |d| <= 1e-4 |want| + 1e-4 rms bound.  s_memtime stamps of one wave and 16 tiles: 9.9 k cycles -- 3.5 k of them the 108 MFMAs, the rest
// the ~1000 other instructions of the tile group (288 of input transform, 144 accumulator reads, 216 of output transform, 36 maxima, 54 LDS reads,
// 70 moves, 12 stores at ~45 cycles): with one wave per SIMD and the fp32 matrix instruction on the vector datapath they ADD.  On executed flops that
// is 0.26 of the MFMA peak where the row-span kernel holds 0.71-0.735: faster, and further from its roofline -- OPT-IN (PVHIP_CONV_STEM_WINO=1).
constexpr int kWTR      = 2;                         // tile rows per band
constexpr int kWRows    = 6 * kWTR + 6;              // raw rows per channel and band: 18
constexpr int kWCopies  = kC * kWRows;               // 54
constexpr int kWPitch   = 288;                       // floats per LDS row (4 of left padding + 256 of a copy fit).  A multiple of 32, NOT of 64: rows one apart (the phases py = 0 / 1) are
                                                     // 32 banks apart, and 18 rows (a channel) / 2 rows are whole 256-byte units: ds_read2st64_b32 fetches a value of channels 0 AND 1
constexpr int kWBufBytes = kWCopies * kWPitch * 4;   // 62208
constexpr int kWThreads = 4 * kWave;

struct StemWinoArgs {
    const float* x;        // [N][3][H][W], W % 4 == 0
    const float* u;        // [4 channel groups][3 steps][36 points][64 lanes]
    float*       y;        // [N][K][OH][OW]
    const float* bias;
    const float* pre_add;
    int N, H, W, OH, OW, K;
    int TY, TX, bands_per_image, bands;
    unsigned x_bytes, y_bytes;
    int act;
    float act_lo, act_hi;
};

typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx3 __attribute__((ext_vector_type(3)));
typedef unsigned uintx3 __attribute__((ext_vector_type(3)));

// one 6-vector through B^T = [[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]] (T = float or a packed pair)
template <class T>
__device__ __forceinline__ void sw_bt(T d0, T d1, T d2, T d3, T d4, T d5, T& o0, T& o1, T& o2, T& o3, T& o4, T& o5) {
    const T c4 = T(4.0f), c5 = T(-5.0f), cm4 = T(-4.0f), c2 = T(2.0f), cm2 = T(-2.0f);
    o0 = __builtin_elementwise_fma(c4, d0, __builtin_elementwise_fma(c5, d2, d4));
    const T a = __builtin_elementwise_fma(cm4, d2, d4), b = __builtin_elementwise_fma(cm4, d1, d3);
    o1 = a + b;
    o2 = a - b;
    const T e = d4 - d2, f = d3 - d1;
    o3 = __builtin_elementwise_fma(c2, f, e);
    o4 = __builtin_elementwise_fma(cm2, f, e);
    o5 = __builtin_elementwise_fma(c4, d1, __builtin_elementwise_fma(c5, d3, d5));
}
// one 6-vector through the A^T of F(3, 4) = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,1]]
template <class T>
__device__ __forceinline__ void sw_at(T m0, T m1, T m2, T m3, T m4, T m5, T& o0, T& o1, T& o2) {
    const T s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    o0 = (m0 + s12) + s34;
    o1 = __builtin_elementwise_fma(T(2.0f), d34, d12);
    o2 = __builtin_elementwise_fma(T(4.0f), s34, s12) + m5;
}

__global__ __launch_bounds__(kWThreads, 1) void conv_stem_wino_kernel(StemWinoArgs a) {
    extern __shared__ __attribute__((aligned(1024))) float stemw_lds[];         // [2][54][256] + 4 floats (the last row's lane 63) + the Add's three constants
    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int kg   = __builtin_amdgcn_readfirstlane(tid / kWave);               // the wave's 16 output channels
    const int t16 = lane & 15, g = lane >> 4, py = g >> 1, px = g & 1;

    const int g_ = (int)gridDim.x, b_ = (int)blockIdx.x;
    const int per = a.bands / g_, extra = a.bands - per * g_;
    int       band = b_ * per + min(b_, extra);
    const int band_end = band + per + (b_ < extra ? 1 : 0);
    if (band >= band_end) return;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    const bool has_k = 16 * kg < a.K;

    // ---- the transformed weights of this wave's 16 channels: 108 registers, loaded once
    float U[3][36];
    {
        const float* const up = a.u + (size_t)kg * 3 * 36 * kWave + lane;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int p = 0; p < 36; ++p) U[s][p] = up[(s * 36 + p) * kWave];
    }
    float bias_l[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int k0 = 16 * kg + 4 * g;                   // accumulator register r: channel k0 + r
    if (a.bias != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_l[r] = k0 + r < a.K ? a.bias[k0 + r] : 0.0f;
    }
    const ActBounds ab = act_bounds(a.act, a.act_lo, a.act_hi);
    const bool colok = lane * 4 < a.W;

    // ---- the copies of one band: instruction i = (channel, raw row), wave w takes i = w, w + 4, ...; lane l: floats 4 l .. 4 l + 3 of the row
    auto issue = [&](int bd, int buf) {
        const int img = bd / a.bands_per_image;
        const int iy0 = (bd - img * a.bands_per_image) * (6 * kWTR) - kPad;
#pragma unroll
        for (int i0 = 0; i0 < kWCopies; i0 += 4) {
            const int i = i0 + kg;
            if (i < kWCopies) {
                const int  c = i / kWRows, rr = i - c * kWRows;
                const int  iy = iy0 + rr;
                const bool ok = colok && (unsigned)iy < (unsigned)a.H;
                const unsigned vo = ok ? (unsigned)((((img * kC + c) * a.H + iy) * a.W + lane * 4) * 4) : kOob;
                lds_dma_b128(xr, stemw_lds + (buf * kWCopies + i) * kWPitch + 4, vo, 0u);
            }
        }
    };
    float* const lds_mean = stemw_lds + 2 * kWCopies * kWPitch + 8;
    auto add_mean = [&](int bd, int buf) {
        if (a.pre_add == nullptr) return;
        const int img = bd / a.bands_per_image;
        const int iy0 = (bd - img * a.bands_per_image) * (6 * kWTR) - kPad;
        constexpr int NR = (kWCopies + 3) / 4;
        floatx4 v[NR];
        bool    live[NR];
        float   mcs[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int i = q * 4 + kg;
            const int c = i / kWRows, rr = i - c * kWRows;
            live[q] = i < kWCopies && (unsigned)(iy0 + rr) < (unsigned)a.H && colok;
            mcs[q] = lds_mean[i < kWCopies ? c : 0];
            if (live[q]) v[q] = *reinterpret_cast<const floatx4*>(stemw_lds + (buf * kWCopies + i) * kWPitch + 4 + lane * 4);
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int i = q * 4 + kg;
            const float mc = mcs[q];
            if (live[q]) {
                v[q][0] = v[q][0] + mc; v[q][1] = v[q][1] + mc; v[q][2] = v[q][2] + mc; v[q][3] = v[q][3] + mc;
                *reinterpret_cast<floatx4*>(stemw_lds + (buf * kWCopies + i) * kWPitch + 4 + lane * 4) = v[q];
            }
        }
    };

    for (int e = tid; e < (2 * kWCopies * kWPitch) / 4; e += kWThreads) reinterpret_cast<floatx4*>(stemw_lds)[e] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    if (tid < kC) lds_mean[tid] = a.pre_add != nullptr ? a.pre_add[tid] : 0.0f;
    __syncthreads();
    issue(band, 0);
    lds_dma_wait_all();
    add_mean(band, 0);
    __syncthreads();
    const unsigned lds0 = (unsigned)(unsigned long)(lds_void_p)stemw_lds;
    const unsigned plane = (unsigned)(a.OH * a.OW * 4);
    int buf = 0;
    for (;;) {
        if (band + 1 < band_end) issue(band + 1, buf ^ 1);
        const int img = band / a.bands_per_image;
        const int ty0 = (band - img * a.bands_per_image) * kWTR;
        const int nt  = min(kWTR, a.TY - ty0) * a.TX;              // tiles of the band
        if (has_k) {
            for (int q0 = 0; q0 < nt; q0 += 16) {
                const int  q = min(q0 + t16, nt - 1);
                const bool live = q0 + t16 < nt;
                const int  tyl = q / a.TX, tx = q - tyl * a.TX;
                // x'(s; py, px; i, j) of this tile: LDS row s * 18 + 6 tyl + 2 i + py, float 6 tx + 2 j + px + 1
                unsigned basej[6];                        // one register per patch column; rows and channels in the immediates (opaque: left visible, hipcc splits
                                                          // an address into a scalar and a lane part and adds them per read)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    basej[j] = lds0 + (unsigned)buf * (unsigned)kWBufBytes + (unsigned)(((6 * tyl + py) * kWPitch + 6 * tx + px + 1 + 2 * j) * 4);
                    asm volatile("" : "+v"(basej[j]));
                }
                floatx4 acc[36];
#pragma unroll
                for (int p = 0; p < 36; ++p) acc[p] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
                // the bias for free: A^T's column of the point x = 1 is (1, 1, 1)^T, so a constant in the accumulator of point (1, 1) comes out of Y = A^T M A on all nine outputs
                acc[7] = floatx4{bias_l[0], bias_l[1], bias_l[2], bias_l[3]};
                // ---- channels 0 and 1 as a packed pair
                {
                    floatx2 d[6][6], t[6][6];
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            // (hipcc fetches rows i, i + 1 of ONE channel per ds_read2st64_b32 and moves the values into channel pairs: 72 moves; reads through
                            // inline asm would need their wait in the same statement -- scripts/check_asm_loads.py -- and expose the LDS latency twice)
                            d[i][j][0] = lds_read_f32(basej[j] + (unsigned)(((0 * kWRows + 2 * i) * kWPitch) * 4));
                            d[i][j][1] = lds_read_f32(basej[j] + (unsigned)(((1 * kWRows + 2 * i) * kWPitch) * 4));
                        }
#pragma unroll
                    for (int j = 0; j < 6; ++j) sw_bt<floatx2>(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], t[0][j], t[1][j], t[2][j], t[3][j], t[4][j], t[5][j]);
#pragma unroll
                    for (int i = 0; i < 6; ++i) sw_bt<floatx2>(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int p = 0; p < 36; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[s][p], d[p / 6][p % 6][s], acc[p], 0, 0, 0);
                }
                // ---- channel 2
                {
                    float d[6][6], t[6][6];
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int j = 0; j < 6; ++j) d[i][j] = lds_read_f32(basej[j] + (unsigned)(((2 * kWRows + 2 * i) * kWPitch) * 4));
#pragma unroll
                    for (int j = 0; j < 6; ++j) sw_bt<float>(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], t[0][j], t[1][j], t[2][j], t[3][j], t[4][j], t[5][j]);
#pragma unroll
                    for (int i = 0; i < 6; ++i) sw_bt<float>(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
#pragma unroll
                    for (int p = 0; p < 36; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[2][p], d[p / 6][p % 6], acc[p], 0, 0, 0);
                }
                // ---- output transform Y = A^T M A of the lane's four channels (pairs r = 0, 1 and 2, 3), bias, activation, stores
                const int ty = ty0 + tyl;
                const int nb = live ? min(3, a.OW - 3 * tx) : 0;                        // valid output columns of the tile
                const unsigned yb = (unsigned)(((img * a.K + k0) * a.OH + 3 * ty) * a.OW + 3 * tx) * 4u;
                // columns first, two channels as a packed pair (the accumulator's register pairs); then the rows per channel, so that the three
                // values of an output row land in three consecutive registers: one 12-byte store, no shuffling
                float o[4][3][3];                                                       // [channel r][row][column]
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    floatx2 tt[3][6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        floatx2 m[6];
#pragma unroll
                        for (int i = 0; i < 6; ++i) m[i] = floatx2{acc[6 * i + j][2 * h], acc[6 * i + j][2 * h + 1]};
                        sw_at<floatx2>(m[0], m[1], m[2], m[3], m[4], m[5], tt[0][j], tt[1][j], tt[2][j]);
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            sw_at<float>(tt[i][0][e], tt[i][1][e], tt[i][2][e], tt[i][3][e], tt[i][4][e], tt[i][5][e], o[2 * h + e][i][0], o[2 * h + e][i][1], o[2 * h + e][i][2]);
                }
                // bias and activation behind wave-uniform branches for all 36 values (launch constants; per value it would be a branch each)
                // three columns as ONE 12-byte store per (channel, row): the lane's offset per row (out of range for a lane that must not store:
                // no tile, a row past the image, a partial tile column -- the buffer drops it), the channel in the scalar offset
                unsigned vrow[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) vrow[i] = (nb == 3 && 3 * ty + i < a.OH) ? yb + (unsigned)(i * a.OW) * 4u : kOob;
                const bool partial = __builtin_amdgcn_ballot_w64(nb == 1 || nb == 2) != 0ull;          // the image's last tile column, when it is a partial one
                auto store_all = [&]() {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(uintx3, floatx3{o[r][i][0], o[r][i][1], o[r][i][2]}), yr, vrow[i], (unsigned)r * plane, 0);
                    if (partial) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            const bool ok = (nb == 1 || nb == 2) && 3 * ty + i < a.OH;
                            const unsigned v0 = ok ? yb + (unsigned)(i * a.OW) * 4u : kOob, v1 = (ok && nb == 2) ? v0 + 4u : kOob;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[r][i][0]), yr, v0, (unsigned)r * plane, 0);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[r][i][1]), yr, v1, (unsigned)r * plane, 0);
                            }
                        }
                    }
                };
                // (each activation stores by itself: merging three register assignments of the 36 values behind the branches costs a copy of each)
                if (a.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < 3; ++j) o[r][i][j] = __builtin_elementwise_maximum(o[r][i][j], 0.0f);
                    store_all();
                } else if (a.act == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                float v = o[r][i][j];
                                v = (v < ab.lo) ? ab.lo : v;
                                o[r][i][j] = (v > ab.hi) ? ab.hi : v;
                            }
                    store_all();
                } else {
                    store_all();
                }
            }
        }
        lds_dma_wait_all();
        if (++band >= band_end) break;
        add_mean(band, buf ^ 1);
        buf ^= 1;
        __syncthreads();
    }
}

// w (K, 3, 7, 7) fp32 -> u[channel group kg][step s][point p = 6 pi + pj][lane]: (G w' G^T)[pi][pj] of channel 16 kg + (lane & 15) and the
// space-to-depth channel (c = s; py, px = lane >> 5, (lane >> 4) & 1), w'(a, b) = w(2 a + py, 2 b + px) (zero past the seventh tap), G = F(3, 4)'s
__global__ __launch_bounds__(kBlock) void conv_stem_wino_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int K) {
    const float G[6][4] = {{0.25f, 0.0f, 0.0f, 0.0f},
                           {-1.0f / 6.0f, -1.0f / 6.0f, -1.0f / 6.0f, -1.0f / 6.0f},
                           {-1.0f / 6.0f, 1.0f / 6.0f, -1.0f / 6.0f, 1.0f / 6.0f},
                           {1.0f / 24.0f, 1.0f / 12.0f, 1.0f / 6.0f, 1.0f / 3.0f},
                           {1.0f / 24.0f, -1.0f / 12.0f, 1.0f / 6.0f, -1.0f / 3.0f},
                           {0.0f, 0.0f, 0.0f, 1.0f}};
    const int total = 4 * 3 * 36 * kWave;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int lane = e & 63, p = (e >> 6) % 36, s = ((e >> 6) / 36) % 3, kg = (e >> 6) / 108;
        const int k = 16 * kg + (lane & 15), gq = lane >> 4, py = gq >> 1, px = gq & 1, pi = p / 6, pj = p - 6 * pi;
        float sum = 0.0f;
        if (k < K) {
            float row[4];                                   // (w' G^T)[a][pj]
            for (int aa = 0; aa < 4; ++aa) {
                float r = 0.0f;
                for (int bb = 0; bb < 4; ++bb) {
                    const int ky = 2 * aa + py, kx = 2 * bb + px;
                    const float wv = (ky < kKH && kx < kKW) ? w[((size_t)(k * kC + s) * kKH + ky) * kKW + kx] : 0.0f;
                    r = __builtin_fmaf(wv, G[pj][bb], r);
                }
                row[aa] = r;
            }
            for (int aa = 0; aa < 4; ++aa) sum = __builtin_fmaf(G[pi][aa], row[aa], sum);
        }
        u[e] = sum;
    }
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

/* conv1 as Winograd F(3x3, 4x4) on the space-to-depth image (ABI v16; conv_stem_wino_kernel above): 7x7 / stride 2 / pad 3 over 3 channels of an
 * UNPADDED (n, 3, h, w) fp32 image, w % 4 == 0 and w <= 224, h and w even, at most 64 output channels, oh = h / 2, ow = w / 2.  NOT the bits of
 * pvhip_conv2d_f32 (another order of summation: the tolerance of the other Winograd forms).  _pack_elems floats of transformed weights. */
int pvhip_conv2d_stem_wino_supported(int c, int h, int w, int k_out, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int oh, int ow) {
    if (c != kC || kh != kKH || kw != kKW || sh != kST || sw != kST || pad_top != kPad || pad_left != kPad || k_out <= 0 || k_out > 64 || k_out % 16 != 0) return 0;
    if (h < 2 || w < 4 || h % 2 != 0 || w % 4 != 0 || w > 224 || oh != h / 2 || ow != w / 2) return 0;
    return 1;
}

long pvhip_conv2d_stem_wino_pack_elems(void) { return 4L * 3 * 36 * kWave; }

int pvhip_conv2d_stem_wino_pack(const float* w_oihw, float* u, int k_out) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(w_oihw != nullptr && u != nullptr && k_out > 0 && k_out <= 64);
    hipLaunchKernelGGL(conv_stem_wino_pack_kernel, dim3(grid_for((size_t)4 * 3 * 36 * kWave)), dim3(kBlock), 0, state().stream, w_oihw, u, k_out);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_conv2d_stem_wino_f32(const float* x, const float* u, float* y, int n, int h, int w, int k_out, int oh, int ow, const float* pre_add,
                               const float* bias, int act, float act_lo, float act_hi) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && act >= 0 && act <= 2);
    if (!pvhip_conv2d_stem_wino_supported(kC, h, w, k_out, kKH, kKW, kST, kST, kPad, kPad, oh, ow))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_stem_wino_f32: shape outside the kernel (ask pvhip_conv2d_stem_wino_supported first)");
    const unsigned long long in_b = (unsigned long long)n * kC * h * w * 4ull, out_b = (unsigned long long)n * k_out * oh * ow * 4ull;
    if (in_b >= (1ull << 31) || out_b >= (1ull << 31)) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_stem_wino_f32: tensor too large");
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && u != nullptr && y != nullptr);
    StemWinoArgs a;
    a.x = x; a.u = u; a.y = y; a.bias = bias; a.pre_add = pre_add;
    a.N = n; a.H = h; a.W = w; a.OH = oh; a.OW = ow; a.K = k_out;
    a.TY = (oh + 2) / 3; a.TX = (ow + 2) / 3;
    a.bands_per_image = (a.TY + kWTR - 1) / kWTR;
    const long bands = (long)n * a.bands_per_image;
    if (bands > 0x3fffffffL) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_stem_wino_f32: too many bands");
    a.bands = (int)bands;
    a.x_bytes = (unsigned)in_b; a.y_bytes = (unsigned)out_b;
    a.act = act; a.act_lo = act_lo; a.act_hi = act_hi;
    const int grid = (int)(bands < kNumCU ? bands : kNumCU);
    const size_t lds = (size_t)2 * kWBufBytes + 64;
    static bool attr_set = false;
    if (!attr_set) {
        PVHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(conv_stem_wino_kernel, dim3((unsigned)grid), dim3(kWThreads), lds, state().stream, a);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // extern "C"
