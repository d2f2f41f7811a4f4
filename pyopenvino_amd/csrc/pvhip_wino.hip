// 3x3 / stride 1 / pad 1 convolution by Winograd's minimal filtering F(2x2, 3x3) on the fp32 matrix cores.
//
// Replaces the same reference function as pvhip_conv.hip (Convolution.py:57-87) for the layers that are 3x3 with unit
// stride and "same" zero padding -- 61 % of GoogLeNet's multiply-adds.  Every 2x2 patch of outputs is computed from
// the 4x4 input patch d around it as  Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A : 16 multiplies per channel and
// patch instead of 36, i.e. 2.25x fewer matrix-core operations for the same convolution (the transforms are additions).
// Rounding differs from the direct sum at the 1e-7 level (fp32 throughout, coefficients 0, +-1, +-1/2).
//
//   * Weights are transformed once, at pack time: U[xi][c][k] = (G g G^T)[xi], xi = 4*i + j, laid out per
//     (block of 32 or 64 output channels, stage of 4 input channels) as the dense [16][4][block] image the kernel stages.
//   * A workgroup of 8 waves owns 64 output channels x 32 patches (32 x 64 when K does not fill 64-channel blocks).
//     Per stage of 4 input channels the waves on duty (the duty rotates from stage to stage) gather the 4x4 patches
//     (lane <-> patch; 16 buffer_load_dword with per-element out-of-range sentinels for the padding, wave-uniform
//     channel offset: no address arithmetic in the loop), transform them in registers (32 additions) and write
//     V[xi][c][patch] to LDS; the U image arrives by LDS-DMA.  Then 16 independent GEMMs
//     D_xi[k][patch] += U_xi[k][c] * V_xi[c][patch] on v_mfma_f32_32x32x2_f32: wave w owns the four xi of row
//     i = w & 3 for the 32x32 tile w >> 2.
//   * Epilogue: the output transform is separable; each wave applies the column half to its own row (registers), the
//     four rows meet in LDS, then bias / activation / store of 2x2 patches.
#include "pvhip_common.h"
#include "pvhip_wino.h"

using namespace pvhip;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kCB = 4;     // input channels per stage

struct WinoArgs {
    const float* x;
    const float* u;
    float*       y;
    const float* bias;
    int N, C, H, W, K;
    int TY, TX, T;          // patches per column / row / in total
    int n_kb, n_stages;
    unsigned x_bytes, u_bytes;
    int   act;
    float act_lo, act_hi;
    int   y_ctotal, y_coff;
};

// U = G g G^T for one (k, c):  G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]]
__global__ __launch_bounds__(kBlock) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int K, int C,
                                                            int n_stages, int kKB) {
    const int kUStage = 16 * kCB * kKB;       // floats of one U stage image [16][4][kKB]
    const int total = ((K + kKB - 1) / kKB) * kKB * C;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int k = e / C, c = e - k * C;
        float g[3][3];
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i / 3][i % 3] = (k < K) ? w[((size_t)k * C + c) * 9 + i] : 0.0f;
        float r[4][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            r[0][q] = g[0][q];
            r[1][q] = 0.5f * (g[0][q] + g[1][q] + g[2][q]);
            r[2][q] = 0.5f * (g[0][q] - g[1][q] + g[2][q]);
            r[3][q] = g[2][q];
        }
        const int kb = k / kKB, kl = k % kKB, s = c / kCB, cl = c % kCB;
        float* up = u + ((size_t)kb * (n_stages + 1) + s) * kUStage + cl * kKB + kl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            up[(i * 4 + 0) * kCB * kKB] = r[i][0];
            up[(i * 4 + 1) * kCB * kKB] = 0.5f * (r[i][0] + r[i][1] + r[i][2]);
            up[(i * 4 + 2) * kCB * kKB] = 0.5f * (r[i][0] - r[i][1] + r[i][2]);
            up[(i * 4 + 3) * kCB * kKB] = r[i][2];
        }
    }
}

// 16 bytes per lane global -> LDS (see pvhip_conv.hip dma_b128: asm on purpose, hipcc would drain vmcnt before the next ds_read)
__device__ __forceinline__ void wino_dma_b128(__amdgpu_buffer_rsrc_t r, float* dst, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lds = (unsigned)(unsigned long)(lds_ptr_t)dst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
#endif
}

// MT x NTL = 32-channel tiles x 32-patch tiles per workgroup: (2, 1) = 64 channels x 32 patches, (1, 2) = 32 x 64,
// (1, 1) = 32 x 32 on four waves.
// The second form gathers (and transforms) each input patch once per 64 output channels instead of once per 32 and is
// used when K fills 64-channel blocks (or nearly: at most 12 % of padding).
template <int MT, int NTL, int WAVES>     // WAVES = 4: every wave owns both 32x32 tiles of its four xi; 8: one tile each
__global__ __launch_bounds__(WAVES * kWave, (WAVES == 8 || MT * NTL == 1) ? 4 : 2) void conv_wino_kernel(WinoArgs a) {
    static_assert(MT * NTL == 2 || (MT * NTL == 1 && WAVES == 4), "two accumulator tiles per xi, or one with four waves");
    constexpr int TILES = MT * NTL;
    static_assert(WAVES == 4 || WAVES == 8, "four or eight waves");
    constexpr int THREADS = WAVES * kWave;
    constexpr int TPW = TILES * 4 / WAVES;    // accumulator tiles per wave and xi
    constexpr unsigned kOob = 0x80000000u;
    constexpr int KB = 32 * MT, NT = 32 * NTL;
    constexpr int U_PIECES = 16 * kCB * KB * 4 / 1024;      // 1-KiB pieces of one U stage image: 8 * MT
    constexpr int U_PER_WAVE = U_PIECES / WAVES;
    struct Stage {                     // one allocation: the epilogue's exchange buffer may run across both arrays
        float Us[2][16][kCB][KB];
        float Vs[2][16][kCB][NT];
    };
    __shared__ __attribute__((aligned(1024))) Stage sm;
    auto& Us = sm.Us;
    auto& Vs = sm.Vs;
    static_assert(sizeof(Stage) == (TILES == 2 ? 48 : 32) * 1024, "48 (32) KB of LDS");

    const int nwg = gridDim.x;
    int       lid;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int kb = lid % a.n_kb;          // the channel blocks of one patch block run back to back on one XCD
    const int tb = lid / a.n_kb;

    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid  = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int HW   = a.H * a.W;
    const int TPI  = a.TY * a.TX;         // patches per image
    const unsigned chan_bytes = (unsigned)HW * 4u;

    // ---- gather duty.  NTL == 2: wave w gathers channel w of the stage for 64 patches (lane <-> patch).
    // NTL == 1: waves 0 and 1 gather two channels each for 32 patches (lane & 31 <-> patch, lane >> 5 picks the channel,
    // whose offset is folded into the lane's addresses); waves 2 and 3 have none.
    // With 8 waves the duty rotates from stage to stage (waves 2g, 2g+1 resp. 4g..4g+3 take the stages with s % groups == g):
    // the transform is vector-ALU work, which the fp32 MFMA cannot overlap, so it is spread over all four SIMDs.
    constexpr int G_WAVES = (NTL == 2) ? 4 : 2;               // waves that gather one stage
    constexpr int G_GROUPS = WAVES / G_WAVES;                  // 1 or 2 (4 waves), 2 or 4 (8 waves)
    const int  g_group = wid / G_WAVES, g_wave = wid % G_WAVES;
    const int  g_patch = (NTL == 2) ? lane : (lane & 31);
    const int  g_chan  = (NTL == 2) ? g_wave : g_wave * 2 + (lane >> 5);   // channel inside the stage (per lane when NTL == 1)
    const bool gathers = true;                                 // every wave has the duty for some stages
    unsigned voff[16];
    {
        const int t = tb * NT + g_patch;
        const bool live = gathers && t < a.T;
        const int n = live ? t / TPI : 0, rem = live ? t - n * TPI : 0;
        const int ty = rem / a.TX, tx = rem - ty * a.TX;
        const int iy0 = 2 * ty - 1, ix0 = 2 * tx - 1;
        // element (r, q) of the 4x4 input patch at rows 2ty-1.., cols 2tx-1..; wraps when the patch starts in the padding
        const unsigned base = (unsigned)(n * a.C * HW + iy0 * a.W + ix0) * 4u + ((NTL == 2) ? 0u : (unsigned)g_chan * chan_bytes);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = live && (unsigned)(iy0 + r) < (unsigned)a.H && (unsigned)(ix0 + q) < (unsigned)a.W;
                voff[r * 4 + q] = ok ? base + (unsigned)(r * a.W + q) * 4u : kOob;
            }
    }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u_bytes, 0x00020000);
    constexpr unsigned u_stage_bytes = 16u * kCB * KB * 4u;
    const unsigned u_base = (unsigned)(kb * (a.n_stages + 1)) * u_stage_bytes;
    const unsigned u_lane = (unsigned)lane * 16u;

    float d[16];
#define PVW_GATHER(s_)                                                                                          \
    if (g_group == (s_) % G_GROUPS) {                                                                           \
        const unsigned soff = (unsigned)((s_) * kCB + ((NTL == 2) ? g_wave : 0)) * chan_bytes;                   \
        _Pragma("unroll") for (int e = 0; e < 16; ++e)                                                           \
            d[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, voff[e], soff, 0));        \
    }
#define PVW_LOAD_U(s_, buf_)                                                                                    \
    {                                                                                                           \
        const unsigned soff = u_base + (unsigned)(s_) * u_stage_bytes;                                           \
        _Pragma("unroll") for (int q = 0; q < U_PER_WAVE; ++q)                                                   \
            wino_dma_b128(ur, &Us[buf_][0][0][0] + (wid + WAVES * q) * 256, u_lane + (unsigned)(wid + WAVES * q) * 1024u, soff); \
    }
    // V = B^T d B,  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]];  written as V[xi][channel][patch]
#define PVW_TRANSFORM_STORE(buf_, s_)                                                                           \
    if (g_group == (s_) % G_GROUPS) {                                                                           \
        float m[16];                                                                                            \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                          \
            m[0 * 4 + q] = d[0 * 4 + q] - d[2 * 4 + q];                                                          \
            m[1 * 4 + q] = d[1 * 4 + q] + d[2 * 4 + q];                                                          \
            m[2 * 4 + q] = d[2 * 4 + q] - d[1 * 4 + q];                                                          \
            m[3 * 4 + q] = d[1 * 4 + q] - d[3 * 4 + q];                                                          \
        }                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
            Vs[buf_][i * 4 + 0][g_chan][g_patch] = m[i * 4 + 0] - m[i * 4 + 2];                                  \
            Vs[buf_][i * 4 + 1][g_chan][g_patch] = m[i * 4 + 1] + m[i * 4 + 2];                                  \
            Vs[buf_][i * 4 + 2][g_chan][g_patch] = m[i * 4 + 2] - m[i * 4 + 1];                                  \
            Vs[buf_][i * 4 + 3][g_chan][g_patch] = m[i * 4 + 1] - m[i * 4 + 3];                                  \
        }                                                                                                       \
    }

    // D_xi for xi = 4*row + j, row = wid & 3; tile = channel half (MT == 2) or patch half (NTL == 2): both (4 waves) or
    // the one of this wave (8 waves: wid >> 2)
    floatx16 acc[4][TPW];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < TPW; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][h][r] = 0.0f;

    const int l31 = lane & 31, lh = lane >> 5;
    const int row = wid & 3, my_tile = wid >> 2;          // my_tile only meaningful with 8 waves

    PVW_GATHER(0);
    PVW_LOAD_U(0, 0);
    PVW_TRANSFORM_STORE(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int s = 0; s < a.n_stages; ++s) {
        const int buf = s & 1;
        PVW_GATHER(s + 1);            // one stage ahead (past the end: the next image's channels / out of range -> unused)
        PVW_LOAD_U(s + 1, buf ^ 1);   // past the end: the spare zero stage of the block
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < kCB / 2; ++kk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xi = row * 4 + j;
                if (TPW == 2) {
                    if (NTL == 2) {
                        const float af = Us[buf][xi][2 * kk + lh][l31];
                        const float b0 = Vs[buf][xi][2 * kk + lh][l31];
                        const float b1 = Vs[buf][xi][2 * kk + lh][(NT - 32) + l31];
                        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, b0, acc[j][0], 0, 0, 0);
                        acc[j][TPW - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, b1, acc[j][TPW - 1], 0, 0, 0);
                    } else {
                        const float a0 = Us[buf][xi][2 * kk + lh][l31];
                        const float a1 = Us[buf][xi][2 * kk + lh][(KB - 32) + l31];
                        const float bf = Vs[buf][xi][2 * kk + lh][l31];
                        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bf, acc[j][0], 0, 0, 0);
                        acc[j][TPW - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bf, acc[j][TPW - 1], 0, 0, 0);
                    }
                } else {
                    const float af = Us[buf][xi][2 * kk + lh][((MT == 2) ? my_tile * 32 : 0) + l31];
                    const float bf = Vs[buf][xi][2 * kk + lh][((NTL == 2) ? my_tile * 32 : 0) + l31];
                    acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[j][0], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        PVW_TRANSFORM_STORE(buf ^ 1, s + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef PVW_GATHER
#undef PVW_LOAD_U
#undef PVW_TRANSFORM_STORE

    // ---- output transform  Y = A^T D A,  A^T = [[1,1,1,0],[0,1,-1,-1]]: columns in registers (this wave holds row i = wid),
    // rows through LDS, one 32-channel x 32-patch tile at a time: Ex[i][b][k][patch].
    float* Ex = (sizeof(sm.Vs) >= 32 * 1024) ? &Vs[0][0][0][0] : &Us[0][0][0][0];     // 4 * 2 * 32 * 32 floats = 32 KB (may span Us and Vs)
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0,
                                                                        a.bias != nullptr ? a.K * 4 : 0, 0x00020000);
    const int OH = a.H, OW = a.W;
#pragma unroll
    for (int h = 0; h < TILES; ++h) {
        if (h == 1) __syncthreads();         // the reads of the first tile are done
        if (TPW == 2 || my_tile == h) {
            const int hh = (TPW == 2) ? h : 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = (r & 3) + 8 * (r >> 2) + 4 * lh;
                Ex[((row * 2 + 0) * 32 + k) * 32 + l31] = acc[0][hh][r] + acc[1][hh][r] + acc[2][hh][r];
                Ex[((row * 2 + 1) * 32 + k) * 32 + l31] = acc[1][hh][r] - acc[2][hh][r] - acc[3][hh][r];
            }
        }
        __syncthreads();
        const int tl = tid & 31;
        const int t  = tb * NT + ((NTL == 2) ? h * 32 : 0) + tl;
        if (t < a.T) {
            const int n = t / TPI, rem = t - n * TPI;
            const int ty = rem / a.TX, tx = rem - ty * a.TX;
            const int oy = 2 * ty, ox = 2 * tx;
#pragma unroll
            for (int q = 0; q < 32 / (THREADS / 32); ++q) {
                const int k  = (tid >> 5) + (THREADS / 32) * q;
                const int kg = kb * KB + ((MT == 2) ? h * 32 : 0) + k;
                if (kg >= a.K) continue;
                float T_[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    T_[i][0] = Ex[((i * 2 + 0) * 32 + k) * 32 + tl];
                    T_[i][1] = Ex[((i * 2 + 1) * 32 + k) * 32 + tl];
                }
                float yv[2][2];
                yv[0][0] = T_[0][0] + T_[1][0] + T_[2][0];
                yv[0][1] = T_[0][1] + T_[1][1] + T_[2][1];
                yv[1][0] = T_[1][0] - T_[2][0] - T_[3][0];
                yv[1][1] = T_[1][1] - T_[2][1] - T_[3][1];
                const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (unsigned)kg * 4u, 0, 0));
                float* __restrict__ yp = a.y + (((size_t)n * a.y_ctotal + a.y_coff + kg) * OH + oy) * OW + ox;
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    if (oy + r2 >= OH) continue;
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        if (ox + c2 >= OW) continue;
                        float v = yv[r2][c2];
                        if (a.bias != nullptr) v = v + bv;
                        if (a.act == 1) v = (v < 0.0f) ? 0.0f : v;
                        else if (a.act == 2) { v = (v < a.act_lo) ? a.act_lo : v; v = (v > a.act_hi) ? a.act_hi : v; }
                        yp[(size_t)r2 * OW + c2] = v;
                    }
                }
            }
        }
    }
}

}  // namespace

namespace pvhip {

bool wino_eligible(int c, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int h, int w, int oh, int ow) {
    const char* e       = getenv("PVHIP_CONV_WINOGRAD");      // "0": the direct kernels (A/B measurements, their tests)
    const bool  enabled = e == nullptr || e[0] != '0';
    return enabled && kh == 3 && kw == 3 && sh == 1 && sw == 1 && pad_top == 1 && pad_left == 1 && oh == h && ow == w &&
           c % kCB == 0 && c >= kCB;
}

// Output channels per workgroup (and per block of the transformed panel): 64 when K is made of whole 64-channel blocks
// (each input patch is then gathered and transformed once per 64 output channels), else 32.
static int wino_kb(int k) {
    if (const char* e = getenv("PVHIP_WINO_KB")) {       // tuning runs only
        const int v = atoi(e);
        if (v == 32 || v == 64) return v;
    }
    return ((k + 63) / 64 * 64) * 100 <= k * 112 ? 64 : 32;      // whole 64-channel blocks, or at most 12 % of padding
}

size_t wino_pack_elems(int k, int c) {
    if (c % kCB != 0) return 0;
    // sized for 64-channel blocks, which also covers the 32-channel layout (ceil(k/32)*32 <= ceil(k/64)*64)
    return (size_t)((k + 63) / 64) * (size_t)(c / kCB + 1) * 16 * kCB * 64;
}

int wino_pack(const float* w_oihw, float* u, int k, int c) {
    const size_t elems = wino_pack_elems(k, c);
    const int    kb    = wino_kb(k);
    hipError_t   e     = hipMemsetAsync(u, 0, elems * sizeof(float), state().stream);     // spare stages, channels >= K
    if (e != hipSuccess) return fail(PVHIP_EHIP, "wino_pack: hipMemsetAsync -> %s", hipGetErrorString(e));
    hipLaunchKernelGGL(wino_pack_kernel, dim3(grid_for((size_t)((k + kb - 1) / kb) * kb * c)), dim3(kBlock), 0, state().stream,
                       w_oihw, u, k, c, c / kCB, kb);
    return PVHIP_OK;
}

int wino_conv(const float* x, const float* u, float* y, int n, int c, int h, int w, int k_out, const float* bias, int act,
              float act_lo, float act_hi, int out_channel_offset, int out_channels_total) {
    const int kb = wino_kb(k_out);
    WinoArgs a;
    a.x = x; a.u = u; a.y = y; a.bias = bias;
    a.N = n; a.C = c; a.H = h; a.W = w; a.K = k_out;
    a.TY = (h + 1) / 2; a.TX = (w + 1) / 2;
    a.T  = n * a.TY * a.TX;
    a.n_kb = (k_out + kb - 1) / kb;
    a.n_stages = c / kCB;
    a.x_bytes = (unsigned)((size_t)n * c * h * w * 4);
    a.u_bytes = (unsigned)(wino_pack_elems(k_out, c) * 4);
    a.act = act; a.act_lo = act_lo; a.act_hi = act_hi;
    a.y_ctotal = out_channels_total; a.y_coff = out_channel_offset;
    // 32-channel blocks run as 32 channels x 32 patches on four waves (more, smaller workgroups: the layers whose K is not
    // made of 64-channel blocks are the small 14x14 ones); PVHIP_WINO_SMALL=0 selects 32 x 64 on eight waves (tuning runs)
    const char* se    = getenv("PVHIP_WINO_SMALL");
    const bool  small = kb == 32 && !(se != nullptr && se[0] == '0');
    const int  nt   = (kb == 64 || small) ? 32 : 64;
    const long n_tb = ((long)a.T + nt - 1) / nt;
    if (n_tb * a.n_kb > 0x7fffffffL) return fail(PVHIP_EUNSUPPORTED, "wino_conv: grid too large");
    const char* we    = getenv("PVHIP_WINO_WAVES");       // tuning runs only
    const int   waves = (we != nullptr && atoi(we) == 4) ? 4 : 8;
    const dim3  grid((unsigned)(n_tb * a.n_kb));
    if (small) hipLaunchKernelGGL((conv_wino_kernel<1, 1, 4>), grid, dim3(256), 0, state().stream, a);
    else if (kb == 64 && waves == 8) hipLaunchKernelGGL((conv_wino_kernel<2, 1, 8>), grid, dim3(512), 0, state().stream, a);
    else if (kb == 64) hipLaunchKernelGGL((conv_wino_kernel<2, 1, 4>), grid, dim3(256), 0, state().stream, a);
    else if (waves == 8) hipLaunchKernelGGL((conv_wino_kernel<1, 2, 8>), grid, dim3(512), 0, state().stream, a);
    else hipLaunchKernelGGL((conv_wino_kernel<1, 2, 4>), grid, dim3(256), 0, state().stream, a);
    return PVHIP_OK;
}

}  // namespace pvhip
