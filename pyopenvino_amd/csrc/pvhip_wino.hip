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
#include <type_traits>

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
    int   balance;          // conv_wino4_kernel: producers placed by SIMD (see the kernel)
    int   n_tiles;          // conv_wino4_kernel: channel blocks x patch blocks, walked by a persistent grid
    int   s_prio;           // conv_wino4s_kernel: producers at wave priority 3, epilogues at 2 (PVHIP_WINO_SHARED_PRIO=0: everything at 0)
    int   s_old;            // conv_wino4s_kernel: producers are waves 0-3 (the oldest) rather than 12-15
    int   s_lag;            // conv_wino4s_kernel: the second consumer group starts kSLag stages behind the first (PVHIP_WINO_SHARED_LAG=0: together)
    int   s_wide;           // conv_wino4s_kernel, ragged extents: whole patches store a row as ONE 16-byte piece (8-byte aligned on 14-wide rows) instead of two 8-byte ones
    int   p_prio;           // conv_wino4_kernel: its two producer waves at wave priority 3
    int   s_order;          // conv_wino4s_kernel: 1 = tiles in channel-pair-major order (a workgroup, and with it an XCD, stays on ONE pair's slice of the transformed weights
                            // across patch blocks); 0 = patch-block-major (the pairs of a patch block follow each other: its patches come out of L2)
    // conv_wino4_kernel: patch index -> (image, patch row, patch column) by multiply-high and shift (w4_magic): the divisors are
    // wave-uniform, but hipcc's own division keeps their reciprocals in VECTOR registers across the main loop -- spilled there
    unsigned tpi_mul, tpi_sh, tx_mul, tx_sh;
};

// n / d for 0 <= n < 2^31 and the (mul, sh) of w4_magic(d): exact (mul = ceil(2^(31+s) / d), s = ceil(log2 d): the error term
// n * (mul * d - 2^(31+s)) stays below 2^(31+s)); mul = 0 stands for d = 1
__device__ __forceinline__ int w4_div(int n, unsigned mul, unsigned sh) {
    return mul != 0u ? (int)(__umulhi((unsigned)n, mul) >> sh) : n;
}
static void w4_magic(unsigned d, unsigned& mul, unsigned& sh) {
    mul = 0u; sh = 0u;
    if (d <= 1u) return;
    unsigned s_ = 0;
    while ((1u << s_) < d) ++s_;
    mul = (unsigned)(((1ull << (31 + s_)) + d - 1) / d);
    sh  = s_ - 1;
}

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

#ifdef PVHIP_DIAG
// ABL = 5 (diagnostic build): per-wave cycle accounts of every 61st workgroup, [wave][segment]; segment 7 counts the workgroups.
// consumers: 0 MFMA segment (LDS reads + MFMA issue), 1 wait for the U DMA, 2 barrier;  producers: 0 gather issue, 1 wait for the older
// gather, 2 transform + LDS writes, 3 barrier.
__device__ unsigned long long g_w4_stamps[8][8];
__device__ unsigned long long g_w4_epi[8][8];      // ABL = 5: the epilogue's phases (write 0, barrier, read + store 0, barrier, write 1, barrier, read + store 1, barrier)
__device__ unsigned g_w4_hw[64][8][2];          // ABL = 5: HW_REG_LDS_ALLOC / HW_REG_HW_ID of the waves of the first 64 workgroups
__device__ unsigned g_w4_hw_ticket;
#define PVW4_NOW() ((ABL == 5) ? (unsigned long long)__builtin_readcyclecounter() : 0ull)
#else
#define PVW4_NOW() 0ull
#endif

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

#ifdef PVHIP_DIAG
    const bool stamp = a.balance == 5;
    unsigned long long st[4] = {0ull, 0ull, 0ull, 0ull};
    const unsigned long long t_entry = stamp ? __builtin_readcyclecounter() : 0ull;
#define PVW_NOW() (stamp ? (unsigned long long)__builtin_readcyclecounter() : 0ull)
#else
#define PVW_NOW() 0ull
#endif
    for (int s = 0; s < a.n_stages; ++s) {
        const int buf = s & 1;
        const unsigned long long t0 = PVW_NOW();
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
        const unsigned long long t1 = PVW_NOW();
        PVW_TRANSFORM_STORE(buf ^ 1, s + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = PVW_NOW();
        __syncthreads();
#ifdef PVHIP_DIAG
        const unsigned long long t3 = PVW_NOW();
        st[0] += t1 - t0; st[1] += t2 - t1; st[2] += t3 - t2;
#else
        (void)t0; (void)t1; (void)t2;
#endif
    }
#ifdef PVHIP_DIAG
    const unsigned long long t_loop = PVW_NOW();
#endif
#undef PVW_GATHER
#undef PVW_LOAD_U
#undef PVW_TRANSFORM_STORE

    // ---- output transform  Y = A^T D A,  A^T = [[1,1,1,0],[0,1,-1,-1]]: columns in registers (this wave holds row i = wid),
    // rows through LDS, one 32-channel x 32-patch tile at a time: Ex[i][k][patch][2 columns] -- the two column results of a
    // (row, channel, patch) are one 8-byte LDS access, and an output row of the 2x2 patch one 8-byte store (even widths).
    typedef float ex2_t __attribute__((ext_vector_type(2)));
    float* Ex = (sizeof(sm.Vs) >= 32 * 1024) ? &Vs[0][0][0][0] : &Us[0][0][0][0];     // 4 * 32 * 32 * 2 floats = 32 KB (may span Us and Vs)
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0,
                                                                        a.bias != nullptr ? a.K * 4 : 0, 0x00020000);
    const int  OH = a.H, OW = a.W;
    const bool pair_stores = (OW & 1) == 0;          // 2*tx + 1 < OW and 8-byte aligned rows
    const ActBounds ab = act_bounds(a.act, a.act_lo, a.act_hi);
#pragma unroll
    for (int h = 0; h < TILES; ++h) {
        if (h == 1) __syncthreads();         // the reads of the first tile are done
        if (TPW == 2 || my_tile == h) {
            const int hh = (TPW == 2) ? h : 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = (r & 3) + 8 * (r >> 2) + 4 * lh;
                ex2_t e;
                e.x = acc[0][hh][r] + acc[1][hh][r] + acc[2][hh][r];
                e.y = acc[1][hh][r] - acc[2][hh][r] - acc[3][hh][r];
                *reinterpret_cast<ex2_t*>(Ex + ((row * 32 + k) * 32 + l31) * 2) = e;
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
                ex2_t T_[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) T_[i] = *reinterpret_cast<const ex2_t*>(Ex + ((i * 32 + k) * 32 + tl) * 2);
                ex2_t yv[2];
                yv[0] = T_[0] + T_[1] + T_[2];
                yv[1] = T_[1] - T_[2] - T_[3];
                const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (unsigned)kg * 4u, 0, 0));
                float* __restrict__ yp = a.y + (((size_t)n * a.y_ctotal + a.y_coff + kg) * OH + oy) * OW + ox;
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    if (oy + r2 >= OH) continue;
                    float ov[2];
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        float v = yv[r2][c2];
                        if (a.bias != nullptr) v = v + bv;
                        v = act_apply(v, ab);
                        ov[c2] = v;
                    }
                    if (pair_stores) {
                        conv_store2(yp + (size_t)r2 * OW, ov[0], ov[1]);
                    } else {
                        conv_store1(yp + (size_t)r2 * OW, ov[0]);
                        if (ox + 1 < OW) conv_store1(yp + (size_t)r2 * OW + 1, ov[1]);
                    }
                }
            }
        }
    }
#ifdef PVHIP_DIAG
    if (stamp && blockIdx.x % 61 == 7 && lane == 0) {
        const unsigned long long t_end = __builtin_readcyclecounter();
        for (int i = 0; i < 3; ++i) atomicAdd(&g_w4_stamps[wid][i], st[i]);
        atomicAdd(&g_w4_stamps[wid][4], t_loop - t_entry - (st[0] + st[1] + st[2]));   // (stamp overhead only)
        atomicAdd(&g_w4_stamps[wid][5], t_end - t_loop);                                 // epilogue
        atomicAdd(&g_w4_stamps[wid][6], t_end - t_entry);                                // from the first stage to the end
        atomicAdd(&g_w4_stamps[wid][7], 1ull);
    }
#endif
#undef PVW_NOW
}


// ---------------------------------------------------------------------------------------------------------------
// F(4x4, 3x3): every 4x4 patch of outputs from the 6x6 input patch around it, 36 multiplies per channel and patch
// instead of 144 -- 4x fewer matrix-core operations than the direct sum, 1.78x fewer than F(2x2, 3x3).  Used for the
// layers whose extents are multiples of 4 and large enough to fill the chip (conv2/3x3 at 56x56, the 28x28 inception
// layers).  fp32 throughout; the transforms have coefficients up to 8 and 1/24, so the result differs from the direct
// sum at the 4e-6 level of the output's maximum (F(2x2): 3e-7; stated tolerance of the path 1e-4).
//
//   * U[xi][c][k] = (G g G^T)[xi], xi = 6*i + j, packed per (block of 32 output channels, stage of 4 input channels)
//     as the dense [36][4][32] image the kernel stages by LDS-DMA.
//   * A workgroup of 8 waves owns 32 output channels x 32 patches (= 512 output pixels).  Waves 0-5 are CONSUMERS:
//     wave i holds the six accumulator tiles D_xi[k][patch], xi = 6*i .. 6*i+5, and issues 12 MFMAs per stage.  Waves
//     6-7 are PRODUCERS: lane <-> (patch, channel of the stage); they gather the 6x6 patch (36 buffer loads, out-of-range
//     sentinels for the rows in the padding, the two border columns zeroed by select), keep TWO stages of gathers in
//     flight, transform V = B^T d B in registers (144 fused multiply-adds) and write V[xi][c][patch] to LDS.  The
//     producers hold no accumulators and the consumers no patches, so the kernel needs 128 registers, not 200.
//   * Epilogue: each consumer applies the column half of Y = A^T D A to its row in registers; the rows meet in LDS
//     (16 channels at a time), then bias / activation and four 16-byte stores per (channel, patch).
constexpr int kXi4 = 36;

// Where element (xi = 6 i + j, channel cl of the stage, output channel kl of the block) of a U stage image lives: in the order the
// consumers of conv_wino4_kernel read it -- consumer i, MFMA m = 6 kk + j of a stage (kk = cl >> 1: reduction step), lane = 32 (cl & 1)
// + kl: [i][m >> 2][lane][m & 3], so that the A operands of four consecutive MFMAs are ONE 16-byte load per lane and a load
// instruction reads 1 KiB of consecutive bytes.  The images never pass through LDS.
__device__ __forceinline__ int wino4_u_offset(int i, int j, int cl, int kl) {
    const int m = (cl >> 1) * 6 + j, lane = (cl & 1) * 32 + kl;
    return ((i * 3 + (m >> 2)) * 64 + lane) * 4 + (m & 3);
}

__global__ __launch_bounds__(kBlock) void wino4_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int K, int C,
                                                             int n_stages) {
    constexpr int KB = 32;
    const int kUStage = kXi4 * kCB * KB;
    const int total = ((K + KB - 1) / KB) * KB * C;
    const float G[6][3] = {{0.25f, 0.0f, 0.0f},
                           {-1.0f / 6.0f, -1.0f / 6.0f, -1.0f / 6.0f},
                           {-1.0f / 6.0f, 1.0f / 6.0f, -1.0f / 6.0f},
                           {1.0f / 24.0f, 1.0f / 12.0f, 1.0f / 6.0f},
                           {1.0f / 24.0f, -1.0f / 12.0f, 1.0f / 6.0f},
                           {0.0f, 0.0f, 1.0f}};
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int k = e / C, c = e - k * C;
        float g[3][3];
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i / 3][i % 3] = (k < K) ? w[((size_t)k * C + c) * 9 + i] : 0.0f;
        float r[6][3];                       // G g
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) r[i][q] = G[i][0] * g[0][q] + G[i][1] * g[1][q] + G[i][2] * g[2][q];
        const int kb = k / KB, kl = k % KB, st = c / kCB, cl = c % kCB;
        float* up = u + ((size_t)kb * (n_stages + 1) + st) * kUStage;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j)
                up[wino4_u_offset(i, j, cl, kl)] = r[i][0] * G[j][0] + r[i][1] * G[j][1] + r[i][2] * G[j][2];
    }
}

// One 6-vector through B^T = [[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]]
__device__ __forceinline__ void wino4_bt(float d0, float d1, float d2, float d3, float d4, float d5, float& o0, float& o1,
                                         float& o2, float& o3, float& o4, float& o5) {
    o0 = __builtin_fmaf(4.0f, d0, __builtin_fmaf(-5.0f, d2, d4));
    const float a = __builtin_fmaf(-4.0f, d2, d4), b = __builtin_fmaf(-4.0f, d1, d3);
    o1 = a + b;
    o2 = a - b;
    const float e = d4 - d2, f = d3 - d1;
    o3 = __builtin_fmaf(2.0f, f, e);
    o4 = __builtin_fmaf(-2.0f, f, e);
    o5 = __builtin_fmaf(4.0f, d1, __builtin_fmaf(-5.0f, d3, d5));
}

// One 6-vector through A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]]
__device__ __forceinline__ void wino4_at(float m0, float m1, float m2, float m3, float m4, float m5, float& o0, float& o1,
                                         float& o2, float& o3) {
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    o0 = (m0 + s12) + s34;
    o1 = __builtin_fmaf(2.0f, d34, d12);
    o2 = __builtin_fmaf(4.0f, s34, s12);
    o3 = __builtin_fmaf(8.0f, d34, d12) + m5;
}

// One 6-vector through the A^T of F(2x2, 5x5) = [[1,1,1,1,1,0],[0,1,-1,2,-2,1]]
__device__ __forceinline__ void wino2_at(float m0, float m1, float m2, float m3, float m4, float m5, float& o0, float& o1) {
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    o0 = (m0 + s12) + s34;
    o1 = __builtin_fmaf(2.0f, d34, d12) + m5;
}

// U = G g G^T for one (k, c) of a 5x5 kernel: G (6x5) row p = scale_p * [1, p, p^2, p^3, p^4] for the points 0, 1, -1, 2, -2
// (scales 1/4, -1/6, -1/6, 1/24, 1/24), last row [0,0,0,0,1]; same panel layout as wino4_pack_kernel.
__global__ __launch_bounds__(kBlock) void wino25_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int K, int C,
                                                              int n_stages) {
    constexpr int KB = 32;
    const int kUStage = kXi4 * kCB * KB;
    const int total = ((K + KB - 1) / KB) * KB * C;
    const float G[6][5] = {{0.25f, 0.0f, 0.0f, 0.0f, 0.0f},
                           {-1.0f / 6.0f, -1.0f / 6.0f, -1.0f / 6.0f, -1.0f / 6.0f, -1.0f / 6.0f},
                           {-1.0f / 6.0f, 1.0f / 6.0f, -1.0f / 6.0f, 1.0f / 6.0f, -1.0f / 6.0f},
                           {1.0f / 24.0f, 2.0f / 24.0f, 4.0f / 24.0f, 8.0f / 24.0f, 16.0f / 24.0f},
                           {1.0f / 24.0f, -2.0f / 24.0f, 4.0f / 24.0f, -8.0f / 24.0f, 16.0f / 24.0f},
                           {0.0f, 0.0f, 0.0f, 0.0f, 1.0f}};
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int k = e / C, c = e - k * C;
        float g[5][5];
#pragma unroll
        for (int i = 0; i < 25; ++i) g[i / 5][i % 5] = (k < K) ? w[((size_t)k * C + c) * 25 + i] : 0.0f;
        float r[6][5];                       // G g
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                float acc = 0.0f;
#pragma unroll
                for (int z = 0; z < 5; ++z) acc += G[i][z] * g[z][q];
                r[i][q] = acc;
            }
        const int kb = k / KB, kl = k % KB, st = c / kCB, cl = c % kCB;
        float* up = u + ((size_t)kb * (n_stages + 1) + st) * kUStage;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float acc = 0.0f;
#pragma unroll
                for (int z = 0; z < 5; ++z) acc += r[i][z] * G[j][z];
                up[wino4_u_offset(i, j, cl, kl)] = acc;
            }
    }
}

typedef float w4_float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float w4_edge(float e, int) { return e; }
__device__ __forceinline__ float w4_edge(w4_float2v e, int i) { return e[i]; }
typedef float w4_float4v __attribute__((ext_vector_type(4)));
// a buffer load as wide as its destination (the WHOLE vector is cast: hipcc (ROCm 7.2) lowers a b128 load whose lanes are cast one
// by one to a dword load)
__device__ __forceinline__ void w4_load(w4_float4v& d, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    d = __builtin_bit_cast(w4_float4v, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void w4_load(w4_float2v& d, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    d = __builtin_bit_cast(w4_float2v, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void w4_load(float& d, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    d = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// ---- the input transform V = B^T d B of the six-point kernels in packed fp32 (v_pk_fma_f32 / v_pk_add_f32: two lanes of arithmetic per
// instruction): 84 (M = 4) / 72 (M = 2) instructions for the 144 scalar operations of wino4_bt twelve times over -- the same
// operations in the same order with the same roundings (a + b is fma(b, 1, a), d4 - d2 is fma(d2, -1, d4): exact products), so the
// bits do not change.  The producers' transform is the long pole of a stage (s_memtime stamps: 2700 cycles of 3500).
__device__ __forceinline__ w4_float2v w4_fma2(w4_float2v a, w4_float2v b, w4_float2v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ w4_float2v w4_splat(float x) { return w4_float2v{x, x}; }

// pass 1: B^T along the rows of a PAIR of columns (elementwise on the pair)
__device__ __forceinline__ void wino4_bt2(w4_float2v d0, w4_float2v d1, w4_float2v d2, w4_float2v d3, w4_float2v d4, w4_float2v d5,
                                          w4_float2v& o0, w4_float2v& o1, w4_float2v& o2, w4_float2v& o3, w4_float2v& o4, w4_float2v& o5) {
    o0 = w4_fma2(w4_splat(4.0f), d0, w4_fma2(w4_splat(-5.0f), d2, d4));
    const w4_float2v a = w4_fma2(w4_splat(-4.0f), d2, d4), b = w4_fma2(w4_splat(-4.0f), d1, d3);
    o1 = a + b;
    o2 = a - b;
    const w4_float2v e = d4 - d2, f = d3 - d1;
    o3 = w4_fma2(w4_splat(2.0f), f, e);
    o4 = w4_fma2(w4_splat(-2.0f), f, e);
    o5 = w4_fma2(w4_splat(4.0f), d1, w4_fma2(w4_splat(-5.0f), d3, d5));
}

// pass 2: B^T along the six columns m0..m5 of one row, the columns arriving as the pairs of pass 1.
// PAIRING 0: pa = {m1, m2}, pb = {m3, m4}, pc = {m0, m5} (M = 4: a lane's own 16 bytes are columns 1..4, columns 0 and 5 come from
// its neighbours);  PAIRING 1: pa = {m0, m1}, pb = {m2, m3}, pc = {m4, m5} (M = 2: own 8 bytes are columns 2..3).
// {a, e} = {m4 - 4 m2, m4 - m2} and {b, f} = {m3 - 4 m1, m3 - m1} are one packed fma each, {o1, o3} = {a + b, e + 2 f} and
// {o2, o4} = {a - b, e - 2 f} too.
template <int PAIRING>
__device__ __forceinline__ void wino4_bt_row(w4_float2v pa, w4_float2v pb, w4_float2v pc, float& o0, float& o1, float& o2, float& o3,
                                             float& o4, float& o5) {
    const w4_float2v k41 = {-4.0f, -1.0f}, k12 = {1.0f, 2.0f}, k12n = {-1.0f, -2.0f};
    w4_float2v ae, bf;
    if (PAIRING == 0) {
        ae = w4_fma2(pa.yy, k41, pb.yy);
        bf = w4_fma2(pa.xx, k41, pb.xx);
        o0 = __builtin_fmaf(4.0f, pc.x, __builtin_fmaf(-5.0f, pa.y, pb.y));
        o5 = __builtin_fmaf(4.0f, pa.x, __builtin_fmaf(-5.0f, pb.x, pc.y));
    } else {
        ae = w4_fma2(pb.xx, k41, pc.xx);
        bf = w4_fma2(pa.yy, k41, pb.yy);
        const w4_float2v o05 = w4_fma2(w4_splat(4.0f), pa, w4_fma2(w4_splat(-5.0f), pb, pc));
        o0 = o05.x;
        o5 = o05.y;
    }
    const w4_float2v o13 = w4_fma2(bf, k12, ae), o24 = w4_fma2(bf, k12n, ae);
    o1 = o13.x; o3 = o13.y;
    o2 = o24.x; o4 = o24.y;
}


// The output transforms of wino4_at / wino2_at on PAIRS (two accumulator registers, or two columns) in packed fp32: the same
// operations in the same order on each half (an fp32 add is exact-rounding identical in v_add_f32 and v_pk_add_f32), half the
// vector instructions -- and every vector instruction of an epilogue is matrix time of the other consumer group.
__device__ __forceinline__ void wino4_at2(w4_float2v m0, w4_float2v m1, w4_float2v m2, w4_float2v m3, w4_float2v m4, w4_float2v m5,
                                          w4_float2v& o0, w4_float2v& o1, w4_float2v& o2, w4_float2v& o3) {
    const w4_float2v s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    o0 = (m0 + s12) + s34;
    o1 = w4_fma2(w4_splat(2.0f), d34, d12);
    o2 = w4_fma2(w4_splat(4.0f), s34, s12);
    o3 = w4_fma2(w4_splat(8.0f), d34, d12) + m5;
}
__device__ __forceinline__ void wino2_at2(w4_float2v m0, w4_float2v m1, w4_float2v m2, w4_float2v m3, w4_float2v m4, w4_float2v m5,
                                          w4_float2v& o0, w4_float2v& o1) {
    const w4_float2v s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    o0 = (m0 + s12) + s34;
    o1 = w4_fma2(w4_splat(2.0f), d34, d12) + m5;
}

// M = 4: F(4x4, 3x3), pad 1.  M = 2: F(2x2, 5x5), pad 2 -- the same six interpolation points, hence the same B^T, the same 36 products
// per channel and patch (for 4 outputs of a 5x5 window: 9 per output instead of 25) and the same kernel; only the gather geometry (an
// aligned 8-byte load per row, TWO columns from each neighbour), the weight transform G (6x5) and the output transform A^T (2x6) differ.
//
// PERSISTENT: the grid is two workgroups per CU and a workgroup walks tiles (32 output channels x 32 patches) L, L + G, L + 2G, ...
// The producers' pipeline (gather two stages ahead, transform one stage ahead) runs ACROSS the tile boundary: in the last two stages
// of a tile they gather stages 0 / 1 of the next tile and transform its stage 0, so that the next tile's main loop starts as soon as
// the epilogue is over (one workgroup per tile spent 15 k cycles of a 50-110 k cycle life between its first instruction and its first
// MFMA, and two workgroups per CU cannot cover that for each other: s_memtime stamps, scripts/stamps_wino4.py).
// RAGGED: extents that are not multiples of M (14x14, 7x7): the last patch of a row / column hangs over the edge; its input columns
// at and beyond the width are zeroed after the load (the row behind them belongs to the next row), rows beyond the height are
// out-of-range offsets, and the epilogue stores only what exists.  F(4x4,3x3) on 14x14 executes 0.33 of the direct multiplies
// (16x16 computed for 14x14) against F(2x2)'s 0.44; on 7x7 0.33 against 0.58.
template <int M, int ABL, bool RAGGED = false>   // ABL: diagnostic ablations (wrong results on purpose): 1 no gather, 2 no transform, 3 no MFMA, 4 no U loads
__global__ __launch_bounds__(512, 4) void conv_wino4_kernel(WinoArgs a) {
    static_assert(M == 4 || M == 2, "F(4x4,3x3) or F(2x2,5x5)");
    constexpr int PAD = (M == 4) ? 1 : 2;               // 6 = M + kernel - 1 input rows / columns starting at M*t - PAD
    constexpr int KB = 32, NT = 32, WAVES = 8, CONSUMERS = 6;
    constexpr unsigned kOob = 0x80000000u;
    // V0 first: the epilogue's exchange area is V1 and the space behind it, and V0 holds stage 0 of the NEXT tile by then.  The
    // weight images U do not pass through LDS: every consumer loads its own MFMA A operands straight into registers (three 16-byte
    // loads per lane and stage from a panel packed in that order: wino4_u_offset) -- an LDS-DMA piece costs its wave 60-185 issue
    // cycles (18 per stage were 0.18 of conv2/3x3's 0.81 ms), a register load a tenth of that, and twelve LDS reads per stage go too.
    struct Stage {
        float V0[kXi4][kCB][NT];
        float V1[kXi4][kCB][NT];
        float ex_tail[(48 - 18) * 256];
    };
    __shared__ __attribute__((aligned(1024))) Stage sm;
    static_assert(sizeof(Stage) == 66 * 1024, "66 KB of LDS (+ 32 bytes of role table): two workgroups per CU");
    const unsigned long long t_entry = PVW4_NOW();
    const unsigned long long r_entry = (ABL == 5) ? __builtin_amdgcn_s_memrealtime() : 0ull;      // 100 MHz
    (void)r_entry;

    const int G = gridDim.x;
    int       L;                 // first tile: workgroups of one XCD (blockIdx & 7) take neighbouring tiles (same patches, other channel blocks)
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = G >> 3, r = G & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int n_tiles = a.n_tiles;
    const int n_eff   = a.n_stages + (a.n_stages & 1);       // an odd stage count runs one more stage on the panel's spare zero image

    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid  = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int l31 = lane & 31, lh = lane >> 5;
    const int HW   = a.H * a.W;
    const int TPI  = a.TY * a.TX;
    const unsigned chan_bytes = (unsigned)HW * 4u;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u_bytes, 0x00020000);
    typedef const __attribute__((address_space(4))) float* const_float_p;      // constant address space: scalar loads
    const const_float_p bias_c = (const_float_p)(unsigned long)a.bias;
    constexpr unsigned u_stage_bytes = (unsigned)kXi4 * kCB * KB * 4u;
    const unsigned u_lane = (unsigned)lane * 16u;

    // the A operands of this consumer's twelve MFMAs of a stage: group g_ = MFMAs 4 g_ .. 4 g_ + 3
#define PVW4_LOAD_U(ua_, so_, g_)      /* so_: byte offset of the stage image (channel block and stage) in the panel */             \
    w4_load(ua_[g_], ur, u_lane + (unsigned)(row * 3 + (g_)) * 1024u, (so_))

    // ---- roles.  Two workgroups share a CU, and waves w and w + 4 of a workgroup share a SIMD: with fixed roles (producers = waves
    // 6, 7) two SIMDs carry four consumers (48 MFMAs per stage of both workgroups) and two carry two consumers and two producers (24).
    // So the producers of a workgroup go on the SIMD pair {0, 1} or {2, 3} by WHICH of the CU's two LDS slots the workgroup got
    // (HW_REG_LDS_ALLOC.LDS_BASE: co-resident workgroups differ in it by construction): every SIMD then carries three consumers and
    // one producer.  SIMD ids from HW_REG_HW_ID; if waves 4-7 are not on four different SIMDs (never seen) the fixed roles are used.
    __shared__ unsigned simd_of[WAVES];
    int  row, pidx;                 // consumer: row of the 6x6 transform domain;  producer: which pair of the stage's channels
    bool producer;
    {
        const unsigned sid  = (__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11))) & 3u;           // HW_ID.SIMD_ID = bits [5:4]
        const unsigned slot = (__builtin_amdgcn_s_getreg(6 | (0 << 6) | (11 << 11)) != 0u) ? 1u : 0u;   // LDS_ALLOC.LDS_BASE != 0
        if (lane == 0) simd_of[wid] = sid;
        __syncthreads();
        const unsigned s4 = simd_of[4], s5 = simd_of[5], s6 = simd_of[6], s7 = simd_of[7];
        const bool spread = a.balance != 0 && ((1u << s4) | (1u << s5) | (1u << s6) | (1u << s7)) == 0xFu;
        if (wid < 4) { producer = false; row = wid; pidx = 0; }
        else if (spread) { producer = (sid >> 1) == slot; row = 4 + (int)(sid & 1u); pidx = (int)(sid & 1u); }
        else { producer = wid >= CONSUMERS; row = wid; pidx = wid - CONSUMERS; }
        producer = __builtin_amdgcn_readfirstlane(producer);
        row      = __builtin_amdgcn_readfirstlane(row);
        pidx     = __builtin_amdgcn_readfirstlane(pidx);
    }

    // ---- output transform Y = A^T D A (A^T: M x 6) of one tile: the column half in registers (consumer i holds row i), the rows meet
    // in LDS, CH channels at a time: Ex[i][CH][patch][c'] = 6 * CH * 32 * M floats = 48 KB (M = 4: 16 channels, M = 2: 32) = V1 and
    // the space behind it; every wave takes part in the second half.  Both roles carry a copy (the registers live across it differ).
    constexpr int CH = (M == 4) ? 16 : 32, PASSES = 32 / CH;
    float* const Ex = &sm.V1[0][0][0];
    const int OH = a.H, OW = a.W;
    const ActBounds ab = act_bounds(a.act, a.act_lo, a.act_hi);
    typedef float exv_t __attribute__((ext_vector_type(M)));      // the M column outputs of a (row, channel, patch): one LDS access
#define PVW4_EPI_WRITE(pass)                                                                                     \
    {                                                                                                            \
        _Pragma("unroll") for (int rr = 0; rr < CH / 2; rr += 2) {          /* two accumulator registers per step: packed fp32 (round 4) */ \
            const int r  = pass * (CH / 2) + rr;                          /* accumulator register */              \
            const int kl = (rr & 3) + 8 * (rr >> 2) + 4 * lh_e;           /* channel inside the pass: 0 .. CH-1 (register r + 1: the next one) */ \
            w4_float2v mm_[6], so2_[4];                                                                          \
            _Pragma("unroll") for (int j = 0; j < 6; ++j) mm_[j] = w4_float2v{acc[j][r], acc[j][r + 1]};         \
            if (M == 4) wino4_at2(mm_[0], mm_[1], mm_[2], mm_[3], mm_[4], mm_[5], so2_[0], so2_[1], so2_[2], so2_[3]); \
            else        wino2_at2(mm_[0], mm_[1], mm_[2], mm_[3], mm_[4], mm_[5], so2_[0], so2_[1]);             \
            exv_t sv0_, sv1_;                                                                                    \
            _Pragma("unroll") for (int c2 = 0; c2 < M; ++c2) { sv0_[c2] = so2_[c2].x; sv1_[c2] = so2_[c2].y; }   \
            *reinterpret_cast<exv_t*>(Ex + ((row * CH + kl) * 32 + tl) * M) = sv0_;                              \
            *reinterpret_cast<exv_t*>(Ex + ((row * CH + kl + 1) * 32 + tl) * M) = sv1_;                          \
        }                                                                                                        \
    }
#define PVW4_EPI_NOWRITE(pass) {}
#define PVW4_EPILOGUE(WRITE_, tile_)                                                                             \
    {                                                                                                            \
        const int kb_e = (tile_) % a.n_kb, tb_e = (tile_) / a.n_kb;                                              \
        /* The lane id is made HERE, by an asm hipcc can neither hoist nor merge: every per-thread value of the epilogue is then   */ \
        /* computed per tile.  Derived from threadIdx they are loop invariants, hipcc keeps them across the main loop, has no      */ \
        /* registers for them there (128), spills them -- and a reload from scratch is a VECTOR MEMORY load: its wait is            */ \
        /* vmcnt(0), i.e. for every store of the previous pass to be acknowledged and every gather in flight to land.              */ \
        int lane_e;                                                                                              \
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));            \
        const int tl = lane_e & 31, lh_e = lane_e >> 5;                            /* 512 threads = 16 channels x 32 patches per sweep */ \
        const int t  = tb_e * NT + tl;                                                                           \
        const int tc = t < a.T ? t : 0;                                                                          \
        const int n_ = w4_div(tc, a.tpi_mul, a.tpi_sh), rem_ = tc - n_ * TPI;                                    \
        const int ty_ = w4_div(rem_, a.tx_mul, a.tx_sh), tx_ = rem_ - ty_ * a.TX;                                \
        float* __restrict__ const yp0 = a.y + (((size_t)n_ * a.y_ctotal + a.y_coff) * OH + M * ty_) * OW + M * tx_; \
        const int  rows_ok = min(M, OH - M * ty_), cols_ok = min(M, OW - M * tx_);                               \
        const bool even_w = (OW & 1) == 0;                                                                       \
        (void)rows_ok; (void)cols_ok; (void)even_w;                                                              \
        _Pragma("unroll") for (int pass = 0; pass < PASSES; ++pass) {                                            \
            const unsigned long long p0_ = PVW4_NOW();                                                           \
            if (pass == 1) __syncthreads();                                                                      \
            const unsigned long long p1_ = PVW4_NOW();                                                           \
            WRITE_(pass);                                                                                        \
            const unsigned long long p2_ = PVW4_NOW();                                                           \
            __syncthreads();                                                                                     \
            const unsigned long long p3_ = PVW4_NOW();                                                           \
            if (pass == 1) ep[3] += p1_ - p0_;                                                                   \
            ep[pass * 4 + 0] += p2_ - p1_;                                                                       \
            ep[pass * 4 + 1] += p3_ - p2_;                                                                       \
            _Pragma("unroll") for (int sweep = 0; sweep < CH / 16; ++sweep) {                                    \
                const int kl = 2 * wid + lh_e + 16 * sweep;                      /* = (tid >> 5) + 16 * sweep */   \
                const int kg = kb_e * KB + pass * CH + kl;                                                       \
                /* the bias by SCALAR loads (the wave's two channels are wave-uniform): a vector load here made every pass wait */ \
                /* for vmcnt(0) -- behind the previous pass's stores and the producers' gathers in flight (stores count in vmcnt) */ \
                float bs0 = -0.0f, bs1 = -0.0f;                  /* no bias: v + -0.0 == v for every v, -0.0 included */ \
                if (a.bias != nullptr) {                                                                         \
                    const int kgs = kb_e * KB + pass * CH + 2 * wid + 16 * sweep;                                \
                    bs0 = bias_c[min(kgs, a.K - 1)];                                                             \
                    bs1 = bias_c[min(kgs + 1, a.K - 1)];                                                         \
                }                                                                                                \
                if (t < a.T && kg < a.K) {                                                                       \
                    const float bv = lh_e ? bs1 : bs0;                                                           \
                    float* __restrict__ yp = yp0 + (size_t)kg * (OH * OW);                                       \
                    exv_t ev[6];                                                                                 \
                    _Pragma("unroll") for (int i = 0; i < 6; ++i) ev[i] = *reinterpret_cast<const exv_t*>(Ex + ((i * CH + kl) * 32 + tl) * M); \
                    float yv[M][M];                                                                              \
                    _Pragma("unroll") for (int c2 = 0; c2 < M; c2 += 2) {       /* two columns per step: packed fp32, the bias add too */ \
                        w4_float2v em_[6], col2_[4];                                                             \
                        _Pragma("unroll") for (int i = 0; i < 6; ++i) em_[i] = w4_float2v{ev[i][c2], ev[i][c2 + 1]}; \
                        if (M == 4) wino4_at2(em_[0], em_[1], em_[2], em_[3], em_[4], em_[5], col2_[0], col2_[1], col2_[2], col2_[3]); \
                        else        wino2_at2(em_[0], em_[1], em_[2], em_[3], em_[4], em_[5], col2_[0], col2_[1]); \
                        _Pragma("unroll") for (int r2 = 0; r2 < M; ++r2) {                                       \
                            const w4_float2v yb_ = col2_[r2] + w4_splat(bv);                                     \
                            yv[r2][c2] = yb_.x;                                                                  \
                            yv[r2][c2 + 1] = yb_.y;                                                              \
                        }                                                                                        \
                    }                                                                                            \
                    /* bias and activation on all M x M values, the bounds behind WAVE-UNIFORM branches: act_apply's two compare-and- */ \
                    /* selects per value were 64 of the ~210 vector instructions of this block (plus 16 selects for "bias or not"),   */ \
                    /* and every vector instruction here competes with the other workgroup's MFMAs for the SIMD's issue slots          */ \
                    if (a.act != 0) {                                                                            \
                        _Pragma("unroll") for (int r2 = 0; r2 < M; ++r2)                                         \
                            _Pragma("unroll") for (int c2 = 0; c2 < M; ++c2) yv[r2][c2] = (yv[r2][c2] < ab.lo) ? ab.lo : yv[r2][c2]; \
                    }                                                                                            \
                    if (a.act == 2) {                                                                            \
                        _Pragma("unroll") for (int r2 = 0; r2 < M; ++r2)                                         \
                            _Pragma("unroll") for (int c2 = 0; c2 < M; ++c2) yv[r2][c2] = (yv[r2][c2] > ab.hi) ? ab.hi : yv[r2][c2]; \
                    }                                                                                            \
                    _Pragma("unroll") for (int r2 = 0; r2 < M; ++r2) {                                           \
                        float ov[M];                                                                             \
                        _Pragma("unroll") for (int c2 = 0; c2 < M; ++c2) ov[c2] = yv[r2][c2];                    \
                        if (RAGGED) {            /* only what exists; pairs where the rows are 8-byte aligned (even width) */ \
                            if (r2 < rows_ok) {                                                                  \
                                _Pragma("unroll") for (int c2 = 0; c2 < M; c2 += 2) {                            \
                                    if (even_w && c2 + 1 < cols_ok) conv_store2(yp + (size_t)r2 * OW + c2, ov[c2], ov[c2 + 1]); \
                                    else {                                                                       \
                                        if (c2 < cols_ok) conv_store1(yp + (size_t)r2 * OW + c2, ov[c2]);                 \
                                        if (c2 + 1 < cols_ok) conv_store1(yp + (size_t)r2 * OW + c2 + 1, ov[c2 + 1]);         \
                                    }                                                                            \
                                }                                                                                \
                            }                                                                                    \
                        } else if (M == 4) conv_store4(yp + (size_t)r2 * OW, ov[0], ov[1], ov[2], ov[3]); \
                        else        conv_store2(yp + (size_t)r2 * OW, ov[0], ov[1]); \
                    }                                                                                            \
                }                                                                                                \
            }                                                                                                    \
            ep[pass * 4 + 2] += PVW4_NOW() - p3_;                                                                \
        }                                                                                                        \
        { const unsigned long long q0_ = PVW4_NOW(); __syncthreads(); ep[7] += PVW4_NOW() - q0_; } /* the exchange area is read out: V1 may be written again */ \
    }

    unsigned long long st[4] = {0ull, 0ull, 0ull, 0ull};      // ABL = 5 only
    unsigned long long ep[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    (void)ep;
    unsigned long long t_epi = 0ull, t_head = 0ull;
    (void)t_epi; (void)t_head;
    if (producer) {
        // ------------------------------------------------------------------ producers
        // round 5: the two producer waves at priority 3 -- their transform is the long pole of a stage and the arbiter prefers the (older) consumers'
        // MFMA streams (lesson 45).  Same box, alternating: the nine 5x5 layers -1 .. -4 % (sum with 3a/3x3 0.612 -> 0.599 ms), same bits; choosing them
        // among the four OLDEST waves on top of it: another -0.4 %, not kept.  PVHIP_TUNE7=1: priority 0 (scripts/time_wino4_prio.py).
        if (a.p_prio != 0) __builtin_amdgcn_s_setprio(3);
        const int g_chan = pidx * 2 + lh;                             // channel inside the stage
        // Lane <-> (patch, channel of the stage).  Columns 1..4 of a patch row are ONE aligned 16-byte load (extents are
        // multiples of 4; consecutive lanes = consecutive patches = consecutive 16-byte pieces: fully coalesced).  Column 0 is
        // the left neighbour's column 4 and column 5 the right neighbour's column 1: they come from the adjacent lane by DPP
        // (wave_shr / wave_shl) when the transform runs, so a stage in flight holds 24 registers, not 36; only the first
        // and the last lane of a 32-patch row load their outer column themselves (one more dword load per row, two active
        // lanes per half).  A row in the padding gets an out-of-range offset (the hardware returns 0); columns in the
        // padding (first / last patch column) are zeroed by select.
        typedef w4_float4v float4v;
        typedef w4_float2v float2v;
        unsigned rowo[6];                 // byte offset of (row, first inner column) of this lane's channel
        unsigned eo[6];                   // edge lanes: offset of the outer column(s) of row r; every other lane: out of range
        bool     zlo, zhi, zlo_n, zhi_n;
        int      nv = M | 256, nv_n = M | 256;    // RAGGED: which columns of the lane's patch exist (see PVW4_ADDRESSES)
        const bool first = l31 == 0, last = l31 == 31;
        // addresses of a tile's patches (a tile past the end: every row out of range -- zeros, no traffic)
#define PVW4_ADDRESSES(tile_, zl_, zh_, nv_)                                                                       \
    {                                                                                                            \
        const int  t    = ((tile_) / a.n_kb) * NT + l31;                                                         \
        const bool live = (tile_) < n_tiles && t < a.T;                                                          \
        const int  tq = live ? t : 0;                                                                            \
        const int  n = w4_div(tq, a.tpi_mul, a.tpi_sh), rem = tq - n * TPI;                                      \
        const int  ty = w4_div(rem, a.tx_mul, a.tx_sh), tx = rem - ty * a.TX;                                    \
        const unsigned base = (unsigned)(n * a.C * HW + g_chan * HW + M * tx) * 4u;                              \
        zl_ = tx == 0;                                                                                           \
        zh_ = tx == a.TX - 1;                                                                                    \
        /* RAGGED: bits 0-7 the inner columns that exist, bit 8: the SECOND column a last lane loads for itself exists (M = 2) */ \
        if (RAGGED) nv_ = min(M, a.W - M * tx) | ((M * tx + M + 1 < a.W) ? 256 : 0);                             \
        /* M = 4: inner columns 1..4 are the lane's own 16 bytes; first lane: column 0 (4 bytes before; on the left border the */ \
        /* value is zeroed anyway: stay in place); last lane: column 5 (16 bytes after; on the right border: 12).               */ \
        /* M = 2: inner columns 2..3 are the lane's own 8 bytes; first lane: columns 0..1 (8 bytes before), last: 4..5 (8 after). */ \
        unsigned eoff;                                                                                           \
        if (M == 4) eoff = first ? (zl_ ? 0u : 0xFFFFFFFCu) : (zh_ ? 12u : 16u);                                 \
        else        eoff = first ? (zl_ ? 0u : 0xFFFFFFF8u) : (zh_ ? 0u : 8u);                                   \
        _Pragma("unroll") for (int r = 0; r < 6; ++r) {                                                          \
            const int iy = M * ty - PAD + r;                                                                     \
            rowo[r] = (live && (unsigned)iy < (unsigned)a.H) ? base + (unsigned)(iy * a.W) * 4u : kOob + 16u;    \
            eo[r]   = (first || last) ? rowo[r] + eoff : kOob;                                                   \
            asm volatile("" : "+v"(rowo[r]), "+v"(eo[r]));                                                       \
        }                                                                                                        \
    }
        // A gather = 12 loads; its row / edge offsets are loop invariants held in registers: with address temporaries inside the loop
        // (the first version) hipcc reused a temporary's register as a load destination and put a wait for the WHOLE previous gather
        // in front of the last loads of the next one -- one stage in flight instead of two (LESSONS.md lesson 23).
        using VecT = typename std::conditional<M == 4, float4v, float2v>::type;      // own (inner) columns of a row
        using EdgT = typename std::conditional<M == 4, float, float2v>::type;        // edge lanes: their outer column(s)
        VecT vA[6], vB[6];
        EdgT eA[6], eB[6];
        // Compiler-tracked loads ON PURPOSE.  Asm loads with hand-counted vmcnt waits were 0-2 % faster, but hipcc is free to COPY
        // a register between the asm that issues a load into it and the asm that waits for it -- it did, in the F(2x2,5x5)
        // instantiation (different physical registers for the same patch in the steady loop and in the tile-boundary code) -- and the
        // copy reads whatever the register held: wrong results whenever the load is slow, i.e. only with other kernels running
        // beside this one (LESSONS.md lesson 24).  With the edge offsets precomputed (no address temporaries in the loop) hipcc's own
        // waits come out where they belong: vmcnt(12) .. after the twelve loads of the next gather.
#define PVW4_GATHER(v_, e_, s_)                                                                                  \
    {                                                                                                            \
        const int se_ = (s_) < a.n_stages ? (s_) : a.n_stages - 1;     /* the padding stage: the last one again (its U is zero) */ \
        const unsigned soff = (unsigned)(se_ * kCB) * chan_bytes;                                                \
        _Pragma("unroll") for (int r = 0; r < 6; ++r) {                                                          \
            if (ABL == 1) {                                                                                      \
                v_[r] = __builtin_bit_cast(float, rowo[r] + soff);                                               \
                e_[r] = 0.0f;                                                                                    \
            } else if (ABL == 6) {          /* no edge loads: what do the six almost empty load instructions of a gather cost? */ \
                w4_load(v_[r], xr, rowo[r], soff);                                                               \
                e_[r] = EdgT{};                                                                                  \
            } else if (ABL == 7) {          /* no patch loads, only the edge loads */                            \
                v_[r] = __builtin_bit_cast(float, rowo[r] + soff);                                               \
                w4_load(e_[r], xr, eo[r], soff);                                                                 \
            } else {                                                                                             \
                w4_load(v_[r], xr, rowo[r], soff);                                                               \
                w4_load(e_[r], xr, eo[r], soff);                                                                 \
            }                                                                                                    \
        }                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    }
#define PVW4_TRANSFORM_STORE(v_, e_, Vb_, zl_, zh_, nv_)                                                              \
    {                                                                                                            \
        /* the six columns of row r as three pairs: lo = from the left neighbour (wave_shr:1; own load in the first lane), */ \
        /* hi = from the right neighbour (wave_shl:1; own load in the last lane), own = the lane's load                     */ \
        float2v pa[6], pb[6], pc[6];                                                                             \
        if (RAGGED) {            /* columns at / beyond the width: zero (before the neighbours read them by DPP) */ \
            _Pragma("unroll") for (int r = 0; r < 6; ++r)                                                        \
                _Pragma("unroll") for (int q = 1; q < M; ++q) v_[r][q] = q < ((nv_) & 255) ? v_[r][q] : 0.0f;   \
        }                                                                                                        \
        _Pragma("unroll") for (int r = 0; r < 6; ++r) {                                                          \
            constexpr int NB = (M == 4) ? 1 : 2;              /* columns taken from each neighbour */             \
            float lo_[NB], hi_[NB];                                                                              \
            _Pragma("unroll") for (int q = 0; q < NB; ++q) {                                                     \
                /* v_mov_b32_dpp, bound_ctrl: a lane without a source (lane 0 / 63) gets 0 -- and takes its own edge load anyway */ \
                const float nl = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (float)v_[r][M - NB + q]), 0x138, 0xf, 0xf, true)); \
                const float nh = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (float)v_[r][q]), 0x130, 0xf, 0xf, true)); \
                lo_[q] = zl_ ? 0.0f : (first ? w4_edge(e_[r], q) : nl);                                          \
                hi_[q] = zh_ ? 0.0f : (last ? ((RAGGED && q >= 1 && !((nv_) & 256)) ? 0.0f : w4_edge(e_[r], q)) : nh); \
            }                                                                                                    \
            if (M == 4) {                                                                                        \
                pa[r] = float2v{v_[r][0], v_[r][1]};                                                             \
                pb[r] = float2v{v_[r][M - 2], v_[r][M - 1]};                                                     \
                pc[r] = float2v{lo_[0], hi_[0]};                                                                 \
            } else {                                                                                             \
                pa[r] = float2v{lo_[0], lo_[NB - 1]};                                                         \
                pb[r] = float2v{v_[r][0], v_[r][1]};                                                             \
                pc[r] = float2v{hi_[0], hi_[NB - 1]};                                                         \
            }                                                                                                    \
        }                                                                                                        \
        float2v ma[6], mb[6], mc[6];                                        /* columns: m = B^T d */              \
        if (ABL == 2) {                                                                                          \
            _Pragma("unroll") for (int r = 0; r < 6; ++r) { ma[r] = pa[r]; mb[r] = pb[r]; mc[r] = pc[r]; }       \
        } else {                                                                                                 \
            wino4_bt2(pa[0], pa[1], pa[2], pa[3], pa[4], pa[5], ma[0], ma[1], ma[2], ma[3], ma[4], ma[5]);       \
            wino4_bt2(pb[0], pb[1], pb[2], pb[3], pb[4], pb[5], mb[0], mb[1], mb[2], mb[3], mb[4], mb[5]);       \
            wino4_bt2(pc[0], pc[1], pc[2], pc[3], pc[4], pc[5], mc[0], mc[1], mc[2], mc[3], mc[4], mc[5]);       \
        }                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) {                     /* rows: V = m B */                   \
            float v0, v1, v2, v3, v4, v5;                                                                        \
            if (ABL == 2) { v0 = mc[i].x; v1 = ma[i].x; v2 = ma[i].y; v3 = mb[i].x; v4 = mb[i].y; v5 = mc[i].y; } \
            else wino4_bt_row<(M == 4) ? 0 : 1>(ma[i], mb[i], mc[i], v0, v1, v2, v3, v4, v5);                    \
            Vb_[i * 6 + 0][g_chan][l31] = v0;                                                                    \
            Vb_[i * 6 + 1][g_chan][l31] = v1;                                                                    \
            Vb_[i * 6 + 2][g_chan][l31] = v2;                                                                    \
            Vb_[i * 6 + 3][g_chan][l31] = v3;                                                                    \
            Vb_[i * 6 + 4][g_chan][l31] = v4;                                                                    \
            Vb_[i * 6 + 5][g_chan][l31] = v5;                                                                    \
        }                                                                                                        \
    }
        int tile = L;
        PVW4_ADDRESSES(tile, zlo, zhi, nv);
        PVW4_GATHER(vA, eA, 0);
        PVW4_GATHER(vB, eB, 1);
        PVW4_TRANSFORM_STORE(vA, eA, sm.V0, zlo, zhi, nv);
        t_head = PVW4_NOW() - t_entry;
        for (;;) {
            __syncthreads();              // V(0) of this tile is in V0
            // stage s: V(s+1) from the gather issued one stage ago, gather of stage s+2 issued now (two stages in flight)
            for (int s = 0; s + 2 < n_eff; s += 2) {
                const unsigned long long t0 = PVW4_NOW();
                PVW4_GATHER(vA, eA, s + 2);
                const unsigned long long t1 = PVW4_NOW();
                const unsigned long long t2 = PVW4_NOW();
                PVW4_TRANSFORM_STORE(vB, eB, sm.V1, zlo, zhi, nv);
                const unsigned long long t3 = PVW4_NOW();
                __syncthreads();
                const unsigned long long t4 = PVW4_NOW();
                PVW4_GATHER(vB, eB, s + 3);
                const unsigned long long u1 = PVW4_NOW();
                const unsigned long long u2 = PVW4_NOW();
                PVW4_TRANSFORM_STORE(vA, eA, sm.V0, zlo, zhi, nv);
                const unsigned long long u3 = PVW4_NOW();
                __syncthreads();
                const unsigned long long u4 = PVW4_NOW();
                st[0] += (t1 - t0) + (u1 - t4); st[1] += (t2 - t1) + (u2 - u1); st[2] += (t3 - t2) + (u3 - u2); st[3] += (t4 - t3) + (u4 - u3);
            }
            // the last two stages: the gathers are stages 0 / 1 of the NEXT tile, and the second transform its stage 0
            PVW4_ADDRESSES(tile + G, zlo_n, zhi_n, nv_n);
            PVW4_GATHER(vA, eA, 0);
            PVW4_TRANSFORM_STORE(vB, eB, sm.V1, zlo, zhi, nv);
            __syncthreads();
            PVW4_GATHER(vB, eB, 1);
            PVW4_TRANSFORM_STORE(vA, eA, sm.V0, zlo_n, zhi_n, nv_n);
            __syncthreads();
            zlo = zlo_n;
            zhi = zhi_n;
            nv  = nv_n;
            const unsigned long long e0 = PVW4_NOW();
            PVW4_EPILOGUE(PVW4_EPI_NOWRITE, tile);
            t_epi += PVW4_NOW() - e0;
            tile += G;
            if (tile >= n_tiles) break;
        }
#undef PVW4_ADDRESSES
#undef PVW4_GATHER
#undef PVW4_TRANSFORM_STORE
    } else {
        // ------------------------------------------------------------------ consumers: one row of the 6x6 transform domain each
        floatx16 acc[6];
        static_assert(offsetof(Stage, V1) == offsetof(Stage, V0) + sizeof(float) * kXi4 * kCB * NT, "V1 follows V0");
        const float* const vbs = &sm.V0[0][0][0] + ((row * 6) * kCB + lh) * NT + l31;      // this lane's B operand of MFMA (j = 0, kk = 0) in V0
        // ABL 8: every workgroup the weights of channel block 0 (how much do the U loads of two workgroups that share a CU, but no weights, cost?)
#define PVW4_U_BASE(tile_) ((ABL == 8) ? 0u : (unsigned)(((tile_) % a.n_kb) * (a.n_stages + 1)) * u_stage_bytes)
        w4_float4v ua[3];
        {
            const unsigned u_first = PVW4_U_BASE(L);
            PVW4_LOAD_U(ua, u_first, 0);
            PVW4_LOAD_U(ua, u_first, 1);
            PVW4_LOAD_U(ua, u_first, 2);
        }
        for (int tile = L; tile < n_tiles; tile += G) {
            const unsigned long long h0 = PVW4_NOW();
            // The last stage of a tile loads stage 0 of the NEXT tile's weights: they arrive during the epilogue.  Loaded after it, the
            // first wait for them would also be a wait for the epilogue's stores (stores count in vmcnt and the counter retires in order).
            const unsigned u_base = PVW4_U_BASE(tile), u_next = PVW4_U_BASE(tile + G);
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
            __syncthreads();
            t_head += PVW4_NOW() - h0;
            for (int s = 0; s < n_eff; ++s) {
                const unsigned long long t0 = PVW4_NOW();
                const float* vb = vbs + (s & 1) * (kXi4 * kCB * NT);         // V1 follows V0: every read below is this address + a constant
                const unsigned so_n = s + 1 < n_eff ? u_base + (unsigned)(s + 1) * u_stage_bytes : u_next;
                // B operands one group (four MFMAs) ahead, A operands of a group reloaded for the next stage as soon as its MFMAs are
                // issued: the sched_barriers keep hipcc from sinking the loads to the end of the stage (it did)
                float bfr[2][4];
#define PVW4_READ_B(dst_, g_)                                                                                   \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                              \
        const int m = 4 * (g_) + e, kk = m / 6, j = m % 6;                                                       \
        dst_[e] = vb[(j * kCB + 2 * kk) * NT];                                                                   \
    }
                PVW4_READ_B(bfr[0], 0);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    if (g < 2) PVW4_READ_B(bfr[(g + 1) & 1], g + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int m = 4 * g + e, j = m % 6;
                        const float af = ua[g][e];
                        const float bf = bfr[g & 1][e];
                        if (ABL == 3) acc[j][0] += af * bf;
                        else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[j], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (ABL != 4) PVW4_LOAD_U(ua, so_n, g);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef PVW4_READ_B
                const unsigned long long t1 = PVW4_NOW();
                const unsigned long long t2 = t1;
                __syncthreads();
                const unsigned long long t3 = PVW4_NOW();
                st[0] += t1 - t0; st[1] += t2 - t1; st[2] += t3 - t2;
            }
            const unsigned long long e0 = PVW4_NOW();
            PVW4_EPILOGUE(PVW4_EPI_WRITE, tile);
            t_epi += PVW4_NOW() - e0;
        }
    }
#undef PVW4_LOAD_U
#undef PVW4_U_BASE
#undef PVW4_EPILOGUE
#undef PVW4_EPI_WRITE
#undef PVW4_EPI_NOWRITE
#ifdef PVHIP_DIAG
    if (ABL == 5 && blockIdx.x % 61 == 7 && lane == 0) {
        const unsigned long long t_end = PVW4_NOW();
        const int who = producer ? CONSUMERS + pidx : row;
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(&g_w4_stamps[who][i], st[i]);
        atomicAdd(&g_w4_stamps[who][4], t_end - t_entry);      // the workgroup's life
        atomicAdd(&g_w4_stamps[who][5], t_epi);                // in epilogues
        atomicAdd(&g_w4_stamps[who][6], t_head);
        atomicAdd(&g_w4_stamps[who][3], producer ? 0ull : (unsigned long long)(__builtin_amdgcn_s_memrealtime() - r_entry));   // consumers: life in 10 ns ticks               // consumers: tile heads (zero, U(0), first barrier); producers: before the first tile
        atomicAdd(&g_w4_stamps[who][7], 1ull);
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&g_w4_epi[who][i], ep[i]);
    }
#endif
}


// ======================================================================================================================
// The six-point kernels with the transformed patches SHARED by 64 output channels and no barrier in the main loop (round 4).
//
// conv_wino4_kernel gathers and transforms every input patch once per 32 output channels: its two producer waves issue ~600
// cycles of vector arithmetic per stage on SIMDs whose matrix pipe the consumers want for 2304 (fp32 MFMA and vector
// instructions of a SIMD never overlap: lessons 25, 35) -- a quarter of the matrix time, and the patch bytes behind it keep the
// CU's L1 miss queue full a third of the time.  Round 3 built the form in which two consumer groups (neighbouring channel
// blocks) read ONE transformed image per stage -- half the gathers and transforms per output -- as one 16-wave workgroup with one
// s_barrier per stage, and it lost (lesson 36): all twelve consumers leave the barrier together, run their MFMA segments on the
// same pipe at the same time, and the pipe then idles through every barrier, which the second workgroup of the CU used to fill.
//
// This form keeps the sharing and drops the barrier.  One workgroup of 16 waves per CU:
//   waves 0-5   consumers of channel block 2 kp     (row i of the 6x6 transform domain each, six accumulator tiles)
//   waves 6-11  consumers of channel block 2 kp + 1
//   waves 12-15 producers: pair p = waves 12 + 2p, 13 + 2p makes the stages q = p (mod 2) (lane <-> patch x channel of the stage)
// (waves are dealt round robin over the SIMDs: every SIMD carries three consumers and one producer).  The transformed images go
// through a RING of four LDS buffers, stage q (numbered across tiles) in buffer q & 3, and two counters per buffer replace the
// barrier: ready[b] counts producer waves that have finished writing (a consumer of stage q waits for 2 (q / 4 + 1)), done[b]
// counts consumer waves that have finished reading (a producer waits for 12 (q / 4) before it overwrites).  A consumer that finishes
// its MFMAs goes on to the next stage at once if its image is there -- nobody waits for the slowest wave of the CU, the twelve
// MFMA streams drift apart by themselves, and while one group is in its epilogue the other runs up to four stages ahead.
// The epilogue is per GROUP (six waves, 384 lanes): exchange area of its own, passes of 12 / 12 / 8 channels (register sets 0-5,
// 6-11, 12-15 of the accumulators: slot = 2 (register - first) + lane half), ordered by a counter barrier of the six waves.
// Same arithmetic in the same order as conv_wino4_kernel: the same bits (tests: test_conv_winograd_shared_v_has_the_bits...).
// For launches with an even number of channel blocks, a stage count that is a multiple of four and enough tiles (wino4_conv).
constexpr int kSRing = 4;
constexpr int kSEx   = 3;      // exchange buffers per consumer group (epilogue passes in flight)
constexpr int kSLag  = 3;      // stages the second consumer group starts behind the first (< kSRing)

// The two counter primitives, in ONE asm statement each (a spin loop or a branch in C++ between a wave's loads and their first use
// makes hipcc's waitcnt pass merge its bookkeeping over the extra control flow and wait for vmcnt(0) in front of the first MFMA of
// every stage: lessons 4, 31).  LDS executes a wave's instructions in order, so a counter increment issued behind a wave's LDS
// writes (reads) is performed behind them: whoever sees the new count sees the data (may overwrite the buffer).  The "memory"
// clobber is the compiler-side fence.  The load of the poll has its wait inside the same statement (lesson 24).
__device__ __forceinline__ void w4s_wait_ge(unsigned lds_addr, unsigned target) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned v, sv;
    asm volatile("1:\n\t"
                 "ds_read_b32 %0, %2\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_readfirstlane_b32 %1, %0\n\t"
                 "s_sub_i32 %1, %1, %3\n\t"
                 "s_cmp_ge_i32 %1, 0\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_sleep 1\n\t"
                 "s_branch 1b\n\t"
                 "2:"
                 : "=&v"(v), "=&s"(sv) : "v"(lds_addr), "s"(target) : "memory", "scc");
#endif
}
#ifdef PVHIP_DIAG
// one poll taken apart: cycles from before the LDS read to its arrival, and from there to behind the v_readfirstlane
__device__ __forceinline__ void w4s_poll_probe(unsigned lds_addr, unsigned long long& lds_cycles, unsigned long long& valu_cycles) {
    (void)lds_addr; (void)lds_cycles; (void)valu_cycles;
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned v, sv;
    unsigned long long ta, tb, tc;
    asm volatile("s_memtime %2\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "ds_read_b32 %0, %5\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "s_memtime %3\n\t"
                 "v_readfirstlane_b32 %1, %0\n\t"
                 "s_nop 0\n\t"
                 "s_memtime %4\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v), "=&s"(sv), "=&s"(ta), "=&s"(tb), "=&s"(tc) : "v"(lds_addr) : "memory");
    lds_cycles += tb - ta;
    valu_cycles += tc - tb;
#endif
}
#define PVS_PWAIT(addr_, target_) { w4s_poll_probe(addr_, st[6], st[5]); w4s_wait_ge(addr_, target_); }
#else
#define PVS_PWAIT(addr_, target_) w4s_wait_ge(addr_, target_)
#endif
__device__ __forceinline__ void w4s_signal(unsigned lds_addr) {        // one increment per WAVE (lane 0)
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "ds_add_u32 %1, %2\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(lds_addr), "v"(1u) : "memory");
#endif
}
// one non-blocking look at a counter (wave-uniform result)
__device__ __forceinline__ unsigned w4s_poll(unsigned lds_addr) {
    unsigned sv = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned v;
    asm volatile("ds_read_b32 %0, %2\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_readfirstlane_b32 %1, %0"
                 : "=&v"(v), "=&s"(sv) : "v"(lds_addr) : "memory");
#endif
    return sv;
}
// the increment with the count BEFORE it returned (wave-uniform): how many waves were here first
__device__ __forceinline__ unsigned w4s_signal_rank(unsigned lds_addr) {
    unsigned sv = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long saved;
    unsigned v;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "ds_add_rtn_u32 %1, %3, %4\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_readfirstlane_b32 %2, %1\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(saved), "=&v"(v), "=&s"(sv) : "v"(lds_addr), "v"(1u) : "memory");
#endif
    return sv;
}
#define PVS_LDS_ADDR(p_) ((unsigned)(unsigned long)(lds_ptr_t)(p_))
#ifdef PVHIP_DIAG
// diagnostic build: cycle accounts per wave of every 16th workgroup of conv_wino4s_kernel, [wave][account] (scripts/stamps_wino4s.py).
// consumers: 0 waiting for the stage's image, 1 LDS reads + MFMAs + weight loads, 2 epilogue work, 3 epilogue counter barriers, 5 stages, 6 tiles
// producers: 0 waiting for a free buffer, 1 transform + LDS writes, 2 gather issue, 5 own stages;  4 = the wave's life, 7 = workgroups
__device__ unsigned long long g_w4s_stamps[16][8];
__device__ unsigned long long g_w4s_simd[16][4];
__device__ unsigned g_w4s_trace[16][96][4];              // workgroup 3: cycle (since its start) at which wave w saw stage q's image / finished stage q (producers: began / finished writing own stage)
#define PVS_TRACE(w_, q_, k_, t_) { if (blockIdx.x == 3 && (q_) < 96u && (threadIdx.x & 63) == 0) g_w4s_trace[w_][q_][k_] = (unsigned)((t_) - t_entry); }          // how often wave w of a stamped workgroup ran on SIMD 0..3 (HW_REG_HW_ID)
#define PVS_NOW() ((unsigned long long)__builtin_readcyclecounter())
#define PVS_DATA_WAIT() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xF7C); __builtin_amdgcn_sched_barrier(0); }      /* vmcnt(12) */
#else
#define PVS_NOW() 0ull
#define PVS_DATA_WAIT() {}
#define PVS_TRACE(w_, q_, k_, t_) {}
#endif

template <int M, bool RAGGED>
__global__ __launch_bounds__(1024, 4) void conv_wino4s_kernel(WinoArgs a) {
    static_assert(M == 4 || M == 2, "F(4x4,3x3) or F(2x2,5x5)");
    constexpr int PAD = (M == 4) ? 1 : 2;
    constexpr int KB = 32, NT = 32, CONS = 12;
    constexpr unsigned kOob = 0x80000000u;
    constexpr int kExBuf = 6 * 4 * 32 * M;            // floats of one exchange buffer: [row][slot][patch][M columns], 4 slots = 2 accumulator registers x 2 lane halves
    struct Smem {
        float    V[kSRing][kXi4][kCB][NT];             // 4 x 18 KB
        float    Ex[2][kSEx][kExBuf];                  // per group: kSEx buffers of 12 KB (M = 4), pass P in buffer P % kSEx
        unsigned ready[kSRing], done[kSRing], exfull[2][kSEx], exfree[2][kSEx];      // [group][buffer]: a wave may be a pass ahead of its neighbours, so the passes of a buffer are counted apart
    };
    __shared__ __attribute__((aligned(1024))) Smem sm;
    static_assert(sizeof(Smem) <= 150 * 1024, "one workgroup per CU");

    const int G = gridDim.x;
    int       L;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = G >> 3, r = G & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int n_kp    = (a.n_kb + 1) >> 1;              // pairs of channel blocks (an odd count: the last pair is one block, its second consumer group only keeps the counters going)
    const int n_tb_s  = a.n_tiles / n_kp;               // patch blocks
#define PVS_PAIR(t_)  (a.s_order ? (t_) / n_tb_s : (t_) % n_kp)
#define PVS_BLOCK(t_) (a.s_order ? (t_) % n_tb_s : (t_) / n_kp)
    const int n_tiles = a.n_tiles;                      // patch blocks x pairs
    const int n_eff   = a.n_stages;                     // a multiple of 4 (wino4_conv)

    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid  = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int l31 = lane & 31, lh = lane >> 5;
    const int HW   = a.H * a.W;
    const int TPI  = a.TY * a.TX;
    const unsigned chan_bytes = (unsigned)HW * 4u;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u_bytes, 0x00020000);
    typedef const __attribute__((address_space(4))) float* const_float_p;
    const const_float_p bias_c = (const_float_p)(unsigned long)a.bias;
    constexpr unsigned u_stage_bytes = (unsigned)kXi4 * kCB * KB * 4u;
    constexpr int      v_buf_floats  = kXi4 * kCB * NT;

    if (tid < kSRing) { sm.ready[tid] = 0u; sm.done[tid] = 0u; }
    if (tid < 2 * kSEx) { (&sm.exfull[0][0])[tid] = 0u; (&sm.exfree[0][0])[tid] = 0u; }
    __syncthreads();                                    // the only barrier of the kernel
    const unsigned long long t_entry = PVS_NOW();
    unsigned long long st[7] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    (void)t_entry; (void)st;

    // Roles: the producers are the OLDEST waves of the workgroup (0-3) unless PVHIP_WINO_SHARED_OLD=0 (then the youngest, 12-15): the
    // SIMD's arbiter prefers older waves, and a producer that only gets issue slots when no consumer has an MFMA ready makes the
    // consumers wait for it (stamps: a v_readfirstlane of a young producer waits ~300 cycles for its slot, single ones thousands).
    const int p_first = a.s_old != 0 ? 0 : CONS;                   // first producer wave
    const int c_first = a.s_old != 0 ? 4 : 0;                      // first consumer wave
    if (wid >= p_first && wid < p_first + 4) {
        // ------------------------------------------------------------------ producers (the gather and the transform of conv_wino4_kernel)
        const int pair = (wid - p_first) >> 1, pidx = (wid - p_first) & 1;
        const int g_chan = pidx * 2 + lh;
        // Producers FIRST.  The SIMD's arbiter picks the oldest wave that has an instruction ready, and the consumers (waves 0-11) are
        // older: with their MFMA streams never interrupted by a barrier, a producer's vector instructions only got issue slots while
        // the consumers were waiting -- for the producers (stamps: 2100 cycles per transform, 560-600 in conv_wino4_kernel; consumers
        // waiting 1.2-1.5 k cycles per stage).  PVHIP_WINO_SHARED_PRIO=0 leaves every wave at priority 0 (A/B runs).
        if (a.s_prio != 0) __builtin_amdgcn_s_setprio(3);
        typedef w4_float4v float4v;
        typedef w4_float2v float2v;
        unsigned rowo[6], eo[6];
        bool     zlo, zhi, zlo_n, zhi_n;
        int      nv = M | 256, nv_n = M | 256;
        const bool first = l31 == 0, last = l31 == 31;
#define PVS_ADDRESSES(tile_, zl_, zh_, nv_)                                                                       \
    {                                                                                                            \
        const int  t    = PVS_BLOCK(tile_) * NT + l31;                                                           \
        const bool live = (tile_) < n_tiles && t < a.T;                                                          \
        const int  tq = live ? t : 0;                                                                            \
        const int  n = w4_div(tq, a.tpi_mul, a.tpi_sh), rem = tq - n * TPI;                                      \
        const int  ty = w4_div(rem, a.tx_mul, a.tx_sh), tx = rem - ty * a.TX;                                    \
        const unsigned base = (unsigned)(n * a.C * HW + g_chan * HW + M * tx) * 4u;                              \
        zl_ = tx == 0;                                                                                           \
        zh_ = tx == a.TX - 1;                                                                                    \
        if (RAGGED) nv_ = min(M, a.W - M * tx) | ((M * tx + M + 1 < a.W) ? 256 : 0);                             \
        unsigned eoff;                                                                                           \
        if (M == 4) eoff = first ? (zl_ ? 0u : 0xFFFFFFFCu) : (zh_ ? 12u : 16u);                                 \
        else        eoff = first ? (zl_ ? 0u : 0xFFFFFFF8u) : (zh_ ? 0u : 8u);                                   \
        _Pragma("unroll") for (int r = 0; r < 6; ++r) {                                                          \
            const int iy = M * ty - PAD + r;                                                                     \
            rowo[r] = (live && (unsigned)iy < (unsigned)a.H) ? base + (unsigned)(iy * a.W) * 4u : kOob + 16u;    \
            eo[r]   = (first || last) ? rowo[r] + eoff : kOob;                                                   \
            asm volatile("" : "+v"(rowo[r]), "+v"(eo[r]));                                                       \
        }                                                                                                        \
    }
        using VecT = typename std::conditional<M == 4, float4v, float2v>::type;
        using EdgT = typename std::conditional<M == 4, float, float2v>::type;
        VecT vA[6], vB[6];
        EdgT eA[6], eB[6];
#define PVS_GATHER(v_, e_, s_)                                                                                   \
    {                                                                                                            \
        const unsigned soff = (unsigned)((s_) * kCB) * chan_bytes;                                               \
        _Pragma("unroll") for (int r = 0; r < 6; ++r) {                                                          \
            w4_load(v_[r], xr, rowo[r], soff);                                                                   \
            w4_load(e_[r], xr, eo[r], soff);                                                                     \
        }                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    }
#define PVS_TRANSFORM_STORE(v_, e_, Vb_, zl_, zh_, nv_)                                                          \
    {                                                                                                            \
        float2v pa[6], pb[6], pc[6];                                                                             \
        if (RAGGED) {                                                                                            \
            _Pragma("unroll") for (int r = 0; r < 6; ++r)                                                        \
                _Pragma("unroll") for (int q = 1; q < M; ++q) v_[r][q] = q < ((nv_) & 255) ? v_[r][q] : 0.0f;   \
        }                                                                                                        \
        _Pragma("unroll") for (int r = 0; r < 6; ++r) {                                                          \
            constexpr int NB = (M == 4) ? 1 : 2;                                                                 \
            float lo_[NB], hi_[NB];                                                                              \
            _Pragma("unroll") for (int q = 0; q < NB; ++q) {                                                     \
                const float nl = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (float)v_[r][M - NB + q]), 0x138, 0xf, 0xf, true)); \
                const float nh = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (float)v_[r][q]), 0x130, 0xf, 0xf, true)); \
                lo_[q] = zl_ ? 0.0f : (first ? w4_edge(e_[r], q) : nl);                                          \
                hi_[q] = zh_ ? 0.0f : (last ? ((RAGGED && q >= 1 && !((nv_) & 256)) ? 0.0f : w4_edge(e_[r], q)) : nh); \
            }                                                                                                    \
            if (M == 4) {                                                                                        \
                pa[r] = float2v{v_[r][0], v_[r][1]};                                                             \
                pb[r] = float2v{v_[r][M - 2], v_[r][M - 1]};                                                     \
                pc[r] = float2v{lo_[0], hi_[0]};                                                                 \
            } else {                                                                                             \
                pa[r] = float2v{lo_[0], lo_[NB - 1]};                                                            \
                pb[r] = float2v{v_[r][0], v_[r][1]};                                                             \
                pc[r] = float2v{hi_[0], hi_[NB - 1]};                                                            \
            }                                                                                                    \
        }                                                                                                        \
        float2v ma[6], mb[6], mc[6];                                                                             \
        wino4_bt2(pa[0], pa[1], pa[2], pa[3], pa[4], pa[5], ma[0], ma[1], ma[2], ma[3], ma[4], ma[5]);           \
        wino4_bt2(pb[0], pb[1], pb[2], pb[3], pb[4], pb[5], mb[0], mb[1], mb[2], mb[3], mb[4], mb[5]);           \
        wino4_bt2(pc[0], pc[1], pc[2], pc[3], pc[4], pc[5], mc[0], mc[1], mc[2], mc[3], mc[4], mc[5]);           \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                          \
            float v0, v1, v2, v3, v4, v5;                                                                        \
            wino4_bt_row<(M == 4) ? 0 : 1>(ma[i], mb[i], mc[i], v0, v1, v2, v3, v4, v5);                         \
            Vb_[((i * 6 + 0) * kCB + g_chan) * NT + l31] = v0;                                                   \
            Vb_[((i * 6 + 1) * kCB + g_chan) * NT + l31] = v1;                                                   \
            Vb_[((i * 6 + 2) * kCB + g_chan) * NT + l31] = v2;                                                   \
            Vb_[((i * 6 + 3) * kCB + g_chan) * NT + l31] = v3;                                                   \
            Vb_[((i * 6 + 4) * kCB + g_chan) * NT + l31] = v4;                                                   \
            Vb_[((i * 6 + 5) * kCB + g_chan) * NT + l31] = v5;                                                   \
        }                                                                                                        \
    }
        // own stage o of a tile is stage s = 2 o + pair; own stages alternate between ring buffers `pair` (o even) and `pair + 2`
        // (scalar offsets turned into addresses where they are used: as pointers they sat in vector registers across the whole loop)
        const int vb_off = __builtin_amdgcn_readfirstlane(pair * v_buf_floats);
#define VbA (&sm.V[0][0][0][0] + vb_off)
#define VbB (&sm.V[0][0][0][0] + vb_off + 2 * v_buf_floats)
        const unsigned readyA = PVS_LDS_ADDR(&sm.ready[0]) + 4u * (unsigned)pair, readyB = readyA + 8u;
        const unsigned doneA  = PVS_LDS_ADDR(&sm.done[0]) + 4u * (unsigned)pair,  doneB  = doneA + 8u;
        unsigned  use = 0u;                              // how often each of this pair's buffers has been filled
        const int n_own = n_eff >> 1;                    // even
        // ---- the second half of the epilogue is the PRODUCERS' work (pair g for consumer group g): a producer spends most of a stage
        // waiting for a free ring buffer, while a consumer group that stores its own tile keeps its six MFMA streams idle for 15-22 k
        // cycles per tile (stamps).  The consumers only apply the column half of Y = A^T D A to two accumulator registers at a time
        // and leave them in one of the group's two exchange buffers (pass P -> buffer P & 1, counter exfull); a producer wave that
        // finds a pass complete while it waits takes its 64 (slot, patch) items out (exfree: the buffer may be written again), applies
        // the row half, bias and activation and stores.  Eight passes per tile; channel of (pass p, register half rr, lane half lh):
        // register r = 2 p + rr -> (r & 3) + 8 (r >> 2) + 4 lh.
        const int OH = a.H, OW = a.W;
        const ActBounds ab = act_bounds(a.act, a.act_lo, a.act_hi);
        typedef float exv_t __attribute__((ext_vector_type(M)));
        const unsigned exfull_a = PVS_LDS_ADDR(&sm.exfull[0][0]) + 4u * kSEx * (unsigned)pair, exfree_a = PVS_LDS_ADDR(&sm.exfree[0][0]) + 4u * kSEx * (unsigned)pair;      // buffer 0; buffer b: + 4 b
        const int      ex_off   = __builtin_amdgcn_readfirstlane(pair * kSEx * kExBuf);
        const unsigned p_total  = L < n_tiles ? 8u * (unsigned)((n_tiles - L + G - 1) / G) : 0u;
        unsigned       p_store  = 0u;                    // passes of this pair's group stored so far
        int            geo_it = -1, geo_ok = 0;          // the lane's output patch of the tile whose passes are being stored
        unsigned       geo_off = 0u;
        const bool     even_w   = (OW & 1) == 0;
        (void)even_w;
#define PVS_STORE_PASS()                                                                                         \
    {                                                                                                            \
        const int it_ = (int)(p_store >> 3), p_ = (int)(p_store & 7u);                                           \
        const int tile_s = L + it_ * G;                                                                          \
        const int kb_e = 2 * PVS_PAIR(tile_s) + pair, tb_e = PVS_BLOCK(tile_s);                                 \
        int lane_e;                                                                                              \
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));            \
        const int tl = lane_e & 31, lh_e = lane_e >> 5;                                                          \
        const int r_  = 2 * p_ + pidx;                                   /* accumulator register of this wave's two slots */ \
        const int kgs = kb_e * KB + (r_ & 3) + 8 * (r_ >> 2);            /* channel of lane half 0; half 1: + 4 */ \
        const int kg  = kgs + 4 * lh_e;                                                                          \
        float bs0 = -0.0f, bs1 = -0.0f;                                                                          \
        if (a.bias != nullptr) {                                                                                 \
            bs0 = bias_c[min(kgs, a.K - 1)];                                                                     \
            bs1 = bias_c[min(kgs + 4, a.K - 1)];                                                                 \
        }                                                                                                        \
        const unsigned pb_ = p_store % (unsigned)kSEx;                                                           \
        const float* const exb = &sm.Ex[0][0][0] + ex_off + (int)pb_ * kExBuf;                                   \
        const int slot = 2 * pidx + lh_e;                                                                        \
        exv_t ev[6];                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) ev[i] = *reinterpret_cast<const exv_t*>(exb + ((i * 4 + slot) * 32 + tl) * M); \
        w4s_signal(exfree_a + 4u * pb_);                 /* LDS executes a wave's instructions in order: the six reads above come before this increment */ \
        if (it_ != geo_it) {              /* the lane's patch of this tile: once per tile, not once per pass (eight passes share it) */ \
            geo_it = it_;                                                                                        \
            const int t = tb_e * NT + tl;                                                                        \
            const int tc = t < a.T ? t : 0;                                                                      \
            const int n_ = w4_div(tc, a.tpi_mul, a.tpi_sh), rem_ = tc - n_ * TPI;                                \
            const int ty_ = w4_div(rem_, a.tx_mul, a.tx_sh), tx_ = rem_ - ty_ * a.TX;                            \
            geo_off = (unsigned)(((n_ * a.y_ctotal + a.y_coff) * OH + M * ty_) * OW + M * tx_);                 \
            geo_ok  = t < a.T ? (min(M, OH - M * ty_) | (min(M, OW - M * tx_) << 4)) : 0;                        \
        }                                                                                                        \
        if (geo_ok != 0 && kg < a.K) {                                                                           \
            const int rows_ok = geo_ok & 15, cols_ok = geo_ok >> 4;                                              \
            (void)rows_ok; (void)cols_ok;                                                                        \
            const float bv = lh_e ? bs1 : bs0;                                                                   \
            float* __restrict__ yp = a.y + ((size_t)geo_off + (size_t)kg * (size_t)(OH * OW));                   \
            float yv[M][M];                                                                                      \
            _Pragma("unroll") for (int c2 = 0; c2 < M; c2 += 2) {                                                \
                w4_float2v em[6], col2[4];                                                                       \
                _Pragma("unroll") for (int i = 0; i < 6; ++i) em[i] = w4_float2v{ev[i][c2], ev[i][c2 + 1]};      \
                if (M == 4) wino4_at2(em[0], em[1], em[2], em[3], em[4], em[5], col2[0], col2[1], col2[2], col2[3]); \
                else        wino2_at2(em[0], em[1], em[2], em[3], em[4], em[5], col2[0], col2[1]);               \
                _Pragma("unroll") for (int r2 = 0; r2 < M; ++r2) {                                               \
                    const w4_float2v yb = col2[r2] + w4_splat(bv);                                               \
                    yv[r2][c2] = yb.x;                                                                           \
                    yv[r2][c2 + 1] = yb.y;                                                                       \
                }                                                                                                \
            }                                                                                                    \
            if (a.act != 0) {                                                                                    \
                _Pragma("unroll") for (int r2 = 0; r2 < M; ++r2)                                                 \
                    _Pragma("unroll") for (int c2 = 0; c2 < M; ++c2) yv[r2][c2] = (yv[r2][c2] < ab.lo) ? ab.lo : yv[r2][c2]; \
            }                                                                                                    \
            if (a.act == 2) {                                                                                    \
                _Pragma("unroll") for (int r2 = 0; r2 < M; ++r2)                                                 \
                    _Pragma("unroll") for (int c2 = 0; c2 < M; ++c2) yv[r2][c2] = (yv[r2][c2] > ab.hi) ? ab.hi : yv[r2][c2]; \
            }                                                                                                    \
            _Pragma("unroll") for (int r2 = 0; r2 < M; ++r2) {                                                   \
                float ov[M];                                                                                     \
                _Pragma("unroll") for (int c2 = 0; c2 < M; ++c2) ov[c2] = yv[r2][c2];                            \
                if (RAGGED) {                                                                                    \
                    if (r2 < rows_ok && M == 4 && cols_ok == M && a.s_wide != 0) {                               \
                        conv_store4(yp + (size_t)r2 * OW, ov[0], ov[1], ov[2], ov[M - 1]);                       \
                    } else if (r2 < rows_ok) {                                                                   \
                        _Pragma("unroll") for (int c2 = 0; c2 < M; c2 += 2) {                                    \
                            if (even_w && c2 + 1 < cols_ok) conv_store2(yp + (size_t)r2 * OW + c2, ov[c2], ov[c2 + 1]); \
                            else {                                                                               \
                                if (c2 < cols_ok) conv_store1(yp + (size_t)r2 * OW + c2, ov[c2]);                \
                                if (c2 + 1 < cols_ok) conv_store1(yp + (size_t)r2 * OW + c2 + 1, ov[c2 + 1]);    \
                            }                                                                                    \
                        }                                                                                        \
                    }                                                                                            \
                } else if (M == 4) conv_store4(yp + (size_t)r2 * OW, ov[0], ov[1], ov[2], ov[3]);                \
                else conv_store2(yp + (size_t)r2 * OW, ov[0], ov[1]);                                            \
            }                                                                                                    \
        }                                                                                                        \
        ++p_store;                                                                                               \
    }
        // a store pass if one is complete; false: nothing to do right now
        auto try_store = [&]() -> bool {
            if (p_store >= p_total) return false;
            if ((int)(w4s_poll(exfull_a + 4u * (p_store % (unsigned)kSEx)) - 6u * (p_store / (unsigned)kSEx + 1u)) < 0) return false;      // all six rows of this pass are in the buffer
            const unsigned long long s0_ = PVS_NOW();
            PVS_STORE_PASS();
            st[6] += PVS_NOW() - s0_;
            return true;
        };
        // wait for a counter; store passes fill the time
#define PVS_WAIT_OR_STORE(flag_, target_)                                                                        \
    while ((int)(w4s_poll(flag_) - (target_)) < 0) {                                                             \
        if (!try_store()) __builtin_amdgcn_s_sleep(1);                                                           \
    }
        int tile = L;
        PVS_ADDRESSES(tile, zlo, zhi, nv);
        PVS_GATHER(vA, eA, pair);
        PVS_GATHER(vB, eB, 2 + pair);
        for (;;) {
            for (int o = 0; o + 2 < n_own; o += 2) {
                const unsigned long long p0 = PVS_NOW();
                PVS_DATA_WAIT();                        /* (diagnostic build: the older gather has landed -- its wait gets an account of its own) */
                const unsigned long long d0 = PVS_NOW();
#ifdef PVHIP_DIAG
                { const unsigned seen_ = *(volatile unsigned*)&sm.done[pair]; PVS_TRACE(wid, 4u * use + (unsigned)pair, 0, t_entry + (unsigned long long)(seen_ * 1000u + (unsigned)CONS * use)); PVS_TRACE(wid, 4u * use + (unsigned)pair, 1, d0); }
#endif
                PVS_WAIT_OR_STORE(doneA, (unsigned)CONS * use);
                const unsigned long long p1 = PVS_NOW();
                PVS_TRANSFORM_STORE(vA, eA, VbA, zlo, zhi, nv);
                w4s_signal(readyA);
                const unsigned long long p2 = PVS_NOW();
                PVS_GATHER(vA, eA, 2 * (o + 2) + pair);
                const unsigned long long p3 = PVS_NOW();
                PVS_DATA_WAIT();
                const unsigned long long d1 = PVS_NOW();
                PVS_WAIT_OR_STORE(doneB, (unsigned)CONS * use);
                const unsigned long long p4 = PVS_NOW();
                PVS_TRANSFORM_STORE(vB, eB, VbB, zlo, zhi, nv);
                w4s_signal(readyB);
                const unsigned long long p5 = PVS_NOW();
                PVS_GATHER(vB, eB, 2 * (o + 3) + pair);
                st[0] += (p1 - d0) + (p4 - d1); st[1] += (p2 - p1) + (p5 - p4); st[2] += (p3 - p2) + (PVS_NOW() - p5);
                st[3] += (d0 - p0) + (d1 - p3);
                { const unsigned qa = 4u * use + (unsigned)pair, qb = qa + 2u; (void)qb; PVS_TRACE(wid, qa, 2, p1); PVS_TRACE(wid, qa, 3, p2); PVS_TRACE(wid, qb, 1, d1); PVS_TRACE(wid, qb, 2, p4); PVS_TRACE(wid, qb, 3, p5); }
                ++use;
            }
            // the last two own stages of the tile: the gathers behind them are own stages 0 / 1 of the NEXT tile
            PVS_ADDRESSES(tile + G, zlo_n, zhi_n, nv_n);
            PVS_WAIT_OR_STORE(doneA, (unsigned)CONS * use);
            PVS_TRANSFORM_STORE(vA, eA, VbA, zlo, zhi, nv);
            w4s_signal(readyA);
            PVS_GATHER(vA, eA, pair);
            PVS_WAIT_OR_STORE(doneB, (unsigned)CONS * use);
            PVS_TRANSFORM_STORE(vB, eB, VbB, zlo, zhi, nv);
            w4s_signal(readyB);
            PVS_GATHER(vB, eB, 2 + pair);
            ++use;
            zlo = zlo_n;
            zhi = zhi_n;
            nv  = nv_n;
            tile += G;
            if (tile >= n_tiles) break;
        }
        while (p_store < p_total) {                      // the last tile's passes
            if (!try_store()) __builtin_amdgcn_s_sleep(1);
        }
#undef PVS_ADDRESSES
#undef PVS_GATHER
#undef PVS_TRANSFORM_STORE
#undef PVS_STORE_PASS
#undef PVS_WAIT_OR_STORE
#undef VbA
#undef VbB
    } else {
        // ------------------------------------------------------------------ consumers
        const int cw  = wid - c_first;                              // 0 .. 11
        const int grp = (int)((unsigned)(5 - cw) >> 31);           // scalar arithmetic on purpose: a select would put it in a vector register (lesson 19)
        const int row = cw - 6 * grp;
        const unsigned u_lane = (unsigned)lane * 16u;
#define PVS_LOAD_U(ua_, so_, g_) w4_load(ua_[g_], ur, u_lane + (unsigned)(row * 3 + (g_)) * 1024u, (so_))
        floatx16 acc[6];
        const float* const vbs = &sm.V[0][0][0][0] + ((row * 6) * kCB + lh) * NT + l31;
        // wave-uniform offsets kept as SCALARS (readfirstlane) and turned into addresses where they are used: as pointers hipcc held them
        // in vector registers across the main loop, spilled them, and a spill reload is a vector memory load whose wait is vmcnt(0) (lesson 34)
        const int ex_off = __builtin_amdgcn_readfirstlane(grp * kSEx * kExBuf);
#define Exg  (&sm.Ex[0][0][0] + ex_off)
        const unsigned exfull_a = PVS_LDS_ADDR(&sm.exfull[0][0]) + 4u * kSEx * (unsigned)grp, exfree_a = PVS_LDS_ADDR(&sm.exfree[0][0]) + 4u * kSEx * (unsigned)grp;
        unsigned pass_n = 0u;                            // exchange passes of this group written so far
        const unsigned ready0 = PVS_LDS_ADDR(&sm.ready[0]), done0 = PVS_LDS_ADDR(&sm.done[0]);
        typedef float exv_t __attribute__((ext_vector_type(M)));
#define PVS_U_BASE(tile_) ((unsigned)(min(2 * PVS_PAIR(tile_) + grp, a.n_kb - 1) * (a.n_stages + 1)) * u_stage_bytes)
        w4_float4v ua[3];
        {
            const unsigned u_first = PVS_U_BASE(L);
            PVS_LOAD_U(ua, u_first, 0);
            PVS_LOAD_U(ua, u_first, 1);
            PVS_LOAD_U(ua, u_first, 2);
        }
        unsigned q = 0u;                                 // stage number across tiles
        // The second group starts kLag stages behind the first and stays there (the ring lets the first run at most kSRing stages
        // ahead): both groups read the same images in the same order, so left alone they reach their epilogues TOGETHER and the matrix
        // pipe idles through both (stamps: 1.2-1.5 k cycles of waiting per stage); staggered, a group's epilogue runs beside the
        // other group's last (first) stages of the tile.
        if (grp == 1 && a.s_lag != 0) w4s_wait_ge(done0 + 4u * (unsigned)(kSLag - 1), 6u);
        for (int tile = L; tile < n_tiles; tile += G) {
            const unsigned u_base = PVS_U_BASE(tile), u_next = PVS_U_BASE(tile + G);
            if (2 * PVS_PAIR(tile) + grp >= a.n_kb) {
                // the unpaired last block of an odd count: this group has no channels in the tile.  It releases every image at once and
                // hands the producers empty passes (they store nothing: kg >= K), so that the counters of both protocols keep their meaning
                for (int s = 0; s < n_eff; ++s, ++q) {
                    const unsigned b = q & (kSRing - 1);
                    w4s_wait_ge(ready0 + 4u * b, 2u * ((q >> 2) + 1u));
                    w4s_signal(done0 + 4u * b);
                }
                for (int p2 = 0; p2 < 8; ++p2) {
                    const unsigned P = pass_n + (unsigned)p2, pb = P % (unsigned)kSEx;
                    if (P >= (unsigned)kSEx) w4s_wait_ge(exfree_a + 4u * pb, 2u * (P / (unsigned)kSEx));
                    w4s_signal(exfull_a + 4u * pb);
                }
                pass_n += 8u;
                PVS_LOAD_U(ua, u_next, 0);           // the next tile finds its first weight image in the registers, as after a tile of its own
                PVS_LOAD_U(ua, u_next, 1);
                PVS_LOAD_U(ua, u_next, 2);
                continue;
            }
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
            for (int s = 0; s < n_eff; ++s, ++q) {
                const unsigned b = q & (kSRing - 1);
                const unsigned long long c0 = PVS_NOW();
                w4s_wait_ge(ready0 + 4u * b, 2u * ((q >> 2) + 1u));
                const unsigned long long c1 = PVS_NOW();
                const float* vb = vbs + b * v_buf_floats;
                const unsigned so_n = s + 1 < n_eff ? u_base + (unsigned)(s + 1) * u_stage_bytes : u_next;
                float bfr[2][4];
#define PVS_READ_B(dst_, g_)                                                                                     \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                              \
        const int m = 4 * (g_) + e, kk = m / 6, j = m % 6;                                                       \
        dst_[e] = vb[(j * kCB + 2 * kk) * NT];                                                                   \
    }
                PVS_READ_B(bfr[0], 0);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    if (g < 2) PVS_READ_B(bfr[(g + 1) & 1], g + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int m = 4 * g + e, j = m % 6;
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[g][e], bfr[g & 1][e], acc[j], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    PVS_LOAD_U(ua, so_n, g);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef PVS_READ_B
                // every B operand of this stage is in registers (the MFMAs above used them): the buffer is released.  The count that comes
                // back says how many of the twelve consumers finished the stage before this wave: the late ones go in front.  The
                // SIMD's arbiter prefers OLDER waves, so the youngest consumer of a SIMD only got the matrix pipe when the others stalled,
                // fell a ring behind, and everybody then waited for the buffers it still held (stamps: 6000 cycles between the first and
                // the last consumer of a stage).
                if (a.s_prio != 0) {
                    const unsigned rank = w4s_signal_rank(done0 + 4u * b) - (unsigned)CONS * (q >> 2);
                    if (rank >= 8u) __builtin_amdgcn_s_setprio(2);
                    else if (rank >= 4u) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                } else {
                    w4s_signal(done0 + 4u * b);
                }
                { const unsigned long long c2 = PVS_NOW(); st[0] += c1 - c0; st[1] += c2 - c1; st[5] += 1; PVS_TRACE(wid, q, 0, c1); PVS_TRACE(wid, q, 1, c2); }
            }
            st[6] += 1;
            // ---- this group's half of the epilogue: the column half of Y = A^T D A on two accumulator registers at a time (packed fp32),
            // left in the group's exchange buffer P & 1 for the producers (PVS_STORE_PASS); the consumers go straight on to the next tile
            const unsigned long long e0_ = PVS_NOW();
            if (a.s_prio != 0) __builtin_amdgcn_s_setprio(3);       // ~300 vector instructions between two tiles of MFMAs: in front of the other group's streams
            {
                int lane_e;
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
#else
                lane_e = 0;
#endif
                const int tl = lane_e & 31, lh_e = lane_e >> 5;
                float* const mine = Exg + ((row * 4 + lh_e) * 32 + tl) * M;
#pragma unroll
                for (int p2 = 0; p2 < 8; ++p2) {
                    const unsigned P = pass_n + (unsigned)p2, pb = P % (unsigned)kSEx;
                    if (P >= (unsigned)kSEx) {            // the buffer's previous pass (P - kSEx) has been taken out by both producer waves: P / kSEx passes x 2 waves
                        const unsigned long long w0_ = PVS_NOW();
                        w4s_wait_ge(exfree_a + 4u * pb, 2u * (P / (unsigned)kSEx));
                        st[3] += PVS_NOW() - w0_;
                    }
                    const int r = 2 * p2;
                    w4_float2v mm[6], so2[4];
#pragma unroll
                    for (int j = 0; j < 6; ++j) mm[j] = w4_float2v{acc[j][r], acc[j][r + 1]};
                    if (M == 4) wino4_at2(mm[0], mm[1], mm[2], mm[3], mm[4], mm[5], so2[0], so2[1], so2[2], so2[3]);
                    else        wino2_at2(mm[0], mm[1], mm[2], mm[3], mm[4], mm[5], so2[0], so2[1]);
                    exv_t sv0, sv1;
#pragma unroll
                    for (int c2 = 0; c2 < M; ++c2) { sv0[c2] = so2[c2].x; sv1[c2] = so2[c2].y; }
                    float* const dst = mine + (int)pb * kExBuf;
                    *reinterpret_cast<exv_t*>(dst) = sv0;                       // slot lh (register r)
                    *reinterpret_cast<exv_t*>(dst + 2 * 32 * M) = sv1;          // slot 2 + lh (register r + 1)
                    w4s_signal(exfull_a + 4u * pb);
                }
                pass_n += 8u;
            }
            if (a.s_prio != 0) __builtin_amdgcn_s_setprio(1);
            st[2] += PVS_NOW() - e0_;
        }
#undef PVS_LOAD_U
#undef PVS_U_BASE
#undef PVS_PAIR
#undef PVS_BLOCK
#undef Exg
    }
#ifdef PVHIP_DIAG
    if ((blockIdx.x & 15) == 3 && lane == 0) {
        st[4] = PVS_NOW() - t_entry;
#pragma unroll
        for (int i = 0; i < 7; ++i) atomicAdd(&g_w4s_stamps[wid][i], st[i]);
        atomicAdd(&g_w4s_stamps[wid][7], 1ull);
        atomicAdd(&g_w4s_simd[wid][(__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11))) & 3u], 1ull);
    }
#endif
}

}  // namespace

namespace pvhip {

bool wino_eligible(int c, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int h, int w, int oh, int ow) {
    return settings().conv_winograd &&      // PVHIP_CONV_WINOGRAD=0: the direct kernels (A/B measurements, their tests)
           kh == 3 && kw == 3 && sh == 1 && sw == 1 && pad_top == 1 && pad_left == 1 && oh == h && ow == w &&
           c % kCB == 0 && c >= kCB;
}

// Output channels per workgroup (and per block of the transformed panel): 64 when K is made of whole 64-channel blocks
// (each input patch is then gathered and transformed once per 64 output channels), else 32.
static int wino_kb(int k) {
    if (settings().wino_kb) return settings().wino_kb;       // PVHIP_WINO_KB: tuning runs only
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
    a.balance = 0; a.n_tiles = 0; a.s_lag = 0; a.s_prio = 0; a.s_old = 0; a.s_order = 0; a.s_wide = 0; a.p_prio = 0;
    a.tpi_mul = a.tpi_sh = a.tx_mul = a.tx_sh = 0u;
#ifdef PVHIP_DIAG
    if (settings().wino4_ablate == 5) a.balance = 5;          // diagnostic build: s_memtime stamps (scripts/stamps_wino.py)
#endif
    // 32-channel blocks run as 32 channels x 32 patches on four waves (more, smaller workgroups: the layers whose K is not
    // made of 64-channel blocks are the small 14x14 ones); PVHIP_WINO_SMALL=0 selects 32 x 64 on eight waves (tuning runs)
    const bool  small = kb == 32 && settings().wino_small;
    const int  nt   = (kb == 64 || small) ? 32 : 64;
    const long n_tb = ((long)a.T + nt - 1) / nt;
    if (n_tb * a.n_kb > 0x7fffffffL) return fail(PVHIP_EUNSUPPORTED, "wino_conv: grid too large");
    const int   waves = settings().wino_waves;         // PVHIP_WINO_WAVES: tuning runs only
    const dim3  grid((unsigned)(n_tb * a.n_kb));
    if (small) hipLaunchKernelGGL((conv_wino_kernel<1, 1, 4>), grid, dim3(256), 0, state().stream, a);
    else if (kb == 64 && waves == 8) hipLaunchKernelGGL((conv_wino_kernel<2, 1, 8>), grid, dim3(512), 0, state().stream, a);
    else if (kb == 64) hipLaunchKernelGGL((conv_wino_kernel<2, 1, 4>), grid, dim3(256), 0, state().stream, a);
    else if (waves == 8) hipLaunchKernelGGL((conv_wino_kernel<1, 2, 8>), grid, dim3(512), 0, state().stream, a);
    else hipLaunchKernelGGL((conv_wino_kernel<1, 2, 4>), grid, dim3(256), 0, state().stream, a);
    return PVHIP_OK;
}

// ---- F(4x4, 3x3)
bool wino4_eligible(int c, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int h, int w, int oh, int ow, int n) {
    const int mode = settings().conv_winograd4;                // PVHIP_CONV_WINOGRAD4: 0 = F(2x2, 3x3) everywhere, 2 = "force"
    if (mode == 0 || !wino_eligible(c, kh, kw, sh, sw, pad_top, pad_left, h, w, oh, ow)) return false;
    const bool ragged = h % 4 != 0 || w % 4 != 0;                 // 14x14, 7x7: the last patches hang over the edge
    if (ragged && !settings().wino_ragged) return false;
    // 512 output pixels per workgroup: worth it where the patch blocks alone give every CU a workgroup; the ragged layers have
    // fewer, larger-than-needed patches and win on the multiply count from 1024 patches on (7x7 at batch 256)
    const long patches = (long)n * ((h + 3) / 4) * ((w + 3) / 4);
    const long min_patches = mode == 2 ? 1 : (ragged ? 1024L : 32L * kNumCU);           // "force": any size (tests)
    return patches >= min_patches;
}

// F(2x2, 5x5): 5x5 / stride 1 / pad 2 ("same"), even extents, C a multiple of 4, enough patches (PVHIP_CONV_WINOGRAD5=0 switches it
// off, =force drops the size rule)
bool wino25_eligible(int c, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int h, int w, int oh, int ow, int n) {
    const int mode = settings().conv_winograd5;
    if (mode == 0 || !settings().conv_winograd) return false;
    if (!(kh == 5 && kw == 5 && sh == 1 && sw == 1 && pad_top == 2 && pad_left == 2 && oh == h && ow == w && c % kCB == 0 && c >= kCB))
        return false;
    const bool ragged = h % 2 != 0 || w % 2 != 0;
    if (ragged && !settings().wino_ragged) return false;
    const long patches = (long)n * ((h + 1) / 2) * ((w + 1) / 2);
    return patches >= (mode == 2 ? 1 : (ragged ? 4096L : 32L * kNumCU));
}

int wino25_pack(const float* w_oihw, float* u, int k, int c) {
    const size_t elems = wino4_pack_elems(k, c);
    hipError_t   e     = hipMemsetAsync(u, 0, elems * sizeof(float), state().stream);
    if (e != hipSuccess) return fail(PVHIP_EHIP, "wino25_pack: hipMemsetAsync -> %s", hipGetErrorString(e));
    hipLaunchKernelGGL(wino25_pack_kernel, dim3(grid_for((size_t)((k + 31) / 32) * 32 * c)), dim3(kBlock), 0, state().stream, w_oihw,
                       u, k, c, c / kCB);
    return PVHIP_OK;
}

size_t wino4_pack_elems(int k, int c) {
    if (c % kCB != 0) return 0;
    return (size_t)((k + 31) / 32) * (size_t)(c / kCB + 1) * kXi4 * kCB * 32;
}

int wino4_pack(const float* w_oihw, float* u, int k, int c) {
    const size_t elems = wino4_pack_elems(k, c);
    hipError_t   e     = hipMemsetAsync(u, 0, elems * sizeof(float), state().stream);
    if (e != hipSuccess) return fail(PVHIP_EHIP, "wino4_pack: hipMemsetAsync -> %s", hipGetErrorString(e));
    hipLaunchKernelGGL(wino4_pack_kernel, dim3(grid_for((size_t)((k + 31) / 32) * 32 * c)), dim3(kBlock), 0, state().stream, w_oihw,
                       u, k, c, c / kCB);
    return PVHIP_OK;
}

int wino4_conv(int m, const float* x, const float* u, float* y, int n, int c, int h, int w, int k_out, const float* bias, int act,
               float act_lo, float act_hi, int out_channel_offset, int out_channels_total) {
    WinoArgs a;
    a.x = x; a.u = u; a.y = y; a.bias = bias;
    a.N = n; a.C = c; a.H = h; a.W = w; a.K = k_out;
    a.TY = (h + m - 1) / m; a.TX = (w + m - 1) / m;
    const bool ragged = h % m != 0 || w % m != 0;
    a.T  = n * a.TY * a.TX;
    a.n_kb = (k_out + 31) / 32;
    a.balance = settings().wino_balance ? 1 : 0;
    a.s_lag = settings().wino_shared_lag;
    a.s_prio = settings().wino_shared_prio;
    a.s_old = settings().wino_shared_old;
    a.p_prio = settings().tune[7] == 1 ? 0 : 1;        // PVHIP_TUNE7=1: no priority for conv_wino4_kernel's producers (A/B runs)
    a.s_wide = settings().tune[5] == 1 ? 0 : 1;       // (PVHIP_TUNE5=1: the old pieces.  Same box, alternating: the 7x7 layers -3 % -- their odd rows stored single floats --, the 14x14 layers +-0.5 %)
    // Tile order of the shared-V form.  MEASURED (scripts/traffic_wino_order.sh, FETCH_SIZE per launch; scripts/time_wino_order.py): channel-pair-major reads
    // 58.5 / 79.8 MB instead of 68.9 / 96.8 on the 7x7 layers 5a / 5b (their transformed weights, 7.4 / 10.6 MB, are the LARGER operand and do not fit an
    // XCD's 4 MB L2; -1.2 % time) and 90-191 MB instead of 70-186 on the 14x14 layers (there the input is the larger one and is then re-read per pair; time
    // within +-1 %).  So: pair-major where twice the weights outweigh the input (5a, 5b).  PVHIP_TUNE3=1: always, =2: never (A/B runs).
    a.s_order = settings().tune[3] == 1 ? 1 : (settings().tune[3] == 2 ? 0 : (2 * wino4_pack_elems(k_out, c) > (size_t)n * c * h * w ? 1 : 0));
    a.n_stages = c / kCB;
    a.x_bytes = (unsigned)((size_t)n * c * h * w * 4);
    a.u_bytes = (unsigned)(wino4_pack_elems(k_out, c) * 4);
    a.act = act; a.act_lo = act_lo; a.act_hi = act_hi;
    a.y_ctotal = out_channels_total; a.y_coff = out_channel_offset;
    const long n_tb = ((long)a.T + 31) / 32;
    if (n_tb * a.n_kb > 0x3fffffffL) return fail(PVHIP_EUNSUPPORTED, "wino4_conv: too many tiles");
    a.n_tiles = (int)(n_tb * a.n_kb);
    w4_magic((unsigned)(a.TY * a.TX), a.tpi_mul, a.tpi_sh);
    w4_magic((unsigned)a.TX, a.tx_mul, a.tx_sh);
    // The shared-V form (conv_wino4s_kernel: one 16-wave workgroup per CU, two channel blocks on one transformed image per stage,
    // counters instead of barriers): an even number of channel blocks, whole groups of four stages, and tiles for every CU.
    // PVHIP_WINO_SHARED=0 never, =2 wherever it applies.
    {
        const int  mode = settings().wino_shared;
        const long tiles_s = n_tb * ((a.n_kb + 1) / 2);
        const bool shape_ok = a.n_kb >= 2 && a.n_stages % 4 == 0 && a.n_stages >= 4 && (a.n_kb % 2 == 0 || mode == 2 || settings().wino_shared_odd);
        // Measured on GoogLeNet's layers at batch 256 (scripts/time_wino_shared.py, same box, alternating): it wins where the main loop is
        // long against the epilogue passes of a tile -- C >= 112: 3b -9..-11 %, 4b -9 %, 4c -7 %, 4d -9 %, 4e -9 %, 5a -9 %, 5b -11..-13 %; the
        // 14x14 layer with C = 96 (4a: -8 %) -- and on conv2/3x3 (C = 64, 4704 tiles: -4..-12 %); it loses on the 28x28 layer with C = 96
        // (3a: +2 %) and is a wash on the 5x5 layers (4 .. 12 stages per tile).
        const bool pays = a.n_stages >= 28 || (a.n_stages >= 24 && ragged) ||
                          (a.n_stages <= 16 && a.n_stages >= 12 && tiles_s >= (long)settings().wino_shared_min_tiles);
        if (mode != 0 && shape_ok && (mode == 2 || pays)) {
            a.n_tiles = (int)tiles_s;
            const dim3 grid_s((unsigned)(tiles_s < kNumCU ? tiles_s : kNumCU));
            if (m == 2) {
                if (ragged) hipLaunchKernelGGL((conv_wino4s_kernel<2, true>), grid_s, dim3(1024), 0, state().stream, a);
                else        hipLaunchKernelGGL((conv_wino4s_kernel<2, false>), grid_s, dim3(1024), 0, state().stream, a);
            } else {
                if (ragged) hipLaunchKernelGGL((conv_wino4s_kernel<4, true>), grid_s, dim3(1024), 0, state().stream, a);
                else        hipLaunchKernelGGL((conv_wino4s_kernel<4, false>), grid_s, dim3(1024), 0, state().stream, a);
            }
            PVHIP_LAUNCH_CHECK();
            return PVHIP_OK;
        }
    }
    // persistent: two workgroups per CU (72 KB of LDS each), each walking tiles L, L + G, ...
    const dim3 grid((unsigned)(a.n_tiles < 2 * kNumCU ? a.n_tiles : 2 * kNumCU));
    if (m == 2) {
        if (ragged) hipLaunchKernelGGL((conv_wino4_kernel<2, 0, true>), grid, dim3(512), 0, state().stream, a);
        else        hipLaunchKernelGGL((conv_wino4_kernel<2, 0>), grid, dim3(512), 0, state().stream, a);
        PVHIP_LAUNCH_CHECK();
        return PVHIP_OK;
    }
    if (ragged) {
        hipLaunchKernelGGL((conv_wino4_kernel<4, 0, true>), grid, dim3(512), 0, state().stream, a);
        PVHIP_LAUNCH_CHECK();
        return PVHIP_OK;
    }
#ifdef PVHIP_DIAG
    switch (settings().wino4_ablate) {      // diagnostic build only (libpvhip_diag.so): results are wrong on purpose
        case 1: hipLaunchKernelGGL((conv_wino4_kernel<4, 1>), grid, dim3(512), 0, state().stream, a); return PVHIP_OK;
        case 2: hipLaunchKernelGGL((conv_wino4_kernel<4, 2>), grid, dim3(512), 0, state().stream, a); return PVHIP_OK;
        case 3: hipLaunchKernelGGL((conv_wino4_kernel<4, 3>), grid, dim3(512), 0, state().stream, a); return PVHIP_OK;
        case 4: hipLaunchKernelGGL((conv_wino4_kernel<4, 4>), grid, dim3(512), 0, state().stream, a); return PVHIP_OK;
        case 5: hipLaunchKernelGGL((conv_wino4_kernel<4, 5>), grid, dim3(512), 0, state().stream, a); return PVHIP_OK;
        case 6: hipLaunchKernelGGL((conv_wino4_kernel<4, 6>), grid, dim3(512), 0, state().stream, a); return PVHIP_OK;
        case 7: hipLaunchKernelGGL((conv_wino4_kernel<4, 7>), grid, dim3(512), 0, state().stream, a); return PVHIP_OK;
        case 8: hipLaunchKernelGGL((conv_wino4_kernel<4, 8>), grid, dim3(512), 0, state().stream, a); return PVHIP_OK;
        default: break;
    }
#endif
    hipLaunchKernelGGL((conv_wino4_kernel<4, 0>), grid, dim3(512), 0, state().stream, a);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // namespace pvhip

#ifdef PVHIP_DIAG
// diagnostic build only: read and clear the cycle accounts of conv_wino4_kernel<4, 5> (PVHIP_WINO4_ABLATE=5); out = 64 counters
extern "C" int pvhip_diag_wino4_hw(unsigned* out) {          // 64 x 8 x 2 words, and the ticket counter is reset
    unsigned zero = 0;
    if (hipDeviceSynchronize() != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w4_hw), 64 * 8 * 2 * sizeof(unsigned)) != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_w4_hw_ticket), &zero, sizeof(zero)) != hipSuccess) return PVHIP_EHIP;
    return PVHIP_OK;
}
extern "C" int pvhip_diag_wino4_epilogue(unsigned long long* out) {      // 8 waves x 8 phases, read and cleared
    unsigned long long zero[64] = {0};
    if (hipDeviceSynchronize() != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w4_epi), sizeof(zero)) != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_w4_epi), zero, sizeof(zero)) != hipSuccess) return PVHIP_EHIP;
    return PVHIP_OK;
}
extern "C" int pvhip_diag_wino4s_stamps(unsigned long long* out) {     // 16 waves x 8 accounts of conv_wino4s_kernel, read and cleared
    unsigned long long zero[128] = {0};
    if (hipDeviceSynchronize() != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w4s_stamps), sizeof(zero)) != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_w4s_stamps), zero, sizeof(zero)) != hipSuccess) return PVHIP_EHIP;
    return PVHIP_OK;
}
extern "C" int pvhip_diag_wino4s_trace(unsigned* out) {       // 16 waves x 96 stages x 2 stamps of workgroup 3 (the last launch)
    if (hipDeviceSynchronize() != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w4s_trace), 16 * 96 * 4 * sizeof(unsigned)) != hipSuccess) return PVHIP_EHIP;
    return PVHIP_OK;
}
extern "C" int pvhip_diag_wino4s_simd(unsigned long long* out) {       // 16 waves x 4 SIMDs: where the waves of the stamped workgroups ran; read and cleared
    unsigned long long zero[64] = {0};
    if (hipDeviceSynchronize() != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w4s_simd), sizeof(zero)) != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_w4s_simd), zero, sizeof(zero)) != hipSuccess) return PVHIP_EHIP;
    return PVHIP_OK;
}
extern "C" int pvhip_diag_wino4_stamps(unsigned long long* out) {
    unsigned long long zero[64] = {0};
    if (hipDeviceSynchronize() != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w4_stamps), sizeof(zero)) != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_w4_stamps), zero, sizeof(zero)) != hipSuccess) return PVHIP_EHIP;
    return PVHIP_OK;
}
#endif
