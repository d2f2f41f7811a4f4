// 3x3 / stride 1 / pad 1 MaxPool followed by a 1x1 convolution (the pool -> pool_proj arm of an inception module) as ONE
// launch: the pooled tensor -- as large as the module's input, written once and read once -- never exists.
//
// Replaces MaxPool.py:41-72 followed by Convolution.py:57-87 for that pair.  The 1x1 convolution is the implicit GEMM of
// pvhip_conv.hip (D[k_out][pixel], v_mfma_f32_32x32x2_f32, weights from the packed K-major panel by LDS-DMA, same
// reduction order: the result carries the bits of the two launches); what changes is where the B tile [16 channels][128
// pixels] of a stage comes from.  Waves 0-3 are CONSUMERS (wave w: pixels 32w .. 32w+31 of the tile, all BM output
// channels).  Waves 4-7 are PRODUCERS: lane <-> (group of VEC adjacent pixels, channel); per channel it loads the three
// rows of its group (one aligned vector each) and the two neighbouring columns, takes the 3x3 max and writes VEC pooled
// values into the B tile.  Zero padding is what the reference pools over (MaxPool.py:53: the pad cells hold 0.0 and take
// part in the max): a row or column outside the image is an out-of-range buffer offset, for which the hardware returns
// 0.0 -- no window logic at all.  NaN wins, as np.max.  The loads of the next stage's groups are issued as soon as a
// group's registers are free, so a whole stage of them is always in flight.
#include <cstdlib>
#include <type_traits>

#include "pvhip_common.h"

using namespace pvhip;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kBK = 16;
constexpr int kRingDepth1 = 2;          // ... for single pixels (odd rows: the 7x7 modules, only with PVHIP_FUSE_POOLCONV=1: 0.080 ms against 0.067 for the two launches; 0.089 with one stage)
constexpr int kRingDepth4 = 1;          // ... for groups of four pixels (the 28x28 modules: two stages need 73 registers -- fewer waves per SIMD -- and measured 6-12 % SLOWER)
constexpr int kRingDepth2 = 2;          // stages of producer loads in flight for groups of two pixels (the 14x14 modules).  Three (168 registers, two workgroups per CU): 0.064 ms on 4a against 0.055 with two -- the waves a CU holds matter as much as the loads in flight

struct PoolConvArgs {
    const float* x;
    const float* wp;      // [kred_pad + spare][kout_pad], rows = input channels (the pointwise panel of pvhip_conv2d_pack_f32)
    float*       y;
    const float* bias;
    int N, C, H, W, K;
    int kout_pad, n_mtiles;
    int P;                // N*H*W
    unsigned x_bytes, wp_bytes;
    int   relu;
    float act_lo, act_hi;
    int y_ctotal, y_coff;
    int prio;             // producers at wave priority 3 (PVHIP_POOLCONV_PRIO=0: everything at 0)
    int abl;              // diagnostic build (PVHIP_CONV_ABLATE bits, wrong results): 1 no activation loads, 2 no pooling arithmetic, 4 no MFMAs, 8 no weight copies, 16 no stores, 32 no loads of the outer columns, 64 s_memtime stamps (pvhip_diag_poolconv_stamps)
};

__device__ __forceinline__ void pc_dma_b128(__amdgpu_buffer_rsrc_t r, float* dst, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lds = (unsigned)(unsigned long)(lds_ptr_t)dst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
#endif
}

typedef float pc_f4 __attribute__((ext_vector_type(4)));
typedef float pc_f2 __attribute__((ext_vector_type(2)));
// a buffer load as wide as its destination (the whole vector is cast: see pvhip_wino.hip)
__device__ __forceinline__ void pc_load(pc_f4& d, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    d = __builtin_bit_cast(pc_f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void pc_load(pc_f2& d, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    d = __builtin_bit_cast(pc_f2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
typedef float pc_f1 __attribute__((ext_vector_type(1)));
__device__ __forceinline__ void pc_load(pc_f1& d, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    d[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// kF16 (FP16 IRs): the same tiles, a stage of 16 channels as ONE v_mfma_f32_32x32x16_f16 per 32-channel tile, both operands rounded to
// fp16 as they are read from LDS (the maximum of the window is taken in fp32, then rounded: what MaxPool followed by pvhip_conv2d_f16_dma does)
// NW (round 5): consumer waves = producer waves = pixels of the tile / 32; NW = 2 (tiles of 64 pixels, twice the workgroups, the same work per wave)
// is an experiment that did not pay -- see conv2d_pooled_impl.
#ifdef PVHIP_DIAG
// diagnostic build, PVHIP_CONV_ABLATE bit 64 (scripts/stamps_poolconv.py): s_memtime accounts of workgroup 77's first consumer and first producer wave, summed over the
// stages: [0] stages, [1] producer pooling (incl. waiting for its loads), [2] producer barrier, [3] producer loop, [4] consumer MFMA section, [5] its wait for the
// weight copy, [6] consumer barrier, [7] consumer loop
__device__ unsigned long long g_pc_stamps[8];
#define PC_NOW() ((abl & 64) ? (unsigned long long)__builtin_readcyclecounter() : 0ull)
#else
#define PC_NOW() 0ull
#endif
template <int BM, int VEC, bool kF16 = false, int NW = 4>
__global__ __launch_bounds__(NW * 128, (BM <= 64 && VEC == 1) || (BM <= 64 && VEC == 4 && kRingDepth4 == 1) ? 8 : (VEC == 2 && kRingDepth2 > 2 ? 2 : 4)) void conv_pool1x1_kernel(PoolConvArgs a) {      // 64 channels: 73 -> 64 registers, four workgroups per CU (3b: -3 %)
    constexpr int BN = 32 * NW, TM = BM / 32, KK = kBK / 2;
    constexpr int CONSUMERS = NW, PRODUCERS = NW;
    constexpr int A_PIECES = kBK * BM * 4 / 1024, A_PER_WAVE = (A_PIECES + CONSUMERS - 1) / CONSUMERS;
    constexpr int GROUPS = BN / VEC;                 // pixel groups per channel row of the tile
    constexpr int CSUB = PRODUCERS * kWave / GROUPS;  // channels the producer lanes cover at once
    constexpr int ITER = kBK / CSUB;                 // producer iterations per stage
    constexpr unsigned kOob = 0x80000000u;
    constexpr bool kWide = VEC == 2;                 // a group of two pixels is loaded as four (the columns around it ride along)
    typedef float vec_t __attribute__((ext_vector_type(VEC)));

    __shared__ __attribute__((aligned(1024))) float As[2][kBK][BM];
    __shared__ __attribute__((aligned(1024))) float Bs[2][kBK][BN];

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

#ifdef PVHIP_DIAG
    const int abl = a.abl;          // scripts/time_poolconv_abl.py: parts switched off
#else
    constexpr int abl = 0;
#endif
    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid  = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int l31 = lane & 31, lh = lane >> 5;
    const int HW = a.H * a.W;
    const int nk = a.C / kBK;
    const unsigned chan_bytes = (unsigned)HW * 4u;

    floatx16 acc[TM];
    if (wid >= CONSUMERS) {
        // ------------------------------------------------------------------ producers
        // (PVHIP_POOLCONV_PRIO: wave priorities were tried here after conv_wino4s_kernel's producers needed them -- producers first
        // costs 11-14 % on the 14x14 modules, consumers first changes nothing: several independent workgroups per CU already interleave)
        if (a.prio == 1) __builtin_amdgcn_s_setprio(3);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
        const int pl = (wid - CONSUMERS) * kWave + lane;       // 0 .. 255
        const int g = pl % GROUPS, cc = pl / GROUPS;
        // byte offsets of this lane's group in rows y-1, y, y+1 (channel cc of the stage; image and pixel folded in); whatever lies
        // outside the image is out of range and reads as the pad value 0.0.  The columns left and right of the group are the
        // neighbouring LANES' outer columns: their column maxima come over by DPP (wave_shr / wave_shl) when the pooling runs, so a
        // pooled value costs 3 / VEC vector loads, not 9 / VEC loads (the first version; at VEC = 2 -- the 14x14 modules -- that made
        // the fused launch slower than the two).  Only the first and the last group of the tile's row of groups load their outer
        // column themselves (one more dword load per row, live in two lanes of GROUPS).
        unsigned offv[3], offe[3];
        bool     zl, zr;                         // the group touches the left / right border: its outer column is padding
        // (VEC = 1, round 4: 128 single-pixel groups span two waves, and a DPP shift ends at the wave: its first and last lane load their outer column too)
        const bool first = g == 0 || (VEC == 1 && lane == 0), last = g == GROUPS - 1 || (VEC == 1 && lane == kWave - 1);
        {
            const int gp = ptile * BN + g * VEC;
            const bool live = gp < a.P;
            const int n = live ? gp / HW : 0, rem = live ? gp - n * HW : 0;
            const int y = rem / a.W, x0 = rem - y * a.W;
            const unsigned base = (unsigned)((n * a.C + cc) * HW + x0) * 4u;
            zl = x0 == 0;
            zr = x0 + VEC >= a.W;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int  iy = y - 1 + r;
                const bool ok = live && (unsigned)iy < (unsigned)a.H;
                offv[r] = ok ? base + (unsigned)(iy * a.W) * 4u : kOob;
                offe[r] = (ok && first && !zl) ? offv[r] - 4u : ((ok && last && !zr) ? offv[r] + (unsigned)VEC * 4u : kOob);
                // VEC = 2 (round 5): ONE 16-byte load per row fetches the group AND both outer columns (from the column left of the group, or from the
                // group itself where that column is padding: no offset below the tensor) -- the launch is bound by the NUMBER of vector-memory
                // instructions its producers issue (ablation: the mostly out-of-range outer-column loads cost as much as the group loads)
                if (kWide) offv[r] = ok ? offv[r] - (zl ? 0u : 4u) : kOob;
            }
        }
        typedef typename std::conditional<kWide, pc_f4, vec_t>::type ring_t;
        // TWO stages of loads in flight (round 5): with one, a stage lasted as long as its loads took to arrive -- s_memtime stamps on 4a: 3.0 k cycles
        // per stage, the consumers 1.3 k of them in their MFMAs and 1.6 k at the barrier waiting for the producers, who were waiting for memory
        // (groups of two pixels only -- the 14x14 modules, few tiles per CU: 128 registers there; the 28x28 modules keep one stage and 64 registers: eight waves per SIMD)
        constexpr int RD = kWide ? kRingDepth2 : (VEC == 4 ? kRingDepth4 : kRingDepth1);
        ring_t ring[RD][ITER][3];            // per slot (stage % RD), iteration and row: the group (kWide: with the columns around it)
        float edge[RD][ITER][3];             //                                          first / last lane: its outer column
#define PVP_LOAD(it_, s_, sl_)                                                                                   \
    {                                                                                                            \
        const int se_ = (s_) < nk ? (s_) : nk - 1;                 /* past the end: the last stage again (unused) */ \
        const unsigned soff = (unsigned)(se_ * kBK + (it_) * CSUB) * chan_bytes;                                 \
        _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                          \
            if (abl & 1) continue;                                                                               \
            pc_load(ring[sl_][it_][r], xr, offv[r], soff);                                                       \
            if (kWide) continue;                                                                                 \
            if (abl & 32) { edge[sl_][it_][r] = 0.0f; continue; }                                                \
            edge[sl_][it_][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, offe[r], soff, 0));  \
        }                                                                                                        \
    }
        // 3x3 max of iteration it_ into the B tile of buffer buf_ (columns first: one NaN-propagating max3 per column, the
        // outer columns' from the neighbour lanes, then a max3 over three adjacent columns per pixel), then the same registers
        // fetch stage s_next_
#define PVP_POOL(it_, buf_, s_next_, sl_)                                                                             \
    {                                                                                                            \
        if (abl & 2) {                                                                                           \
            vec_t q_;                                                                                            \
            _Pragma("unroll") for (int i = 0; i < VEC; ++i) q_[i] = ring[sl_][it_][1][i];                             \
            *reinterpret_cast<vec_t*>(&Bs[buf_][(it_) * CSUB + cc][g * VEC]) = q_;                               \
            PVP_LOAD(it_, s_next_, sl_);                                                                              \
            continue;                                                                                            \
        }                                                                                                        \
        if (kWide) {                                                                                             \
            float q_[4];                 /* column maxima of the four loaded columns */                          \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) q_[c] = max3_nan(ring[sl_][it_][0][c], ring[sl_][it_][1][c], ring[sl_][it_][2][c]); \
            const float own0_ = zl ? q_[0] : q_[1], own1_ = zl ? q_[1] : q_[2];                                  \
            const float lft_ = zl ? 0.0f : q_[0], rgt_ = zr ? 0.0f : (zl ? q_[2] : q_[3]);                       \
            vec_t o2_;                                                                                           \
            o2_[0] = max3_nan(lft_, own0_, own1_);                                                               \
            o2_[VEC - 1] = max3_nan(own0_, own1_, rgt_);                                                         \
            *reinterpret_cast<vec_t*>(&Bs[buf_][(it_) * CSUB + cc][g * VEC]) = o2_;                              \
            PVP_LOAD(it_, s_next_, sl_);                                                                              \
            continue;                                                                                            \
        }                                                                                                        \
        float cm_[VEC + 2];                  /* column maxima; a NaN in the column IS the maximum (max3_nan): no bookkeeping beside it */ \
        _Pragma("unroll") for (int c = 0; c < VEC; ++c) cm_[c + 1] = max3_nan(ring[sl_][it_][0][c], ring[sl_][it_][1][c], ring[sl_][it_][2][c]); \
        const float em_ = max3_nan(edge[sl_][it_][0], edge[sl_][it_][1], edge[sl_][it_][2]);                                    \
        const float lm_ = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, cm_[VEC]), 0x138, 0xf, 0xf, true)); \
        const float rm_ = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, cm_[1]), 0x130, 0xf, 0xf, true)); \
        cm_[0]       = zl ? 0.0f : (first ? em_ : lm_);                                                          \
        cm_[VEC + 1] = zr ? 0.0f : (last ? em_ : rm_);                                                           \
        vec_t o_;                                                                                                \
        _Pragma("unroll") for (int i = 0; i < VEC; ++i) o_[i] = max3_nan(cm_[i], cm_[i + 1], cm_[i + 2]);        \
        *reinterpret_cast<vec_t*>(&Bs[buf_][(it_) * CSUB + cc][g * VEC]) = o_;                                   \
        PVP_LOAD(it_, s_next_, sl_);                                                                                  \
    }
#pragma unroll
        for (int q = 0; q < RD; ++q)
#pragma unroll
            for (int it = 0; it < ITER; ++it) PVP_LOAD(it, q, q);
#pragma unroll
        for (int it = 0; it < ITER; ++it) PVP_POOL(it, 0, RD, 0);              // B(0) from slot 0, which then fetches stage RD
        __syncthreads();
        unsigned long long tp_pool = 0ull, tp_bar = 0ull;
        const unsigned long long tp_begin = PC_NOW();
        for (int s = 0; s < nk; s += RD) {                                     // stage q sits in slot q % RD and goes into B tile q & 1
#pragma unroll
            for (int d = 0; d < RD; ++d) {
                if (s + d < nk) {
                    const int bn = (s + d + 1) & 1;
                    const unsigned long long t0_ = PC_NOW();
#pragma unroll
                    for (int it = 0; it < ITER; ++it) PVP_POOL(it, bn, s + d + 1 + RD, (d + 1) % RD);     // B(s+d+1); past the end: an unused tile
                    const unsigned long long t1_ = PC_NOW();
                    __syncthreads();
                    tp_pool += t1_ - t0_; tp_bar += PC_NOW() - t1_;
                }
            }
        }
#ifdef PVHIP_DIAG
        if ((abl & 64) && blockIdx.x == 77 && wid == CONSUMERS && lane == 0) {
            g_pc_stamps[0] = (unsigned long long)nk; g_pc_stamps[1] = tp_pool; g_pc_stamps[2] = tp_bar; g_pc_stamps[3] = PC_NOW() - tp_begin;
        }
#endif
        (void)tp_pool; (void)tp_bar; (void)tp_begin;
#undef PVP_LOAD
#undef PVP_POOL
    } else {
        // ------------------------------------------------------------------ consumers
        if (a.prio == 2) __builtin_amdgcn_s_setprio(3);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, a.wp_bytes, 0x00020000);
        unsigned avoff[A_PER_WAVE];
#pragma unroll
        for (int q = 0; q < A_PER_WAVE; ++q) {
            const int f = (wid + CONSUMERS * q) * 256 + lane * 4;
            avoff[q]    = (unsigned)((f / BM) * a.kout_pad + m0 + (f % BM)) * 4u;
        }
        const unsigned a_stage_bytes = (unsigned)(kBK * a.kout_pad) * 4u;
#define PVP_LOAD_A(s_, buf_)                                                                                     \
    _Pragma("unroll") for (int q = 0; q < A_PER_WAVE; ++q)                                                       \
        if (A_PIECES % CONSUMERS == 0 || wid + CONSUMERS * q < A_PIECES)                                         \
            if (!(abl & 8)) pc_dma_b128(wr, &As[buf_][0][0] + (wid + CONSUMERS * q) * 256, avoff[q], (unsigned)(s_) * a_stage_bytes);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        PVP_LOAD_A(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int b_col = wid * 32 + l31;
        unsigned long long tc_mfma = 0ull, tc_vm = 0ull, tc_bar = 0ull;
        const unsigned long long tc_begin = PC_NOW();
        for (int s = 0; s < nk; ++s) {
            const int buf = s & 1;
            const unsigned long long t0_ = PC_NOW();
            PVP_LOAD_A(s + 1, buf ^ 1);          // past the end: the spare zero stages of the panel
            __builtin_amdgcn_sched_barrier(0);
            if (kF16) {
                typedef _Float16 half8 __attribute__((ext_vector_type(8)));
                half8 b8;               // MFMA operand layout: lane (column l31, half lh) holds reduction rows 8 lh .. 8 lh + 7
#pragma unroll
                for (int q = 0; q < 8; ++q) b8[q] = (_Float16)Bs[buf][8 * lh + q][b_col];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    half8 a8;
#pragma unroll
                    for (int q = 0; q < 8; ++q) a8[q] = (_Float16)As[buf][8 * lh + q][l31 + i * 32];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                continue;
            }
            float af[2][TM], bf[2];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[0][i] = As[buf][lh][l31 + i * 32];
            bf[0] = Bs[buf][lh][b_col];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk + 1 < KK) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[nxt][i] = As[buf][2 * (kk + 1) + lh][l31 + i * 32];
                    bf[nxt] = Bs[buf][2 * (kk + 1) + lh][b_col];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) if (!(abl & 4)) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur], acc[i], 0, 0, 0);
                if (kk + 1 < KK) __builtin_amdgcn_sched_group_barrier(0x100, TM + 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t1_ = PC_NOW();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t2_ = PC_NOW();
            __syncthreads();
            tc_mfma += t1_ - t0_; tc_vm += t2_ - t1_; tc_bar += PC_NOW() - t2_;
        }
#ifdef PVHIP_DIAG
        if ((abl & 64) && blockIdx.x == 77 && wid == 0 && lane == 0) {
            g_pc_stamps[4] = tc_mfma; g_pc_stamps[5] = tc_vm; g_pc_stamps[6] = tc_bar; g_pc_stamps[7] = PC_NOW() - tc_begin;
        }
#endif
        (void)tc_mfma; (void)tc_vm; (void)tc_bar; (void)tc_begin;
#undef PVP_LOAD_A

        // epilogue (consumers only): bias, activation, NCHW stores -- 128-byte runs of consecutive pixels
        const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0,
                                                                            a.bias != nullptr ? a.K * 4 : 0, 0x00020000);
        const int gp = ptile * BN + wid * 32 + l31;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row0 = m0 + i * 32 + 4 * lh;
            float     bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (unsigned)(row0 + (r & 3) + 8 * (r >> 2)) * 4u, 0, 0));
            if (gp >= a.P) continue;
            const int n = gp / HW, rem = gp - n * HW;
            float* __restrict__ yp = a.y + ((size_t)n * a.y_ctotal + a.y_coff + row0) * HW + rem;
            float vv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) vv[r] = acc[i][r];
            bias_act_n<16>(vv, bv, a.bias != nullptr, a.relu, act_bounds(a.relu, a.act_lo, a.act_hi));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                if (row0 + dr < a.K && !(abl & 16)) conv_store1(yp + (size_t)dr * HW, vv[r]);
            }
        }
    }
}

inline int round_up_int(int v, int q) { return (v + q - 1) / q * q; }

bool pooled_supported(int n, int c, int h, int w, int k_out) {
    // Rows of whole 16-byte groups (28x28 modules) or 8-byte groups (14x14); PVHIP_FUSE_POOLCONV=4 keeps it to the former (A/B runs).
    // Round 4: odd widths (7x7 modules) with single-pixel groups, for 65 .. 128 output channels (the 128-channel form has the
    // registers for eight iterations of loads in flight) only with PVHIP_FUSE_POOLCONV=1: on GoogLeNet's 7x7 modules the fused launch
    // takes 0.089 ms against 0.021 + 0.043 for the two (three dword loads per pooled value), so the default keeps them apart
    if (settings().fuse_poolconv == 0) return false;
    const int min_vec = settings().fuse_poolconv == 4 ? 4 : (settings().fuse_poolconv == 2 ? 2 : 1);
    if (w % min_vec != 0) return false;
    if (n <= 0 || c < kBK || c % kBK != 0 || h <= 0 || w <= 0 || k_out <= 0 || k_out > 128) return false;
    if (w % 2 != 0 && k_out <= 64) return false;
    if ((unsigned long long)n * c * h * w >= (1ull << 29)) return false;
    return true;
}

}  // namespace

extern "C" {

#ifdef PVHIP_DIAG
// diagnostic build only: the cycle accounts of conv_pool1x1_kernel (PVHIP_CONV_ABLATE bit 64); out = 8 counters
int pvhip_diag_poolconv_stamps(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pc_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return PVHIP_EHIP;
    return PVHIP_OK;
}
#endif

int pvhip_conv2d_pooled_supported(int n, int c, int h, int w, int k_out) { return pooled_supported(n, c, h, w, k_out) ? 1 : 0; }

static int conv2d_pooled_impl(const float* x, const float* wpack, float* y, int n, int c, int h, int w, int k_out, const float* bias,
                              int act, int out_channel_offset, int out_channels_total, float act_lo, float act_hi, bool f16) {
    PVHIP_REQUIRE_INIT();
    if (!pooled_supported(n, c, h, w, k_out))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_pooled_f32: shape outside the fused kernel (ask pvhip_conv2d_pooled_supported first)");
    PVHIP_CHECK_ARG(x != nullptr && wpack != nullptr && y != nullptr && act >= 0 && act <= 2);
    PVHIP_CHECK_ARG(out_channels_total == 0 || (out_channel_offset >= 0 && out_channel_offset + k_out <= out_channels_total));
    const unsigned long long out_e = (unsigned long long)n * (out_channels_total > 0 ? out_channels_total : k_out) * h * w;
    if (out_e >= (1ull << 31)) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_pooled_f32: output exceeds 2^31 elements");
    PoolConvArgs a;
    constexpr int kTabSpare = 2 * kBK, kPanelSpare = 2 * kBK, kKoutAlign = 128;      // the panel layout of pvhip_conv2d_pack_f32
    const int kred_pad = round_up_int(c, kBK);
    a.kout_pad = round_up_int(k_out, kKoutAlign);
    a.x = x; a.wp = wpack + 2 * (kred_pad + kTabSpare); a.y = y; a.bias = bias;
    a.N = n; a.C = c; a.H = h; a.W = w; a.K = k_out;
    a.P = n * h * w;
    a.x_bytes  = (unsigned)((unsigned long long)n * c * h * w * 4ull);
    a.wp_bytes = (unsigned)((size_t)(kred_pad + kPanelSpare) * a.kout_pad * sizeof(float));
    a.relu = act; a.act_lo = act_lo; a.act_hi = act_hi;
    a.y_ctotal = out_channels_total > 0 ? out_channels_total : k_out;
    a.y_coff   = out_channels_total > 0 ? out_channel_offset : 0;
    a.abl = 0;
    a.prio = settings().poolconv_prio >= 0 ? settings().poolconv_prio : ((w % 4 != 0) ? 1 : 0);      // (see pvhip_common.h)
#ifdef PVHIP_DIAG
    a.abl = settings().conv_ablate;
#endif
    int bm = k_out <= 32 ? 32 : (k_out <= 64 ? 64 : 128);
    if (settings().tune[0] == 32 || settings().tune[0] == 64) bm = settings().tune[0] < bm ? settings().tune[0] : bm;      // experiment: narrower channel tiles = more workgroups
    a.n_mtiles = (k_out + bm - 1) / bm;
    // Tiles of 64 pixels (NW = 2: twice the workgroups) only with PVHIP_TUNE1=2 (A/B runs; scripts/time_poolconv_bn.py).  MEASURED, same box: the 14x14
    // modules at batch 256 (392 tiles of 128 pixels for 256 CUs) 0.064-0.067 ms either way, 4e and 3b 15-30 % SLOWER on the narrow tiles: a CU's
    // rate over these stages does not depend on how many workgroups share it (16 images: one tile per CU takes 0.037 ms = 1.25 us per stage,
    // the load latency; 256 images: 0.066 ms at 1.5 tiles per CU) -- the bound is not occupancy.
    const int  n_pt128 = (a.P + 127) / 128;
    const bool narrow  = !f16 && w % 2 == 0 && settings().tune[1] == 2;
    const int  n_ptiles = narrow ? (a.P + 63) / 64 : n_pt128;
    const dim3 grid((unsigned)(a.n_mtiles * n_ptiles)), block(narrow ? 256 : 512);
    const bool v4 = w % 4 == 0;
    if (narrow) {
#define PVP_NARROW(BM_)                                                                                                       \
        { if (v4) hipLaunchKernelGGL((conv_pool1x1_kernel<BM_, 4, false, 2>), grid, block, 0, state().stream, a);             \
          else    hipLaunchKernelGGL((conv_pool1x1_kernel<BM_, 2, false, 2>), grid, block, 0, state().stream, a); }
        if (bm == 32) PVP_NARROW(32)
        else if (bm == 64) PVP_NARROW(64)
        else PVP_NARROW(128)
#undef PVP_NARROW
        PVHIP_LAUNCH_CHECK();
        return PVHIP_OK;
    }
#define PVP_LAUNCH(BM_)                                                                                           \
    do {                                                                                                          \
        if (f16) {                                                                                                \
            if (v4) hipLaunchKernelGGL((conv_pool1x1_kernel<BM_, 4, true>), grid, block, 0, state().stream, a);   \
            else    hipLaunchKernelGGL((conv_pool1x1_kernel<BM_, 2, true>), grid, block, 0, state().stream, a);   \
        } else {                                                                                                  \
            if (v4) hipLaunchKernelGGL((conv_pool1x1_kernel<BM_, 4>), grid, block, 0, state().stream, a);         \
            else    hipLaunchKernelGGL((conv_pool1x1_kernel<BM_, 2>), grid, block, 0, state().stream, a);         \
        }                                                                                                         \
    } while (0)
    if (w % 2 != 0) {                   // odd width: single-pixel groups, 128-channel tiles only (pooled_supported)
        if (f16) hipLaunchKernelGGL((conv_pool1x1_kernel<128, 1, true>), grid, block, 0, state().stream, a);
        else     hipLaunchKernelGGL((conv_pool1x1_kernel<128, 1>), grid, block, 0, state().stream, a);
    } else if (bm == 32) PVP_LAUNCH(32);
    else if (bm == 64) PVP_LAUNCH(64);
    else PVP_LAUNCH(128);
#undef PVP_LAUNCH
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_conv2d_pooled_f32(const float* x, const float* wpack, float* y, int n, int c, int h, int w, int k_out, const float* bias,
                            int act, int out_channel_offset, int out_channels_total, float act_lo, float act_hi) {
    return conv2d_pooled_impl(x, wpack, y, n, c, h, w, k_out, bias, act, out_channel_offset, out_channels_total, act_lo, act_hi, false);
}

int pvhip_conv2d_pooled_f16(const float* x, const float* wpack, float* y, int n, int c, int h, int w, int k_out, const float* bias,
                            int act, int out_channel_offset, int out_channels_total, float act_lo, float act_hi) {
    return conv2d_pooled_impl(x, wpack, y, n, c, h, w, k_out, bias, act, out_channel_offset, out_channels_total, act_lo, act_hi, true);
}

}  // extern "C"
