// Convolution as an implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
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
#include <cstring>

#include "pvhip_common.h"
#include "pvhip_wino.h"

using namespace pvhip;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kBK       = 16;   // reduction rows per stage
constexpr int kTabSpare   = 2 * kBK;   // padding rows after the gather table (prefetch / unrolled look-ahead)
constexpr int kPanelSpare = 2 * kBK;   // zero rows after the weight panel
constexpr int kKoutAlign = 128;  // packed panel width is a multiple of this
constexpr int kMaxConvDests = PVHIP_MAX_CONV_DESTS;

struct ConvArgs {
    const float* x;
    const int*   ktab;  // [2][kred_pad + 2*kBK] : byte offsets c*H*W + r*W + s, then bit indices r*kw + s
    const float* wp;    // [kred_pad][kout_pad]
    float*       y;
    const float* bias;  // optional [K]
    int N, C, H, W, K, OH, OW;
    int sh, sw, pt, pl, kh, kw;
    unsigned x_bytes, wp_bytes;
    int kred_pad, kout_pad;
    int P;              // N*OH*OW
    int n_mtiles, n_ptiles;
    int   relu;             // epilogue activation: 0 none, 1 ReLU (ReLU.py:11), 2 clamp to [act_lo, act_hi] (Clamp.py:11)
    float act_lo, act_hi;
    int y_ctotal, y_coff;   // channels of the tensor y points into, and this convolution's first channel in it
    // Several convolutions of the same input as one launch (conv_igemm_dma_kernel only): the panel holds their output
    // channels one after the other, each range padded to whole 32-channel tiles; a tile belongs to one range and stores
    // into that range's tensor.  nseg == 0: the single destination above.
    int nseg;
    struct Seg {
        float* y;
        int m_begin, k;     // first panel row of the range, real output channels in it
        int ctotal, coff;   // as y_ctotal / y_coff
        int layout;         // 0: fp32 NCHW; 1 (f16 form only): fp16, channels blocked by eight ([n][ceil16(k) / 8][oh * ow][8]: pvhip_conv_dest)
    } seg[kMaxConvDests];
};

// One gather element of the im2col tile: returns x[n, c, ih0 + r, iw0 + s] or 0 for a padding cell.
// koff / rs are the wave-uniform table entry of the reduction row (byte offset c*H*W + r*W + s, and
// the bit index r*kw + s -- or (r << 8) | s on the compare path), `inb` is the lane's in-bounds bit
// mask over (r, s), `xoff` the lane's byte offset of x[n, 0, ih0, iw0].  An out-of-bounds cell becomes
// an out-of-range buffer offset (the hardware returns 0): no branch, no select on the data.
template <bool kMask>
__device__ __forceinline__ float gather_one(__amdgpu_buffer_rsrc_t xr, int koff, int rs, unsigned long long inb,
                                            unsigned xoff, int ih0, int iw0, int H, int W) {
    unsigned bit;
    if (kMask) {
        bit = (unsigned)(inb >> rs) & 1u;   // padding rows carry rs = 63, a bit that is never set
    } else {
        const int r = rs >> 8, s = rs & 0xff;   // padding rows carry r = 0x7fff
        bit = (((unsigned)(ih0 + r) < (unsigned)H) & ((unsigned)(iw0 + s) < (unsigned)W)) ? 1u : 0u;
    }
    // The whole offset goes through the VGPR: the 32-bit wrap of xoff + koff is what makes a window that
    // starts in the top/left padding (xoff "negative") land on the right element.  A padding cell gets
    // the offset 2^31, which is >= num_records (the host checks x_bytes <= 2^31) and, unlike an all-ones
    // offset, cannot wrap back into range when the address unit adds the access size.
    const unsigned off = bit ? (xoff + (unsigned)koff) : 0x80000000u;
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, 0, 0));
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool kMask>
__global__ __launch_bounds__(kBlock, 2) void conv_igemm_kernel(ConvArgs a) {
    static_assert(WAVES_M * WAVES_N == kBlock / kWave, "4 waves per workgroup");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1 && WM % 32 == 0 && WN % 32 == 0, "wave tile is a multiple of 32x32");
    static_assert(BN % kWave == 0 && kBlock % BN == 0, "a wave gathers whole reduction rows");
    constexpr int B_LOADS    = kBK * BN / kBlock;   // reduction rows gathered per lane per stage
    constexpr int A_F4_TOTAL = kBK * BM / 4;
    constexpr int A_F4       = (A_F4_TOTAL + kBlock - 1) / kBlock;
    constexpr int KK         = kBK / 2;             // MFMA steps per stage

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
    const int prow0 = __builtin_amdgcn_readfirstlane(tid / BN) * B_LOADS;  // first reduction row of this wave
    int                ih0 = 0, iw0 = 0;
    unsigned           xoff = 0;
    unsigned long long inb  = 0;
    {
        const int gp = ptile * BN + pc;
        if (gp < a.P) {
            const int n   = gp / OHW;
            const int rem = gp - n * OHW;
            const int oy  = rem / a.OW;
            const int ox  = rem - oy * a.OW;
            ih0           = oy * a.sh - a.pt;
            iw0           = ox * a.sw - a.pl;
            xoff          = (unsigned)(n * a.C * HW + ih0 * a.W + iw0) * 4u;
            if (kMask) {
                for (int r = 0; r < a.kh; ++r)
                    for (int s = 0; s < a.kw; ++s)
                        if ((unsigned)(ih0 + r) < (unsigned)a.H && (unsigned)(iw0 + s) < (unsigned)a.W)
                            inb |= 1ull << (r * a.kw + s);
            }
        } else {
            ih0 = INT_MIN / 2;  // every bounds test fails; the mask stays 0
        }
    }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);

    float  breg[B_LOADS];
    float4 areg[A_F4];
    int    tko[B_LOADS], trs[B_LOADS];   // table entries of the NEXT stage to gather (scalar registers)
    const int* __restrict__ tab_rs = a.ktab + a.kred_pad + kTabSpare;

#define PV_LOAD_ENT(kt_)                                                              \
    {                                                                                 \
        const int* __restrict__ tp = a.ktab + (kt_) * kBK + prow0;                    \
        const int* __restrict__ tq = tab_rs + (kt_) * kBK + prow0;                    \
        _Pragma("unroll") for (int j = 0; j < B_LOADS; ++j) { tko[j] = tp[j]; trs[j] = tq[j]; } \
    }
#define PV_GATHER(j_) breg[j_] = gather_one<kMask>(xr, tko[j_], trs[j_], inb, xoff, ih0, iw0, a.H, a.W)
#define PV_LOAD_A(kt_)                                                                \
    _Pragma("unroll") for (int j = 0; j < A_F4; ++j) {                                \
        const int f = tid + j * kBlock;                                               \
        if (A_F4_TOTAL % kBlock == 0 || f < A_F4_TOTAL) {                             \
            const int arow = f / (BM / 4), ac4 = f % (BM / 4);                        \
            areg[j] = *reinterpret_cast<const float4*>(a.wp + (size_t)((kt_) * kBK + arow) * a.kout_pad + m0 + ac4 * 4); \
        }                                                                             \
    }
#define PV_STORE_TILES(buf_)                                                          \
    {                                                                                 \
        _Pragma("unroll") for (int j = 0; j < B_LOADS; ++j) Bs[buf_][prow0 + j][pc] = breg[j]; \
        _Pragma("unroll") for (int j = 0; j < A_F4; ++j) {                            \
            const int f = tid + j * kBlock;                                           \
            if (A_F4_TOTAL % kBlock == 0 || f < A_F4_TOTAL) {                         \
                const int arow = f / (BM / 4), ac4 = f % (BM / 4);                    \
                *reinterpret_cast<float4*>(&As[buf_][arow][ac4 * 4]) = areg[j];       \
            }                                                                         \
        }                                                                             \
    }

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
    // prologue: stage 0 into LDS buffer 0; table entries of stage 1 into scalar registers
    PV_LOAD_ENT(0);
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j) PV_GATHER(j);
    PV_LOAD_A(0);
    PV_STORE_TILES(0);
    PV_LOAD_ENT(1);   // the table has one spare stage of padding rows at its end
    __syncthreads();

    // The loop body has no conditionals: the last iteration gathers and stages one stage past the end
    // (table rows there are padding rows -> the loads read as 0; the weight panel has one spare zero
    // stage), which keeps every wait counter of the body exact.
    //
    // Order inside one stage (pinned with sched_barrier so the scheduler cannot sink the global loads
    // behind the MFMAs, which would expose their whole latency before the LDS write):
    //   1. LDS reads of the first MFMA step of stage kt          (latency hidden behind 2.)
    //   2. all global loads of stage kt+1 (A panel + gather)      (in flight during 3.)
    //   3. MFMA steps of stage kt, operands of step kk+1 read from LDS before the MFMAs of step kk
    //   4. table prefetch for stage kt+2, LDS write of stage kt+1, barrier
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = As[buf][lh][a_col + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = Bs[buf][lh][b_col + j * 32];
        PV_LOAD_A(kt + 1);
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) PV_GATHER(j);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < KK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nxt][i] = As[buf][2 * (kk + 1) + lh][a_col + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[nxt][j] = Bs[buf][2 * (kk + 1) + lh][b_col + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
            // emit this step as: LDS reads of step kk+1, then the MFMAs of step kk
            if (kk + 1 < KK) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        PV_LOAD_ENT(kt + 2);   // consumed one whole stage later
        PV_STORE_TILES(buf ^ 1);
        __syncthreads();
    }
#undef PV_LOAD_ENT
#undef PV_GATHER
#undef PV_LOAD_A
#undef PV_STORE_TILES

    // ---- epilogue: accumulator register r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gp = ptile * BN + wn * WN + j * 32 + l31;
        if (gp >= a.P) continue;
        const int    n    = gp / OHW;
        const int    rem  = gp - n * OHW;
        float* __restrict__ yp = a.y + ((size_t)n * a.y_ctotal + a.y_coff) * OHW + rem;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ko = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ko < a.K) {
                    float v = acc[i][j][r];
                    if (a.bias != nullptr) v = v + a.bias[ko];
                    v = act_apply(v, act_bounds(a.relu, a.act_lo, a.act_hi));
                    conv_store1(yp + (size_t)ko * OHW, v);
                }
            }
        }
    }
}

#ifdef PVHIP_DIAG   // predecessor kernel, kept for A/B runs in the diagnostic build only (PVHIP_CONV_KERNEL=lds)
// ---------------------------------------------------------------------------------------------------
// (r,s)-major variant of the LDS-tiled kernel, used whenever C is a multiple of the stage depth (16).
//
// On gfx950 the fp32 MFMA executes on the SIMD's vector FMA lanes: VALU instructions do not co-issue with
// it (SQ_VALU_MFMA_COEXEC_CYCLES == 0 on this kernel family), so every VALU instruction in the reduction
// loop is MFMA time lost.  The c-major reduction order needs ~5 VALU per gathered element (window-bit
// test, offset add, select).  Ordering the reduction (r,s)-major instead -- row = (r*kw + s)*C + c, the
// weight panel is packed to match -- makes the window tap constant over the C/16 stages of one (r,s):
// the lane's byte offset `voff` (or the out-of-range sentinel when the tap falls in the padding) is
// computed once per tap, and the 16 channel rows of a stage differ only by a wave-uniform soffset
// c*H*W*4 handled by the scalar unit.  The gather is then buffer_load_dword voff, soffset with ZERO
// VALU instructions per element.  Everything else (LDS staging, stage order, tiles) is as in
// conv_igemm_kernel.  Epilogue: bias is fetched with range-checked buffer loads (no per-element bounds code).
template <int BM, int BN, int WAVES_M, int WAVES_N, bool kPW>   // kPW: pointwise (1x1, stride 1, no padding, H*W % 4 == 0)
__global__ __launch_bounds__(kBlock, 2) void conv_igemm_rs_kernel(ConvArgs a) {
    static_assert(WAVES_M * WAVES_N == kBlock / kWave, "4 waves per workgroup");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1 && WM % 32 == 0 && WN % 32 == 0, "wave tile is a multiple of 32x32");
    static_assert(BN % kWave == 0 && kBlock % BN == 0, "a wave gathers whole reduction rows");
    constexpr int B_LOADS    = kBK * BN / kBlock;         // dword gather: rows per lane per stage
    constexpr int B_LOADS4   = kBK * BN / 4 / kBlock;     // pointwise: 16-byte loads per lane per stage
    constexpr int QUADS      = BN / 4;                    // pixel quads per tile row
    constexpr int ROWS_PASS  = kBlock / QUADS;            // tile rows covered by one pass of the workgroup
    constexpr int A_F4_TOTAL = kBK * BM / 4;
    constexpr int A_F4       = (A_F4_TOTAL + kBlock - 1) / kBlock;
    constexpr int KK         = kBK / 2;
    constexpr unsigned kOob  = 0x80000000u;
    static_assert(QUADS % 32 == 0 || QUADS == 32, "a half wave covers whole tile rows");

    __shared__ __attribute__((aligned(16))) float As[2][kBK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][kBK][BN];

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

    const int OHW = a.OH * a.OW;
    const int HW  = a.H * a.W;
    const unsigned chan_bytes = (unsigned)HW * 4u;
    // dword gather: lane <-> one pixel, B_LOADS consecutive rows; pointwise: lane <-> one pixel quad of one row
    const int pc    = kPW ? (tid % QUADS) * 4 : tid % BN;
    const int prow0 = kPW ? 0 : __builtin_amdgcn_readfirstlane(tid / BN) * B_LOADS;
    const int qrow  = tid / QUADS;                       // pointwise: tile row of pass 0 (wave-uniform up to lane>>5)
    unsigned           xoff = 0;
    unsigned long long inb  = 0;     // bit (r*kw + s): tap inside the image for this lane's pixel
    {
        const int gp = ptile * BN + pc;
        if (gp < a.P) {
            const int n   = gp / OHW;
            const int rem = gp - n * OHW;
            if (kPW) {
                xoff = (unsigned)(n * a.C * HW + rem) * 4u;
                inb  = 1ull;
            } else {
                const int oy  = rem / a.OW;
                const int ox  = rem - oy * a.OW;
                const int ih0 = oy * a.sh - a.pt;
                const int iw0 = ox * a.sw - a.pl;
                xoff          = (unsigned)(n * a.C * HW + ih0 * a.W + iw0) * 4u;
                for (int r = 0; r < a.kh; ++r)
                    for (int s = 0; s < a.kw; ++s)
                        if ((unsigned)(ih0 + r) < (unsigned)a.H && (unsigned)(iw0 + s) < (unsigned)a.W)
                            inb |= 1ull << (r * a.kw + s);
            }
        }
    }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const int* __restrict__ rstab = a.ktab;            // [kh*kw + spare]: (r*W + s)*4
    const int      ncs        = a.C / kBK;              // channel stages per tap
    const int      nrs        = a.kh * a.kw;

    float  breg[kPW ? 1 : B_LOADS];
    float4 breg4[kPW ? B_LOADS4 : 1];
    float4 areg[A_F4];

    // state of the stage being LOADED (one ahead of the stage being multiplied)
    int      rs_l = 0, cs_l = 0;
    unsigned voff = (inb & 1ull) ? xoff + (unsigned)rstab[0] : kOob;
    // pointwise: the lane's row inside a pass differs between the two half-waves only when a wave spans two
    // tile rows (QUADS == 32); that lane-constant part is folded into voff, the wave-uniform part is added to
    // the scalar row base.  Lanes whose pixel quad lies past the tensor read quad 0 (their columns are never
    // stored): plain global loads have no range check.
    const int qrow_u = __builtin_amdgcn_readfirstlane(qrow);          // row of lane 0 of this wave
    if (kPW) voff = ((inb & 1ull) ? xoff : 0u) + (unsigned)(qrow - qrow_u) * chan_bytes;

#define PV2_GATHER()                                                                                    \
    if (kPW) {                                                                                          \
        /* plain 16-byte global loads: uniform row base (scalar) + the lane's 32-bit byte offset.  (The     \
           16-byte raw-buffer-load builtins of this toolchain lower to a single dword load.) */            \
        _Pragma("unroll") for (int j = 0; j < B_LOADS4; ++j) {                                          \
            const char* rowp = reinterpret_cast<const char*>(a.x) +                                      \
                               (size_t)(cs_l * kBK + qrow_u + j * ROWS_PASS) * chan_bytes;               \
            breg4[j] = *reinterpret_cast<const float4*>(rowp + voff);                                   \
        }                                                                                               \
    } else {                                                                                            \
        const unsigned sbase = (unsigned)(cs_l * kBK + prow0) * chan_bytes;                             \
        _Pragma("unroll") for (int j = 0; j < B_LOADS; ++j)                                             \
            breg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, voff, sbase + (unsigned)j * chan_bytes, 0)); \
    }
#define PV2_ADVANCE()                                                                                   \
    if (++cs_l == ncs) {                                                                                \
        cs_l = 0;                                                                                       \
        ++rs_l;                                                                                         \
        if (kPW) {                                                                                      \
            --rs_l;                                     /* single tap: the look-ahead past the end re-reads stage 0 (unused) */ \
        } else {                                                                                        \
            const unsigned ro = (unsigned)rstab[rs_l];  /* spare zero entries past the last tap */      \
            voff = (rs_l < nrs && ((inb >> rs_l) & 1ull)) ? xoff + ro : kOob;                           \
        }                                                                                               \
    }
#define PV2_LOAD_A(kt_)                                                               \
    _Pragma("unroll") for (int j = 0; j < A_F4; ++j) {                                \
        const int f = tid + j * kBlock;                                               \
        if (A_F4_TOTAL % kBlock == 0 || f < A_F4_TOTAL) {                             \
            const int arow = f / (BM / 4), ac4 = f % (BM / 4);                        \
            areg[j] = *reinterpret_cast<const float4*>(a.wp + (size_t)((kt_) * kBK + arow) * a.kout_pad + m0 + ac4 * 4); \
        }                                                                             \
    }
#define PV2_STORE_TILES(buf_)                                                         \
    {                                                                                 \
        if (kPW) {                                                                    \
            _Pragma("unroll") for (int j = 0; j < B_LOADS4; ++j)                      \
                *reinterpret_cast<float4*>(&Bs[buf_][qrow + j * ROWS_PASS][pc]) = breg4[j]; \
        } else {                                                                      \
            _Pragma("unroll") for (int j = 0; j < B_LOADS; ++j) Bs[buf_][prow0 + j][pc] = breg[j]; \
        }                                                                             \
        _Pragma("unroll") for (int j = 0; j < A_F4; ++j) {                            \
            const int f = tid + j * kBlock;                                           \
            if (A_F4_TOTAL % kBlock == 0 || f < A_F4_TOTAL) {                         \
                const int arow = f / (BM / 4), ac4 = f % (BM / 4);                    \
                *reinterpret_cast<float4*>(&As[buf_][arow][ac4 * 4]) = areg[j];       \
            }                                                                         \
        }                                                                             \
    }

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

    const int nk = nrs * ncs;
    PV2_GATHER();
    PV2_ADVANCE();
    PV2_LOAD_A(0);
    PV2_STORE_TILES(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = As[buf][lh][a_col + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = Bs[buf][lh][b_col + j * 32];
        PV2_LOAD_A(kt + 1);
        PV2_GATHER();           // stage kt+1 (past the end: every lane reads the out-of-range sentinel -> 0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < KK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nxt][i] = As[buf][2 * (kk + 1) + lh][a_col + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[nxt][j] = Bs[buf][2 * (kk + 1) + lh][b_col + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
            if (kk + 1 < KK) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        PV2_STORE_TILES(buf ^ 1);
        PV2_ADVANCE();
        __syncthreads();
    }
#undef PV2_GATHER
#undef PV2_ADVANCE
#undef PV2_LOAD_A
#undef PV2_STORE_TILES

    // ---- epilogue: accumulator register r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0,
                                                                        a.bias != nullptr ? a.K * 4 : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row0 = m0 + wm * WM + i * 32 + 4 * lh;      // this lane's rows: row0 + (r&3) + 8*(r>>2)
        float     bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // range-checked dword loads: channels >= K (and a null bias: 0 records) read as 0.  (16-byte buffer
            // loads through this descriptor return the first dword in all four lanes of the result on gfx950.)
            bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                br, (unsigned)(row0 + (r & 3) + 8 * (r >> 2)) * 4u, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gp = ptile * BN + wn * WN + j * 32 + l31;
            if (gp >= a.P) continue;
            const int n   = gp / OHW;
            const int rem = gp - n * OHW;
            float* __restrict__ yp = a.y + ((size_t)n * a.y_ctotal + a.y_coff + row0) * OHW + rem;
            float vv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) vv[r] = acc[i][j][r];
            bias_act_n<16>(vv, bv, a.bias != nullptr, a.relu, act_bounds(a.relu, a.act_lo, a.act_hi));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                if (row0 + dr < a.K) conv_store1(yp + (size_t)dr * OHW, vv[r]);
            }
        }
    }
}
#endif  // PVHIP_DIAG

// ---------------------------------------------------------------------------------------------------
// LDS-DMA variant of the (r,s)-major kernel: both tiles go global -> LDS with `buffer_load ... lds`, no
// staging registers and no ds_write pass.
//
// The K-major LDS images make this natural.  One wave-instruction of the im2col gather fills 64
// consecutive floats of one tile row (lane <-> pixel, per-lane source offset `voff` or the out-of-range
// sentinel -> the hardware writes 0, wave-uniform channel row in soffset, wave-uniform LDS destination in
// M0); the weight tile [16][BM] is a dense copy of BM/16 one-KiB pieces (16 bytes per lane).  A stage is
// 8 gather instructions + at most 2 weight pieces per wave; the loads of stage t+1 are issued before the
// MFMAs of stage t and waited for only at the end of it.
typedef const __attribute__((address_space(4))) int* const_int_p;

// Global -> LDS loads: lds_dma_b32 / lds_dma_b128 / lds_dma_wait_all of pvhip_common.h (asm statements on purpose, see there).
#define dma_b32 lds_dma_b32
#define dma_b128 lds_dma_b128
#define dma_wait_all lds_dma_wait_all

// kPW (pointwise: 1x1, stride 1, no padding, H*W % 4 == 0, (r,s)-major): the im2col tile is then a plain copy of 16
// channel rows x 128 consecutive pixels, moved as 8 one-KiB pieces (16 bytes per lane, two tile rows per piece) instead
// of 32 dword gathers per stage.
// kValid (c-major only): no padding and every window inside the tensor -- no window test at all, the tap's offset rides in the scalar
// offset of the load: ZERO vector instructions per gathered row instead of five (shift, and, compare, select, add), which were 45 of
// conv1's 61 issue slots per 16 MFMAs.  Padded layers reach it through pvhip_pad2d_f32 (the Convolution plugin pads once per launch).
// kF16 (FP16 IRs, (r,s)-major only): the SAME fp32 tiles in LDS, but a stage of 16 channels is ONE v_mfma_f32_32x32x16_f16 per 32-channel
// tile: a lane reads its eight reduction rows of both operands, rounds them to fp16 (to nearest even; the constants of an FP16 IR are fp16
// values already) and the matrix cores accumulate in fp32 -- 32 MFMA cycles per tile and stage instead of 512, after which the launch is
// as fast as its LDS-DMA copies (Convolution.py:57-87 computed in numpy float16 by the reference, common_def.py:13-17).
template <int BM, bool kRS, bool kPW = false, bool kValid = false, bool kF16 = false>   // kRS: (r,s)-major reduction order (C % 16 == 0); else c-major with the window-bit table
__global__ __launch_bounds__(kBlock, 2) void conv_igemm_dma_kernel(ConvArgs a) {
    static_assert(!kPW || kRS, "the pointwise copy uses the (r,s)-major panel");
    static_assert(!kValid || !kRS, "the test-free gather is the c-major one");
    constexpr int BN = 128, TM = BM / 32, KK = kBK / 2;
    constexpr int A_PIECES = kBK * BM * 4 / 1024;          // 1-KiB wave-instructions per weight tile
    constexpr int A_PER_WAVE = (A_PIECES + 3) / 4;
    constexpr int B_LOADS = kBK * BN / kBlock;             // gather instructions per wave per stage
    constexpr unsigned kOob = 0x80000000u;
    static_assert(BM % 32 == 0 && A_PIECES >= 1, "weight tile is whole 1-KiB pieces");

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

    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid  = __builtin_amdgcn_readfirstlane(tid / kWave);

    const int OHW = a.OH * a.OW;
    const int HW  = a.H * a.W;
    const unsigned chan_bytes = (unsigned)HW * 4u;
    const int phalf = (wid & 1) * 64;                      // this wave's 64 pixels of the tile
    const int prow0 = (wid >> 1) * B_LOADS;                // and its 8 reduction rows of every stage
    unsigned           xoff = kValid ? kOob : 0u;           // kValid: a pixel past the end reads zeros through its voffset
    unsigned long long inb  = 0;
    {
        const int gp = ptile * BN + phalf + lane;
        if (gp < a.P) {
            const int n   = gp / OHW;
            const int rem = gp - n * OHW;
            const int oy  = rem / a.OW;
            const int ox  = rem - oy * a.OW;
            const int ih0 = oy * a.sh - a.pt;
            const int iw0 = ox * a.sw - a.pl;
            xoff          = (unsigned)(n * a.C * HW + ih0 * a.W + iw0) * 4u;
            for (int r = 0; r < a.kh; ++r)
                for (int s = 0; s < a.kw; ++s)
                    if ((unsigned)(ih0 + r) < (unsigned)a.H && (unsigned)(iw0 + s) < (unsigned)a.W)
                        inb |= 1ull << (r * a.kw + s);
        }
    }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, a.wp_bytes, 0x00020000);
    // The gather tables were written by conv_pack_kernel in an earlier launch and are constant here: read them
    // through the constant address space so that the loads stay scalar (s_load) next to the asm statements.
    const const_int_p rstab  = (const_int_p)(unsigned long)a.ktab;
    const const_int_p tab_rs = rstab + a.kred_pad + kTabSpare;          // c-major: window-bit index of every row
    const int ncs = kRS ? a.C / kBK : 1;
    const int nrs = a.kh * a.kw;

    // weight pieces of this wave: piece q covers floats [q*256, q*256 + 256) of the [16][BM] image
    unsigned avoff[A_PER_WAVE];
#pragma unroll
    for (int q = 0; q < A_PER_WAVE; ++q) {
        const int f = (wid + 4 * q) * 256 + lane * 4;
        avoff[q]    = (unsigned)((f / BM) * a.kout_pad + m0 + (f % BM)) * 4u;
    }
    const unsigned a_stage_bytes = (unsigned)(kBK * a.kout_pad) * 4u;

    // pointwise: lane -> (row parity inside a piece, group of 4 consecutive pixels); a group never straddles two images
    unsigned pwoff = kOob;
    if (kPW) {
        const int gp = ptile * BN + (lane & 31) * 4;
        if (gp < a.P) {
            const int n = gp / HW, hw = gp - n * HW;
            pwoff = (unsigned)(n * a.C * HW + hw) * 4u + (unsigned)(lane >> 5) * chan_bytes;
        }
    }

    int      rs_l = 0, cs_l = 0, kt_l = 0;      // stage being loaded
    unsigned voff = kRS ? ((inb & 1ull) ? xoff + (unsigned)rstab[0] : kOob) : 0u;
    // c-major: the 8 table entries (byte offset c*H*W + r*W + s, window bit r*kw + s; padding rows carry bit 63,
    // never set) of this wave's rows of the stage being loaded, in scalar registers
    int tko[B_LOADS], trs[B_LOADS];
#define PV3_LOAD_ENT(kt_)                                                                                 \
    if (!kRS) {                                                                                           \
        const const_int_p tp = rstab + (kt_) * kBK + prow0;                                               \
        const const_int_p tq = tab_rs + (kt_) * kBK + prow0;                                              \
        _Pragma("unroll") for (int j = 0; j < B_LOADS; ++j) { tko[j] = tp[j]; trs[j] = tq[j]; }           \
    }
    PV3_LOAD_ENT(0);

#define PV3_ISSUE(buf_)                                                                                   \
    {                                                                                                     \
        if (kPW) {                                                                                        \
            _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                 \
                dma_b128(xr, &Bs[buf_][2 * (wid + 4 * q)][0], pwoff,                                      \
                         (unsigned)(cs_l * kBK + 2 * (wid + 4 * q)) * chan_bytes);                        \
        } else if (kRS) {                                                                                 \
            const unsigned sbase = (unsigned)(cs_l * kBK + prow0) * chan_bytes;                           \
            _Pragma("unroll") for (int j = 0; j < B_LOADS; ++j)                                           \
                dma_b32(xr, &Bs[buf_][prow0 + j][phalf], voff, sbase + (unsigned)j * chan_bytes);         \
        } else {                                                                                          \
            if (kValid && kt_l + 1 < nk) {        /* every row of the stage is a tap, every tap is inside the tensor */ \
                _Pragma("unroll") for (int j = 0; j < B_LOADS; ++j)                                       \
                    dma_b32(xr, &Bs[buf_][prow0 + j][phalf], xoff, (unsigned)tko[j]);                     \
            } else {                              /* (kValid: the last stage and the spare one -- their padding rows read zeros) */ \
                _Pragma("unroll") for (int j = 0; j < B_LOADS; ++j) {                                     \
                    const bool tap = kValid ? trs[j] != 63 : (bool)((unsigned)(inb >> trs[j]) & 1u);      \
                    const unsigned off = tap ? xoff + (unsigned)tko[j] : kOob;                            \
                    dma_b32(xr, &Bs[buf_][prow0 + j][phalf], off, 0u);                                    \
                }                                                                                         \
            }                                                                                             \
        }                                                                                                 \
        _Pragma("unroll") for (int q = 0; q < A_PER_WAVE; ++q)                                            \
            if (A_PIECES % 4 == 0 || wid + 4 * q < A_PIECES)                                              \
                dma_b128(wr, &As[buf_][0][0] + (wid + 4 * q) * 256, avoff[q], (unsigned)kt_l * a_stage_bytes); \
    }
#define PV3_ADVANCE()                                                                                     \
    ++kt_l;                                                                                               \
    PV3_LOAD_ENT(kt_l);                                                                                   \
    if (kRS && ++cs_l == ncs) {                                                                           \
        cs_l = 0;                                                                                         \
        ++rs_l;                                                                                           \
        const unsigned ro = (unsigned)rstab[rs_l];  /* spare zero entries past the last tap */            \
        voff = (rs_l < nrs && ((inb >> rs_l) & 1ull)) ? xoff + ro : kOob;                                 \
    }

    floatx16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int l31 = lane & 31, lh = lane >> 5;
    const int b_col = wid * 32 + l31;

    const int nk = kRS ? nrs * ncs : a.kred_pad / kBK;
    PV3_ISSUE(0);
    PV3_ADVANCE();
    dma_wait_all();
    __syncthreads();

    // c-major: the LAST stage is taken out of the loop.  Its rows past C*kh*kw are zero rows of the panel: only the k-steps that hold
    // a real row are multiplied (conv1 of GoogLeNet: 147 rows = 9 stages + 3 rows, 2 of the last stage's 8 k-steps -- 148 of 160
    // MFMA steps), and nothing is copied for a stage behind it.
    const int nk_loop = (kRS || kF16) ? nk : nk - 1;          // (the f16 form: a stage is ONE matrix instruction, nothing to skip)
    for (int kt = 0; kt < nk_loop; ++kt) {
        const int buf = kt & 1;
        if (kF16) {
            typedef _Float16 half8 __attribute__((ext_vector_type(8)));
            PV3_ISSUE(buf ^ 1);     // stage kt+1 (past the end: the spare zero stages)
            __builtin_amdgcn_sched_barrier(0);
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
            PV3_ADVANCE();
            dma_wait_all();
            __syncthreads();
            continue;
        }
        float af[2][TM], bf[2];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = As[buf][lh][l31 + i * 32];
        bf[0] = Bs[buf][lh][b_col];
        PV3_ISSUE(buf ^ 1);     // stage kt+1 (past the end: the spare zero stages)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < KK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nxt][i] = As[buf][2 * (kk + 1) + lh][l31 + i * 32];
                bf[nxt] = Bs[buf][2 * (kk + 1) + lh][b_col];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur], acc[i], 0, 0, 0);
            if (kk + 1 < KK) __builtin_amdgcn_sched_group_barrier(0x100, TM + 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        PV3_ADVANCE();
        dma_wait_all();
        __syncthreads();
    }
    if (!kRS && !kF16) {
        const int buf   = (nk - 1) & 1;
        const int steps = min(KK, (a.C * nrs - (nk - 1) * kBK + 1) >> 1);     // k-steps of the last stage with a real reduction row
        for (int kk = 0; kk < steps; ++kk) {
            float afl[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) afl[i] = As[buf][2 * kk + lh][l31 + i * 32];
            const float bfl = Bs[buf][2 * kk + lh][b_col];
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(afl[i], bfl, acc[i], 0, 0, 0);
        }
    }
#undef PV3_ISSUE
#undef PV3_LOAD_ENT
#undef PV3_ADVANCE

    // The destination table is read from the argument block HERE, through a pointer the compiler cannot see through:
    // as ordinary arguments its 30 dwords would be loaded at kernel entry and held in scalar registers across the
    // reduction loop, which has none to spare (the LDS-DMA statements need their operands in SGPRs).
    typedef const __attribute__((address_space(4))) ConvArgs* kernarg_p;
    kernarg_p ka = (kernarg_p)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    const int nseg_l = ka->nseg;
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0,
                                                                        a.bias != nullptr ? a.K * 4 : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row0 = m0 + i * 32 + 4 * lh;
        float     bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                br, (unsigned)(row0 + (r & 3) + 8 * (r >> 2)) * 4u, 0, 0));
        const int gp = ptile * BN + wid * 32 + l31;
        if (gp >= a.P) continue;
        const int n   = gp / OHW;
        const int rem = gp - n * OHW;
        float* __restrict__ yb = a.y;
        int yct = a.y_ctotal, ycoff = a.y_coff, klim = a.K;
        int c8_blocks = 0, c8_first = 0;    // a range stored as blocked fp16: its channel blocks, and the first block of this workgroup's tile
        if (nseg_l > 0) {                   // the destination range this 32-channel tile belongs to (workgroup-uniform)
            int sg = 0;
            for (int q = 1; q < nseg_l; ++q) sg = (m0 + i * 32 >= ka->seg[q].m_begin) ? q : sg;
            yb    = ka->seg[sg].y;
            yct   = ka->seg[sg].ctotal;
            ycoff = ka->seg[sg].coff - ka->seg[sg].m_begin;
            klim  = ka->seg[sg].m_begin + ka->seg[sg].k;
            if (kF16 && ka->seg[sg].layout == 1) {
                c8_blocks = (ka->seg[sg].k + 15) / 16 * 2;
                c8_first  = (m0 - ka->seg[sg].m_begin) / 8;      // ranges begin at whole 32-channel tiles
            }
        }
        float* __restrict__ yp = yb + ((size_t)n * yct + ycoff + row0) * OHW + rem;
        float vv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] = acc[i][r];
        bias_act_n<16>(vv, bv, a.bias != nullptr, a.relu, act_bounds(a.relu, a.act_lo, a.act_hi));
        if (kF16 && c8_blocks > 0) {
            // the reader is pvhip_conv2d_f16_c8: fp16 (round to nearest even: the value the reader would round this output to anyway),
            // the eight channels of a pixel in one 16-byte piece.  Registers 4 g .. 4 g + 3 are channels 8 g + 4 lh .. + 3 of the tile:
            // half a piece per lane, the two lane halves of a wave fill it.  Channels past the range's count up to a whole 16-channel
            // stage are written too: zero rows of the panel, zero bias -- zeros, which is what the reader's zero weights expect.
            typedef _Float16 half4v __attribute__((ext_vector_type(4)));
            _Float16* const yh = reinterpret_cast<_Float16*>(yb);
            const int blk0 = c8_first + i * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (blk0 + g >= c8_blocks) continue;
                half4v h;
#pragma unroll
                for (int j = 0; j < 4; ++j) h[j] = (_Float16)vv[4 * g + j];
                *reinterpret_cast<half4v*>(yh + (((size_t)n * c8_blocks + blk0 + g) * OHW + rem) * 8 + 4 * lh) = h;
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (row0 + dr < klim) conv_store1(yp + (size_t)dr * OHW, vv[r]);
        }
    }
}

#ifdef PVHIP_DIAG   // alternative kernel, kept for A/B runs and ablations in the diagnostic build only (PVHIP_CONV_KERNEL=wave)
// ---------------------------------------------------------------------------------------------------
// Wave-direct variant: no LDS staging of operands and no barriers in the reduction loop.
//
// With one VGPR per fp32 MFMA operand and a wave tile of (32*TM) output channels x (32*TN) pixels, the B
// (im2col) elements a wave needs are needed by no other wave of the workgroup, so staging them through
// LDS only adds writes, reads and a barrier per stage.  Here every lane gathers exactly the operand
// element the MFMA wants from it -- lane l supplies B[k = 2*step + (l>>5)][pixel = l&31] -- straight
// into registers, and likewise A[k][k_out = l&31] from the packed panel (128-byte runs; the panel is a
// few hundred KB and lives in L1/L2).  Two register sets alternate (stage t+1 loads are in flight under
// the MFMAs of stage t); waits are counted vmcnt.  The (byte offset, window bit) table of the reduction
// rows is copied to LDS once per workgroup and read per lane with ds_read_b64 (both lane halves read one
// address each: broadcast, conflict-free).  The 4 waves of a workgroup take consecutive output-channel
// tiles of the same pixel tile, so their B loads hit in L1.
template <int TM, int TN, bool kMask, int ABLATE = 0>   // ABLATE (diagnostic builds only): 1 = no B gather, 2 = no A loads
__global__ __launch_bounds__(kBlock, 2) void conv_wave_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) int2 tab[];   // [kred_pad + kTabSpare] {koff bytes, rs}
    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid  = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int l31 = lane & 31, lh = lane >> 5;
    const int tab_n = a.kred_pad + kTabSpare;
    for (int i = tid; i < tab_n; i += kBlock) tab[i] = make_int2(a.ktab[i], a.ktab[tab_n + i]);
    __syncthreads();

    // ---- this wave's tile
    const long n_tiles = (long)a.n_mtiles * a.n_ptiles;
    long       t;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        t = (long)lid * (kBlock / kWave) + wid;
    }
    if (t >= n_tiles) return;   // no barrier after this point
    const int mt = (int)(t % a.n_mtiles);
    const int pt = (int)(t / a.n_mtiles);
    const int m0 = mt * (32 * TM);
    const int p0 = pt * (32 * TN);

    const int OHW = a.OH * a.OW, HW = a.H * a.W;
    int                ih0[TN], iw0[TN];
    unsigned           xoff[TN];
    unsigned long long inb[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gp = p0 + j * 32 + l31;
        ih0[j] = INT_MIN / 2; iw0[j] = 0; xoff[j] = 0; inb[j] = 0;
        if (gp < a.P) {
            const int n   = gp / OHW;
            const int rem = gp - n * OHW;
            const int oy  = rem / a.OW;
            const int ox  = rem - oy * a.OW;
            ih0[j]        = oy * a.sh - a.pt;
            iw0[j]        = ox * a.sw - a.pl;
            xoff[j]       = (unsigned)(n * a.C * HW + ih0[j] * a.W + iw0[j]) * 4u;
            if (kMask) {
                for (int r = 0; r < a.kh; ++r)
                    for (int s = 0; s < a.kw; ++s)
                        if ((unsigned)(ih0[j] + r) < (unsigned)a.H && (unsigned)(iw0[j] + s) < (unsigned)a.W)
                            inb[j] |= 1ull << (r * a.kw + s);
            }
        }
    }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, a.wp_bytes, 0x00020000);
    const unsigned woff = (unsigned)(lh * a.kout_pad + m0 + l31) * 4u;   // lane part of the panel offset
    const unsigned wrow = (unsigned)a.kout_pad * 4u;                     // bytes per panel row

    float areg[2][TM][kBK / 2], breg[2][TN][kBK / 2];

#define PV_WLOAD(set_, kt_)                                                                               \
    {                                                                                                     \
        const int row0 = (kt_) * kBK;                                                                     \
        _Pragma("unroll") for (int s = 0; s < kBK / 2; ++s) {                                             \
            const int2 e = tab[row0 + 2 * s + lh];                                                        \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                \
                breg[set_][j][s] = (ABLATE & 1) ? __builtin_bit_cast(float, (xoff[j] & 0xffffu) | 0x3f800000u) \
                                                : gather_one<kMask>(xr, e.x, e.y, inb[j], xoff[j], ih0[j], iw0[j], a.H, a.W); \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                \
                areg[set_][i][s] = (ABLATE & 2) ? __builtin_bit_cast(float, (woff & 0xffffu) | 0x3f800000u)   \
                                                : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(        \
                    wr, woff + (unsigned)(i * 128), (unsigned)(row0 + 2 * s) * wrow, 0));                 \
        }                                                                                                 \
    }
#define PV_WMMA(set_)                                                                                     \
    _Pragma("unroll") for (int s = 0; s < kBK / 2; ++s)                                                   \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                    \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[set_][i][s], breg[set_][j][s], acc[i][j], 0, 0, 0);

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Stages are processed in pairs with two register sets; an odd tail runs one stage past the end
    // (padding rows of the table read as 0, the panel has spare zero stages).
    const int nk = a.kred_pad / kBK;
    PV_WLOAD(0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
        PV_WLOAD(1, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        PV_WMMA(0);
        __builtin_amdgcn_sched_barrier(0);
        PV_WLOAD(0, kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        PV_WMMA(1);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef PV_WLOAD
#undef PV_WMMA

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gp = p0 + j * 32 + l31;
        if (gp >= a.P) continue;
        const int n   = gp / OHW;
        const int rem = gp - n * OHW;
        float* __restrict__ yp = a.y + ((size_t)n * a.y_ctotal + a.y_coff) * OHW + rem;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ko = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ko < a.K) {
                    float v = acc[i][j][r];
                    if (a.bias != nullptr) v = v + a.bias[ko];
                    v = act_apply(v, act_bounds(a.relu, a.act_lo, a.act_hi));
                    conv_store1(yp + (size_t)ko * OHW, v);
                }
            }
        }
    }
}

#endif  // PVHIP_DIAG

// Is the (r,s)-major reduction order (conv_igemm_rs_kernel) used for this weight shape?
inline bool rs_major(int c, int kh, int kw) { return c % kBK == 0 && kh * kw < 64; }

__global__ __launch_bounds__(kBlock) void conv_pack_kernel(const float* __restrict__ w, int* __restrict__ ktab,
                                                            float* __restrict__ wp, int K, int C, int kh, int kw, int H,
                                                            int W, int kred, int kred_pad, int kout_pad, int rsmajor) {
    const size_t total  = (size_t)(kred_pad + kPanelSpare) * kout_pad;   // includes the spare zero stages
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const bool   mask   = kh * kw < 64;
    const int    tab_n  = kred_pad + kTabSpare;                  // spare stages of padding rows
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int kr = (int)(e / kout_pad);   // panel row
        const int ko = (int)(e % kout_pad);
        float     v  = 0.0f;
        if (kr < kred && ko < K) {
            // source reduction index in OIHW order is (c*kh + r)*kw + s
            int src = kr;
            if (rsmajor) {                    // panel row = (r*kw + s)*C + c
                const int rs = kr / C, c = kr - rs * C;
                src          = c * (kh * kw) + rs;
            }
            v = w[(size_t)ko * kred + src];
        }
        wp[e] = v;
    }
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)(2 * tab_n); e += stride) {
        const int i = (int)e;
        int       v = 0;
        if (rsmajor) {
            // table[rs] = byte offset of tap (r, s) inside one channel plane; zeros after the last tap
            if (i < kh * kw) v = ((i / kw) * W + (i % kw)) * 4;
        } else if (i < tab_n) {
            if (i < kred) {
                const int s = i % kw, t = i / kw, r = t % kh, c = t / kh;
                v = (c * H * W + r * W + s) * 4;
            }
        } else {
            const int kr = i - tab_n;
            v            = mask ? 63 : (0x7fff << 8);
            if (kr < kred) {
                const int s = kr % kw, r = (kr / kw) % kh;
                v           = mask ? (r * kw + s) : ((r << 8) | s);
            }
        }
        ktab[i] = v;
    }
}

inline int round_up_int(int v, int q) { return (v + q - 1) / q * q; }

// The LDS-DMA kernel serves every window of fewer than 64 taps; conv_igemm_kernel (register-staged, compare path) is the one general
// fallback for larger windows.  The diagnostic build also carries the predecessors: PVHIP_CONV_KERNEL=lds selects the register-staged
// conv_igemm_rs_kernel / conv_igemm_kernel<.., true> for A/B measurements (scripts/tune_conv.py, tests/diag_variants.py).
#ifdef PVHIP_DIAG
inline bool dma_enabled() { return settings().conv_kernel != 1; }
#else
inline bool dma_enabled() { return true; }
#endif

template <int BM, int BN, int WAVES_M, int WAVES_N, bool kF16 = false>
void launch_conv(const ConvArgs& a, int n_ptiles) {
    if (BN == 128 && WAVES_M == 1 && dma_enabled() && (rs_major(a.C, a.kh, a.kw) || a.kh * a.kw < 64)) {
        const size_t dyn = (size_t)settings().conv_lds_pad_kb * 1024;     // tuning: extra dynamic LDS caps workgroups per CU
        const bool pw = rs_major(a.C, a.kh, a.kw) && a.kh == 1 && a.kw == 1 && a.sh == 1 && a.sw == 1 && a.pt == 0 && a.pl == 0 &&
                        a.OH == a.H && a.OW == a.W && (a.H * a.W) % 4 == 0 && !settings().conv_nopw;
        if (kF16) {           // (r,s)-major layers, and c-major ones (conv1: C = 3) -- through the padding pass without a window test
            if (pw) hipLaunchKernelGGL((conv_igemm_dma_kernel<BM, true, true, false, true>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), dyn, state().stream, a);
            else if (rs_major(a.C, a.kh, a.kw))
                hipLaunchKernelGGL((conv_igemm_dma_kernel<BM, true, false, false, true>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), dyn, state().stream, a);
            else if (a.pt == 0 && a.pl == 0 && (a.OH - 1) * a.sh + a.kh <= a.H && (a.OW - 1) * a.sw + a.kw <= a.W && !settings().conv_novalid)
                hipLaunchKernelGGL((conv_igemm_dma_kernel<BM, false, false, true, true>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), dyn, state().stream, a);
            else
                hipLaunchKernelGGL((conv_igemm_dma_kernel<BM, false, false, false, true>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), dyn, state().stream, a);
            return;
        }
        if (pw)
            hipLaunchKernelGGL((conv_igemm_dma_kernel<BM, true, true>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), dyn, state().stream, a);
        else if (rs_major(a.C, a.kh, a.kw))
            hipLaunchKernelGGL((conv_igemm_dma_kernel<BM, true>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), dyn, state().stream, a);
        else if (a.pt == 0 && a.pl == 0 && (a.OH - 1) * a.sh + a.kh <= a.H && (a.OW - 1) * a.sw + a.kw <= a.W && !settings().conv_novalid)
            hipLaunchKernelGGL((conv_igemm_dma_kernel<BM, false, false, true>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), dyn, state().stream, a);
        else
            hipLaunchKernelGGL((conv_igemm_dma_kernel<BM, false>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), dyn, state().stream, a);
    }
#ifdef PVHIP_DIAG
    else if (rs_major(a.C, a.kh, a.kw)) {
        const bool pointwise = a.kh == 1 && a.kw == 1 && a.sh == 1 && a.sw == 1 && a.pt == 0 && a.pl == 0 &&
                               a.OH == a.H && a.OW == a.W && (a.H * a.W) % 4 == 0 && settings().conv_pw16;   // 16-byte gather measured slower: opt-in
        const size_t dyn = (size_t)settings().conv_lds_pad_kb * 1024;
        if (pointwise)
            hipLaunchKernelGGL((conv_igemm_rs_kernel<BM, BN, WAVES_M, WAVES_N, true>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock),
                               dyn, state().stream, a);
        else
            hipLaunchKernelGGL((conv_igemm_rs_kernel<BM, BN, WAVES_M, WAVES_N, false>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock),
                               dyn, state().stream, a);
    }
    else if (a.kh * a.kw < 64)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, true>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), 0,
                           state().stream, a);
#endif
    else   // windows of 64 taps and more: the general fallback (the in-bounds test is a compare per element instead of a window-bit mask)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, false>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock),
                           0, state().stream, a);
}

}  // namespace

extern "C" {

size_t pvhip_conv2d_pack_elems(int k_out, int c, int kh, int kw) {
    if (k_out <= 0 || c <= 0 || kh <= 0 || kw <= 0) return 0;
    const size_t kred_pad = (size_t)round_up_int(c * kh * kw, kBK);
    const size_t kout_pad = (size_t)round_up_int(k_out, kKoutAlign);
    size_t elems = 2 * (kred_pad + kTabSpare) + (kred_pad + kPanelSpare) * kout_pad;   // two tables, then the weight panel (both with spare stages)
    if (kh == 3 && kw == 3) elems += wino_pack_elems(k_out, c) + wino4_pack_elems(k_out, c);
    if (kh == 1 && kw == 1) elems += pw_pack_elems(k_out, c);         // 1x1: the fragment-ordered panel of the pointwise kernel
    if (kh == 5 && kw == 5) elems += wino4_pack_elems(k_out, c);       // 5x5: the F(2x2,5x5) panel   // 3x3: the Winograd-transformed panels ride along (stride / pad are not known yet)
    return elems;
}

int pvhip_conv2d_pack_f32(const float* w_oihw, float* wpack, int k_out, int c, int kh, int kw, int h, int w) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(w_oihw != nullptr && wpack != nullptr);
    PVHIP_CHECK_ARG(k_out > 0 && c > 0 && kh > 0 && kw > 0 && h > 0 && w > 0);
    if (kh >= 256 || kw >= 256 || (unsigned long long)c * h * w >= (1ull << 29))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_pack_f32: C=%d kh=%d kw=%d H=%d W=%d outside table encoding", c, kh, kw, h, w);
    const int kred = c * kh * kw, kred_pad = round_up_int(kred, kBK), kout_pad = round_up_int(k_out, kKoutAlign);
    int*   ktab = reinterpret_cast<int*>(wpack);
    float* wp   = wpack + 2 * (kred_pad + kTabSpare);
    hipLaunchKernelGGL(conv_pack_kernel, dim3(grid_for((size_t)(kred_pad + kPanelSpare) * kout_pad)), dim3(kBlock), 0, state().stream,
                       w_oihw, ktab, wp, k_out, c, kh, kw, h, w, kred, kred_pad, kout_pad, rs_major(c, kh, kw) ? 1 : 0);
    if (kh == 1 && kw == 1 && pw_pack_elems(k_out, c) > 0) {
        const int rc = pw_pack(w_oihw, wp + (size_t)(kred_pad + kPanelSpare) * kout_pad, k_out, c);
        if (rc) return rc;
    }
    if (kh == 5 && kw == 5 && wino4_pack_elems(k_out, c) > 0) {          // used or not: wino25_eligible (extents, batch)
        const int rc = wino25_pack(w_oihw, wp + (size_t)(kred_pad + kPanelSpare) * kout_pad, k_out, c);
        if (rc) return rc;
    }
    if (kh == 3 && kw == 3 && wino_pack_elems(k_out, c) > 0) {
        float* const u2 = wp + (size_t)(kred_pad + kPanelSpare) * kout_pad;
        int rc = wino_pack(w_oihw, u2, k_out, c);
        if (rc) return rc;
        {                                     // the F(4x4, 3x3) panel behind it (whether it is used depends on extents and batch: wino4_eligible)
            rc = wino4_pack(w_oihw, u2 + wino_pack_elems(k_out, c), k_out, c);
            if (rc) return rc;
        }
    }
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

static int conv2d_impl(const float* x, const float* wpack, float* y, int n, int c, int h, int w, int k_out, int kh,
                       int kw, int oh, int ow, int sh, int sw, int pad_top, int pad_left, const float* bias, int relu,
                       int out_channel_offset, int out_channels_total, float act_lo, float act_hi, bool f16 = false, bool c8_out = false) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && c > 0 && h > 0 && w > 0 && k_out > 0 && kh > 0 && kw > 0 && oh >= 0 && ow >= 0);
    PVHIP_CHECK_ARG(sh > 0 && sw > 0 && pad_top >= 0 && pad_left >= 0);
    if (kh >= 256 || kw >= 256)
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f32: kh=%d kw=%d outside table encoding", kh, kw);
    PVHIP_CHECK_ARG(out_channels_total == 0 || (out_channel_offset >= 0 && out_channel_offset + k_out <= out_channels_total));
    const unsigned long long in_e = (unsigned long long)n * c * h * w,
                             out_e = (unsigned long long)n * (out_channels_total > 0 ? out_channels_total : k_out) * oh * ow;
    if (in_e >= (1ull << 29) || out_e >= (1ull << 31))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f32: input exceeds 2^29 elements (buffer offsets below 2^31) or output 2^31");
    if (out_e == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && wpack != nullptr && y != nullptr);

    ConvArgs a;
    a.kred_pad = round_up_int(c * kh * kw, kBK);
    a.kout_pad = round_up_int(k_out, kKoutAlign);
    a.x        = x;
    a.ktab     = reinterpret_cast<const int*>(wpack);
    a.wp       = wpack + 2 * (a.kred_pad + kTabSpare);
    a.y        = y;
    a.bias     = bias;
    a.N = n; a.C = c; a.H = h; a.W = w; a.K = k_out; a.OH = oh; a.OW = ow;
    a.sh = sh; a.sw = sw; a.pt = pad_top; a.pl = pad_left; a.kh = kh; a.kw = kw;
    a.x_bytes = (unsigned)(in_e * 4ull);
    a.P    = n * oh * ow;
    a.relu = relu;
    a.act_lo = act_lo;
    a.act_hi = act_hi;
    a.y_ctotal = out_channels_total > 0 ? out_channels_total : k_out;
    a.y_coff   = out_channels_total > 0 ? out_channel_offset : 0;
    a.nseg     = 0;
    if (c8_out) {        // pvhip_conv2d_f16_dma_c8: the output as fp16 blocked by eight channels (one range: the whole panel)
        PVHIP_CHECK_ARG(f16 && out_channels_total == 0 && (relu == 0 || relu == 1));
        a.nseg = 1;
        a.seg[0].y = y; a.seg[0].m_begin = 0; a.seg[0].k = k_out; a.seg[0].ctotal = k_out; a.seg[0].coff = 0; a.seg[0].layout = 1;
    }

    a.wp_bytes = (unsigned)((size_t)(a.kred_pad + kPanelSpare) * a.kout_pad * sizeof(float));

    if (f16) {      // pvhip_conv2d_f16_dma: every layer on the LDS-DMA kernel's f16 form (the matrix work is 16x cheaper: no Winograd, 64-channel tiles)
        if (!(rs_major(c, kh, kw) || kh * kw < 64) || !dma_enabled())
            return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_dma: C %% 16 == 0 or a window of fewer than 64 taps required (C=%d, %dx%d)", c, kh, kw);
        // the activation tile of a stage is re-read once per channel tile (through L2, which is what this form is bound by): wide tiles
        int bm = k_out > 64 ? 128 : (k_out > 32 ? 64 : 32);
        if (settings().f16_bm) bm = settings().f16_bm;          // PVHIP_CONV_F16_BM: tuning runs
        a.n_mtiles = (k_out + bm - 1) / bm;
        const int n_ptiles = (a.P + 127) / 128;
        if (bm == 128)     launch_conv<128, 128, 1, 4, true>(a, n_ptiles);
        else if (bm == 64) launch_conv<64, 128, 1, 4, true>(a, n_ptiles);
        else               launch_conv<32, 128, 1, 4, true>(a, n_ptiles);
        PVHIP_LAUNCH_CHECK();
        return PVHIP_OK;
    }
    // ---- 3x3 / stride 1 / same padding: Winograd F(2x2, 3x3), 2.25x fewer matrix-core operations (pvhip_wino.hip)
    if (pw_eligible(c, kh, kw, sh, sw, pad_top, pad_left, h, w, oh, ow)) {
        const PwDest d{y, 0, k_out, a.y_ctotal, a.y_coff};
        const int rc = pw_conv(x, a.wp + (size_t)(a.kred_pad + kPanelSpare) * a.kout_pad, n, c, h * w, k_out, bias, relu, act_lo, act_hi, 1, &d);
        if (rc) return rc;
        PVHIP_LAUNCH_CHECK();
        return PVHIP_OK;
    }
    if (wino25_eligible(c, kh, kw, sh, sw, pad_top, pad_left, h, w, oh, ow, n)) {
        const int rc = wino4_conv(2, x, a.wp + (size_t)(a.kred_pad + kPanelSpare) * a.kout_pad, y, n, c, h, w, k_out, bias, relu, act_lo,
                                  act_hi, a.y_coff, a.y_ctotal);
        if (rc) return rc;
        PVHIP_LAUNCH_CHECK();
        return PVHIP_OK;
    }
    if (wino4_eligible(c, kh, kw, sh, sw, pad_top, pad_left, h, w, oh, ow, n)) {
        const int rc = wino4_conv(4, x, a.wp + (size_t)(a.kred_pad + kPanelSpare) * a.kout_pad + wino_pack_elems(k_out, c), y, n, c, h, w,
                                  k_out, bias, relu, act_lo, act_hi, a.y_coff, a.y_ctotal);
        if (rc) return rc;
        PVHIP_LAUNCH_CHECK();
        return PVHIP_OK;
    }
    if (wino_eligible(c, kh, kw, sh, sw, pad_top, pad_left, h, w, oh, ow)) {
        const int rc = wino_conv(x, a.wp + (size_t)(a.kred_pad + kPanelSpare) * a.kout_pad, y, n, c, h, w, k_out, bias, relu, act_lo,
                                 act_hi, a.y_coff, a.y_ctotal);
        if (rc) return rc;
        PVHIP_LAUNCH_CHECK();
        return PVHIP_OK;
    }

#ifdef PVHIP_DIAG
    // ---- wave-direct kernel (PVHIP_CONV_KERNEL=wave, PVHIP_CONV_WTILE=TMxTN in units of 32)
    const size_t tab_bytes = (size_t)(a.kred_pad + kTabSpare) * sizeof(int2);
    if (settings().conv_kernel == 2 && tab_bytes <= 60 * 1024 && !rs_major(c, kh, kw)) {
        const int tm = settings().wtile_m, tn = settings().wtile_n;
        a.n_mtiles = (k_out + 32 * tm - 1) / (32 * tm);
        a.n_ptiles = (a.P + 32 * tn - 1) / (32 * tn);
        const long n_tiles = (long)a.n_mtiles * a.n_ptiles;
        const int  grid    = (int)((n_tiles + 3) / 4);
        const bool mask    = kh * kw < 64;
#define PV_WAVE_LAUNCH(TM_, TN_)                                                                                    \
    do {                                                                                                            \
        if (mask) hipLaunchKernelGGL((conv_wave_kernel<TM_, TN_, true>), dim3(grid), dim3(kBlock), tab_bytes, state().stream, a);  \
        else hipLaunchKernelGGL((conv_wave_kernel<TM_, TN_, false>), dim3(grid), dim3(kBlock), tab_bytes, state().stream, a);      \
    } while (0)
#ifdef PVHIP_DIAG
        if (const int v = settings().conv_ablate) {   // diagnostic build only (libpvhip_diag.so): results are wrong on purpose
            if (tm == 2 && tn == 1) {
                if (v == 1) hipLaunchKernelGGL((conv_wave_kernel<2, 1, true, 1>), dim3(grid), dim3(kBlock), tab_bytes, state().stream, a);
                else if (v == 2) hipLaunchKernelGGL((conv_wave_kernel<2, 1, true, 2>), dim3(grid), dim3(kBlock), tab_bytes, state().stream, a);
                else hipLaunchKernelGGL((conv_wave_kernel<2, 1, true, 3>), dim3(grid), dim3(kBlock), tab_bytes, state().stream, a);
            } else {
                if (v == 1) hipLaunchKernelGGL((conv_wave_kernel<2, 2, true, 1>), dim3(grid), dim3(kBlock), tab_bytes, state().stream, a);
                else if (v == 2) hipLaunchKernelGGL((conv_wave_kernel<2, 2, true, 2>), dim3(grid), dim3(kBlock), tab_bytes, state().stream, a);
                else hipLaunchKernelGGL((conv_wave_kernel<2, 2, true, 3>), dim3(grid), dim3(kBlock), tab_bytes, state().stream, a);
            }
            PVHIP_LAUNCH_CHECK();
            return PVHIP_OK;
        }
#endif
        if (tm == 1 && tn == 1) PV_WAVE_LAUNCH(1, 1);
        else if (tm == 1 && tn == 2) PV_WAVE_LAUNCH(1, 2);
        else if (tm == 2 && tn == 1) PV_WAVE_LAUNCH(2, 1);
        else if (tm == 2 && tn == 2) PV_WAVE_LAUNCH(2, 2);
        else if (tm == 4 && tn == 1) PV_WAVE_LAUNCH(4, 1);
        else if (tm == 1 && tn == 4) PV_WAVE_LAUNCH(1, 4);
        else return fail(PVHIP_EINVAL, "pvhip_conv2d_f32: unsupported PVHIP_CONV_WTILE %dx%d", tm, tn);
#undef PV_WAVE_LAUNCH
        PVHIP_LAUNCH_CHECK();
        return PVHIP_OK;
    }

#endif  // PVHIP_DIAG

    // ---- tile selection (calibrated with scripts/tune_conv.py on the GoogLeNet shapes at batch 256):
    // 128-pixel tiles; 64 output channels per tile when that wastes less than half a tile and still
    // leaves >= 4 workgroups per CU, else 32.  PVHIP_CONV_TILE=BMxBN overrides (tuning runs only).
    int bm = (k_out % 64 == 0 || k_out % 64 > 32) ? 64 : 32, bn = 128;
    if (bm == 64 && (long)((a.P + 127) / 128) * ((k_out + 63) / 64) < 4 * kNumCU) bm = 32;
    if (kh == 1 && kw == 1) bm = 32;      // 1x1 layers: the smaller tile wins on every GoogLeNet shape (more workgroups per CU)
#ifdef PVHIP_DIAG
    if (settings().tile_bm > 0) { bm = settings().tile_bm; bn = settings().tile_bn; }      // PVHIP_CONV_TILE: tuning runs only
#endif
    a.n_mtiles       = (k_out + bm - 1) / bm;
    const int n_ptiles = (a.P + bn - 1) / bn;

#ifdef PVHIP_DIAG
    if (bm == 128 && bn == 256) launch_conv<128, 256, 2, 2>(a, n_ptiles);
    else if (bm == 128 && bn == 128 && dma_enabled()) launch_conv<128, 128, 1, 4>(a, n_ptiles);
    else if (bm == 128 && bn == 128) launch_conv<128, 128, 2, 2>(a, n_ptiles);
    else if (bm == 64 && bn == 256) launch_conv<64, 256, 1, 4>(a, n_ptiles);
    else if (bm == 32 && bn == 256) launch_conv<32, 256, 1, 4>(a, n_ptiles);
    else
#endif
    if (bm == 64) launch_conv<64, 128, 1, 4>(a, n_ptiles);
    else launch_conv<32, 128, 1, 4>(a, n_ptiles);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_conv2d_f32(const float* x, const float* wpack, float* y, int n, int c, int h, int w, int k_out, int kh, int kw,
                     int oh, int ow, int sh, int sw, int pad_top, int pad_left, const float* bias, int relu,
                     int out_channel_offset, int out_channels_total, float act_lo, float act_hi) {
    return conv2d_impl(x, wpack, y, n, c, h, w, k_out, kh, kw, oh, ow, sh, sw, pad_top, pad_left, bias, relu, out_channel_offset,
                       out_channels_total, act_lo, act_hi);
}

int pvhip_conv2d_f16_dma_supported(int c, int kh, int kw) { return (c > 0 && kh > 0 && kw > 0 && (rs_major(c, kh, kw) || kh * kw < 64) && dma_enabled()) ? 1 : 0; }

int pvhip_conv2d_f16_dma(const float* x, const float* wpack, float* y, int n, int c, int h, int w, int k_out, int kh, int kw,
                         int oh, int ow, int sh, int sw, int pad_top, int pad_left, const float* bias, int relu,
                         int out_channel_offset, int out_channels_total, float act_lo, float act_hi) {
    return conv2d_impl(x, wpack, y, n, c, h, w, k_out, kh, kw, oh, ow, sh, sw, pad_top, pad_left, bias, relu, out_channel_offset,
                       out_channels_total, act_lo, act_hi, true);
}

int pvhip_conv2d_f16_dma_c8(const float* x, const float* wpack, void* yb, int n, int c, int h, int w, int k_out, int kh, int kw,
                            int oh, int ow, int sh, int sw, int pad_top, int pad_left, const float* bias, int act) {
    return conv2d_impl(x, wpack, static_cast<float*>(yb), n, c, h, w, k_out, kh, kw, oh, ow, sh, sw, pad_top, pad_left, bias, act, 0, 0, 0.0f, 0.0f,
                       true, true);
}

int pvhip_conv2d_kernel_kind(int n, int c, int h, int w, int k_out, int kh, int kw, int oh, int ow, int sh, int sw, int pad_top, int pad_left) {
    // (the row-span kernel is an entry of its own, pvhip_conv2d_stem_f32: the caller pads the image to the row length it asks for)
    if (settings().conv_stem && pvhip_conv2d_stem_f32_supported(c, h, w, k_out, kh, kw, sh, sw, pad_top, pad_left, oh, ow) > 0 &&
        (unsigned long long)n * k_out * oh * ow * 4ull < (1ull << 31) && (unsigned long long)n * c * (h + 6) * 256ull * 4ull < (1ull << 31))
        return (settings().conv_stem_wino && pvhip_conv2d_stem_wino_supported(c, h, w, k_out, kh, kw, sh, sw, pad_top, pad_left, oh, ow) > 0)
                   ? PVHIP_CONV_KIND_STEM_WINO : PVHIP_CONV_KIND_STEM;
    if (pw_eligible(c, kh, kw, sh, sw, pad_top, pad_left, h, w, oh, ow)) return PVHIP_CONV_KIND_POINTWISE;
    if (wino25_eligible(c, kh, kw, sh, sw, pad_top, pad_left, h, w, oh, ow, n)) return PVHIP_CONV_KIND_WINO_F2_5X5;
    if (wino4_eligible(c, kh, kw, sh, sw, pad_top, pad_left, h, w, oh, ow, n)) return PVHIP_CONV_KIND_WINO_F4_3X3;
    if (wino_eligible(c, kh, kw, sh, sw, pad_top, pad_left, h, w, oh, ow)) return PVHIP_CONV_KIND_WINO_F2_3X3;
    return PVHIP_CONV_KIND_IGEMM;
}

int pvhip_conv2d_multi_supported(int c, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int n_dest) {
    return (n_dest >= 1 && n_dest <= kMaxConvDests && kh == 1 && kw == 1 && sh == 1 && sw == 1 && pad_top == 0 && pad_left == 0 &&
            rs_major(c, kh, kw)) ? 1 : 0;
}

static int conv2d_multi_impl(const float* x, const float* wpack, int n, int c, int h, int w, int kh, int kw, int oh, int ow, int sh,
                             int sw, int pad_top, int pad_left, const float* bias, int act, float act_lo, float act_hi, int n_dest,
                             const pvhip_conv_dest* dests, bool f16) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && c > 0 && h > 0 && w > 0 && oh >= 0 && ow >= 0 && dests != nullptr);
    if (!pvhip_conv2d_multi_supported(c, kh, kw, sh, sw, pad_top, pad_left, n_dest) || oh != h || ow != w)
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_multi_f32: only 1x1 / stride 1 / unpadded convolutions with C %% 16 == 0 and at most %d destinations",
                    kMaxConvDests);
    ConvArgs a;
    int k_panel = 0;
    unsigned long long out_max = 0;
    for (int i = 0; i < n_dest; ++i) {
        const pvhip_conv_dest& d = dests[i];
        PVHIP_CHECK_ARG(d.y != nullptr && d.k > 0);
        PVHIP_CHECK_ARG(d.channels_total == 0 || (d.channel_offset >= 0 && d.channel_offset + d.k <= d.channels_total));
        a.seg[i].y       = d.y;
        a.seg[i].m_begin = k_panel;
        a.seg[i].k       = d.k;
        a.seg[i].ctotal  = d.channels_total > 0 ? d.channels_total : d.k;
        a.seg[i].coff    = d.channels_total > 0 ? d.channel_offset : 0;
        a.seg[i].layout  = d.layout;
        if (d.layout != 0) {            // fp16, channels blocked by eight: the f16 form only, a tensor of its own, zeros survive the activation
            PVHIP_CHECK_ARG(d.layout == 1 && f16 && d.channels_total == 0 && (act == 0 || act == 1));
        }
        k_panel += round_up_int(d.k, 32);
        const unsigned long long oe = (unsigned long long)n * a.seg[i].ctotal * oh * ow;
        if (oe > out_max) out_max = oe;
    }
    const unsigned long long in_e = (unsigned long long)n * c * h * w;
    if (in_e >= (1ull << 29) || out_max >= (1ull << 31))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_multi_f32: input exceeds 2^29 elements or an output 2^31");
    if (in_e == 0 || oh == 0 || ow == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && wpack != nullptr);
    if (!f16 && pw_eligible(c, kh, kw, sh, sw, pad_top, pad_left, h, w, oh, ow)) {
        PwDest pd[kMaxConvDests];
        for (int i = 0; i < n_dest; ++i) pd[i] = PwDest{a.seg[i].y, a.seg[i].m_begin, a.seg[i].k, a.seg[i].ctotal, a.seg[i].coff};
        const int kred_pad = round_up_int(c, kBK), kout_pad = round_up_int(k_panel, kKoutAlign);
        const float* ap = wpack + 2 * (kred_pad + kTabSpare) + (size_t)(kred_pad + kPanelSpare) * kout_pad;
        const int rc = pw_conv(x, ap, n, c, h * w, k_panel, bias, act, act_lo, act_hi, n_dest, pd);
        if (rc) return rc;
        PVHIP_LAUNCH_CHECK();
        return PVHIP_OK;
    }
    a.nseg     = n_dest;
    a.kred_pad = round_up_int(c, kBK);
    a.kout_pad = round_up_int(k_panel, kKoutAlign);
    a.x = x;
    a.ktab = reinterpret_cast<const int*>(wpack);
    a.wp   = wpack + 2 * (a.kred_pad + kTabSpare);
    a.y    = dests[0].y;
    a.bias = bias;
    a.N = n; a.C = c; a.H = h; a.W = w; a.K = k_panel; a.OH = oh; a.OW = ow;
    a.sh = 1; a.sw = 1; a.pt = 0; a.pl = 0; a.kh = 1; a.kw = 1;
    a.x_bytes = (unsigned)(in_e * 4ull);
    a.P = n * oh * ow;
    a.relu = act; a.act_lo = act_lo; a.act_hi = act_hi;
    a.y_ctotal = a.seg[0].ctotal; a.y_coff = a.seg[0].coff;
    a.wp_bytes = (unsigned)((size_t)(a.kred_pad + kPanelSpare) * a.kout_pad * sizeof(float));
    int bm = settings().multi_bm;             // PVHIP_CONV_MULTI_BM: tuning runs only
    if (f16) bm = settings().f16_bm ? settings().f16_bm : (k_panel > 64 ? 128 : (k_panel > 32 ? 64 : 32));      // wide tiles: the input tile is re-read per channel tile
    a.n_mtiles = (k_panel + bm - 1) / bm;          // a 64-channel tile may straddle two ranges: the epilogue looks the range up per 32 channels
    const int n_ptiles = (a.P + 127) / 128;
    const bool pw = (h * w) % 4 == 0;
    if (f16) {           // FP16 IRs: the f16 form of the same kernel (pvhip_conv2d_f16_dma), one launch for the module's 1x1 convolutions
        const dim3 grid(a.n_mtiles * n_ptiles);
        if (bm == 128) {
            if (pw) hipLaunchKernelGGL((conv_igemm_dma_kernel<128, true, true, false, true>), grid, dim3(kBlock), 0, state().stream, a);
            else    hipLaunchKernelGGL((conv_igemm_dma_kernel<128, true, false, false, true>), grid, dim3(kBlock), 0, state().stream, a);
        } else if (bm == 64) {
            if (pw) hipLaunchKernelGGL((conv_igemm_dma_kernel<64, true, true, false, true>), grid, dim3(kBlock), 0, state().stream, a);
            else    hipLaunchKernelGGL((conv_igemm_dma_kernel<64, true, false, false, true>), grid, dim3(kBlock), 0, state().stream, a);
        } else {
            if (pw) hipLaunchKernelGGL((conv_igemm_dma_kernel<32, true, true, false, true>), grid, dim3(kBlock), 0, state().stream, a);
            else    hipLaunchKernelGGL((conv_igemm_dma_kernel<32, true, false, false, true>), grid, dim3(kBlock), 0, state().stream, a);
        }
        PVHIP_LAUNCH_CHECK();
        return PVHIP_OK;
    }
    if (bm == 128) {
        if (pw) hipLaunchKernelGGL((conv_igemm_dma_kernel<128, true, true>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), 0, state().stream, a);
        else    hipLaunchKernelGGL((conv_igemm_dma_kernel<128, true, false>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), 0, state().stream, a);
    } else if (bm == 64) {
        if (pw) hipLaunchKernelGGL((conv_igemm_dma_kernel<64, true, true>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), 0, state().stream, a);
        else    hipLaunchKernelGGL((conv_igemm_dma_kernel<64, true, false>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), 0, state().stream, a);
    } else {
        if (pw) hipLaunchKernelGGL((conv_igemm_dma_kernel<32, true, true>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), 0, state().stream, a);
        else    hipLaunchKernelGGL((conv_igemm_dma_kernel<32, true, false>), dim3(a.n_mtiles * n_ptiles), dim3(kBlock), 0, state().stream, a);
    }
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_conv2d_multi_f32(const float* x, const float* wpack, int n, int c, int h, int w, int kh, int kw, int oh, int ow, int sh,
                           int sw, int pad_top, int pad_left, const float* bias, int act, float act_lo, float act_hi, int n_dest,
                           const pvhip_conv_dest* dests) {
    return conv2d_multi_impl(x, wpack, n, c, h, w, kh, kw, oh, ow, sh, sw, pad_top, pad_left, bias, act, act_lo, act_hi, n_dest, dests, false);
}

int pvhip_conv2d_multi_f16_dma(const float* x, const float* wpack, int n, int c, int h, int w, int kh, int kw, int oh, int ow, int sh,
                               int sw, int pad_top, int pad_left, const float* bias, int act, float act_lo, float act_hi, int n_dest,
                               const pvhip_conv_dest* dests) {
    return conv2d_multi_impl(x, wpack, n, c, h, w, kh, kw, oh, ow, sh, sw, pad_top, pad_left, bias, act, act_lo, act_hi, n_dest, dests, true);
}

}  // extern "C"
