// FP16 IRs, second step of the fp16 tensors (the reference holds EVERY tensor of an FP16 IR in float16, common_def.py:13-17):
// convolutions that read fp16 blocked by eight channels ("c8", pvhip_f16c8.hip) and WRITE it -- the inception modules of GoogLeNet from
// the module input to the channel Concat without an fp32 tensor in between:
//   * several 1x1 convolutions of the same input as one launch (the 1x1 / 3x3_reduce / 5x5_reduce arms), each output-channel tile storing
//     into the tensor of its member: an fp16 c8 tensor of its own, a channel range of the module's c8 Concat buffer, or (fallback) fp32 NCHW;
//   * the 3x3 / 5x5 convolutions behind the reduce arms, writing their range of the Concat buffer;
//   * MaxPool 3x3 / stride 1 / pad 1 + pool_proj: the pooled operand is the maximum of the nine shifted LDS reads (v_pk_maximum3_f16: a NaN
//     wins, as for np.max; the zeros the copy wrote for cells outside the image are the zero padding MaxPool.py:53 pads with).
// Same roles as conv_f16_c8_kernel (a producer wave copies whole rows by LDS-DMA, four consumer waves own a 32-channel tile each), with
//   * LDS rows of 8 / 16 / 32 / 64 pixels (the smallest power of two that holds W + 2 pad): one copy instruction moves 8 / 4 / 2 / 1 rows,
//     a 14-wide stage is 2-3 instructions per channel block instead of 9;
//   * 1x1 windows: a stage is FOUR 16-channel steps (eight channel blocks), so that a barrier is worth 16 MFMAs;
//   * the MFMA with the WEIGHT fragment first: an accumulator is [channel][pixel], a lane holds four consecutive channels of its pixel
//     per register group -- 8 bytes of a c8 piece, which the other lane half completes to 16.
#include "pvhip_common.h"

using namespace pvhip;

namespace {

typedef float    floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

constexpr int kMMaxThreads = 512;   // four consumer waves + one to four producer waves (C8mArgs.nprod)
constexpr int kMMaxBuf  = 6;
constexpr int kMSteps   = 4;        // 16-channel steps per stage of a 1x1 window
constexpr unsigned kOob = 0x80000000u;

struct C8mSeg {
    void* y;
    int   m_begin, k;        // first panel row of the member, its real output channels
    int   layout;            // 0: fp32 NCHW; 1: fp16 c8
    int   ctotal, coff;      // channels of the tensor y points into and this member's first channel in it (c8: multiples of 8)
    int   nblk;              // c8: channel blocks this member writes (a tensor of its own: ceil16(k) / 8, zeros included; a range: k / 8)
};

struct C8mArgs {
    const _Float16* xb;      // [N][CB][H*W][8]
    const _Float16* wf;      // [n_mtiles][CB / 2][taps][tm][64 lanes][8 halves]
    const float*    bias;    // [Kp] or null
    int N, CB, H, W, Kp;
    int tm, n_mtiles, tiles_per_image, n_tiles;
    int R, rows, rows_pad, stride_sh;      // output rows per tile; LDS rows a stage needs / holds (a multiple of 64 >> stride_sh) per channel block; log2 of the LDS row length in pixels
    int n_ins, nbuf, stages;               // copy instructions per stage; stage buffers; stages per tile
    int reg_copy;                          // producers copy through registers (global load + ds_write) instead of LDS-DMA
    int tpw2;                              // consumers own two channel tiles and half the pixel blocks each (groups of three or four tiles)
    int abl;                               // diagnostic build (PVHIP_CONV_ABLATE bits; wrong results on purpose): 1 every consumer loads channel tile 0's weight fragments (one
                                           // set of lines per workgroup: how much is the L2 -> CU traffic of the weights?), 2 no weight loads at all, 4 no stores,
                                           // 16 no copies  (nothing inside the tap loop: a uniform branch there costs a vmcnt(0) per tap, lesson 51 -- an ablation of the
                                           // LDS reads that way made the whole launch 25 % slower and measured itself)
    int nprod;                             // producer waves: the copy instructions of a stage are dealt out to them (a wave issues one per ~200 cycles)
    unsigned x_bytes, wf_bytes;
    int   act;
    float lo, hi;
    int   nseg;
    C8mSeg seg[PVHIP_MAX_CONV_DESTS];
};

__device__ __forceinline__ void c8m_wait_vmcnt(int n) {     // until at most n of this wave's copies are in flight (n: wave-uniform)
#if defined(__HIP_DEVICE_COMPILE__)
    switch (n < 62 ? n : 62) {
#define PVM_CASE(k_) case k_: asm volatile("s_waitcnt vmcnt(" #k_ ")" ::: "memory"); break;
        PVM_CASE(1) PVM_CASE(2) PVM_CASE(3) PVM_CASE(4) PVM_CASE(5) PVM_CASE(6) PVM_CASE(7) PVM_CASE(8) PVM_CASE(9) PVM_CASE(10) PVM_CASE(11) PVM_CASE(12) PVM_CASE(13) PVM_CASE(14) PVM_CASE(15) PVM_CASE(16) PVM_CASE(17) PVM_CASE(18) PVM_CASE(19) PVM_CASE(20) PVM_CASE(21) PVM_CASE(22) PVM_CASE(23) PVM_CASE(24) PVM_CASE(25) PVM_CASE(26) PVM_CASE(27) PVM_CASE(28) PVM_CASE(29) PVM_CASE(30) PVM_CASE(31) PVM_CASE(32) PVM_CASE(33) PVM_CASE(34) PVM_CASE(35) PVM_CASE(36) PVM_CASE(37) PVM_CASE(38) PVM_CASE(39) PVM_CASE(40) PVM_CASE(41) PVM_CASE(42) PVM_CASE(43) PVM_CASE(44) PVM_CASE(45) PVM_CASE(46) PVM_CASE(47) PVM_CASE(48) PVM_CASE(49) PVM_CASE(50) PVM_CASE(51) PVM_CASE(52) PVM_CASE(53) PVM_CASE(54) PVM_CASE(55) PVM_CASE(56) PVM_CASE(57) PVM_CASE(58) PVM_CASE(59) PVM_CASE(60) PVM_CASE(61) PVM_CASE(62)
#undef PVM_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#endif
}

// element-wise maximum of three fp16 x 8 vectors, a NaN in any of them wins (np.max, MaxPool.py:70)
__device__ __forceinline__ half8 pk_max3_nan(half8 a, half8 b, half8 c) {
    uint4v x = __builtin_bit_cast(uint4v, a), y = __builtin_bit_cast(uint4v, b), z = __builtin_bit_cast(uint4v, c), d;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned r;
        asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(x[i]), "v"(y[i]), "v"(z[i]));
        d[i] = r;
    }
#else
    d = x; (void)y; (void)z;
#endif
    return __builtin_bit_cast(half8, d);
}

// The consumer side of conv_f16_c8m_kernel: TW channel tiles (tile0 ..) x NBW pixel blocks (blk0 ..) of the workgroup's tile.
template <int KS, int NBW, bool POOL, int TW>
__device__ __forceinline__ void c8m_consume(const C8mArgs& a, const char* const c8m_lds, const int lane, const int tile0, const int blk0, const int mt,
                                            const int img, const int oy0, const int npx) {
    constexpr int TAPSW = KS * KS;
    constexpr int TAPS  = KS == 1 ? kMSteps : TAPSW;
    constexpr int BLK   = KS == 1 ? 2 * kMSteps : 2;
    constexpr int RING  = KS == 1 ? kMSteps : KS;
    const int l31 = lane & 31, lh = lane >> 5;
    const int HW     = a.H * a.W;
    const int ncs16  = a.CB >> 1;
    const int stride = 1 << a.stride_sh;
    const unsigned blk_bytes = (unsigned)(a.rows_pad * stride) * 16u;
    const unsigned buf_bytes = blk_bytes * BLK;
    const int S = a.stages;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.wf), 0, a.wf_bytes, 0x00020000);

    unsigned pixoff[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int p  = 32 * (blk0 + nb) + l31;
        const int pc = p < a.R * a.W ? p : 0;
        const int pr = pc / a.W, px = pc - pr * a.W;
        pixoff[nb] = (unsigned)lh * blk_bytes + (unsigned)((pr * stride + px) * 16);
    }
    floatx16 acc[TW][NBW];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][nb][r] = 0.0f;

#ifdef PVHIP_DIAG
    const int abl = a.abl;
#else
    constexpr int abl = 0;
#endif
    unsigned wlane[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) wlane[i] = (abl & 2) ? kOob : (unsigned)lane * 16u + ((abl & 1) ? 0u : (unsigned)(tile0 + i) * 1024u);
    // fragment of MFMA step t of stage cs: window taps (KS > 1): 16-channel step cs, tap t; 1x1: 16-channel step cs * 4 + t (past the
    // last one: an out-of-range offset, zeros -- the copy wrote zeros for those channel blocks too)
#define PVM_LOAD_A(dst_, cs_, t_)                                                                                \
    {                                                                                                            \
        const int c16_ = KS == 1 ? (cs_) * kMSteps + (t_) : (cs_);                                               \
        const int tap_ = KS == 1 ? 0 : (t_);                                                                     \
        const unsigned so_ = (unsigned)((((mt * ncs16 + min(c16_, ncs16 - 1)) * TAPSW + tap_) * a.tm) * 1024);   \
        _Pragma("unroll") for (int i = 0; i < TW; ++i)                                                           \
            dst_[i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wr, c16_ < ncs16 ? wlane[i] : kOob, so_, 0)); \
    }
    half8 af[RING][TW];
#pragma unroll
    for (int s = 0; s < RING; ++s) PVM_LOAD_A(af[s], 0, s);
    unsigned sb = 0;
    for (int cs = 0; cs < S; ++cs) {
        asm volatile("s_barrier" ::: "memory");
        const char* const buf = c8m_lds + sb * buf_bytes;
        sb = sb + 1 == (unsigned)a.nbuf ? 0u : sb + 1;
        const int cs_x = cs + 1 < S ? cs + 1 : cs;                       // the stage the ring's tail prefetches (the last stage: its own again, unused)
        half8 b8[2][NBW];
        // pixel operand of MFMA step t: window taps: channel blocks (0, 1), shifted by the tap; 1x1: channel blocks (2 t, 2 t + 1);
        // pooled: the maximum over the 3x3 window of those
#define PVM_READ_B(dst_, t_)                                                                                     \
    _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb) {                                                         \
        if (POOL) {                                                                                              \
            const char* const p0_ = buf + pixoff[nb] + (unsigned)(2 * (t_)) * blk_bytes;                         \
            half8 m_[3];                                                                                         \
            _Pragma("unroll") for (int g = 0; g < 3; ++g)                                                        \
                m_[g] = pk_max3_nan(*reinterpret_cast<const half8*>(p0_ + (unsigned)((g * stride + 0) * 16)),   \
                                    *reinterpret_cast<const half8*>(p0_ + (unsigned)((g * stride + 1) * 16)),   \
                                    *reinterpret_cast<const half8*>(p0_ + (unsigned)((g * stride + 2) * 16)));  \
            dst_[nb] = pk_max3_nan(m_[0], m_[1], m_[2]);                                                         \
        } else if (KS == 1) {                                                                                    \
            dst_[nb] = *reinterpret_cast<const half8*>(buf + pixoff[nb] + (unsigned)(2 * (t_)) * blk_bytes);     \
        } else {                                                                                                 \
            dst_[nb] = *reinterpret_cast<const half8*>(buf + pixoff[nb] + (unsigned)((((t_) / KS) * stride + ((t_) % KS)) * 16)); \
        }                                                                                                        \
    }
        if (!POOL) PVM_READ_B(b8[0], 0);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            // window taps / 1x1: the operands of step t + 1 are read in front of the MFMAs of step t; pooled: step by step (nine reads and
            // four maxima per operand: two steps of them in flight are registers the accumulators do not leave)
            if (POOL) { PVM_READ_B(b8[t & 1], t); }
            else if (t + 1 < TAPS) { PVM_READ_B(b8[(t + 1) & 1], t + 1); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) acc[i][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t % RING][i], b8[t & 1][nb], acc[i][nb], 0, 0, 0);
            if (t + RING < TAPS) PVM_LOAD_A(af[t % RING], cs, t + RING)
            else                 PVM_LOAD_A(af[t % RING], cs_x, t + RING - TAPS)
            __builtin_amdgcn_sched_barrier(0);
        }
#undef PVM_READ_B
    }
#undef PVM_LOAD_A

    // ---- epilogue: register 4 g + j of accumulator (i, nb) is panel row row0t + 8 g + 4 lh + j of tile pixel 32 (blk0 + nb) + l31
    const ActBounds ab = act_bounds(a.act, a.lo, a.hi);
    typedef const __attribute__((address_space(4))) float* const_float_p;
    const const_float_p bias_c = (const_float_p)(unsigned long)a.bias;
    // the member table is read through the kernarg pointer (indexed as an ordinary argument array it would be copied to scratch)
    typedef const __attribute__((address_space(4))) C8mArgs* kernarg_p;
    kernarg_p ka = (kernarg_p)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    const int P0 = oy0 * a.W + 32 * blk0 + l31;
    const int p0 = 32 * blk0 + l31;                                      // the lane's pixel of block 0 inside the tile
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        const int row0t = (mt * a.tm + tile0 + i) * 32;                  // first panel row of this channel tile
        int sg = 0;
        for (int q = 1; q < a.nseg; ++q) sg = row0t >= ka->seg[q].m_begin ? q : sg;
        const int m_rel = row0t - ka->seg[sg].m_begin;                   // multiple of 32
        const int kreal = ka->seg[sg].k;
        if (ka->seg[sg].layout == 1) {
            _Float16* const yh = static_cast<_Float16*>(ka->seg[sg].y);
            const int cbt = ka->seg[sg].ctotal >> 3, cb0 = (ka->seg[sg].coff + m_rel) >> 3, nblk = ka->seg[sg].nblk - (m_rel >> 3);
            // Two register groups at a time: group g0 = 2 gp holds channels 8 g0 + 4 lh .. + 3 of the lane's pixel, g1 = g0 + 1 the next block's.
            // v_permlane32_swap exchanges the upper lane half of one register with the lower half of another: afterwards a lane of the lower
            // half holds ALL EIGHT channels of block g0 and its partner in the upper half all eight of block g1 -- one 16-byte store per
            // lane instead of two 8-byte ones (a wave's stores leave at ~170 cycles apiece whatever their width: lesson 53).
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                float bs0[2][4], bs1[2][4];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bs0[h2][j] = bs1[h2][j] = -0.0f;
                        if (a.bias != nullptr) {
                            bs0[h2][j] = bias_c[min(row0t + 8 * (2 * gp + h2) + j, a.Kp - 1)];
                            bs1[h2][j] = bias_c[min(row0t + 8 * (2 * gp + h2) + 4 + j, a.Kp - 1)];
                        }
                    }
                const int  g_mine = 2 * gp + lh;
                const bool g_ok   = g_mine < nblk;
                _Float16* const yb = yh + (((size_t)img * cbt + cb0 + g_mine) * HW + P0) * 8;
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    unsigned w0[2], w1[2];                                // the lane's four channels of block g0 (w0) and g1 (w1) as two dwords each
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        half4 hv;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float v = acc[i][nb][4 * (2 * gp + h2) + j] + (lh ? bs1[h2][j] : bs0[h2][j]);
                            if (a.act != 0) v = (v < ab.lo) ? ab.lo : v;
                            if (a.act == 2) v = (v > ab.hi) ? ab.hi : v;
                            hv[j] = (_Float16)v;
                        }
                        const uint2 u = __builtin_bit_cast(uint2, hv);
                        if (h2 == 0) { w0[0] = u.x; w0[1] = u.y; } else { w1[0] = u.x; w1[1] = u.y; }
                    }
                    uint4v piece;
                    {
                        const auto s0 = __builtin_amdgcn_permlane32_swap(w0[0], w1[0], false, false);   // s0[0]: lower half own g0, upper half partner's g1; s0[1]: lower half partner's g0, upper half own g1
                        const auto s1 = __builtin_amdgcn_permlane32_swap(w0[1], w1[1], false, false);
                        piece[0] = s0[0]; piece[1] = s1[0]; piece[2] = s0[1]; piece[3] = s1[1];         // channels 0-3 (two dwords), then 4-7
                    }
                    if (g_ok && p0 + 32 * nb < npx && !(abl & 4)) *reinterpret_cast<uint4v*>(yb + (size_t)(32 * nb) * 8) = piece;
                }
            }
        } else {
            float* const yf = static_cast<float*>(ka->seg[sg].y) + ((size_t)img * ka->seg[sg].ctotal + ka->seg[sg].coff + m_rel) * HW + P0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                float bs0 = -0.0f, bs1 = -0.0f;
                if (a.bias != nullptr) {
                    bs0 = bias_c[min(row0t + dr, a.Kp - 1)];
                    bs1 = bias_c[min(row0t + dr + 4, a.Kp - 1)];
                }
                const int kk = m_rel + dr + 4 * lh;
                if (kk >= kreal) continue;
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    float v = acc[i][nb][r] + (lh ? bs1 : bs0);
                    if (a.act != 0) v = (v < ab.lo) ? ab.lo : v;
                    if (a.act == 2) v = (v > ab.hi) ? ab.hi : v;
                    if (p0 + 32 * nb < npx) conv_store1(yf + (size_t)(dr + 4 * lh) * HW + 32 * nb, v);
                }
            }
        }
    }
}

// KS: window (1, 3, 5); NB: 32-pixel blocks of a tile (2 / 4); POOL (KS == 1): MaxPool 3x3 / stride 1 / pad 1 in front of the 1x1 window
template <int KS, int NB, bool POOL>
__global__ __launch_bounds__(kMMaxThreads, (KS == 5 && NB == 4) ? 3 : 4) void conv_f16_c8m_kernel(C8mArgs a) {
    static_assert(!POOL || KS == 1, "the pooled form is a 1x1 convolution");
    constexpr int PADG  = POOL ? 1 : (KS - 1) / 2;                       // halo of the copied rows
    constexpr int BLK   = KS == 1 ? 2 * kMSteps : 2;                     // channel blocks per stage
    extern __shared__ __attribute__((aligned(1024))) char c8m_lds[];     // [nbuf][BLK][rows_pad][stride][16 bytes]

    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid  = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int HW     = a.H * a.W;
    const int stride = 1 << a.stride_sh;
    const unsigned blk_bytes = (unsigned)(a.rows_pad * stride) * 16u;
    const unsigned buf_bytes = blk_bytes * BLK;
    const int S = a.stages;

    // tile = (channel group, image, row tile), the channel groups of a pixel tile back to back; XCD-aware as in conv_f16_c8_kernel
    const int G = gridDim.x;
    int       tile;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = G >> 3, r = G & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt  = tile % a.n_mtiles, pt = tile / a.n_mtiles;
    const int img = pt / a.tiles_per_image;
    const int oy0 = (pt - img * a.tiles_per_image) * a.R;
    const int npx = min(a.R, a.H - oy0) * a.W;

    if (wid >= 4) {
        // ------------------------------------------------------------------ producers: copy instruction k of a stage (channel block k / q_n,
        // rows (k % q_n) * rpi ..) belongs to producer k % nprod
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.xb), 0, a.x_bytes, 0x00020000);
        const int  pw    = wid - 4;
        const int  rpi   = 64 >> a.stride_sh;                            // rows per copy instruction
        const int  lrow  = lane >> a.stride_sh;
        const int  x     = (lane & (stride - 1)) - PADG;
        const bool colok = x >= 0 && x < a.W;
        const int  iy0   = oy0 - PADG;
        const int  q_n   = a.rows_pad / rpi;
        const int  mine  = (a.n_ins - pw + a.nprod - 1) / a.nprod;       // this wave's copy instructions per stage
        int s_i = 0;
        auto issue_next = [&]() {
            char* const dst = c8m_lds + (unsigned)(s_i % a.nbuf) * buf_bytes;
            for (int k = pw; k < a.n_ins; k += a.nprod) {
#ifdef PVHIP_DIAG
                if (a.abl & 16) continue;
#endif
                const int  b    = k / q_n, q = k - b * q_n;
                const int  gb   = s_i * BLK + b;
                const bool bok  = gb < a.CB;
                const unsigned plane = (unsigned)((img * a.CB + (bok ? gb : 0)) * HW);
                const int  rr = q * rpi + lrow;
                const int  iy = iy0 + rr;
                const bool ok = bok && colok && rr < a.rows && iy >= 0 && iy < a.H;
                const unsigned vo = ok ? (plane + (unsigned)(iy * a.W + x)) * 16u : kOob;
                lds_dma_b128(xr, reinterpret_cast<float*>(dst + (unsigned)b * blk_bytes + (unsigned)q * 1024u), vo, 0u);
            }
            ++s_i;
        };
        if (a.reg_copy != 0) {
            // The other copy path (PVHIP_CONV_F16_C8_REG=1, up to eight copy instructions per wave and stage): global -> registers ->
            // ds_write_b128.  Loads of stage s + 1 fly while the consumers work on stage s, the writes go to the other buffer before
            // B(s + 1).  Ordinary loads leave a wave faster than LDS-DMA instructions do; they hold registers and lgkm slots instead.
            constexpr int MAXI = 8;
            uint4v regs[MAXI];
            auto load_stage = [&](int st) {
#pragma unroll
                for (int i = 0; i < MAXI; ++i) {
                    const int  k    = pw + i * a.nprod;
                    const int  kc   = k < a.n_ins ? k : 0;
                    const int  b    = kc / q_n, q = kc - b * q_n;
                    const int  gb   = st * BLK + b;
                    const bool bok  = gb < a.CB && k < a.n_ins;
                    const unsigned plane = (unsigned)((img * a.CB + (bok ? gb : 0)) * HW);
                    const int  rr = q * rpi + lrow;
                    const int  iy = iy0 + rr;
                    const bool ok = bok && colok && rr < a.rows && iy >= 0 && iy < a.H;
                    const unsigned vo = ok ? (plane + (unsigned)(iy * a.W + x)) * 16u : kOob;
                    regs[i] = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(xr, vo, 0u, 0));
                }
            };
            auto write_stage = [&](int st) {
                char* const dst = c8m_lds + (unsigned)(st % a.nbuf) * buf_bytes + (unsigned)lane * 16u;
#pragma unroll
                for (int i = 0; i < MAXI; ++i) {
                    const int k = pw + i * a.nprod;
                    if (k < a.n_ins) {
                        const int b = k / q_n, q = k - b * q_n;
                        *reinterpret_cast<uint4v*>(dst + (unsigned)b * blk_bytes + (unsigned)q * 1024u) = regs[i];
                    }
                }
            };
            load_stage(0);
            write_stage(0);
            for (int s = 0; s < S; ++s) {
                if (s + 1 < S) load_stage(s + 1);
                __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): this wave's LDS writes of stage s are done
                asm volatile("s_barrier" ::: "memory");              // B(s)
                if (s + 1 < S) write_stage(s + 1);
            }
            return;
        }
        while (s_i < S && s_i < a.nbuf) issue_next();
        for (int s = 0; s < S; ++s) {
            c8m_wait_vmcnt((s_i - s - 1) * mine);
            asm volatile("s_barrier" ::: "memory");
            if (s >= 1 && s_i < S) issue_next();
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    // One channel tile and NB pixel blocks per wave -- or (C8mArgs.tpw2: PVHIP_CONV_F16_C8_TPW=2, 1x1 windows) TWO channel tiles and NB / 2
    // pixel blocks: a pixel operand read from LDS then feeds two MFMAs, for a second weight fragment per step.  MEASURED, NOT KEPT: same
    // box, alternating, whole FP16 pass: the nine sibling launches 0.67 -> 0.71 ms (3b 0.109 -> 0.120), with the 3x3 / 5x5 layers in that
    // form too 0.887 -> 0.923 / 0.190 -> 0.200 -- half the LDS reads per MFMA buy nothing: the consumers' LDS reads are not what these
    // launches wait for (lesson 54).
    if (KS == 1 && !POOL && a.tpw2 != 0) {
        constexpr int NBH = NB / 2;
        const int tile0 = 2 * (wid & 1), blk0 = (wid >> 1) * NBH;
        int nt = 0;
        if (tile0 < a.tm && (mt * a.tm + tile0) * 32 < a.Kp) nt = (tile0 + 1 < a.tm && (mt * a.tm + tile0 + 1) * 32 < a.Kp) ? 2 : 1;
        if (nt == 2)      c8m_consume<KS, NBH, POOL, 2>(a, c8m_lds, lane, tile0, blk0, mt, img, oy0, npx);
        else if (nt == 1) c8m_consume<KS, NBH, POOL, 1>(a, c8m_lds, lane, tile0, blk0, mt, img, oy0, npx);
        else for (int s = 0; s < S; ++s) asm volatile("s_barrier" ::: "memory");
        return;
    }
    if (wid < a.tm && (mt * a.tm + wid) * 32 < a.Kp) c8m_consume<KS, NB, POOL, 1>(a, c8m_lds, lane, wid, 0, mt, img, oy0, npx);
    else for (int s = 0; s < S; ++s) asm volatile("s_barrier" ::: "memory");
}

// MaxPool 3x3 (any stride / padding, MaxPool.py:41-72) on fp16 c8 tensors: one lane = one (image, channel block, output pixel), nine
// 16-byte loads (cells of the zero padding: zeros; cells past the padded edge: not in the window), NaN wins, one 16-byte store.
__global__ __launch_bounds__(kBlock) void maxpool3x3_c8_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, int n, int cb, int h, int w,
                                                               int oh, int ow, int sh, int sw, int pt, int pl, int hp, int wp) {
    const size_t total = (size_t)n * cb * oh * ow;
    const half8 zero = half8{0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int    ox = (int)(e % ow);
        size_t       f  = e / ow;
        const int    oy = (int)(f % oh);
        const size_t plane = f / oh;                                  // image * cb + block
        const half8* const xp = reinterpret_cast<const half8*>(x) + plane * (size_t)(h * w);
        half8 rowm[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int py = oy * sh + r;                               // row in the padded image
            half8 v[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int px = ox * sw + s;
                const int iy = py - pt, ix = px - pl;
                const bool inwin = py < hp && px < wp;                // the window is clipped at the padded edge (MaxPool.py:63-66)
                const bool inimg = iy >= 0 && iy < h && ix >= 0 && ix < w;
                // a cell outside the window repeats the window's first cell (always inside: oy * sh < hp), a padding cell is zero
                const int cy = inwin ? iy : oy * sh - pt, cx = inwin ? ix : ox * sw - pl;
                const bool cimg = inwin ? inimg : (cy >= 0 && cy < h && cx >= 0 && cx < w);
                v[s] = cimg ? xp[(size_t)cy * w + cx] : zero;
            }
            rowm[r] = pk_max3_nan(v[0], v[1], v[2]);
        }
        reinterpret_cast<half8*>(y)[e] = pk_max3_nan(rowm[0], rowm[1], rowm[2]);
    }
}

// MaxPool 3x3 followed by LRN over five channels (MaxPool.py:41-72 then LRN.py:10-22; GoogLeNet's pool1/3x3_s2 -> pool1/norm1) on fp16 c8
// tensors, one launch: a lane owns one output pixel and walks the channel blocks -- nine 16-byte loads and the NaN-propagating maxima per
// block (the pooled tensor never exists), the pooled values of three consecutive blocks in registers as fp32, because the window of
// channel 8 b + q reaches two channels into the neighbour blocks; squares summed in ascending channel order, d^-beta as in lrn_div.
template <int BETA_MODE>
__global__ __launch_bounds__(kBlock) void maxpool3x3_lrn_c8_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, int n, int cb, int c, int h, int w,
                                                                   int oh, int ow, int sh, int sw, int pt, int pl, int hp, int wp, float alpha,
                                                                   float beta, float bias) {
    const size_t total = (size_t)n * oh * ow;
    const half8 zero = half8{0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(e % ow);
        const size_t f = e / ow;
        const int oy = (int)(f % oh), im = (int)(f / oh);
        // the nine cells of the window: offsets inside a plane, or -1 for a zero cell of the padding; a cell past the padded edge repeats the first
        int off[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
                const int py = oy * sh + r, px = ox * sw + s2;
                const bool inwin = py < hp && px < wp;
                const int cy = (inwin ? py : oy * sh) - pt, cx = (inwin ? px : ox * sw) - pl;
                off[3 * r + s2] = (cy >= 0 && cy < h && cx >= 0 && cx < w) ? cy * w + cx : -1;
            }
        const half8* const xi = reinterpret_cast<const half8*>(x) + (size_t)im * cb * (h * w);
        half8* const yo = reinterpret_cast<half8*>(y) + (size_t)im * cb * (oh * ow) + (size_t)oy * ow + ox;
        auto pooled = [&](int b, float (&dst)[8]) {
            if (b < 0 || b >= cb) {
#pragma unroll
                for (int q = 0; q < 8; ++q) dst[q] = 0.0f;
                return;
            }
            const half8* const xp = xi + (size_t)b * (h * w);
            half8 v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] = off[t] >= 0 ? xp[off[t]] : zero;
            const half8 m = pk_max3_nan(pk_max3_nan(v[0], v[1], v[2]), pk_max3_nan(v[3], v[4], v[5]), pk_max3_nan(v[6], v[7], v[8]));
#pragma unroll
            for (int q = 0; q < 8; ++q) dst[q] = (8 * b + q < c) ? (float)m[q] : 0.0f;
        };
        float prev[8], cur[8], nxt[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) prev[q] = 0.0f;
        pooled(0, cur);
        for (int b = 0; b < cb; ++b) {
            pooled(b + 1, nxt);
            float ext[12];                        // channels 8 b - 2 .. 8 b + 9
            ext[0] = prev[6]; ext[1] = prev[7];
#pragma unroll
            for (int q = 0; q < 8; ++q) ext[2 + q] = cur[q];
            ext[10] = nxt[0]; ext[11] = nxt[1];
            half8 o;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float s_ = ext[q] * ext[q];
#pragma unroll
                for (int t = 1; t < 5; ++t) s_ = s_ + ext[q + t] * ext[q + t];
                o[q] = (_Float16)lrn_div(ext[q + 2], bias + alpha * s_, beta, BETA_MODE);
            }
            yo[(size_t)b * (oh * ow)] = o;
#pragma unroll
            for (int q = 0; q < 8; ++q) { prev[q] = cur[q]; cur[q] = nxt[q]; }
        }
    }
}

// LRN over five channels followed by MaxPool 3x3 (LRN.py:10-22 then MaxPool.py:41-72; GoogLeNet's conv2/norm2 -> pool2/3x3_s2) on fp16 c8
// tensors, one launch: the LRN tensor never exists.  A workgroup owns one image and a band of pooled rows, i.e. the input rows those
// windows touch; a lane owns up to PPT pixels of the band and walks the channel blocks with the values of three consecutive blocks in
// registers as fp32 (the window of channel 8 b + q reaches two channels into the neighbour blocks; squares summed in ascending channel
// order), leaves the normalised block of its pixels in LDS as fp16 (16 bytes per pixel: the fp16 rounding the reference's float16 LRN
// tensor has too), and after a barrier the workgroup pools the block: nine 16-byte LDS reads per output (a cell of the padding: zeros; a
// cell past the padded edge repeats the first), a NaN wins.
template <int BETA_MODE, int PPT>
__global__ __launch_bounds__(kBlock) void lrn_maxpool3x3_c8_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, int n, int cb, int c, int h, int w,
                                                                   int oh, int ow, int sh, int sw, int pt, int pl, int hp, int wp, int R, int n_bands,
                                                                   float alpha, float beta, float bias) {
    extern __shared__ __attribute__((aligned(16))) char lp_lds[];
    half8* const tile = reinterpret_cast<half8*>(lp_lds);                // [band pixels]
    const half8 zero = half8{0, 0, 0, 0, 0, 0, 0, 0};
    const int tid  = threadIdx.x;
    const int img  = blockIdx.x / n_bands, band = blockIdx.x - img * n_bands;
    const int oy0  = band * R, oy1 = min(oh, oy0 + R);
    const int iy_lo = max(0, oy0 * sh - pt), iy_hi = min(h, (oy1 - 1) * sh + 3 - pt);
    const int band_px = (iy_hi - iy_lo) * w;
    const int hw = h * w, ohw = oh * ow;
    const half8* const xi = reinterpret_cast<const half8*>(x) + (size_t)img * cb * hw + (size_t)iy_lo * w;
    // ---- pooling geometry of this lane's output (one per lane: R * ow <= 256)
    const int  oyl = tid / ow, ox = tid - oyl * ow;
    const bool pact = oyl < oy1 - oy0;
    int off[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
            const int oy = oy0 + (pact ? oyl : 0), oxx = pact ? ox : 0;
            const int py = oy * sh + r, px = oxx * sw + s2;
            const bool inwin = py < hp && px < wp;
            const int cy = (inwin ? py : oy * sh) - pt, cx = (inwin ? px : oxx * sw) - pl;
            off[3 * r + s2] = (cy >= 0 && cy < h && cx >= 0 && cx < w) ? (cy - iy_lo) * w + cx : -1;
        }
    half8* const yo = reinterpret_cast<half8*>(y) + (size_t)img * cb * ohw + (size_t)(oy0 + (pact ? oyl : 0)) * ow + (pact ? ox : 0);

    float prev[PPT][8], cur[PPT][8], nxt[PPT][8];
    auto load_block = [&](int b, float (&dst)[PPT][8]) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int p = tid + i * kBlock;
            half8 v = zero;
            if (b < cb && p < band_px) v = xi[(size_t)b * hw + p];
#pragma unroll
            for (int q = 0; q < 8; ++q) dst[i][q] = (8 * b + q < c) ? (float)v[q] : 0.0f;
        }
    };
#pragma unroll
    for (int i = 0; i < PPT; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) prev[i][q] = 0.0f;
    load_block(0, cur);
    for (int b = 0; b < cb; ++b) {
        load_block(b + 1, nxt);
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int p = tid + i * kBlock;
            float ext[12];
            ext[0] = prev[i][6]; ext[1] = prev[i][7];
#pragma unroll
            for (int q = 0; q < 8; ++q) ext[2 + q] = cur[i][q];
            ext[10] = nxt[i][0]; ext[11] = nxt[i][1];
            half8 o;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float s_ = ext[q] * ext[q];
#pragma unroll
                for (int t = 1; t < 5; ++t) s_ = s_ + ext[q + t] * ext[q + t];
                o[q] = (_Float16)lrn_div(ext[q + 2], bias + alpha * s_, beta, BETA_MODE);
            }
            if (p < band_px) tile[p] = o;
        }
        __syncthreads();
        if (pact) {
            half8 v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] = off[t] >= 0 ? tile[off[t]] : zero;
            yo[(size_t)b * ohw] = pk_max3_nan(pk_max3_nan(v[0], v[1], v[2]), pk_max3_nan(v[3], v[4], v[5]), pk_max3_nan(v[6], v[7], v[8]));
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PPT; ++i)
#pragma unroll
            for (int q = 0; q < 8; ++q) { prev[i][q] = cur[i][q]; cur[i][q] = nxt[i][q]; }
    }
}

// AvgPool (AvgPool.py:41-59: the mean of in[y s : min(h - 1, y s + kh), x s : min(w - 1, x s + kw)], no padding -- GoogLeNet's 7x7 pool
// averages the top-left 6x6) on an fp16 c8 tensor; the output is fp32 NCHW (the classifier behind it is dense): one lane per (image,
// channel block, output pixel), a sequential fp32 sum in the window's row-major order, as avgpool2d_kernel.
__global__ __launch_bounds__(kBlock) void avgpool_c8_kernel(const _Float16* __restrict__ x, float* __restrict__ y, int n, int cb, int c, int h, int w, int oh,
                                                            int ow, int kh, int kw, int sh, int sw) {
    const size_t total = (size_t)n * cb * oh * ow;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(e % ow);
        size_t    f  = e / ow;
        const int oy = (int)(f % oh); f /= oh;
        const int b = (int)(f % cb), im = (int)(f / cb);
        const half8* const xp = reinterpret_cast<const half8*>(x) + ((size_t)im * cb + b) * (size_t)(h * w);
        const int y0 = oy * sh, x0 = ox * sw;
        const int y1 = min(y0 + kh, h - 1), x1 = min(x0 + kw, w - 1);
        float sum[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        int   cnt = 0;
        for (int iy = y0; iy < y1; ++iy)
            for (int ix = x0; ix < x1; ++ix) {
                const half8 v = xp[iy * w + ix];
#pragma unroll
                for (int q = 0; q < 8; ++q) sum[q] += (float)v[q];
                ++cnt;
            }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int ch = 8 * b + q;
            // (rounded to fp16: the reference's AvgPool of a float16 tensor returns float16 -- AvgPool.py:57-58, res dtype = input dtype; kept in an fp32 tensor)
            if (ch < c) y[(((size_t)im * c + ch) * oh + oy) * ow + ox] = cnt > 0 ? (float)(_Float16)(sum[q] / (float)cnt) : NAN;
        }
    }
}

inline int c8m_blocks(int c) { return (c + 15) / 16 * 2; }
inline int c8m_mtiles(int k) { return (k + 127) / 128; }
inline int c8m_tm(int k) { const int t32 = (k + 31) / 32, nm = c8m_mtiles(k); return (t32 + nm - 1) / nm; }

struct C8mTile { int R, tiles, rows, rows_pad, stride_sh, nb; };
inline bool c8m_tile(int h, int w, int padg, C8mTile& t) {
    if (h <= 0 || w <= 0 || w + 2 * padg > 64) return false;
    int sh = 3;
    while ((1 << sh) < w + 2 * padg) ++sh;
    t.stride_sh = sh;
    int r = 128 / w;
    if (r > h) r = h;
    if (r < 1) return false;
    t.tiles = (h + r - 1) / r;
    t.R     = (h + t.tiles - 1) / t.tiles;
    t.tiles = (h + t.R - 1) / t.R;
    t.rows  = t.R + 2 * padg;
    const int rpi = 64 >> sh;
    t.rows_pad = (t.rows + rpi - 1) / rpi * rpi;
    t.nb = t.R * w <= 64 ? 2 : 4;
    return true;
}

template <int KS, bool POOL>
void launch_c8m(const C8mArgs& a, int grid, int nb, size_t lds) {
    const int threads = 64 * (4 + a.nprod);
    if (nb == 2) hipLaunchKernelGGL((conv_f16_c8m_kernel<KS, 2, POOL>), dim3(grid), dim3(threads), lds, state().stream, a);
    else         hipLaunchKernelGGL((conv_f16_c8m_kernel<KS, 4, POOL>), dim3(grid), dim3(threads), lds, state().stream, a);
}

// MaxPool 3x3 -> LRN -> 1x1 convolution (+ bias, ReLU) on blocked fp16 tensors as ONE launch (round 5; the FP16-IR twin of
// maxpool3x3_lrn_conv1x1_kernel of pvhip_norm.hip: GoogLeNet's pool1 -> norm1 -> conv2/3x3_reduce).  maxpool3x3_lrn_c8_kernel holds, block
// after block, the eight normalised channels of ITS pixel as one 16-byte piece -- two B operands of v_mfma_f32_32x32x4_2b_f16, whose two blocks
// take four k-values per LANE (lane l: column l & 31 of block l >> 5): D[k][pixel] += W[k][4 c'..4 c' + 3] . lrn[4 c'..][pixel], fp32
// accumulation.  A = the weights rounded to fp16 once (the constants of an FP16 IR are fp16 values) and transposed into LDS as
// wl[c / 4][k][4].  Epilogue as conv_f16_c8m_kernel's: bias, ReLU, fp16, v_permlane32_swap between the lane halves, one 16-byte blocked
// piece per lane -- for the pixel of lane 32 b + (l & 31), whose output index comes over by a wave shuffle.
template <int BETA_MODE, int KT>
__global__ __launch_bounds__(kBlock) void maxpool3x3_lrn_conv1x1_c8_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, const float* __restrict__ cw,
                                                                           const float* __restrict__ cbias, int n, int cb, int c, int h, int w, int oh, int ow,
                                                                           int sh, int sw, int pt, int pl, int hp, int wp, float alpha, float beta, float bias,
                                                                           int k_out, int act) {
    typedef float floatx32 __attribute__((ext_vector_type(32)));
    constexpr int KP = 32 * KT;
    __shared__ __attribute__((aligned(16))) half4 wl[16][KP];                 // [input channel / 4][output channel]: cb <= 8 blocks = 16 quads
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    for (int e = tid; e < 2 * cb * KP; e += kBlock) {
        const int q4 = e / KP, k = e - q4 * KP;
        half4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (k < k_out && 4 * q4 + j < c) ? (_Float16)cw[(size_t)k * c + 4 * q4 + j] : (_Float16)0.0f;
        wl[q4][k] = v;
    }
    __syncthreads();
    const size_t total = (size_t)n * oh * ow;
    const half8 zero = half8{0, 0, 0, 0, 0, 0, 0, 0};
    const int ohw = oh * ow;
    const int kbt = (k_out + 15) / 16 * 2;                                    // blocks of the output tensor
    const int l31 = lane & 31, lh = lane >> 5;
    for (size_t e0 = (size_t)blockIdx.x * blockDim.x; e0 < total; e0 += (size_t)gridDim.x * blockDim.x) {       // (workgroup-uniform: every lane takes part in the MFMAs)
        const size_t e = e0 + tid;
        const bool live = e < total;
        const size_t ee = live ? e : 0;
        const int ox = (int)(ee % ow);
        const size_t f = ee / ow;
        const int oy = (int)(f % oh), im = (int)(f / oh);
        int off[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
                const int py = oy * sh + r, px = ox * sw + s2;
                const bool inwin = py < hp && px < wp;
                const int cy = (inwin ? py : oy * sh) - pt, cx = (inwin ? px : ox * sw) - pl;
                off[3 * r + s2] = (cy >= 0 && cy < h && cx >= 0 && cx < w) ? cy * w + cx : -1;
            }
        const half8* const xi = reinterpret_cast<const half8*>(x) + (size_t)im * cb * (h * w);
        auto pooled = [&](int b, float (&dst)[8]) {
            if (b < 0 || b >= cb) {
#pragma unroll
                for (int q = 0; q < 8; ++q) dst[q] = 0.0f;
                return;
            }
            const half8* const xp = xi + (size_t)b * (h * w);
            half8 v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] = off[t] >= 0 ? xp[off[t]] : zero;
            const half8 m = pk_max3_nan(pk_max3_nan(v[0], v[1], v[2]), pk_max3_nan(v[3], v[4], v[5]), pk_max3_nan(v[6], v[7], v[8]));
#pragma unroll
            for (int q = 0; q < 8; ++q) dst[q] = (8 * b + q < c) ? (float)m[q] : 0.0f;
        };
        floatx32 acc[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 32; ++r) acc[kt][r] = 0.0f;
        float prev[8], cur[8], nxt[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) prev[q] = 0.0f;
        pooled(0, cur);
        for (int b = 0; b < cb; ++b) {
            pooled(b + 1, nxt);
            float ext[12];                        // channels 8 b - 2 .. 8 b + 9
            ext[0] = prev[6]; ext[1] = prev[7];
#pragma unroll
            for (int q = 0; q < 8; ++q) ext[2 + q] = cur[q];
            ext[10] = nxt[0]; ext[11] = nxt[1];
            half4 o[2];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float s_ = ext[q] * ext[q];
#pragma unroll
                for (int t = 1; t < 5; ++t) s_ = s_ + ext[q + t] * ext[q + t];
                const _Float16 v = (_Float16)lrn_div(ext[q + 2], bias + alpha * s_, beta, BETA_MODE);      // (the fp16 value maxpool3x3_lrn_c8_kernel stores)
                o[q >> 2][q & 3] = live ? v : (_Float16)0.0f;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
                    acc[kt] = __builtin_amdgcn_mfma_f32_32x32x4f16(wl[2 * b + s2][32 * kt + l31], o[s2], acc[kt], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) { prev[q] = cur[q]; cur[q] = nxt[q]; }
        }
        // ---- epilogue: register 16 b2 + 4 g + j of acc[kt] at lane (l31, lh) = output channel 32 kt + 8 g + 4 lh + j of the pixel of lane 32 b2 + l31
        const long opix = live ? (long)im * kbt * ohw + (long)oy * ow + ox : -1;
        long opix_b[2];
        opix_b[0] = ((long)__shfl((int)(opix >> 32), l31, kWave) << 32) | (unsigned)__shfl((int)opix, l31, kWave);
        opix_b[1] = ((long)__shfl((int)(opix >> 32), 32 + l31, kWave) << 32) | (unsigned)__shfl((int)opix, 32 + l31, kWave);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                float bs[2][4];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int kk = 32 * kt + 8 * (2 * gp + h2) + 4 * lh + j;
                        bs[h2][j] = (cbias != nullptr && kk < k_out) ? cbias[kk] : 0.0f;
                    }
                const int blk = 4 * kt + 2 * gp + lh;            // the block this lane holds after the swap
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    unsigned w0[2], w1[2];
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        half4 hv;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float v = acc[kt][16 * b2 + 4 * (2 * gp + h2) + j] + bs[h2][j];
                            if (act != 0) v = __builtin_elementwise_maximum(v, 0.0f);
                            hv[j] = (_Float16)v;
                        }
                        const uint2 u = __builtin_bit_cast(uint2, hv);
                        if (h2 == 0) { w0[0] = u.x; w0[1] = u.y; } else { w1[0] = u.x; w1[1] = u.y; }
                    }
                    const auto s0 = __builtin_amdgcn_permlane32_swap(w0[0], w1[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(w0[1], w1[1], false, false);
                    uint4v piece;
                    piece[0] = s0[0]; piece[1] = s1[0]; piece[2] = s0[1]; piece[3] = s1[1];
                    if (opix_b[b2] >= 0 && blk < kbt)
                        *reinterpret_cast<uint4v*>(y + ((size_t)opix_b[b2] + (size_t)blk * ohw) * 8) = piece;
                }
            }
    }
}

}  // namespace

extern "C" {

int pvhip_conv2d_f16_c8_multi_supported(int c, int h, int w, int kh, int kw, int pool, int n_dest) {
    if (c <= 0 || kh != kw || (kh != 1 && kh != 3 && kh != 5) || n_dest < 1 || n_dest > PVHIP_MAX_CONV_DESTS) return 0;
    if (pool && kh != 1) return 0;
    if (n_dest > 1 && kh != 1) return 0;
    C8mTile t;
    return c8m_tile(h, w, pool ? 1 : (kh - 1) / 2, t) ? 1 : 0;
}

int pvhip_conv2d_f16_c8_multi(const void* xb, const float* wf, int n, int c, int h, int w, int kh, int kw, int pool,
                              const float* bias, int act, float act_lo, float act_hi, int n_dest, const pvhip_conv_dest* dests) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && dests != nullptr);
    if (!pvhip_conv2d_f16_c8_multi_supported(c, h, w, kh, kw, pool, n_dest))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_c8_multi: stride-1 \"same\" 1x1 (one or several members, optionally behind a 3x3 / 1 / 1 MaxPool) / 3x3 / 5x5 windows over short rows");
    C8mArgs a;
    int k_panel = 0;
    unsigned long long out_max = 0;
    for (int i = 0; i < n_dest; ++i) {
        const pvhip_conv_dest& d = dests[i];
        PVHIP_CHECK_ARG(d.y != nullptr && d.k > 0 && (d.layout == 0 || d.layout == 1));
        PVHIP_CHECK_ARG(d.channels_total == 0 || (d.channel_offset >= 0 && d.channel_offset + d.k <= d.channels_total));
        C8mSeg& sgm = a.seg[i];
        sgm.y = d.y; sgm.m_begin = k_panel; sgm.k = d.k; sgm.layout = d.layout;
        if (d.layout == 1) {
            PVHIP_CHECK_ARG(act == 0 || act == 1);                      // the zeros of the padding channels must survive the activation
            if (d.channels_total > 0) {
                PVHIP_CHECK_ARG(d.k % 8 == 0 && d.channel_offset % 8 == 0 && d.channels_total % 16 == 0);
                sgm.ctotal = d.channels_total; sgm.coff = d.channel_offset; sgm.nblk = d.k / 8;
            } else {
                sgm.ctotal = (d.k + 15) / 16 * 16; sgm.coff = 0; sgm.nblk = sgm.ctotal / 8;
            }
        } else {
            sgm.ctotal = d.channels_total > 0 ? d.channels_total : d.k;
            sgm.coff   = d.channels_total > 0 ? d.channel_offset : 0;
            sgm.nblk   = 0;
        }
        k_panel += (d.k + 31) / 32 * 32;
        const unsigned long long oe = (unsigned long long)n * sgm.ctotal * h * w;
        if (oe > out_max) out_max = oe;
    }
    const int cb = c8m_blocks(c);
    const unsigned long long in_b = (unsigned long long)n * cb * h * w * 16ull;
    if (in_b >= (1ull << 31) || out_max >= (1ull << 31)) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_c8_multi: input exceeds 2^31 bytes or an output 2^31 elements");
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(xb != nullptr && wf != nullptr);
    C8mTile t;
    c8m_tile(h, w, pool ? 1 : (kh - 1) / 2, t);
    a.xb = static_cast<const _Float16*>(xb); a.wf = reinterpret_cast<const _Float16*>(wf); a.bias = bias;
    a.N = n; a.CB = cb; a.H = h; a.W = w; a.Kp = k_panel;
    a.tm = c8m_tm(k_panel); a.n_mtiles = c8m_mtiles(k_panel);
    a.tiles_per_image = t.tiles;
    a.R = t.R; a.rows = t.rows; a.rows_pad = t.rows_pad; a.stride_sh = t.stride_sh;
    const int blk = kh == 1 ? 2 * kMSteps : 2;
    const int ncs16 = cb / 2;
    a.stages = kh == 1 ? (ncs16 + kMSteps - 1) / kMSteps : ncs16;
    a.n_ins  = blk * (t.rows_pad / (64 >> t.stride_sh));
    const size_t stage = (size_t)blk * t.rows_pad * (1 << t.stride_sh) * 16;
    int nbuf = (int)(52 * 1024 / stage);
    if (nbuf > kMMaxBuf) nbuf = kMMaxBuf;
    if (nbuf > a.stages) nbuf = a.stages;
    if (nbuf < 1) nbuf = 1;
    if (settings().f16_c8_reg != 0 && nbuf < 2) nbuf = 2;
    {
        const int knob = settings().f16_c8_prod;                                // PVHIP_CONV_F16_C8_PROD=1..4 (tuning runs); default: ~3 copy instructions per wave and stage
        int np = knob > 0 ? knob : (a.n_ins + 2) / 3;                           // (GoogLeNet at batch 256, 1x1 family: one producer 1.51 ms, by sixes 1.08, four 0.98)
        a.nprod = np < 1 ? 1 : (np > 4 ? 4 : np);
    }
    const int per_wave = (a.n_ins + a.nprod - 1) / a.nprod;
    a.reg_copy = (settings().f16_c8_reg != 0 && per_wave <= 8) ? 1 : 0;
    if ((nbuf - 1) * per_wave > 62) nbuf = 62 / per_wave + 1;
    if (stage * nbuf > 64 * 1024) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_c8_multi: a stage of %zu bytes does not fit", stage);
    a.nbuf = nbuf;
    a.x_bytes  = (unsigned)in_b;
    a.wf_bytes = (unsigned)(pvhip_conv2d_f16_c8_pack_elems(k_panel, c, kh, kw) * 4);
    a.act = act; a.lo = act_lo; a.hi = act_hi;
    a.nseg = n_dest;
    {
        const int knob = settings().f16_c8_tpw;                                 // PVHIP_CONV_F16_C8_TPW=2: two channel tiles per consumer wave for 1x1 windows (measured slower: off)
        a.tpw2 = (kh == 1 && knob == 2 && a.tm >= 2) ? 1 : 0;
    }
    a.abl = 0;
#ifdef PVHIP_DIAG
    a.abl = settings().conv_ablate;
#endif
    const long tiles = (long)n * t.tiles * a.n_mtiles;
    if (tiles > 0x3fffffffL) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_c8_multi: too many tiles");
    a.n_tiles = (int)tiles;
    const size_t lds = stage * nbuf;
    if (pool)         launch_c8m<1, true>(a, (int)tiles, t.nb, lds);
    else if (kh == 1) launch_c8m<1, false>(a, (int)tiles, t.nb, lds);
    else if (kh == 3) launch_c8m<3, false>(a, (int)tiles, t.nb, lds);
    else              launch_c8m<5, false>(a, (int)tiles, t.nb, lds);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_maxpool3x3_c8(const void* x, void* y, int n, int c, int h, int w, int oh, int ow, int sh, int sw,
                        int pad_top, int pad_left, int pad_bottom, int pad_right) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && c > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && sh > 0 && sw > 0 && pad_top >= 0 && pad_left >= 0 && pad_bottom >= 0 && pad_right >= 0);
    const int hp = h + pad_top + pad_bottom, wp = w + pad_left + pad_right;
    PVHIP_CHECK_ARG((oh - 1) * sh < hp && (ow - 1) * sw < wp);
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr);
    const int cb = c8m_blocks(c);
    const size_t total = (size_t)n * cb * oh * ow;
    hipLaunchKernelGGL(maxpool3x3_c8_kernel, dim3(grid_for(total)), dim3(kBlock), 0, state().stream, static_cast<const _Float16*>(x),
                       static_cast<_Float16*>(y), n, cb, h, w, oh, ow, sh, sw, pad_top, pad_left, hp, wp);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

/* ... and with the 1x1 / stride 1 / unpadded convolution behind the LRN in the same launch (ABI v15; the FP16-IR twin of pvhip_maxpool_lrn_conv1x1_f32):
 * x: c8 (n, c, h, w), c <= 64; w_oihw: the (k_out, c, 1, 1) fp32 weights (holding fp16 values), k_out <= 64; y: c8 (n, k_out, oh, ow); act none / ReLU. */
int pvhip_maxpool3x3_lrn_conv1x1_c8_supported(int c, int k_out, int size) { return (c > 0 && c <= 64 && k_out > 0 && k_out <= 64 && size == 5) ? 1 : 0; }

int pvhip_maxpool3x3_lrn_conv1x1_c8(const void* x, const float* w_oihw, void* y, int n, int c, int h, int w, int oh, int ow, int sh, int sw,
                                    int pad_top, int pad_left, int pad_bottom, int pad_right, int size, float alpha, float beta, float bias,
                                    int k_out, const float* conv_bias, int act) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && c > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && sh > 0 && sw > 0 && pad_top >= 0 && pad_left >= 0 && pad_bottom >= 0 && pad_right >= 0);
    PVHIP_CHECK_ARG(act == 0 || act == 1);
    if (!pvhip_maxpool3x3_lrn_conv1x1_c8_supported(c, k_out, size))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_maxpool3x3_lrn_conv1x1_c8: a window of five channels, at most 64 channels either side");
    const int hp = h + pad_top + pad_bottom, wp = w + pad_left + pad_right;
    PVHIP_CHECK_ARG((oh - 1) * sh < hp && (ow - 1) * sw < wp);
    if ((unsigned long long)n * 64 * oh * ow >= (1ull << 31)) return fail(PVHIP_EUNSUPPORTED, "pvhip_maxpool3x3_lrn_conv1x1_c8: tensor too large");
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr && w_oihw != nullptr);
    const int cb = c8m_blocks(c);
    const int bm = lrn_beta_mode(beta, bias);
    const size_t total = (size_t)n * oh * ow;
    const int kt = (k_out + 31) / 32;
#define PVM_PLC(BM_, KT_) hipLaunchKernelGGL((maxpool3x3_lrn_conv1x1_c8_kernel<BM_, KT_>), dim3(grid_for(total)), dim3(kBlock), 0, state().stream,                   \
                                             static_cast<const _Float16*>(x), static_cast<_Float16*>(y), w_oihw, conv_bias, n, cb, c, h, w, oh, ow, sh, sw, pad_top, \
                                             pad_left, hp, wp, alpha, beta, bias, k_out, act)
    if (bm == 4) { if (kt == 1) PVM_PLC(4, 1); else PVM_PLC(4, 2); }
    else if (bm == 1) { if (kt == 1) PVM_PLC(1, 1); else PVM_PLC(1, 2); }
    else { if (kt == 1) PVM_PLC(0, 1); else PVM_PLC(0, 2); }
#undef PVM_PLC
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_maxpool3x3_lrn_c8(const void* x, void* y, int n, int c, int h, int w, int oh, int ow, int sh, int sw,
                            int pad_top, int pad_left, int pad_bottom, int pad_right, int size, float alpha, float beta, float bias) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && c > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && sh > 0 && sw > 0 && pad_top >= 0 && pad_left >= 0 && pad_bottom >= 0 && pad_right >= 0);
    if (size != 5) return fail(PVHIP_EUNSUPPORTED, "pvhip_maxpool3x3_lrn_c8: a window of five channels");
    const int hp = h + pad_top + pad_bottom, wp = w + pad_left + pad_right;
    PVHIP_CHECK_ARG((oh - 1) * sh < hp && (ow - 1) * sw < wp);
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr);
    const int cb = c8m_blocks(c);
    const int bm = lrn_beta_mode(beta, bias);
    const size_t total = (size_t)n * oh * ow;
#define PVM_PL(BM_) hipLaunchKernelGGL((maxpool3x3_lrn_c8_kernel<BM_>), dim3(grid_for(total)), dim3(kBlock), 0, state().stream, static_cast<const _Float16*>(x), \
                                       static_cast<_Float16*>(y), n, cb, c, h, w, oh, ow, sh, sw, pad_top, pad_left, hp, wp, alpha, beta, bias)
    switch (bm) {
        case 4: PVM_PL(4); break;
        case 1: PVM_PL(1); break;
        case 2: PVM_PL(2); break;
        case 3: PVM_PL(3); break;
        default: PVM_PL(0); break;
    }
#undef PVM_PL
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_avgpool_c8(const void* x, float* y, int n, int c, int h, int w, int oh, int ow, int kh, int kw, int sh, int sw) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && c > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0);
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr);
    const int cb = c8m_blocks(c);
    hipLaunchKernelGGL(avgpool_c8_kernel, dim3(grid_for((size_t)n * cb * oh * ow)), dim3(kBlock), 0, state().stream, static_cast<const _Float16*>(x), y, n, cb, c,
                       h, w, oh, ow, kh, kw, sh, sw);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_lrn_maxpool3x3_c8_supported(int h, int w, int oh, int ow, int sh, int sw, int pad_top, int pad_left, int size) {
    if (size != 5 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || sh <= 0 || sw <= 0 || ow > kBlock) return 0;
    int R = kBlock / ow;                                     // one pooled output per lane
    if (R > oh) R = oh;
    while (R > 1 && ((R - 1) * sh + 3) * w > 4 * kBlock) --R;        // and at most four band pixels per lane
    return (((R - 1) * sh + 3) * w <= 4 * kBlock) ? R : 0;
}

int pvhip_lrn_maxpool3x3_c8(const void* x, void* y, int n, int c, int h, int w, int size, float alpha, float beta, float bias,
                            int oh, int ow, int sh, int sw, int pad_top, int pad_left, int pad_bottom, int pad_right) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && c > 0 && pad_top >= 0 && pad_left >= 0 && pad_bottom >= 0 && pad_right >= 0);
    const int R = pvhip_lrn_maxpool3x3_c8_supported(h, w, oh, ow, sh, sw, pad_top, pad_left, size);
    if (R == 0) return fail(PVHIP_EUNSUPPORTED, "pvhip_lrn_maxpool3x3_c8: a window of five channels, pooled rows of at most %d outputs, bands of at most %d pixels", kBlock, 4 * kBlock);
    const int hp = h + pad_top + pad_bottom, wp = w + pad_left + pad_right;
    PVHIP_CHECK_ARG((oh - 1) * sh < hp && (ow - 1) * sw < wp);
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr);
    const int cb = c8m_blocks(c);
    const int n_bands = (oh + R - 1) / R;
    const int rows_in = (R - 1) * sh + 3;
    const int ppt = (rows_in * w + kBlock - 1) / kBlock;
    const size_t lds = (size_t)rows_in * w * 16;
    const int bm = lrn_beta_mode(beta, bias) == 4 ? 4 : (lrn_beta_mode(beta, bias) == 1 ? 1 : 0);
    const dim3 grid((unsigned)(n * n_bands));
#define PVM_LP(BM_, PPT_) hipLaunchKernelGGL((lrn_maxpool3x3_c8_kernel<BM_, PPT_>), grid, dim3(kBlock), lds, state().stream, static_cast<const _Float16*>(x), \
                                             static_cast<_Float16*>(y), n, cb, c, h, w, oh, ow, sh, sw, pad_top, pad_left, hp, wp, R, n_bands, alpha, beta, bias)
#define PVM_LP_P(BM_) { if (ppt <= 1) PVM_LP(BM_, 1); else if (ppt <= 2) PVM_LP(BM_, 2); else PVM_LP(BM_, 4); }
    if (bm == 4) PVM_LP_P(4) else if (bm == 1) PVM_LP_P(1) else PVM_LP_P(0)
#undef PVM_LP_P
#undef PVM_LP
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // extern "C"
