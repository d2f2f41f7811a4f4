// The fourth f16 convolution kernel for FP16 IRs (Convolution.py:57-87 computed in numpy float16 by the reference,
// common_def.py:13-17): stride-1 "same" 3x3 / 5x5 (and 1x1) windows whose INPUT is an fp16 tensor in HBM with the channels blocked
// by eight -- [N][C/8][H*W][8 halves], "c8" -- which is what the producing 1x1 convolution of an inception module writes when this
// kernel is its only reader (pvhip_conv_dest.layout = 1).  The reference holds every tensor of an FP16 IR in float16; here the
// 3x3_reduce / 5x5_reduce tensors are the first that are stored so, and in the layout the f16 matrix cores want:
//   * the eight channels of a pixel are ONE 16-byte piece, which is the MFMA operand of a lane (v_mfma_f32_32x32x16_f16: lane
//     (pixel n, half h) supplies reduction rows 8 h .. 8 h + 7).  No conversion pass, no transposition, half the bytes;
//   * a workgroup owns R whole image rows (R * W <= 128 pixels) of one image and up to 128 output channels.  Per stage of 16 input
//     channels its PRODUCER wave copies the R + 2 pad input rows of the two channel blocks into LDS by LDS-DMA, one 1-KiB
//     instruction per row: lane l fetches image column l - pad, and a column or row outside the image is an out-of-range offset --
//     the hardware writes zeros, so the zero padding costs no test and no arithmetic; every tap is then a constant shift;
//   * the four CONSUMER waves own one 32-channel tile each: per tap one 16-byte weight fragment per lane straight from L2 (packed
//     once, the next tap row's in flight) and one ds_read_b128 + MFMA per 32-pixel block.
// Why a producer wave.  vmcnt retires in order: in pvhip_conv2d_f16_span every wave issued its share of the next stage's copies
// (HBM latency) and then its weight loads (L2 latency) -- and the first wait for a weight fragment was a wait for the copy in
// front of it.  Here the copies are alone in the producer's queue, three stage buffers deep, and the consumers' queues hold
// nothing but weight fragments; one s_barrier per stage joins them.
#include "pvhip_common.h"

using namespace pvhip;

namespace {

typedef float    floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int kC8Threads = 320;     // four consumer waves + the producer wave
constexpr int kC8MaxBuf  = 8;       // stage buffers: as many as fit (C8Args.nbuf)
constexpr int kC8MaxRows = 10;      // LDS rows per channel block and stage (R + 2 pad): 3 x 2 x 10 KB = 60 KB
constexpr unsigned kOob  = 0x80000000u;

struct C8Args {
    const _Float16* xb;      // [N][CB][H*W][8]
    const _Float16* wf;      // [n_mtiles][CB / 2][taps][tm][64 lanes][8 halves]  (the span kernel's fragments)
    float*          y;
    const float*    bias;
    int N, CB, H, W, K;
    int tm, n_mtiles, tiles_per_image, n_tiles;
    int R, rows;             // output rows per tile; LDS rows per channel block = R + 2 pad
    int nbuf;                // stage buffers in LDS (2 .. kC8MaxBuf)
    int store_vec;           // output pixels per store: 4 / 2 / 1 by the alignment of a tile's first pixel in its channel plane
    unsigned x_bytes, wf_bytes;
    int   act;
    float lo, hi;
    int   y_ctotal, y_coff;
};

__device__ __forceinline__ void c8_wait_vmcnt(int n) {      // until at most n of this wave's copies are in flight (n: wave-uniform, even)
#if defined(__HIP_DEVICE_COMPILE__)
    switch (n < 62 ? n : 62) {
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
        case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
        case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
        case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
        case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break;
        case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
        case 42: asm volatile("s_waitcnt vmcnt(42)" ::: "memory"); break;
        case 44: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
        case 46: asm volatile("s_waitcnt vmcnt(46)" ::: "memory"); break;
        case 48: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
        case 50: asm volatile("s_waitcnt vmcnt(50)" ::: "memory"); break;
        case 52: asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); break;
        case 54: asm volatile("s_waitcnt vmcnt(54)" ::: "memory"); break;
        case 56: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
        case 58: asm volatile("s_waitcnt vmcnt(58)" ::: "memory"); break;
        case 60: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
        case 62: asm volatile("s_waitcnt vmcnt(62)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // fewer in flight than allowed: correct, only slower
    }
#endif
}

#ifdef PVHIP_DIAG
// diagnostic build: cycle accounts of conv_f16_c8_kernel, summed over every 16th workgroup (scripts/stamps_f16_c8.py).
// consumer wave 0: 0 entry -> first barrier passed, 1 waiting at the later barriers, 2 stages (reads + MFMAs), 3 epilogue, 4 life, 5 workgroups
// producer: 8 issuing the first copies, 9 waiting for copies (vmcnt), 10 waiting at barriers, 11 life
__device__ unsigned long long g_c8_stamps[16];
#define PVC8_NOW() __builtin_readcyclecounter()
#define PVC8_STAMP(i_, v_) { if ((blockIdx.x & 15) == 3 && lane == 0) atomicAdd(&g_c8_stamps[i_], (unsigned long long)(v_)); }
#else
#define PVC8_NOW() 0ull
#define PVC8_STAMP(i_, v_) {}
#endif

// KS: window (1, 3, 5); NB: 32-pixel blocks of a tile (2: up to 64 pixels, 4: up to 128)
// PERSISTENT: the grid is as many workgroups as the chip holds and a workgroup walks tiles L, L + G, ...; the stages of its tiles are
// ONE sequence for the producer, which is two stages ahead across tile boundaries -- the first copies of a tile (HBM latency, and a
// tile is only C / 16 stages long) fly while the consumers finish and store the tile before.
template <int KS, int NB>
__global__ __launch_bounds__(kC8Threads, (KS == 5 && NB == 4) ? 3 : 4) void conv_f16_c8_kernel(C8Args a) {
    constexpr int TAPS = KS * KS, PAD = (KS - 1) / 2;
    extern __shared__ __attribute__((aligned(1024))) char c8_lds[];      // [nbuf][2 channel blocks][rows][64 pixels][16 bytes]

    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid  = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int l31 = lane & 31, lh = lane >> 5;
    const int HW   = a.H * a.W;
    const int ncs  = a.CB >> 1;
    const int rows = a.rows;
    const unsigned buf_bytes = 2u * (unsigned)rows * 1024u;

    // tiles = (channel group, image, row tile), the channel groups of a pixel tile back to back; the workgroups of one XCD
    // (blockIdx & 7) take neighbouring tiles, so that the groups of a pixel tile find it in that XCD's L2
    const int G = gridDim.x;
    int       L;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = G >> 3, r = G & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int my_tiles = L < a.n_tiles ? (a.n_tiles - L + G - 1) / G : 0;
    const int S = my_tiles * ncs;                                        // stages of this workgroup

    if (wid == 4) {
        // ------------------------------------------------------------------ producer: the copies of every stage
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.xb), 0, a.x_bytes, 0x00020000);
        const int  x     = lane - PAD;                                   // image column of this lane's piece
        const bool colok = x >= 0 && x < a.W;
        const int  n_ins = 2 * rows;
        int s_i = 0, it_i = 0, cs_i = 0, img_i = 0, iy0_i = 0;          // the next stage to copy
        auto issue_next = [&]() {
            if (cs_i == 0) {
                const int pt = (L + it_i * G) / a.n_mtiles;
                img_i = pt / a.tiles_per_image;
                iy0_i = (pt - img_i * a.tiles_per_image) * a.R - PAD;
            }
            char* const dst = c8_lds + (unsigned)(s_i % a.nbuf) * buf_bytes;
            for (int cb = 0; cb < 2; ++cb) {
                const unsigned plane = (unsigned)((img_i * a.CB + 2 * cs_i + cb) * HW);
                for (int rr = 0; rr < rows; ++rr) {
                    const int  iy = iy0_i + rr;
                    const bool ok = colok && iy >= 0 && iy < a.H;
                    const unsigned vo = ok ? (plane + (unsigned)(iy * a.W + x)) * 16u : kOob;
                    lds_dma_b128(xr, reinterpret_cast<float*>(dst + (cb * rows + rr) * 1024), vo, 0u);
                }
            }
            ++s_i;
            if (++cs_i == ncs) { cs_i = 0; ++it_i; }
        };
        // nbuf stage buffers: stage s + nbuf - 1 is copied as soon as the consumers have left stage s - 1 (B(s)) -- for a tile of up to nbuf
        // stages everything is in flight at once: the copies are HBM / Infinity-Cache latency, a stage is ~1200 cycles of MFMAs
        const int nbuf = a.nbuf;
        const unsigned long long p0 = PVC8_NOW();
        (void)p0;
        while (s_i < S && s_i < nbuf) issue_next();
        unsigned long long p1 = PVC8_NOW(), tw = 0ull, tb = 0ull;
        PVC8_STAMP(8, p1 - p0);
        for (int s = 0; s < S; ++s) {
            const unsigned long long q0 = PVC8_NOW();
            c8_wait_vmcnt((s_i - s - 1) * n_ins);                        // stage s has landed (the younger ones may fly)
            const unsigned long long q1 = PVC8_NOW();
            asm volatile("s_barrier" ::: "memory");                      // B(s): the consumers have also finished stage s - 1 ...
            const unsigned long long q2 = PVC8_NOW();
            tw += q1 - q0; tb += q2 - q1;
            if (s >= 1 && s_i < S) issue_next();                         // ... whose buffer is the one stage s - 1 + nbuf goes to
        }
        PVC8_STAMP(9, tw); PVC8_STAMP(10, tb); PVC8_STAMP(11, PVC8_NOW() - p0);
        (void)p1; (void)tw; (void)tb;
        return;
    }

    // ---------------------------------------------------------------------- consumers: one 32-channel tile each
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.wf), 0, a.wf_bytes, 0x00020000);
    unsigned pixoff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int p  = 32 * nb + l31;
        const int pc = p < a.R * a.W ? p : 0;
        const int pr = pc / a.W, px = pc - pr * a.W;
        pixoff[nb] = (unsigned)(((lh * rows + pr) * 64 + px) * 16);      // tap (0, 0) of the pixel: LDS row pr, column px (= image column px - pad)
    }
    const ActBounds ab = act_bounds(a.act, a.lo, a.hi);
    const unsigned wlane = (unsigned)lane * 16u + (unsigned)wid * 1024u;
#define PVC8_LOAD_A(dst_, vo_, mt_, cs_, tap_)                                                                   \
    dst_ = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wr, vo_, (unsigned)(((((mt_) * ncs + (cs_)) * TAPS + (tap_)) * a.tm) * 1024), 0))
#define PVC8_HAVE(mt_) (wid < a.tm && ((mt_) * a.tm + wid) * 32 < a.K)
    // weight fragments: a ring of RING taps.  Tap t of a stage sits in slot t % RING; when its MFMAs are issued the slot takes tap
    // t + RING (of the next stage / the next tile where that is past this one): RING - 1 taps of MFMAs cover the load
    constexpr int RING = KS;                          // a tap row ahead (a whole 3x3 stage ahead, 36 registers, measured the same)
    half8 af[RING];
#pragma unroll
    for (int s = 0; s < RING; ++s) af[s] = half8{0, 0, 0, 0, 0, 0, 0, 0};
    if (my_tiles > 0 && PVC8_HAVE(L % a.n_mtiles)) {
#pragma unroll
        for (int s = 0; s < RING; ++s) PVC8_LOAD_A(af[s], wlane, L % a.n_mtiles, 0, s);
    }
    unsigned sb = 0;                                                     // stage buffer of the next stage
    const unsigned long long c0 = PVC8_NOW();
    unsigned long long t_first = 0ull, t_bar = 0ull, t_stage = 0ull, t_epi = 0ull;
    (void)c0; (void)t_first; (void)t_bar; (void)t_stage; (void)t_epi;
    for (int it = 0; it < my_tiles; ++it) {
        const int tile = L + it * G;
        const int mt   = tile % a.n_mtiles, pt = tile / a.n_mtiles;
        const int img  = pt / a.tiles_per_image;
        const int oy0  = (pt - img * a.tiles_per_image) * a.R;
        const int npx  = min(a.R, a.H - oy0) * a.W;                      // output pixels of this tile (whole rows: plane index oy0 * W + p)
        const bool have = PVC8_HAVE(mt);
        const int  mt_n = (tile + G) % a.n_mtiles;                       // the next tile's channel group
        const bool have_n = it + 1 < my_tiles && PVC8_HAVE(mt_n);
        floatx16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
        if (!have) {                                                     // a wave without a channel tile in this group only keeps the barriers
            for (int cs = 0; cs < ncs; ++cs) {
                asm volatile("s_barrier" ::: "memory");
                sb = sb + 1 == (unsigned)a.nbuf ? 0u : sb + 1;
            }
            if (have_n) {
#pragma unroll
                for (int s = 0; s < RING; ++s) PVC8_LOAD_A(af[s], wlane, mt_n, 0, s);
            }
            continue;
        }
        // No branch inside a stage: with `if (have)` around the MFMAs and the loads of every tap hipcc's wait insertion met a join per tap
        // and put s_waitcnt vmcnt(0) in front of each tap's MFMAs -- every weight fragment waited for one tap after its issue.  The
        // prefetch across the tile boundary of a wave whose next group has no tile for it loads zeros through an out-of-range offset.
        const unsigned wl_n = have_n ? wlane : kOob;
        for (int cs = 0; cs < ncs; ++cs) {
            const unsigned long long s0 = PVC8_NOW();
            asm volatile("s_barrier" ::: "memory");                      // B(s): this stage is in its buffer
            const unsigned long long s1 = PVC8_NOW();
            if (it == 0 && cs == 0) t_first = s1 - c0; else t_bar += s1 - s0;
            const char* const buf = c8_lds + sb * buf_bytes;
            sb = sb + 1 == (unsigned)a.nbuf ? 0u : sb + 1;
            const bool     last  = cs + 1 == ncs;
            const unsigned wl_x  = last ? wl_n : wlane;                  // the fragments past this stage: the next stage's, or the next tile's first
            const int      mt_x  = last ? mt_n : mt, cs_x = last ? 0 : cs + 1;
            // pixel operands one tap ahead: the reads of tap t + 1 are issued in front of the MFMAs of tap t
            half8 b8[2][NB];
#define PVC8_READ_B(dst_, t_)                                                                                    \
    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                            \
        dst_[nb] = *reinterpret_cast<const half8*>(buf + pixoff[nb] + (unsigned)((((t_) / KS) * 64 + ((t_) % KS)) * 16));
            PVC8_READ_B(b8[0], 0);
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                if (t + 1 < TAPS) PVC8_READ_B(b8[(t + 1) & 1], t + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b8[t & 1][nb], af[t % RING], acc[nb], 0, 0, 0);
                if (t + RING < TAPS) PVC8_LOAD_A(af[t % RING], wlane, mt, cs, t + RING);
                else                 PVC8_LOAD_A(af[t % RING], wl_x, mt_x, cs_x, t + RING - TAPS);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef PVC8_READ_B
            t_stage += PVC8_NOW() - s1;
        }
        const unsigned long long e0 = PVC8_NOW();
        (void)e0;
        if (!have) continue;
        // ---- epilogue.  The MFMA's FIRST operand is the pixel fragment, so an accumulator is [pixel][channel]: lane (l31, lh) holds
        // channel (mt * tm + wid) * 32 + l31 and register 4 g + j is pixel 32 nb + 8 g + 4 lh + j of the tile -- four CONSECUTIVE
        // pixels of one channel plane per register group: one 16-byte store where the alignment allows (H * W and the tile's first
        // pixel multiples of four: the 56- and 28-wide layers), else two 8-byte or four 4-byte ones.  (With the channels in the
        // registers -- 64 dword stores per wave and tile -- the epilogue took 11 k of a workgroup's 27 k cycles: a wave's stores
        // leave at ~170 cycles apiece whatever their width.)
        const int  kch = (mt * a.tm + wid) * 32 + l31;
        const bool kok = kch < a.K;
        const float bv = (a.bias != nullptr && kok) ? a.bias[kch] : -0.0f;
        float* __restrict__ const yk = a.y + (((size_t)img * a.y_ctotal + a.y_coff + (kok ? kch : 0)) * HW + oy0 * a.W + 4 * lh);
        const int vec = a.store_vec;                                     // 4, 2 or 1 (wave-uniform)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = acc[nb][4 * g + j] + bv;
                    if (a.act != 0) v[j] = (v[j] < ab.lo) ? ab.lo : v[j];
                    if (a.act == 2) v[j] = (v[j] > ab.hi) ? ab.hi : v[j];
                }
                const int p = 32 * nb + 8 * g + 4 * lh;                  // the first of the four pixels
                if (!kok) continue;
                if (vec == 4) {
                    if (p < npx) conv_store4(yk + 32 * nb + 8 * g, v[0], v[1], v[2], v[3]);
                } else if (vec == 2) {
                    if (p < npx) conv_store2(yk + 32 * nb + 8 * g, v[0], v[1]);
                    if (p + 2 < npx) conv_store2(yk + 32 * nb + 8 * g + 2, v[2], v[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (p + j < npx) conv_store1(yk + 32 * nb + 8 * g + j, v[j]);
                }
            }
        }
        t_epi += PVC8_NOW() - e0;
    }
    if (wid == 0) {
        PVC8_STAMP(0, t_first); PVC8_STAMP(1, t_bar); PVC8_STAMP(2, t_stage); PVC8_STAMP(3, t_epi); PVC8_STAMP(4, PVC8_NOW() - c0); PVC8_STAMP(5, 1);
    }
#undef PVC8_LOAD_A
#undef PVC8_HAVE
}

// w (K, C, ks, ks) fp32 -> fp16 fragments [mt][cs][tap][i][lane][q]: channel (mt * TM + i) * 32 + lane % 32, input channel
// cs * 16 + 8 * (lane / 32) + q; output channels past K and input channels past C (C is padded to whole stages) are zero
__global__ __launch_bounds__(kBlock) void conv_f16_c8_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ wf, int K, int C,
                                                                  int ncs, int taps, int tm, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(e & 7), lane = (int)((e >> 3) & 63);
        size_t    f = e >> 9;                       // ((mt * ncs + cs) * taps + tap) * tm + i
        const int i = (int)(f % tm);   f /= tm;
        const int tap = (int)(f % taps); f /= taps;
        const int cs = (int)(f % ncs);
        const int mt = (int)(f / ncs);
        const int k = (mt * tm + i) * 32 + (lane & 31), c = cs * 16 + 8 * (lane >> 5) + q;
        wf[e] = (k < K && c < C) ? (_Float16)w[((size_t)k * C + c) * taps + tap] : (_Float16)0.0f;
    }
}

// NCHW fp32 <-> c8 fp16 (round to nearest even; channels past C are zeros): the boundary of the blocked layout (tests, and a reader
// that is not pvhip_conv2d_f16_c8)
__global__ __launch_bounds__(kBlock) void c8_from_f32_kernel(const float* __restrict__ x, _Float16* __restrict__ xb, int n, int c, int cb, int hw) {
    const size_t total = (size_t)n * cb * hw;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int    p = (int)(e % hw);
        const size_t f = e / hw;
        const int    b = (int)(f % cb), im = (int)(f / cb);
        half8 v;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int ch = 8 * b + q;
            v[q] = ch < c ? (_Float16)x[((size_t)im * c + ch) * hw + p] : (_Float16)0.0f;
        }
        reinterpret_cast<half8*>(xb)[e] = v;
    }
}
__global__ __launch_bounds__(kBlock) void c8_to_f32_kernel(const _Float16* __restrict__ xb, float* __restrict__ x, int n, int c, int cb, int hw) {
    const size_t total = (size_t)n * c * hw;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int    p = (int)(e % hw);
        const size_t f = e / hw;
        const int    ch = (int)(f % c), im = (int)(f / c);
        x[e] = (float)xb[(((size_t)im * cb + (ch >> 3)) * hw + p) * 8 + (ch & 7)];
    }
}

inline int c8_blocks(int c) { return (c + 15) / 16 * 2; }             // channel blocks of a c8 tensor: whole 16-channel stages
inline int c8_mtiles(int k) { return (k + 127) / 128; }
inline int c8_tm(int k) { const int t32 = (k + 31) / 32, nm = c8_mtiles(k); return (t32 + nm - 1) / nm; }

struct C8Tile { int R, tiles, rows, nb; };
inline bool c8_tile(int h, int w, int ks, C8Tile& t) {
    const int pad = (ks - 1) / 2;
    if (h <= 0 || w <= 0 || w + 2 * pad > 64) return false;           // lane l fetches column l - pad: the row and its padding in 64 lanes
    int r = 128 / w;
    if (r > kC8MaxRows - 2 * pad) r = kC8MaxRows - 2 * pad;
    if (r > h) r = h;
    if (r < 1) return false;
    t.tiles = (h + r - 1) / r;
    t.R     = (h + t.tiles - 1) / t.tiles;
    t.tiles = (h + t.R - 1) / t.R;
    t.rows  = t.R + 2 * pad;
    t.nb    = t.R * w <= 64 ? 2 : 4;
    return true;
}

template <int KS>
void launch_c8(const C8Args& a, int grid, int nb, size_t lds) {
    if (nb == 2) hipLaunchKernelGGL((conv_f16_c8_kernel<KS, 2>), dim3(grid), dim3(kC8Threads), lds, state().stream, a);
    else         hipLaunchKernelGGL((conv_f16_c8_kernel<KS, 4>), dim3(grid), dim3(kC8Threads), lds, state().stream, a);
}

}  // namespace

extern "C" {

size_t pvhip_c8_f16_elems(int n, int c, int h, int w) {                // FLOATS a c8 tensor of these logical dims occupies
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return 0;
    return (size_t)n * c8_blocks(c) * h * w * 4;                       // 8 halves = 4 floats per (block, pixel)
}

int pvhip_c8_f16_from_f32(const float* x, void* xb, int n, int c, int h, int w) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && c > 0 && h > 0 && w > 0);
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && xb != nullptr);
    const size_t total = (size_t)n * c8_blocks(c) * h * w;
    hipLaunchKernelGGL(c8_from_f32_kernel, dim3(grid_for(total)), dim3(kBlock), 0, state().stream, x, static_cast<_Float16*>(xb), n, c, c8_blocks(c), h * w);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_c8_f16_to_f32(const void* xb, float* x, int n, int c, int h, int w) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && c > 0 && h > 0 && w > 0);
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && xb != nullptr);
    hipLaunchKernelGGL(c8_to_f32_kernel, dim3(grid_for((size_t)n * c * h * w)), dim3(kBlock), 0, state().stream, static_cast<const _Float16*>(xb), x, n, c,
                       c8_blocks(c), h * w);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_conv2d_f16_c8_supported(int c, int h, int w, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int oh, int ow) {
    if (c <= 0 || kh != kw || (kh != 1 && kh != 3 && kh != 5) || sh != 1 || sw != 1) return 0;
    const int pad = (kh - 1) / 2;
    if (pad_top != pad || pad_left != pad || oh != h || ow != w) return 0;
    C8Tile t;
    return c8_tile(h, w, kh, t) ? 1 : 0;
}

size_t pvhip_conv2d_f16_c8_pack_elems(int k_out, int c, int kh, int kw) {          // FLOATS of the fragment panel
    if (k_out <= 0 || c <= 0 || kh <= 0 || kw <= 0) return 0;
    const size_t halves = (size_t)c8_mtiles(k_out) * (c8_blocks(c) / 2) * kh * kw * c8_tm(k_out) * 512;
    return (halves + 1) / 2;
}

int pvhip_conv2d_f16_c8_pack(const float* w_oihw, float* wf, int k_out, int c, int kh, int kw) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(w_oihw != nullptr && wf != nullptr && k_out > 0 && c > 0 && kh > 0 && kh == kw && kh <= 5);
    const int    ncs   = c8_blocks(c) / 2;
    const size_t total = (size_t)c8_mtiles(k_out) * ncs * kh * kw * c8_tm(k_out) * 512;
    hipLaunchKernelGGL(conv_f16_c8_pack_kernel, dim3(grid_for(total)), dim3(kBlock), 0, state().stream, w_oihw, reinterpret_cast<_Float16*>(wf), k_out, c,
                       ncs, kh * kw, c8_tm(k_out), total);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_conv2d_f16_c8(const void* xb, const float* wf, float* y, int n, int c, int h, int w, int k_out, int kh, int kw,
                        const float* bias, int act, int out_channel_offset, int out_channels_total, float act_lo, float act_hi) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && k_out > 0);
    const int pad = (kh - 1) / 2;
    if (!pvhip_conv2d_f16_c8_supported(c, h, w, kh, kw, 1, 1, pad, pad, h, w))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_c8: stride-1 \"same\" 1x1 / 3x3 / 5x5 windows over rows of at most %d pixels", 64 - 2 * pad);
    PVHIP_CHECK_ARG(out_channels_total == 0 || (out_channel_offset >= 0 && out_channel_offset + k_out <= out_channels_total));
    const int cb = c8_blocks(c);
    const unsigned long long in_b = (unsigned long long)n * cb * h * w * 16ull,
                             out_e = (unsigned long long)n * (out_channels_total > 0 ? out_channels_total : k_out) * h * w;
    if (in_b >= (1ull << 31) || out_e >= (1ull << 31)) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_c8: input exceeds 2^31 bytes or output 2^31 elements");
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(xb != nullptr && wf != nullptr && y != nullptr);
    C8Tile t;
    c8_tile(h, w, kh, t);
    C8Args a;
    a.xb = static_cast<const _Float16*>(xb); a.wf = reinterpret_cast<const _Float16*>(wf); a.y = y; a.bias = bias;
    a.N = n; a.CB = cb; a.H = h; a.W = w; a.K = k_out;
    a.tm = c8_tm(k_out);
    a.n_mtiles = c8_mtiles(k_out);
    a.tiles_per_image = t.tiles;
    a.R = t.R; a.rows = t.rows;
    a.x_bytes  = (unsigned)in_b;
    a.wf_bytes = (unsigned)(pvhip_conv2d_f16_c8_pack_elems(k_out, c, kh, kw) * 4);
    a.act = act; a.lo = act_lo; a.hi = act_hi;
    a.y_ctotal = out_channels_total > 0 ? out_channels_total : k_out;
    a.y_coff   = out_channels_total > 0 ? out_channel_offset : 0;
    const long tiles = (long)n * t.tiles * a.n_mtiles;
    if (tiles > 0x3fffffffL) return fail(PVHIP_EUNSUPPORTED, "pvhip_conv2d_f16_c8: too many tiles");
    a.n_tiles = (int)tiles;
    // Two grids.  ONE TILE PER WORKGROUP, as many of its stages in flight as LDS holds: the short tiles (5x5: C / 16 = 1-3 stages; conv2)
    // and the 7-wide layers.  PERSISTENT, two workgroups per CU walking tiles with the producer ahead across tile boundaries: 3x3 layers of
    // six or more stages (scripts/time_f16_c8.py: 3b 0.149 -> 0.131 ms, 4e 0.104 -> 0.087; conv2 and every 5x5 layer 1-40 % slower: a
    // consumer's stores count in vmcnt like its loads and retire in order, so a tile's first weight fragments wait for the stores of the
    // tile before -- a price that long tiles earn back, short ones do not).  PVHIP_CONV_F16_C8_WGS=n > 0: persistent with n per CU; -1: never.
    const int    knob  = settings().f16_c8_wgs;
    const int    ncs   = cb / 2;
    const int    wgs   = knob > 0 ? knob : ((knob == 0 && kh == 3 && ncs >= 6 && w >= 14) ? 2 : 0);      // 0: one tile per workgroup
    const size_t stage = (size_t)2 * t.rows * 1024;
    const size_t budget = stage > 13 * 1024 ? 72 * 1024 : 52 * 1024;           // two / three workgroups per CU
    int nbuf = (int)(budget / stage);
    if (nbuf > kC8MaxBuf) nbuf = kC8MaxBuf;
    if (wgs == 0 && nbuf > ncs) nbuf = ncs;
    if (nbuf < 2) nbuf = 2;
    if ((nbuf - 1) * 2 * t.rows > 62) nbuf = 62 / (2 * t.rows) + 1;           // the producer counts its copies in vmcnt (6 bits)
    a.nbuf = nbuf;
    a.store_vec = ((h * w) % 4 == 0 && (t.R * w) % 4 == 0) ? 4 : (((h * w) % 2 == 0 && (t.R * w) % 2 == 0) ? 2 : 1);
    const size_t lds = stage * nbuf;
    const long resident = wgs > 0 ? (long)kNumCU * wgs : tiles;
    const int  grid = (int)(tiles < resident ? tiles : resident);
    if (kh == 1)      launch_c8<1>(a, grid, t.nb, lds);
    else if (kh == 3) launch_c8<3>(a, grid, t.nb, lds);
    else              launch_c8<5>(a, grid, t.nb, lds);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

#ifdef PVHIP_DIAG
int pvhip_diag_c8_stamps(unsigned long long* out) {       // 16 accounts of conv_f16_c8_kernel (see g_c8_stamps), read and cleared
    unsigned long long zero[16] = {0};
    if (hipDeviceSynchronize() != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c8_stamps), sizeof(zero)) != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_c8_stamps), zero, sizeof(zero)) != hipSuccess) return PVHIP_EHIP;
    return PVHIP_OK;
}
#endif

}  // extern "C"
