// The network's first convolution -- 7x7, stride 2, 3 input channels (GoogLeNet conv1/7x7_s2: 60 GFLOP and 0.8 GB of output at
// batch 256) -- as a persistent kernel whose reduction loop contains nothing but LDS reads with immediate offsets and MFMAs.
//
// Replaces the same reference function as pvhip_conv.hip (Convolution.py:57-87).  With 3 input channels the im2col tile of the
// general kernel is 147 rows that are shifted copies of a few input rows: 75 KB of gathers through L1 per 128 pixels, each with
// its own window test.  Here a workgroup keeps
//   * the WHOLE weight panel in LDS (loaded once per workgroup; the workgroup then walks pixel tiles), laid out per reduction
//     step as [step][k parity][64 output channels], step = (c, s, pair of kernel rows j): the two k of one
//     v_mfma_f32_32x32x2_f32 are kernel rows 2j and 2j+1 of the same (c, s) (row 7 = zero weights: 84 steps for 147 taps),
//   * the zero-padded INPUT patch of its 8 x 16 output pixels (3 x 21 x 37 floats, row pitch 40, plus a row of zeros per channel), double-buffered: the patch
//     of the next tile is fetched into registers while the current one is multiplied.
// An Add of a per-channel constant in front of the convolution (GoogLeNet's data/mean) can ride in the patch fetch: x + m[c] for
// the elements inside the image (the same fp32 add the Add node does), zeros for the padding.
// The B operand of step (c, s, j) for lane (pixel, parity lh) is then patch[c][2*py + 2j + lh][2*px + s]: a per-lane base that never
// changes plus a compile-time offset -- ds_read_b32 with an immediate, no address arithmetic, no window test (padding is zeros in
// LDS), no global loads in the loop.  The A operand likewise.  D[k][pixel] as in pvhip_conv.hip: wave w owns output rows 2w, 2w+1
// of the tile, lane & 31 the pixel, so NCHW stores are 64-byte runs.
#include <cstdlib>

#include "pvhip_common.h"

using namespace pvhip;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kKH = 7, kKW = 7, kC = 3, kST = 2;
constexpr int kTR = 8, kTC = 16;                        // output tile: 8 rows x 16 columns = 128 pixels
constexpr int kPRin = (kTR - 1) * kST + kKH;            // 21 input rows
constexpr int kPR = kPRin + 1;                          // + one row of zeros: kernel row 7 (zero weights) of the last tile row reads it,
                                                        // and 0 * (whatever LDS holds there) must not be NaN
constexpr int kPC = (kTC - 1) * kST + kKW;              // 37 input columns
constexpr int kPitch = 40;
constexpr int kPatch = kC * kPR * kPitch;               // 2520 floats
constexpr int kSteps = kC * kKW * 4;                    // 84: (c, s, row pair j)
constexpr int kPanel = kSteps * 2 * 64;                 // 10752 floats = 42 KiB
constexpr int kPerThread = (kPatch + kBlock - 1) / kBlock;   // 10 patch elements per thread

struct StemArgs {
    const float* x;
    const float* wl;      // [84][2][64]
    float*       y;
    const float* bias;
    const float* pre_add;   // optional [3]: added to every input element inside the image (the Add node in front of the convolution)
    int N, H, W, K, OH, OW;
    int pt, pl;
    int tiles_y, tiles_x, n_tiles;
    int   act;
    float act_lo, act_hi;
    int y_ctotal, y_coff;
    unsigned wl_bytes, y_bytes;
};

__global__ __launch_bounds__(kBlock) void stem_pack_kernel(const float* __restrict__ w, float* __restrict__ wl, int K) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < kPanel; e += gridDim.x * blockDim.x) {
        const int ch = e & 63, lh = (e >> 6) & 1, step = e >> 7;
        const int j = step & 3, cs = step >> 2, s = cs % kKW, c = cs / kKW;
        const int r = 2 * j + lh;
        wl[e] = (r < kKH && ch < K) ? w[((size_t)(ch * kC + c) * kKH + r) * kKW + s] : 0.0f;
    }
}

__device__ __forceinline__ void stem_dma_b128(__amdgpu_buffer_rsrc_t r, float* dst, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lds = (unsigned)(unsigned long)(lds_ptr_t)dst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
#endif
}

// What the epilogue of one tile needs besides the accumulators: per 32-channel half the lane's byte offset in y (or out of
// range), and the arguments of the activation.
struct StemOut {
    unsigned voff[2];
    unsigned plane;        // bytes between channels
    int      k, act;
    float    act_lo, act_hi;
    bool     has_bias;
};

__device__ __forceinline__ void stem_store_one(__amdgpu_buffer_rsrc_t yr, const StemOut& o, const floatx16 (&acc)[2], const float (&bv)[2][16],
                                               int lh, int i, int r) {
    const int dr = (r & 3) + 8 * (r >> 2);
    float v = acc[i][r];
    if (o.has_bias) v = v + bv[i][r];
    if (o.act == 1) v = (v < 0.0f) ? 0.0f : v;
    else if (o.act == 2) { v = (v < o.act_lo) ? o.act_lo : v; v = (v > o.act_hi) ? o.act_hi : v; }
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, (i * 32 + 4 * lh + dr < o.k) ? o.voff[i] : 0x80000000u,
                                          (unsigned)dr * o.plane, 0);
}

// Steps STEP .. 83 of the reduction, fully unrolled with compile-time LDS offsets: read the operands of step STEP+1 into
// the other register set, then issue the two MFMAs of step STEP; the sched_group_barriers keep that order in the
// instruction stream (3 LDS reads, then 2 MFMAs), so an MFMA waits only for reads issued a whole step earlier.
// The 32 output stores of the PREVIOUS tile ride along, one every other step: all workgroups run in step (same work per
// tile), so an epilogue of its own would be a burst of 0.8 GB of stores with idle matrix cores, then matrix cores with an
// idle memory system; spread over the next tile's reduction the stores cost no time of their own.
template <int STEP, bool STORES>
__device__ __forceinline__ void stem_steps(const float* __restrict__ P, const float* __restrict__ A, float (&bfr)[2], float (&a0r)[2],
                                           float (&a1r)[2], floatx16 (&acc)[2], __amdgpu_buffer_rsrc_t yr, const StemOut& po,
                                           const floatx16 (&prev)[2], const float (&bv)[2][16], int lh) {
    constexpr int cur = STEP & 1, nxt = cur ^ 1;
    constexpr bool store_here = STORES && STEP >= 8 && STEP < 8 + 64 && (STEP % 2 == 0);
    if constexpr (STEP + 1 < kSteps) {
        constexpr int n_ = STEP + 1, j_ = n_ & 3, cs_ = n_ >> 2, s_ = cs_ % kKW, c_ = cs_ / kKW;
        bfr[nxt] = P[c_ * kPR * kPitch + 2 * j_ * kPitch + s_];
        a0r[nxt] = A[n_ * 128];
        a1r[nxt] = A[n_ * 128 + 32];
    }
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0r[cur], bfr[cur], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1r[cur], bfr[cur], acc[1], 0, 0, 0);
    if constexpr (store_here) stem_store_one(yr, po, prev, bv, lh, ((STEP - 8) / 2) / 16, ((STEP - 8) / 2) % 16);
    if constexpr (STEP + 1 < kSteps) {
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        if constexpr (store_here) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
        stem_steps<STEP + 1, STORES>(P, A, bfr, a0r, a1r, acc, yr, po, prev, bv, lh);
    }
}

template <int ABL>   // diagnostic ablations (wrong results on purpose): 1 no MFMA loop, 2 no output stores, 3 no patch fetch
__global__ __launch_bounds__(kBlock, 2) void conv_stem7x7_kernel(StemArgs a) {
    __shared__ __attribute__((aligned(1024))) float Wl[kPanel];
    __shared__ __attribute__((aligned(16))) float Patch[2][kPatch];

    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid  = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int l31 = lane & 31, lh = lane >> 5;
    const int HW = a.H * a.W, OHW = a.OH * a.OW;

    // weight panel -> LDS, once
    {
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wl), 0, a.wl_bytes, 0x00020000);
        for (int q = wid; q < kPanel / 256; q += kBlock / kWave) stem_dma_b128(wr, Wl + q * 256, (unsigned)lane * 16u + (unsigned)q * 1024u, 0u);
    }

    // this thread's elements of an input patch: (channel, row, column) of idx = tid + 256*i, and their offset from the
    // patch origin in the tensor; a patch that lies wholly inside the image (most do) needs no window test
    int pr[kPerThread], pq[kPerThread], rel[kPerThread];
    float padd[kPerThread];
    bool used[kPerThread];
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) {
        const int idx = tid + kBlock * i;
        const int c_ = idx / (kPR * kPitch);
        pr[i] = (idx % (kPR * kPitch)) / kPitch;
        pq[i] = idx % kPitch;
        used[i] = idx < kPatch && pq[i] < kPC && pr[i] < kPRin;
        rel[i] = c_ * HW + pr[i] * a.W + pq[i];
        padd[i] = (a.pre_add != nullptr && c_ < kC) ? a.pre_add[c_] : 0.0f;
    }
    float pre[kPerThread];
#define PVS_FETCH(t_)                                                                                       \
    {                                                                                                       \
        const int tx_ = (t_) % a.tiles_x, r_ = (t_) / a.tiles_x, ty_ = r_ % a.tiles_y, n_ = r_ / a.tiles_y; \
        const int iy0 = ty_ * kTR * kST - a.pt, ix0 = tx_ * kTC * kST - a.pl;                               \
        const float* __restrict__ xo = a.x + ((long)n_ * kC * HW + (long)iy0 * a.W + ix0);                  \
        if (iy0 >= 0 && ix0 >= 0 && iy0 + kPRin <= a.H && ix0 + kPC <= a.W) {                                 \
            _Pragma("unroll") for (int i = 0; i < kPerThread; ++i) pre[i] = used[i] ? xo[rel[i]] + padd[i] : 0.0f; \
        } else {                                                                                            \
            _Pragma("unroll") for (int i = 0; i < kPerThread; ++i) {                                        \
                const int iy = iy0 + pr[i], ix = ix0 + pq[i];                                               \
                const bool ok = used[i] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;    \
                pre[i] = ok ? xo[rel[i]] + padd[i] : 0.0f;                                                  \
            }                                                                                               \
        }                                                                                                   \
    }
#define PVS_STORE(buf_)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < kPerThread; ++i)                                                  \
        if (tid + kBlock * i < kPatch) Patch[buf_][tid + kBlock * i] = pre[i];

    // operand bases: pixel (py, px) = (2*wid + (l31 >> 4), l31 & 15) of the tile; kernel-row parity lh
    const int py = 2 * wid + (l31 >> 4), px = l31 & 15;
    const int b_base = (kST * py + lh) * kPitch + kST * px;
    const int a_base = lh * 64 + l31;

    int t = blockIdx.x, cur = 0;
    if (t < a.n_tiles) {
        PVS_FETCH(t);
        PVS_STORE(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0,
                                                                        a.bias != nullptr ? a.K * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    // this lane's 32 output channels never change: their bias values are loaded once
    float bv[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            bv[i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (unsigned)(i * 32 + 4 * lh + (r & 3) + 8 * (r >> 2)) * 4u, 0, 0));
    StemOut po;
    po.voff[0] = po.voff[1] = 0x80000000u;        // no previous tile yet: its stores are out of range (dropped by the hardware)
    po.plane = (unsigned)OHW * 4u;
    po.k = a.K; po.act = a.act; po.act_lo = a.act_lo; po.act_hi = a.act_hi; po.has_bias = a.bias != nullptr;
    floatx16 prev[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) prev[i][r] = 0.0f;

    for (; t < a.n_tiles; t += gridDim.x) {
        const int tn = t + (int)gridDim.x;
        if (ABL != 3 && tn < a.n_tiles) PVS_FETCH(tn);

        floatx16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        const float* __restrict__ P = Patch[cur] + b_base;
        const float* __restrict__ A = Wl + a_base;
        // operands of step k+1 are read from LDS before the MFMAs of step k are issued (two register sets); the previous
        // tile's outputs are stored along the way
        float bfr[2], a0r[2], a1r[2];
        bfr[0] = P[0];
        a0r[0] = A[0];
        a1r[0] = A[32];
        if (ABL == 1) { acc[0][0] = bfr[0] + a0r[0]; acc[1][0] = a1r[0]; }
        else if (ABL == 2) stem_steps<0, false>(P, A, bfr, a0r, a1r, acc, yr, po, prev, bv, lh);
        else stem_steps<0, true>(P, A, bfr, a0r, a1r, acc, yr, po, prev, bv, lh);

        // this tile becomes the previous one: where its outputs go
        {
            const int tx_ = t % a.tiles_x, r_ = t / a.tiles_x, ty_ = r_ % a.tiles_y, n_ = r_ / a.tiles_y;
            const int oy = ty_ * kTR + py, ox = tx_ * kTC + px;
            const bool inside = oy < a.OH && ox < a.OW;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                po.voff[i] = inside ? (unsigned)((n_ * a.y_ctotal + a.y_coff + i * 32 + 4 * lh) * OHW + oy * a.OW + ox) * 4u : 0x80000000u;
                prev[i] = acc[i];
            }
        }
        if (ABL != 3 && tn < a.n_tiles) PVS_STORE(cur ^ 1);
        __syncthreads();          // the next patch is in place; everybody is done with the current one
        cur ^= 1;
    }
    // the last tile's outputs
    if (ABL != 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) stem_store_one(yr, po, prev, bv, lh, i, r);
    }
#undef PVS_FETCH
#undef PVS_STORE
}

}  // namespace

namespace pvhip {

bool stem_eligible(int c, int kh, int kw, int sh, int sw, int k_out, int pad_top, int pad_left, unsigned long long out_elems) {
    if (out_elems >= (1ull << 29)) return false;          // 32-bit byte offsets in the buffer stores
    // Opt-in (PVHIP_CONV_STEM=1).  Alone on the chip this kernel beats the general one (0.74 against 0.83 ms on conv1), but in the
    // forward pass with several requests in flight it LOSES 2 % of the images/s: its persistent workgroups hold 128 KB of every
    // CU's LDS for the whole launch, so the bandwidth-bound kernels of the other requests cannot move in beside it.
    if (!settings().conv_stem) return false;
    return c == kC && kh == kKH && kw == kKW && sh == kST && sw == kST && k_out <= 64 && pad_top >= 0 && pad_left >= 0;
}

size_t stem_pack_elems(int k, int c, int kh, int kw) { return (c == kC && kh == kKH && kw == kKW && k <= 64) ? (size_t)kPanel : 0; }

int stem_pack(const float* w_oihw, float* wl, int k) {
    hipLaunchKernelGGL(stem_pack_kernel, dim3(grid_for(kPanel)), dim3(kBlock), 0, state().stream, w_oihw, wl, k);
    return PVHIP_OK;
}

int stem_conv(const float* x, const float* pre_add, const float* wl, float* y, int n, int h, int w, int k_out, int oh, int ow, int pad_top,
              int pad_left, const float* bias, int act, float act_lo, float act_hi, int out_channel_offset, int out_channels_total) {
    StemArgs a;
    a.x = x; a.wl = wl; a.y = y; a.bias = bias; a.pre_add = pre_add;
    a.N = n; a.H = h; a.W = w; a.K = k_out; a.OH = oh; a.OW = ow; a.pt = pad_top; a.pl = pad_left;
    a.tiles_y = (oh + kTR - 1) / kTR; a.tiles_x = (ow + kTC - 1) / kTC;
    const long tiles = (long)n * a.tiles_y * a.tiles_x;
    if (tiles > 0x7fffffffL) return fail(PVHIP_EUNSUPPORTED, "stem_conv: too many tiles");
    a.n_tiles = (int)tiles;
    a.act = act; a.act_lo = act_lo; a.act_hi = act_hi;
    a.y_ctotal = out_channels_total; a.y_coff = out_channel_offset;
    a.wl_bytes = (unsigned)(kPanel * sizeof(float));
    a.y_bytes  = (unsigned)((size_t)n * out_channels_total * oh * ow * sizeof(float));      // pvhip_conv2d_f32 keeps outputs below 2^31 elements... and this kernel below 2^31 bytes
    const int per_cu = settings().stem_wg > 0 ? settings().stem_wg : 2;     // PVHIP_STEM_WG: tuning runs only
    const int grid = (int)(tiles < (long)per_cu * kNumCU ? tiles : (long)per_cu * kNumCU);
#ifdef PVHIP_DIAG
    switch (settings().stem_ablate) {     // diagnostic build only (libpvhip_diag.so): results are wrong on purpose
        case 1: hipLaunchKernelGGL(conv_stem7x7_kernel<1>, dim3(grid), dim3(kBlock), 0, state().stream, a); return PVHIP_OK;
        case 2: hipLaunchKernelGGL(conv_stem7x7_kernel<2>, dim3(grid), dim3(kBlock), 0, state().stream, a); return PVHIP_OK;
        case 3: hipLaunchKernelGGL(conv_stem7x7_kernel<3>, dim3(grid), dim3(kBlock), 0, state().stream, a); return PVHIP_OK;
        default: break;
    }
#endif
    hipLaunchKernelGGL(conv_stem7x7_kernel<0>, dim3(grid), dim3(kBlock), 0, state().stream, a);
    return PVHIP_OK;
}

}  // namespace pvhip
