// Pointwise (1x1 / stride 1 / unpadded) convolution on the fp32 matrix cores: the 1x1, 3x3_reduce, 5x5_reduce and pool_proj
// arms of the inception modules (32 % of GoogLeNet's convolution time), the SSD-MobileNet pointwise layers.
//
// Replaces Convolution.py:57-87 (im2col + np.dot) for that geometry.  D[k_out][pixel] = sum_c W[k_out][c] * X[c][pixel]
// with v_mfma_f32_32x32x2_f32, reduction in ascending c, two products per instruction: the bits of conv_igemm_dma_kernel.
//
// What bounded the general LDS-DMA kernel on these layers was not the matrix cores but the instructions around them: per 8
// MFMAs a wave issued 16 ds_read_b32, 2-3 LDS-DMA pieces, a vmcnt(0) and a barrier, and every 32-channel tile of a pixel
// tile streamed the SAME activation tile into LDS again (6-20 times through L2).  Here:
//   * a workgroup (4 waves) owns 128 pixels and a chunk of four 32-channel tiles, wave w owns tile w.  The activation tile
//     [16 channels][128 pixels] of a stage is brought into LDS ONCE per chunk (LDS-DMA, 16 bytes per lane, no staging
//     registers) and read by all four waves: T/4 times through L2 instead of T times;
//   * MFMA column n of accumulator j is pixel 4n + j: a lane's B operands of a reduction step for its 4 accumulators
//     are 4 consecutive floats of one LDS row -- ONE ds_read_b128 (conflict-free) feeds 4 MFMAs -- and an accumulator
//     register of the 4 accumulators is 4 consecutive pixels of one channel: the epilogue stores 16 bytes per lane;
//   * the weights never pass through LDS: no two waves of a workgroup share any, so each wave loads its own fragments
//     straight into registers from a panel packed in fragment order (pw_pack_kernel: one coalesced 16-byte load per lane
//     = 4 reduction steps), one stage ahead.
// Per 32 MFMAs a wave now issues 8 ds_read_b128, 2 global loads, 2 LDS-DMA pieces and one barrier, and a workgroup
// holds 32 KB of LDS: four stage buffers, three stages of activations in flight (see the pipeline comment in the kernel).
#include <cstdlib>
#include <cstring>

#include "pvhip_common.h"
#include "pvhip_wino.h"

using namespace pvhip;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int kBK = 16;            // input channels per stage
constexpr unsigned kOob = 0x80000000u;

struct PwArgs {
    const float* x;
    const float* ap;       // fragment-ordered weight panel [T][S][2][64][4]
    const float* bias;     // optional [32*T] (panel rows)
    int C, HW, P, S, T, nchunk;
    unsigned x_bytes;
    int   act;             // 0 none, 1 ReLU (ReLU.py:11), 2 clamp (Clamp.py:11)
    float lo, hi;
    int   stagger;         // 10-ns ticks between the start times of the workgroups that share a CU at launch (0 = none)
    int   nseg;
    PwDest seg[PVHIP_MAX_CONV_DESTS];
};

// Panel element (((t*S + s)*2 + h)*64 + lane)*4 + i  =  W[32t + (lane & 31)][16s + 8h + 2i + (lane >> 5)]: what lane `lane`
// feeds v_mfma_f32_32x32x2_f32 as its A operand in reduction steps 4h .. 4h+3 of stage s.  Rows >= k are zero.
__global__ __launch_bounds__(kBlock) void pw_pack_kernel(const float* __restrict__ w, float* __restrict__ ap, int k, int c, int T, int S) {
    const size_t total  = (size_t)T * S * 512;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int i = (int)(e & 3), lane = (int)((e >> 2) & 63), h = (int)((e >> 8) & 1);
        const size_t ts = e >> 9;
        const int s = (int)(ts % S), t = (int)(ts / S);
        const int row = 32 * t + (lane & 31), col = 16 * s + 8 * h + 2 * i + (lane >> 5);
        ap[e] = (row < k) ? w[(size_t)row * c + col] : 0.0f;
    }
}

// The MFMAs of one stage for a wave that owns TN pixel columns per lane: reduction step kk takes its A operand from the wave's
// fragment registers and ONE LDS read of TN consecutive floats (read one step ahead) as the B operands of its TN accumulators.
template <int TN>
__device__ __forceinline__ void pw_mfma_stage(const float* bs, const float4v (&fa)[2], floatx16 (&acc)[4]) {
    typedef float bvec_t __attribute__((ext_vector_type(TN)));
    bvec_t bq[2];
    bq[0] = *reinterpret_cast<const bvec_t*>(bs);
#pragma unroll
    for (int kk = 0; kk < kBK / 2; ++kk) {
        const int c = kk & 1;
        if (kk + 1 < kBK / 2) bq[c ^ 1] = *reinterpret_cast<const bvec_t*>(bs + 2 * (kk + 1) * 128);
        const float av = fa[kk >> 2][kk & 3];
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq[c][j], acc[j], 0, 0, 0);
        if (kk + 1 < kBK / 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, TN, 0);
    }
}

// Bias / activation / store of a wave's TN accumulators: register r of accumulator j is channel row0 + (r&3) + 8*(r>>2) of
// pixel gp0 + j; with whole 4-pixel groups (kVec) the TN pixels of a lane are one TN*4-byte store.
// ACT: 0 none, 1 a lower bound (ReLU), 2 both bounds (Clamp) -- act_apply without the compare-and-select pairs that do nothing for the
// launch's activation (pw_store picks the instantiation behind a wave-uniform branch: two of the five vector instructions per value).
template <int ACT>
__device__ __forceinline__ float pw_act(float v, const ActBounds& b) {
    if (ACT >= 1) v = (v < b.lo) ? b.lo : v;
    if (ACT == 2) v = (v > b.hi) ? b.hi : v;
    return v;
}
template <int TN, bool kVec, int ACT>
__device__ __forceinline__ void pw_store_act(const PwArgs& a, const floatx16 (&acc)[4], const float (&bv)[16], float* __restrict__ yb, int yct,
                                             int ycoff, int klim, int row0, int gp0) {
    typedef float bvec_t __attribute__((ext_vector_type(TN)));
    const ActBounds ab = act_bounds(a.act, a.lo, a.hi);          // bv[] holds -0.0 where there is no bias
    if (kVec) {
        if (gp0 >= a.P) return;
        const int n = gp0 / a.HW, hw = gp0 - n * a.HW;
        float* __restrict__ yp = yb + ((size_t)n * yct + ycoff + row0) * a.HW + hw;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (row0 + dr < klim) {
                bvec_t v;
#pragma unroll
                for (int j = 0; j < TN; ++j) v[j] = pw_act<ACT>(acc[j][r] + bv[r], ab);
                conv_storev(reinterpret_cast<bvec_t*>(yp + (size_t)dr * a.HW), v);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gp = gp0 + j;
            if (gp >= a.P) continue;
            const int n = gp / a.HW, hw = gp - n * a.HW;
            float* __restrict__ yp = yb + ((size_t)n * yct + ycoff + row0) * a.HW + hw;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                if (row0 + dr < klim) {
                    conv_store1(yp + (size_t)dr * a.HW, pw_act<ACT>(acc[j][r] + bv[r], ab));
                }
            }
        }
    }
}
template <int TN, bool kVec>
__device__ __forceinline__ void pw_store(const PwArgs& a, const floatx16 (&acc)[4], const float (&bv)[16], float* __restrict__ yb, int yct,
                                         int ycoff, int klim, int row0, int gp0) {
    if (a.act == 0) pw_store_act<TN, kVec, 0>(a, acc, bv, yb, yct, ycoff, klim, row0, gp0);
    else if (a.act == 1) pw_store_act<TN, kVec, 1>(a, acc, bv, yb, yct, ycoff, klim, row0, gp0);
    else pw_store_act<TN, kVec, 2>(a, acc, bv, yb, yct, ycoff, klim, row0, gp0);
}

// One workgroup = 4 waves, one per SIMD (with one wave per channel tile and 5-7 waves per workgroup, two waves of a workgroup
// shared a SIMD and the others waited for them at every barrier: 15-35 % lost) = 128 pixels x a chunk of channel tiles:
//   TN = 4: wave w owns channel tile 4*chunk + w and all 128 pixels          (4 accumulators, one ds_read_b128 per 4 MFMAs);
//           in the last chunk of a panel whose tile count is not a multiple of 4 the waves without a tile only help to copy
//   TN = 2: panels of two tiles: wave w owns tile w/2 and pixels 64*(w%2) .. +63    (2 accumulators, ds_read_b64)
//   TN = 1: panels of one tile:  wave w owns the tile and pixels 32*w .. +31        (1 accumulator, ds_read_b32)
// MFMA column n of accumulator j is pixel TN*n + j of the wave's pixels.  (TN is a template argument, one kernel per form: as
// a run-time value, or as three inlined bodies, hipcc keeps a set of accumulators per form and copies between them.)
// ABL (diagnostic build only, results wrong on purpose): 1 = the activation copies read nothing (a descriptor of 0 records: zeros
// land in LDS), 2 = no epilogue stores, 4 = no MFMAs.
#ifdef PVHIP_DIAG
__device__ unsigned long long g_pw_stamps[8];      // ABL = 8: cycles of every 61st workgroup's wave 0: prologue, main loop, epilogue, count; [4] 10 ns ticks
#define PW_NOW() ((ABL == 8) ? (unsigned long long)__builtin_readcyclecounter() : 0ull)
#else
#define PW_NOW() 0ull
#endif
template <int TN, bool kVec, int ABL = 0, int NB = 4>      // NB: LDS stage buffers (4: three stages of copies in flight, 32 KB; 3: two, 24 KB -- six workgroups per CU)
__global__ __launch_bounds__(kBlock, NB == 3 ? 6 : 4) void conv_pw_kernel(PwArgs a) {
    const unsigned long long pw_t0 = PW_NOW();
#ifdef PVHIP_DIAG
    const unsigned long long pw_r0 = (ABL == 8) ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
    constexpr int BN = 128;
    constexpr int NP = kVec ? 2 : 8;                     // LDS-DMA instructions per wave and stage (1 KiB or 256 B each)
    __shared__ __attribute__((aligned(1024))) float Bs[NB][kBK][BN];

    // XCD-aware tile order: the channel chunks of one pixel tile run back to back on one XCD (the chunks after the first
    // find the activation tile in that L2)
    const int nwg = gridDim.x;
    int       lid;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int chunk = lid % a.nchunk;
    const int p0    = (lid / a.nchunk) * BN;
    const int lane  = threadIdx.x & (kWave - 1);
    const int wid   = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int l31 = lane & 31, lh = lane >> 5;
    constexpr int tn = TN, pc = 4 / TN;                  // pixel sub-tiles per workgroup tile
    const int  t_raw  = TN * chunk + wid / pc;           // this wave's channel tile
    const bool active = t_raw < a.T;                     // wave-uniform; a wave without a tile only helps to copy
    const int  t      = active ? t_raw : a.T - 1;
    const int  sub    = wid % pc;                        // and its pixels: p0 + sub * 32 * TN ...
    const unsigned chan_bytes = (unsigned)a.HW * 4u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (ABL & 1) ? 0 : a.x_bytes, 0x00020000);

    // ---- this lane's part of an LDS-DMA instruction: byte offset of its pixel(s) in channel row 0 of the stage (+ its row
    // inside the piece); pixels past the tensor are out of range and land as 0
    unsigned dvoff[kVec ? 1 : 2];
    if (kVec) {
        const int gp = p0 + 4 * l31;                      // a group of 4 pixels never straddles two images (HW % 4 == 0)
        dvoff[0] = kOob;
        if (gp < a.P) {
            const int n = gp / a.HW, hw = gp - n * a.HW;
            dvoff[0] = (unsigned)(n * a.C * a.HW + hw) * 4u + (unsigned)lh * chan_bytes;
        }
    } else {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int gp = p0 + hf * 64 + lane;
            dvoff[hf] = kOob;
            if (gp < a.P) {
                const int n = gp / a.HW, hw = gp - n * a.HW;
                dvoff[hf] = (unsigned)(n * a.C * a.HW + hw) * 4u;
            }
        }
    }
    // vec: wave w copies pieces w and w + 4 (piece q = tile rows 2q, 2q+1); dword: wave w copies rows 4w .. 4w+3, two halves each
#define PW_ISSUE(stage_, buf_)                                                                               \
    {                                                                                                        \
        if (kVec) {                                                                                          \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                  \
                const int q = wid + 4 * i;                                                                   \
                lds_dma_b128(xr, &Bs[buf_][0][0] + q * 256, dvoff[0], (unsigned)((stage_) * kBK + 2 * q) * chan_bytes); \
            }                                                                                                \
        } else {                                                                                             \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                  \
                const int row = 4 * wid + (i >> 1);                                                          \
                lds_dma_b32(xr, &Bs[buf_][0][0] + row * BN + (i & 1) * 64, dvoff[(!kVec) ? (i & 1) : 0],     \
                            (unsigned)((stage_) * kBK + row) * chan_bytes);                                  \
            }                                                                                                \
        }                                                                                                    \
    }

    // ---- software pipeline.  Iteration j issues  [A(j+2): 2 loads into registers]  [B(j+3): NP LDS-DMA pieces]  and, after its
    // MFMAs, makes sure A(j+1) has landed.  Four LDS buffers: stage j is read while j+1, j+2, j+3 land; B(j+3) overwrites the buffer
    // stage j-1 was read from, behind the barrier every wave passes after its last read of it.
    //
    // The weight loads are ORDINARY loads (hipcc tracks them and places their s_waitcnt itself); only the LDS-DMA copies, which have no
    // register destination, are asm statements.  Rounds 1-2 issued the weight loads from asm too and waited for them in a second asm
    // statement with a hand-counted vmcnt: between the two statements hipcc considers the destination register defined and may copy it
    // before the data has landed (LESSONS.md lesson 24: wrong images at batch 256 next to other streams, in pvhip_wino.hip).  With tracked
    // loads (buffer-load builtins: scalar panel offset, one lane offset) there is no such window, by construction: the vector-memory
    // counter retires in order and hipcc counts only ITS loads, a
    // subsequence of the real queue (the asm copies are invisible to it), so the vmcnt it emits can only wait for MORE than it needs.
    // What that costs: its vmcnt(2) at the end of iteration j ("A(j+2) may still fly") lets only the last two operations of the
    // iteration -- LDS-DMA pieces of B(j+3) -- stay in flight, everything issued before must have landed: one stage of latency hiding
    // where the hand-counted vmcnt(2*NP+2) kept three (measured equal: latency was never this kernel's limit, lesson 18).
    // B(j+1), which the next iteration reads after the barrier, is older than all of that (and has its own explicit wait, which names
    // no register).  Three register sets rotate by name.
    const int S = a.S;
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ap), 0, (unsigned)(a.T * S) * 2048u, 0x00020000);
    const unsigned avoff = (unsigned)lane * 16u;
    const int boff = sub * 32 * tn + tn * l31;           // this lane's first pixel column in a tile row

    floatx16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    float4v ra[3][2];
#define PW_LOAD_A(set_, stage_)                                                                              \
    {                                                                                                        \
        const int sa_ = (stage_) < S ? (stage_) : S - 1;     /* past the end: the last stage again (never consumed) */ \
        const unsigned so_ = (unsigned)(t * S + sa_) * 2048u;                                                \
        ra[set_][0] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(ar, avoff, so_, 0));           \
        ra[set_][1] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(ar, avoff + 1024u, so_, 0));   \
    }
    // a use of the set hipcc can see: its own s_waitcnt for the two loads goes in front of it
#if defined(__HIP_DEVICE_COMPILE__)
#define PW_WAIT_A(set_) asm volatile("" : "+v"(ra[set_][0]), "+v"(ra[set_][1]) :: "memory")
#else
#define PW_WAIT_A(set_)
#endif
#define PW_STAGE(s_, cur_, ld_, nx_)                                                                         \
    {                                                                                                        \
        asm volatile("s_barrier" ::: "memory");                                                              \
        PW_LOAD_A(ld_, (s_) + 2);                                                                            \
        {                                                                                                    \
            const int sd_ = (s_) + NB - 1 < S ? (s_) + NB - 1 : S - 1;                                       \
            PW_ISSUE(sd_, ((s_) + NB - 1) % NB);                                                             \
        }                                                                                                    \
        if (active && !(ABL & 4)) pw_mfma_stage<TN>(&Bs[(s_) % NB][lh][boff], ra[cur_], acc);  \
        PW_WAIT_A(nx_);                                                                                      \
        /* this wave's pieces of B(j+1) must be in LDS before the next barrier.  Younger than them are B(j+2) and B(j+3) -- and */ \
        /* the weight loads A(j+1), A(j+2), which are NOT counted: hipcc deletes the loads of the stages past the end, and a */ \
        /* count that relied on them would let two pieces of B(j+1) fly there.  Not counting them only waits for more.  (No  */ \
        /* register is involved: a hand-counted wait is safe here, and hipcc's own wait above has usually satisfied it.)     */ \
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"((NB - 2) * NP) : "memory");                                \
    }

    // prologue: A(0), A(1), B(0), B(1), B(2); A(0) and B(0) must have landed
    PW_LOAD_A(0, 0);
    PW_LOAD_A(1, 1);
    PW_ISSUE(0, 0);
    { const int s1 = 1 < S ? 1 : S - 1; PW_ISSUE(s1, 1); }
    if (NB == 4) { const int s2 = 2 < S ? 2 : S - 1; PW_ISSUE(s2, 2); }
    PW_WAIT_A(0);
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"((NB - 2) * NP) : "memory");     // B(0) of this wave (B(1), B(2) may fly)
    const unsigned long long pw_t1 = PW_NOW();

    // whole triples in the loop, the one or two stages left behind it: with `break`s inside the loop hipcc's structurizer gave the
    // loop ONE latch that the break paths reach too, and its vmcnt bookkeeping then entered the header with "set 0 may still be in
    // flight" (true on the path that leaves behind the second stage, which never comes back): a vmcnt(3) in front of the first MFMA
    // of every third stage, i.e. a wait for a load issued a few instructions earlier.
    int s = 0;
    for (; s + 3 <= S; s += 3) {
        PW_STAGE(s, 0, 2, 1);
        PW_STAGE(s + 1, 1, 0, 2);
        PW_STAGE(s + 2, 2, 1, 0);
    }
    if (s < S) {
        PW_STAGE(s, 0, 2, 1);
        if (s + 1 < S) PW_STAGE(s + 1, 1, 0, 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the look-ahead copies must not outlive the workgroup's LDS
    const unsigned long long pw_t2 = PW_NOW();
    (void)pw_t0; (void)pw_t1; (void)pw_t2;
#undef PW_STAGE
#undef PW_WAIT_A
#undef PW_LOAD_A
#undef PW_ISSUE
    if (!active) return;
    if (ABL & 2) {                                       // keep the accumulators alive, store nothing
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)                      // (on the host pass a 64-byte "v" operand silently drops the kernel's stub)
            asm volatile("" :: "v"(acc[j]));
#endif
        }
        return;
    }

    // ---- epilogue.  The destination table is read from the argument block HERE, through a pointer the compiler cannot see
    // through (as plain arguments its dwords would be held in scalar registers across the loop).
    typedef const __attribute__((address_space(4))) PwArgs* kernarg_p;
    kernarg_p ka = (kernarg_p)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    int sg = 0;
    const int nseg = ka->nseg;
    for (int q = 1; q < nseg; ++q) sg = (32 * t >= ka->seg[q].m_begin) ? q : sg;
    float* __restrict__ yb = ka->seg[sg].y;
    const int yct   = ka->seg[sg].ctotal;
    const int ycoff = ka->seg[sg].coff - ka->seg[sg].m_begin;
    const int klim  = ka->seg[sg].m_begin + ka->seg[sg].k;

    const int row0 = 32 * t + 4 * lh;                     // this lane's channels: row0 + (r&3) + 8*(r>>2)
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0,
                                                                        a.bias != nullptr ? a.T * 32 * 4 : 0, 0x00020000);
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (unsigned)(row0 + (r & 3) + 8 * (r >> 2)) * 4u, 0, 0));
        if (a.bias == nullptr) bv[r] = -0.0f;
    }

    if (ABL == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long pw_tb = PW_NOW();
    (void)pw_tb;
    const int gp0 = p0 + boff;
    pw_store<TN, kVec>(a, acc, bv, yb, yct, ycoff, klim, row0, gp0);
#ifdef PVHIP_DIAG
    if (ABL == 8 && blockIdx.x % 61 == 7 && threadIdx.x == 0) {
        const unsigned long long pw_t3 = PW_NOW();
        atomicAdd(&g_pw_stamps[5], pw_tb - pw_t2);
        atomicAdd(&g_pw_stamps[0], pw_t1 - pw_t0);
        atomicAdd(&g_pw_stamps[1], pw_t2 - pw_t1);
        atomicAdd(&g_pw_stamps[2], pw_t3 - pw_t2);
        atomicAdd(&g_pw_stamps[3], 1ull);
        atomicAdd(&g_pw_stamps[4], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - pw_r0));
    }
#endif
}

template <int TN>
void launch_pw(const PwArgs& a, bool vec, int grid) {
    hipStream_t st = state().stream;
#ifdef PVHIP_DIAG
    if (vec && TN == 4) switch (settings().pw_ablate) {  // diagnostic build only: wrong on purpose
        case 1: hipLaunchKernelGGL((conv_pw_kernel<4, true, 1>), dim3(grid), dim3(kBlock), 0, st, a); return;
        case 2: hipLaunchKernelGGL((conv_pw_kernel<4, true, 2>), dim3(grid), dim3(kBlock), 0, st, a); return;
        case 3: hipLaunchKernelGGL((conv_pw_kernel<4, true, 3>), dim3(grid), dim3(kBlock), 0, st, a); return;
        case 4: hipLaunchKernelGGL((conv_pw_kernel<4, true, 4>), dim3(grid), dim3(kBlock), 0, st, a); return;
        case 6: hipLaunchKernelGGL((conv_pw_kernel<4, true, 6>), dim3(grid), dim3(kBlock), 0, st, a); return;
        default: break;
    }
    if (vec && settings().pw_ablate == 8) { hipLaunchKernelGGL((conv_pw_kernel<TN, true, 8>), dim3(grid), dim3(kBlock), 0, st, a); return; }   // s_memtime stamps
#endif
    // Three stage buffers (24 KB: six workgroups per CU instead of four, two stages of copies in flight instead of three), PVHIP_TUNE2=3 only.  MEASURED
    // (scripts/time_pw.py, same box): alone, the 28x28 sibling launches gain 5 % (3a 0.148 -> 0.141 ms, 3b 0.281 -> 0.267), the 14x14 ones lose up to 8 %; with
    // a rule that takes the former the step does not move (54.96 / 54.57 / 55.13 k images/s against 54.97 / 54.41 / 55.04): not the default.
    if (vec && TN == 2 && settings().tune[2] == 3) {
        hipLaunchKernelGGL((conv_pw_kernel<TN, true, 0, 3>), dim3(grid), dim3(kBlock), 0, st, a);
        return;
    }
    if (vec) hipLaunchKernelGGL((conv_pw_kernel<TN, true>), dim3(grid), dim3(kBlock), 0, st, a);
    else     hipLaunchKernelGGL((conv_pw_kernel<TN, false>), dim3(grid), dim3(kBlock), 0, st, a);
}

}  // namespace

namespace pvhip {

bool pw_eligible(int c, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int h, int w, int oh, int ow) {
    return kh == 1 && kw == 1 && sh == 1 && sw == 1 && pad_top == 0 && pad_left == 0 && oh == h && ow == w && c % kBK == 0 &&
           settings().conv_pointwise;
}

size_t pw_pack_elems(int k, int c) {
    if (k <= 0 || c <= 0 || c % kBK) return 0;
    return (size_t)((k + 31) / 32) * (c / kBK) * 512;
}

int pw_pack(const float* w_oihw, float* ap, int k, int c) {
    const int T = (k + 31) / 32, S = c / kBK;
    hipLaunchKernelGGL(pw_pack_kernel, dim3(grid_for((size_t)T * S * 512)), dim3(kBlock), 0, state().stream, w_oihw, ap, k, c, T, S);
    return PVHIP_OK;
}

int pw_conv(const float* x, const float* ap, int n, int c, int hw, int k_panel, const float* bias, int act, float act_lo, float act_hi,
            int ndest, const PwDest* dests) {
    PwArgs a;
    a.x = x; a.ap = ap; a.bias = bias;
    a.C = c; a.HW = hw; a.P = n * hw; a.S = c / kBK; a.T = (k_panel + 31) / 32;
    a.x_bytes = (unsigned)((unsigned long long)n * c * hw * 4ull);
    a.act = act; a.lo = act_lo; a.hi = act_hi;
    a.nseg = ndest;
    for (int i = 0; i < ndest; ++i) a.seg[i] = dests[i];
    // Channel tiles per workgroup.  Measured on the GoogLeNet shapes at batch 256 (scripts/time_pw.py): two (64 pixels per wave,
    // 6 waves per SIMD by registers) beats four (128 pixels per wave, 4 waves per SIMD) on every panel -- more, smaller
    // workgroups interleave their first loads and their stores with each other's MFMAs -- and one (32 pixels per wave) wins
    // where two would leave the chip short of workgroups, or leave half a workgroup idle (odd T) on a small layer.
    const long ptiles = (a.P + 127) / 128;
    const long grid2  = ptiles * ((a.T + 1) / 2);
    int tn = 2;
    if (a.T == 1 || grid2 < 4L * kNumCU || ((a.T & 1) && grid2 < 16L * kNumCU)) tn = 1;
    if (settings().pw_tn == 4 && a.T >= 3) tn = 4;      // PVHIP_PW_TN: tuning runs only
    if (settings().pw_tn == 2 && a.T >= 2) tn = 2;
    if (settings().pw_tn == 1) tn = 1;
    a.nchunk = (a.T + tn - 1) / tn;
    // tile time of a four-tile workgroup sharing its SIMDs with three others: S stages x 32 MFMAs x 64 cycles x 4 at ~2.1 GHz;
    // the four workgroups of a CU start a quarter of that apart (PVHIP_PW_STAGGER: percent of that quarter, tuning runs)
    a.stagger = 0;
    if ((long)((a.P + 127) / 128) * a.nchunk > 8L * kNumCU)
        a.stagger = (int)((double)a.S * 32.0 * 64.0 * 4.0 / 2100.0 * 100.0 / 4.0 * settings().pw_stagger_pct / 100.0);
    const int grid = ((a.P + 127) / 128) * a.nchunk;
    if (tn == 4) launch_pw<4>(a, hw % 4 == 0, grid);
    else if (tn == 2) launch_pw<2>(a, hw % 4 == 0, grid);
    else launch_pw<1>(a, hw % 4 == 0, grid);
    return PVHIP_OK;
}

}  // namespace pvhip

#ifdef PVHIP_DIAG
// diagnostic build only: read and clear the cycle accounts of conv_pw_kernel<.., .., 8> (PVHIP_PW_ABLATE=8); out = 8 counters
extern "C" int pvhip_diag_pw_stamps(unsigned long long* out) {
    unsigned long long zero[8] = {0};
    if (hipDeviceSynchronize() != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pw_stamps), sizeof(zero)) != hipSuccess) return PVHIP_EHIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_pw_stamps), zero, sizeof(zero)) != hipSuccess) return PVHIP_EHIP;
    return PVHIP_OK;
}
#endif
