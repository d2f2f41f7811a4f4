// Internal helpers shared by the libpvhip translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/pvhip.h"

namespace pvhip {

constexpr int kMaxStreams = PVHIP_MAX_STREAMS;

struct State {
    bool        ready  = false;
    int         device = -1;
    hipStream_t stream = nullptr;                 // the CURRENT stream: every launch and copy goes here
    hipStream_t streams[kMaxStreams] = {};        // streams[0] is created by pvhip_init, the others on first select
    int         current = 0;
    bool        forked  = false;                  // a stream other than 0 has been used since the last full sync
    bool        capturing = false;                // between pvhip_graph_begin_capture and _end_capture: nothing executes yet
};
State& state();

// Every PVHIP_* environment variable the library understands, parsed ONCE by pvhip_init (and again by
// pvhip_settings_reload, for tests and tuning scripts that flip them): no getenv on any launch path.
struct Settings {
    int generation = 0;          // bumped by every (re)load: host-side plan caches key on it
    // ---- product switches
    int  conv_kernel    = 0;     // PVHIP_CONV_KERNEL (diagnostic build only): 0 = LDS-DMA kernels (default), 1 = "lds" (register-staged), 2 = "wave"
    bool conv_winograd  = true;  // PVHIP_CONV_WINOGRAD=0: direct kernels for the 3x3 / 5x5 layers
    int  conv_winograd4 = 1;     // PVHIP_CONV_WINOGRAD4: 0 off, 1 by size rule, 2 ("force") any size
    int  conv_winograd5 = 1;     // PVHIP_CONV_WINOGRAD5: likewise for F(2x2,5x5)
    bool conv_pointwise = true;  // PVHIP_CONV_POINTWISE=0: 1x1 layers on the general LDS-DMA kernel
    bool conv_stem      = true;  // PVHIP_CONV_STEM=0: a 7x7 / 2 first convolution over three channels on the general LDS-DMA kernel, not from row spans (pvhip_stem.hip)
    bool conv_stem_wino = false; // PVHIP_CONV_STEM_WINO=1: that layer as Winograd F(3x3,4x4) on the space-to-depth image (0.34 of the multiplies, 0.50 ms against 0.56: faster, but not the bits of the general kernel and 0.26 of the MFMA peak on executed flops -- measured, opt-in)
    int  poolconv_prio  = -1;    // PVHIP_POOLCONV_PRIO=0|1|2: no wave priority | priority 3 for the producers | for the consumers of conv_pool1x1_kernel; unset: the producers where they keep two stages of loads in flight (groups of two pixels: 4a .. 4e -7..-8 %; with one stage it cost 11-14 %, on the 28x28 modules it still costs 5-15 %)
    int  fuse_poolconv  = 2;     // PVHIP_FUSE_POOLCONV: 0 off, 1 any width (odd widths: 65-128 output channels, single-pixel groups -- measured slower than the two launches on the 7x7 modules: 0.089 against 0.064 ms), 2 (default) rows of whole 16- or 8-byte groups, 4 only 16-byte groups (A/B runs)
    bool pool3          = true;  // PVHIP_POOL3=0: the one-shot MaxPool kernel for 3x3 windows too
    int  stream_nt      = 1;     // PVHIP_STREAM_NT: nontemporal loads / stores in the streaming kernels: 0 never, 1 from 64 MiB moved on, 2 always
    int  stream_wg      = 16;    // PVHIP_STREAM_WG: workgroups per CU of the grid-stride streaming kernels
    // ---- tuning runs (scripts/): defaults are what the product uses
    int  tile_bm = 0, tile_bn = 0;         // PVHIP_CONV_TILE=BMxBN
    int  wtile_m = 2, wtile_n = 1;         // PVHIP_CONV_WTILE=TMxTN (wave kernel, units of 32)
    int  conv_lds_pad_kb = 0;              // PVHIP_CONV_LDS_PAD_KB: extra dynamic LDS caps workgroups per CU
    bool conv_nopw = false;                // PVHIP_CONV_NOPW: general kernel without its pointwise copy
    int  f16_bm = 0;                       // PVHIP_CONV_F16_BM=32|64|128: channel tile of the f16 LDS-DMA form (tuning runs)
    int  f16_c8_wgs = 0;                   // PVHIP_CONV_F16_C8_WGS=n: pvhip_conv2d_f16_c8 on a persistent grid of n workgroups per CU (0: by layer shape; -1: one tile per workgroup)
    int  f16_c8_prod = 0;                  // PVHIP_CONV_F16_C8_PROD=1..4: producer waves of pvhip_conv2d_f16_c8_multi (0: by the stage's copy instructions)
    int  f16_c8_reg = 0;                   // PVHIP_CONV_F16_C8_REG=1: the producers of pvhip_conv2d_f16_c8_multi copy through registers (tuning runs)
    int  f16_c8_tpw = 0;                   // PVHIP_CONV_F16_C8_TPW=2: pvhip_conv2d_f16_c8_multi's 1x1 launches with two channel tiles per consumer wave (measured slower: off)
    bool lrnpool_wave = false;             // PVHIP_LRNPOOL_WAVE=1: LRN + MaxPool 3x3 / 2 on the barrier-free wave form (measured: 0.180 ms against 0.171 for the workgroup form; same bits)
    int  dwconv_cols = 1;                  // PVHIP_DWCONV_COLS=0: depthwise 3x3 on the one-shot LDS kernel, 2: with an output stage (A/B)
    bool conv_novalid = false;             // PVHIP_CONV_NOVALID: c-major gather with the window test even where no window leaves the tensor (A/B)
    bool conv_pw16 = false;                // PVHIP_CONV_PW: 16-byte gather of the register-staged kernel
    int  multi_bm = 32;                    // PVHIP_CONV_MULTI_BM
    int  pw_tn = 0;                        // PVHIP_PW_TN=1|2: channel tiles per workgroup of the pointwise kernel (0 = by panel size)
    int  pw_stagger_pct = 0;               // PVHIP_PW_STAGGER: start-time stagger of the pointwise kernel, percent of its rule (0 = off)
    int  pool3_kb = 16, pool3_stage = 1, pool3_wg = 0, pool3_g = 0, pool3_s = 0, pool3_band = 0;   // PVHIP_POOL3_KB/_STAGE/_WG/_CFG
    bool pool3_verbose = false;
    int  pool_lds_kb = 16;                 // PVHIP_POOL_LDS_KB
    int  wino_kb = 0;                      // PVHIP_WINO_KB=32|64
    bool wino_small = true;                // PVHIP_WINO_SMALL=0
    int  wino_waves = 8;                   // PVHIP_WINO_WAVES=4
    bool wino_ragged = true;               // PVHIP_WINO_RAGGED=0: the six-point Winograd kernel only where the extents are multiples of its patch (A/B runs)
    bool wino_balance = true;              // PVHIP_WINO_BALANCE=0: fixed producer waves in the six-point Winograd kernel (A/B runs)
    int  wino_shared = 1;                  // PVHIP_WINO_SHARED=0|1|2: the shared-V form of the six-point kernels never / by the tile rule / wherever it applies
    int  wino_shared_prio = 1;             // PVHIP_WINO_SHARED_PRIO=0: no wave priorities in the shared-V form (A/B runs)
    int  wino_shared_old = 1;              // PVHIP_WINO_SHARED_OLD=0: the producers of the shared-V form are the youngest waves (A/B runs)
    bool wino_shared_odd = true;           // PVHIP_WINO_SHARED_ODD=0: the shared-V form only for an even number of channel blocks
    int  wino_shared_lag = 0;              // PVHIP_WINO_SHARED_LAG=1: the second consumer group of the shared-V form starts three stages behind the first (A/B runs: no gain, the ring bounds the stagger anyway)
    int  wino_shared_min_tiles = 2048;      // PVHIP_WINO_SHARED_MIN_TILES: tiles (patch blocks x channel-block pairs) from which the rule picks it for launches of 12-16 stages
    // ---- wrong-on-purpose ablations: honoured only by the diagnostic build (make diag -> libpvhip_diag.so, -DPVHIP_DIAG)
    int  conv_ablate = 0, wino4_ablate = 0, pw_ablate = 0, stem_ablate = 0;
    // PVHIP_TUNE0 .. PVHIP_TUNE7: A/B knobs of the measurements this round kept (0 = the product's choice; every alternative is bit-identical):
    //   0: 32 | 64 = narrower channel tiles of MaxPool + pool_proj          1: 2 = its pixel tiles of 64 instead of 128 (1 = never)
    //   3: shared-V Winograd tile order, 1 = channel-pair-major always, 2 = never     4: 1 = conv1's stores as 16 x 64-byte pieces (not whole rows through LDS)
    //   5: 1 = ragged Winograd patches store 8-byte / 4-byte pieces           6: 1 = LRN + MaxPool pools one output per lane (not four)
    //   2: 3 = the pointwise kernel with three stage buffers (six workgroups per CU; measured neutral in the step)      7: 1 = conv_wino4_kernel's producer waves at priority 0 (not 3)
    int  tune[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool pool3_tuning() const { return pool3_kb != 16 || pool3_stage != 1 || pool3_wg != 0 || pool3_g != 0; }
};
const Settings& settings();
void            load_settings();

// MatMul.py:9-17 on the split-K fp32-MFMA kernel (pvhip_matmul.hip); round_f16: both operands rounded to fp16 first (FP16 IRs).
int matmul_impl(const float* a, const float* b, float* c, int m, int n, int k, int trans_a, int trans_b, int round_f16);

// Records a formatted message for pvhip_last_error() and returns `code`.
int fail(int code, const char* fmt, ...);

constexpr int kWave      = 64;    // CDNA wavefront
constexpr int kNumCU     = 256;   // MI355X
constexpr int kBlock     = 256;   // default workgroup: 4 waves, one per SIMD
constexpr int kMaxBlocks = kNumCU * 8;

inline int grid_for(size_t work_items, int per_block = kBlock) {
    size_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > (size_t)kMaxBlocks) b = kMaxBlocks;
    return (int)b;
}

// Division by a run-time-uniform divisor as multiply-high + shift (exact for n < 2^31).
struct FastDiv {
    unsigned mul, shift, d;
};
inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f{0u, 0u, d};
    if (d <= 1) return f;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;                       // 2^(s-1) < d <= 2^s
    const unsigned long long k = 31ull + s;
    f.mul   = (unsigned)(((1ull << k) + d - 1) / d);   // ceil(2^k / d) in [2^31, 2^32)
    f.shift = s - 1;
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
    return f.d <= 1 ? n : (__umulhi(n, f.mul) >> f.shift);
}

// IEEE 754-2019 maximum of three (gfx950: v_maximum3_f32): a NaN operand gives NaN -- np.max's rule -- in ONE instruction.  v_max3_f32
// is maxNum (a NaN operand LOSES): next to it every window needed unordered compares, an OR chain and a select (a third of the pooling
// arithmetic of conv_pool1x1_kernel, which runs on the SIMDs that run the MFMAs).  Two operands: max3_nan(a, b, b).
// Through the builtin, not inline asm: an asm statement is opaque to hipcc's hazard recognizer, and a value it writes one or two issue
// slots before a v_mfma reads it as an operand arrives STALE (gfx950 wants two wait states between a VALU write and the matrix
// instruction's read; round 5, a pooled value fed straight into v_mfma_f32_32x32x1_2b_f32: three of four pixel columns wrong).
__device__ __forceinline__ float max3_nan(float a, float b, float c) {
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);
}


// Fused bias + activation of the convolution epilogues WITHOUT branches.  `act` (0 none, 1 ReLU, 2 Clamp) is a launch constant,
// but written as `if (act == 1) ... else if (act == 2) ...` per value hipcc kept scalar branches around every one of the 16-32
// values of a lane (the pointwise kernel's epilogue was 1262 instructions with ~300 branches and took a fifth of a workgroup's
// life).  Two compare-and-selects against bounds computed once give the same bits: ReLU is (v < 0) ? 0 : v with hi = +inf,
// none has lo = -inf; NaN and -0.0 pass through as before.  A missing bias is -0.0 (v + -0.0 == v for every v, -0.0 included).
struct ActBounds {
    float lo, hi;
};
__device__ __forceinline__ ActBounds act_bounds(int act, float lo, float hi) {
    ActBounds b;
    b.lo = act == 1 ? 0.0f : (act == 2 ? lo : -__builtin_inff());
    b.hi = act == 2 ? hi : __builtin_inff();
    return b;
}
__device__ __forceinline__ float act_apply(float v, const ActBounds& b) {
    v = (v < b.lo) ? b.lo : v;
    v = (v > b.hi) ? b.hi : v;
    return v;
}
// The same for N accumulator values of one lane, behind WAVE-UNIFORM branches (the activation and the bias pointer are launch
// constants): per value act_apply is two compares and two selects and "bias or not" one more select -- five vector instructions
// of which a ReLU layer with a bias needs two, and every vector instruction of a convolution kernel is matrix time lost
// (profiles/r03_issue_mix.md).  The operations that remain are the same ones on the same values: the same bits.
template <int N>
__device__ __forceinline__ void bias_act_n(float (&v)[N], const float (&b)[N], bool has_bias, int act, const ActBounds& ab) {
    if (has_bias) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = v[i] + b[i];
    }
    if (act == 1) {
        // ReLU as ONE instruction per value (round 5): the IEEE 754-2019 maximum with +0.0 (v_maximum3_f32; a NaN stays a NaN).  It differs
        // from ReLU.py:9-12's np.where(x < 0, 0, x) for one value only, an exact -0.0 (kept there, +0.0 here) -- which a sum that starts from
        // a +0.0 accumulator can never be (+0.0 + -0.0 = +0.0 under round-to-nearest), nor that sum plus a bias (x + y = -0.0 only for
        // x = y = -0.0).  The compare-and-select form is three instructions with its wait state, and vector instructions are matrix time.
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = __builtin_elementwise_maximum(v[i], 0.0f);
    } else if (act != 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = (v[i] < ab.lo) ? ab.lo : v[i];
    }
    if (act == 2) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = (v[i] > ab.hi) ? ab.hi : v[i];
    }
}

// (bias + alpha * sum)^beta of the LRN kernels.  beta_mode: 1 -> d^0.75 as sqrt(d)*sqrt(sqrt(d)) (two correctly
// rounded roots), 2 -> d^0.5, 3 -> d, 0 -> powf.
__device__ __forceinline__ float lrn_pow_f(float d, float beta, int beta_mode) {
    if (beta_mode == 1) {
        const float s = sqrtf(d);
        return s * sqrtf(s);
    }
    if (beta_mode == 2) return sqrtf(d);
    if (beta_mode == 3) return d;
    return powf(d, beta);
}

// x / (bias + alpha * sum)^beta.  Mode 4 is d^-0.75 = rsq(d) * rsq(sqrt(d)) on the hardware's 1-ulp square-root
// instructions (3 transcendental issues instead of two IEEE square roots and an IEEE division, ~1/3 of the VALU work of
// an LRN element; a few ulp from the exact quotient, far inside the 1e-4 of the path).  It is chosen on the host only
// when bias keeps d in the normal range, where those instructions are accurate.
__device__ __forceinline__ float lrn_div(float x, float d, float beta, int beta_mode) {
    // d^-0.75 = 2^(-0.75 log2 d): TWO transcendental issues (v_log_f32, v_exp_f32; ~1 ulp each: 3e-7 relative for d up to 100) where
    // rsq(d) * rsq(sqrt(d)) took three -- they are quarter-rate, and the LRN + MaxPool launch is vector-ALU-bound beside its stream
    if (beta_mode == 4) return x * __builtin_amdgcn_exp2f(-0.75f * __builtin_amdgcn_logf(d));
    return x / lrn_pow_f(d, beta, beta_mode);
}

inline int lrn_beta_mode(float beta, float bias) {
    if (beta == 0.75f) return bias >= 1e-20f ? 4 : 1;
    if (beta == 0.5f) return 2;
    if (beta == 1.0f) return 3;
    return 0;
}

// Output stores of the convolution epilogues: ORDINARY stores.  Nontemporal ones (kConvStoreNT = true) were measured in round 3, on the
// idea that the 4 MB a round of tiles writes per XCD evicts the input patches and weight fragments its workgroups re-read from the 4 MB
// L2 they share (profiles/r03a: the six-point Winograd kernel fetches 3.5x its input from HBM): -5 % images/s, the Winograd
// F(4x4,3x3) launches 2.05 -> 2.38 ms, the pointwise sibling launches +2 %, conv1 unchanged -- a lane's 16-byte pieces of different
// rows are merged into whole lines by L2 when they are ordinary stores and go to memory one by one when they are not.  The switch stays
// for A/B builds.
constexpr bool kConvStoreNT = false;
// kConvStoreSC1: write-through stores that do not keep their line in the XCD's L2 (`sc1`; MI355X guide, "stores of each flavour"), the
// other way to keep a round's output from evicting inputs and weights.  Measured too: -3.5 % images/s (Winograd 1.90 -> 2.16 ms, the
// others +2..4 %).  The output rows of these layers are 224 / 112 / 56 / 28 bytes long: a 128-byte line holds pieces of several rows
// that different store instructions write, and only ordinary stores are merged in L2 before they leave it.  asm statements (a store
// has no destination register: nothing for hipcc to copy early); off.
constexpr bool kConvStoreSC1 = false;
typedef float pv_f4v __attribute__((ext_vector_type(4)));
typedef float pv_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void conv_store1(float* p, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (kConvStoreSC1) { asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); return; }
#endif
    if (kConvStoreNT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
__device__ __forceinline__ void conv_store2(float* p, float a, float b) {
    pv_f2v v; v.x = a; v.y = b;
#if defined(__HIP_DEVICE_COMPILE__)
    if (kConvStoreSC1) { asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); return; }
#endif
    if (kConvStoreNT) __builtin_nontemporal_store(v, reinterpret_cast<pv_f2v*>(p));
    else *reinterpret_cast<pv_f2v*>(p) = v;
}
__device__ __forceinline__ void conv_store4(float* p, float a, float b, float c, float d) {
    pv_f4v v; v.x = a; v.y = b; v.z = c; v.w = d;
#if defined(__HIP_DEVICE_COMPILE__)
    if (kConvStoreSC1) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); return; }
#endif
    if (kConvStoreNT) __builtin_nontemporal_store(v, reinterpret_cast<pv_f4v*>(p));
    else *reinterpret_cast<pv_f4v*>(p) = v;
}
template <class V>
__device__ __forceinline__ void conv_storev(V* p, V v) {       // V: a float ext_vector_type of 1, 2 or 4
#if defined(__HIP_DEVICE_COMPILE__)
    if (kConvStoreSC1) {
        if constexpr (sizeof(V) == 4) { const float f_ = v[0]; asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(f_) : "memory"); }
        else if constexpr (sizeof(V) == 8) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
        return;
    }
#endif
    if (kConvStoreNT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// Global -> LDS loads (LDS-DMA), written as asm statements on purpose: hipcc treats the builtin form as a store to all
// of LDS and puts s_waitcnt vmcnt(0) in front of the next ds_read, which would serialise the loads of stage t+1 with the
// MFMAs of stage t.  The asm loads are invisible to its counters; a kernel waits for them itself (lds_dma_wait_all, or a
// counted s_waitcnt) before the barrier that publishes the stage.  `dst` is wave-uniform: lane l's bytes land at dst + l*4
// (b32) or dst + l*16 (b128); an out-of-range source offset writes 0.
typedef __attribute__((address_space(3))) void* lds_void_p;
__device__ __forceinline__ void lds_dma_b32(__amdgpu_buffer_rsrc_t r, float* dst, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lds = (unsigned)(unsigned long)(lds_void_p)dst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
#endif
}
__device__ __forceinline__ void lds_dma_b128(__amdgpu_buffer_rsrc_t r, float* dst, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lds = (unsigned)(unsigned long)(lds_void_p)dst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
#endif
}
__device__ __forceinline__ void lds_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

}  // namespace pvhip

#define PVHIP_REQUIRE_INIT()                                                          \
    do {                                                                              \
        if (!pvhip::state().ready)                                                    \
            return pvhip::fail(PVHIP_ENOTINIT, "%s: pvhip_init() not called", __func__); \
    } while (0)

#define PVHIP_HIP(call)                                                               \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess)                                                         \
            return pvhip::fail(PVHIP_EHIP, "%s: %s -> %s", __func__, #call,           \
                               hipGetErrorString(e_));                                \
    } while (0)

#define PVHIP_CHECK_ARG(cond)                                                         \
    do {                                                                              \
        if (!(cond))                                                                  \
            return pvhip::fail(PVHIP_EINVAL, "%s: argument check failed: %s",         \
                               __func__, #cond);                                      \
    } while (0)

// After a kernel launch: surface launch-configuration errors immediately.
#define PVHIP_LAUNCH_CHECK()                                                          \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess)                                                         \
            return pvhip::fail(PVHIP_EHIP, "%s: kernel launch -> %s", __func__,       \
                               hipGetErrorString(e_));                                \
    } while (0)
