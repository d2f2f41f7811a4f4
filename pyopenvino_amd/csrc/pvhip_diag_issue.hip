// DIAGNOSTIC BUILD ONLY (make diag -> libpvhip_diag.so): what shares the SIMD's vector issue with what.  scripts/issue_mix.py runs
// MFMA streams (fp32 32x32x2 or bf16 32x32x16) and packed-fp32 vector streams alone, interleaved in one wave, and on different waves of
// one SIMD, and prints the times: "fp32 MFMA + vector = the sum" is what the convolution kernels (Convolution.py:57-87 replacements) are
// built around (LESSONS.md lesson 1); whether the bf16 matrix instructions behave the same decides what a split-bf16 contraction
// (fp32 operands as three bf16 terms, six products, fp32 accumulation) could buy.  Not part of the product library.
#include "pvhip_common.h"

using namespace pvhip;

namespace {

typedef float  floatx16 __attribute__((ext_vector_type(16)));
typedef float  float2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// One iteration: 8 fp32 MFMAs (8 x 64 cycles) or 16 bf16 MFMAs (16 x 8 passes) on four independent accumulators, and / or 64 packed FMAs
// on 16 independent chains.  split = 1: waves 0-3 of the workgroup run the MFMAs and waves 4-7 the vector stream (waves w and w + 4
// share a SIMD); split = 0: every wave runs what is asked for, interleaved by the compiler's scheduler hints.
// VALU: 0 none, 1: 64 v_pk_fma_f32, 2: 128 v_fma_f32 (the same arithmetic unpacked), 3: 64 v_fma_f32, 4: 64 v_max3_f32, 5: 64 v_mov_b32 dpp
template <int MFMA, int VALU, bool SPLIT>       // MFMA: 0 none, 1 fp32 32x32x2, 2 bf16 32x32x16
__global__ __launch_bounds__(512) void issue_mix_kernel(float* out, int iters) {
    const int  wave    = threadIdx.x >> 6;
    const bool do_mfma = MFMA != 0 && (!SPLIT || wave < 4);
    const bool do_valu = VALU != 0 && (!SPLIT || wave >= 4);
    constexpr int NV = VALU == 2 ? 128 : 64;
    floatx16 acc[4];
    float2v  ch[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) ch[i] = float2v{(float)threadIdx.x * 1e-3f + i, 1.0f};
    const float   af = (float)(threadIdx.x & 7) * 0.125f, bf = 1.0f + (float)(threadIdx.x & 3);
    bf16x8        a8, b8;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(af + i); b8[i] = (__bf16)(bf - i); }
    const float2v m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
    // asm statements: the streams are exactly what is written here (as builtins hipcc turned most packed FMAs into scalar ones and
    // added 80 moves per iteration to honour the scheduling hints).  Four accumulators in turn: an MFMA never waits for the previous one.
#define PV_MFMA(i_)                                                                                                             \
    {                                                                                                                           \
        if (MFMA == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i_]) : "v"(af), "v"(bf));              \
        else           asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i_]) : "v"(a8), "v"(b8));           \
    }
#define PV_PKFMA(i_)                                                                                                            \
    {                                                                                                                           \
        if (VALU == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(ch[(i_) & 15]) : "v"(m), "v"(c));                       \
        else if (VALU == 2 || VALU == 3) {                                                                                      \
            if ((i_) & 16) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ch[(i_) & 15].y) : "v"(m.y), "v"(c.y));               \
            else           asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ch[(i_) & 15].x) : "v"(m.x), "v"(c.x));               \
        } else if (VALU == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(ch[(i_) & 15].x) : "v"(m.x), "v"(c.x));           \
        else asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(ch[(i_) & 15].x) : "v"(ch[((i_) + 1) & 15].y)); \
    }
    for (int it = 0; it < iters; ++it) {
        if (do_mfma && !do_valu) {
#pragma unroll
            for (int q = 0; q < (MFMA == 1 ? 8 : 16); ++q) PV_MFMA(q & 3);
        } else if (do_valu && !do_mfma) {
#pragma unroll
            for (int q = 0; q < NV; ++q) PV_PKFMA(q & 31);
        } else if (do_mfma && do_valu) {
            constexpr int NM = MFMA == 1 ? 8 : 16, PER = NV / NM;      // one MFMA, then its share of the vector instructions
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                PV_MFMA(q & 3);
#pragma unroll
                for (int e = 0; e < PER; ++e) PV_PKFMA((q * PER + e) & 31);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results are read below (hipcc does not see MFMAs in asm)
#undef PV_MFMA
#undef PV_PKFMA
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += ch[i].x + ch[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MFMA, int VALU>
void launch_mix(float* out, int iters, int split, int blocks) {
    if (split) hipLaunchKernelGGL((issue_mix_kernel<MFMA, VALU, true>), dim3(blocks), dim3(512), 0, state().stream, out, iters);
    else       hipLaunchKernelGGL((issue_mix_kernel<MFMA, VALU, false>), dim3(blocks), dim3(512), 0, state().stream, out, iters);
}

}  // namespace

// mfma: 0 none, 1 fp32 32x32x2, 2 bf16 32x32x16; valu: 0 none, 1: 64 packed fp32 FMAs per iteration, 2-5: see the kernel; split: MFMAs on waves 0-3, vector stream on
// waves 4-7 of each 8-wave workgroup (else every wave runs both); out: blocks * 512 floats.
extern "C" int pvhip_diag_issue_mix(float* out, int mfma, int valu, int split, int iters, int blocks) {
    if (out == nullptr || iters <= 0 || blocks <= 0 || mfma < 0 || mfma > 2) return fail(PVHIP_EINVAL, "pvhip_diag_issue_mix: bad arguments");
    if (mfma == 0 && !valu) return fail(PVHIP_EINVAL, "pvhip_diag_issue_mix: nothing to run");
    if (valu < 0 || valu > 5) return fail(PVHIP_EINVAL, "pvhip_diag_issue_mix: valu 0..5");
#define PV_MIX(M_)                                                           \
    switch (valu) {                                                          \
        case 0: launch_mix<M_, 0>(out, iters, split, blocks); break;         \
        case 1: launch_mix<M_, 1>(out, iters, split, blocks); break;         \
        case 2: launch_mix<M_, 2>(out, iters, split, blocks); break;         \
        case 3: launch_mix<M_, 3>(out, iters, split, blocks); break;         \
        case 4: launch_mix<M_, 4>(out, iters, split, blocks); break;         \
        default: launch_mix<M_, 5>(out, iters, split, blocks); break;        \
    }
    if (mfma == 0) PV_MIX(0) else if (mfma == 1) PV_MIX(1) else PV_MIX(2)
#undef PV_MIX
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}
