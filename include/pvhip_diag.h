/* Entry points of the DIAGNOSTIC build only (make -C pyopenvino_amd/csrc diag -> libpvhip_diag.so, compiled with -DPVHIP_DIAG).
 *
 * libpvhip_diag.so exports everything include/pvhip.h declares plus what is declared here: measurement probes, the predecessor
 * convolution kernels kept for A/B runs (PVHIP_CONV_KERNEL=lds|wave, PVHIP_CONV_TILE, PVHIP_CONV_PW), the wrong-on-purpose
 * ablation paths (PVHIP_*_ABLATE) and in-kernel cycle stamps.  None of it is in the product library; nothing in pyopenvino_amd/
 * needs it.  Users: bench.py (roofline.sustained), scripts/, tests/diag_variants.py.                                          */
#ifndef PVHIP_DIAG_H
#define PVHIP_DIAG_H
#include "pvhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement utility, no reference counterpart: TFLOP/s this device SUSTAINS on v_mfma_f32_32x32x2_f32 alone (one wave per
 * SIMD, operands in registers, random data) and the shader clock it holds meanwhile (s_memtime / s_memrealtime) -- the chip
 * lowers its clock under matrix load, so this is the ceiling bench.py quotes beside the 157.3 TFLOP/s of 2.4 GHz.
 * mode 1: a second wave per SIMD issues v_fma_f32 only (fp32 matrix and vector instructions of a SIMD do not overlap);
 * mode 2: the v_mfma_f32_16x16x4_f32 shape (same flops per cycle; measured: the same sustained rate).  The clock ramps up over
 * the first milliseconds after idle: call it a few times and take the best.                                                 */
int pvhip_mfma_ceiling_f32(int mode, int iters, double* tflops, double* clock_ghz);

/* scripts/sweep_stream.py: one float4 copy (relu == 0) or ReLU stream over n floats (n % 4 == 0) with `unroll` (1|2|4|8) 16-byte loads per
 * lane issued before the first store; layout 0: the loads of a lane a whole grid apart, 1: adjacent pieces of the lane's workgroup, 2:
 * adjacent words of the lane; nt != 0: nontemporal loads and stores; blocks x threads as given.                                  */
int pvhip_diag_stream_f32(const float* x, float* y, unsigned long long n, int relu, int unroll, int nt, int layout, int blocks, int threads);

/* scripts/issue_mix.py: MFMA streams (mfma: 0 none, 1 fp32 32x32x2, 2 bf16 32x32x16) and / or a packed-fp32 vector stream (valu != 0),
 * in every wave (split = 0) or the MFMAs on waves 0-3 and the vector stream on waves 4-7 of each 8-wave workgroup (split = 1: one of each
 * per SIMD); iters iterations of 8 fp32 / 16 bf16 MFMAs and 64 packed FMAs; out: blocks * 512 floats.                                 */
int pvhip_diag_issue_mix(float* out, int mfma, int valu, int split, int iters, int blocks);

/* scripts/time_pw.py --ablate 8: read and clear the cycle accounts of conv_pw_kernel<.., .., 8> (PVHIP_PW_ABLATE=8); out = 8 counters. */
int pvhip_diag_pw_stamps(unsigned long long* out);

/* scripts/stamps_wino4.py (PVHIP_WINO4_ABLATE=5): where the waves of conv_wino4_kernel ran (64 x 8 x 2 words of HW_ID / LDS_ALLOC; the
 * ticket counter is reset), and its s_memtime stamps (64 counters; cleared).                                                        */
/* scripts/stamps_poolconv.py (PVHIP_CONV_ABLATE=64): s_memtime accounts of conv_pool1x1_kernel, workgroup 77, summed over its stages: out[0] stages, [1] producer
 * pooling (incl. waiting for its loads), [2] producer barrier, [3] producer loop, [4] consumer MFMA section, [5] its wait for the weight copy, [6] consumer barrier,
 * [7] consumer loop (cycles).                                                                                                            */
int pvhip_diag_poolconv_stamps(unsigned long long* out);
int pvhip_diag_wino4_hw(unsigned* out);
int pvhip_diag_wino4_stamps(unsigned long long* out);
int pvhip_diag_wino4s_stamps(unsigned long long* out);
/* conv_f16_c8_kernel: 16 cycle accounts summed over every 16th workgroup (pvhip_f16c8.hip g_c8_stamps), read and cleared */
int pvhip_diag_c8_stamps(unsigned long long* out);
int pvhip_diag_wino4s_simd(unsigned long long* out);
int pvhip_diag_wino4s_trace(unsigned* out);                  /* workgroup 3 of the last launch: [wave][stage < 96][saw the image / finished] in cycles since its start */        /* the same workgroups: 16 waves x 4 SIMDs, how often wave w ran on SIMD s */      /* conv_wino4s_kernel (shared-V form): 16 waves x 8 cycle accounts, read and cleared */
/* the same run's epilogue phases: [wave 0..7][write 0, barrier, read + store 0, barrier, write 1, barrier, read + store 1, barrier]  */
int pvhip_diag_wino4_epilogue(unsigned long long* out);

#ifdef __cplusplus
}
#endif
#endif
