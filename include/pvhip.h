/*
 * pvhip.h -- C ABI of libpvhip.so: the MI355X (gfx950) numeric back end that sits under the
 * pyopenvino op-plugin boundary `compute(node, inputs, kernel_type, debug)`.
 *
 * The reference (yas-sim/pyopenvino) is pure Python and has no FFI of its own, so these entry points
 * are what a binding for its per-layer hot path binds: one function per `kernel_<Op>_*` body of the
 * reference plugins, plus device memory / stream / event plumbing and one RCCL gather.  Each
 * declaration cites the reference function it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns 0 on success, a negative PVHIP_E* code otherwise; it never throws and
 *     never synchronises the stream unless its comment says so; pvhip_last_error() returns a
 *     human-readable description of the last failure on this thread.
 *   - all tensor pointers are DEVICE pointers obtained from pvhip_malloc, contiguous, fp32 unless
 *     noted; activations are NCHW, convolution weights OIHW, MatMul operands row-major 2-D.
 *   - every kernel is enqueued on the library's single compute stream of the device selected by
 *     pvhip_init (one host thread / process per GPU).
 *   - sizes are element counts (not bytes) unless the name says bytes.
 */
#ifndef PVHIP_H
#define PVHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVHIP_OK            0
#define PVHIP_EHIP         -1   /* a HIP runtime call failed                                  */
#define PVHIP_EINVAL       -2   /* argument rejected on the host (shape / attribute check)    */
#define PVHIP_ENOTINIT     -3   /* pvhip_init has not been called                             */
#define PVHIP_ECOMM        -4   /* RCCL failure / library not loadable                        */
#define PVHIP_EUNSUPPORTED -5   /* configuration outside what the kernels implement           */

#define PVHIP_ABI_VERSION   16

/* ---------------------------------------------------------------- runtime plumbing ---------- */
/* No reference counterpart: the reference computes in host numpy arrays (inference_engine.py:245-256
 * hands ndarray between plugins).  These give plugins a device-resident tensor to hand over instead. */
int         pvhip_abi_version(void);
const char* pvhip_last_error(void);
int         pvhip_device_count(int* count);
int         pvhip_init(int device);                     /* select device, create the compute stream, read the PVHIP_* variables */
int         pvhip_settings_reload(void);                /* read the PVHIP_* environment variables again (they are parsed once, by
                                                           pvhip_init, never on a launch path): tests and tuning scripts that flip one */
int         pvhip_shutdown(void);                       /* free pool, destroy stream                 */
int         pvhip_device_name(char* buf, size_t buflen);
int         pvhip_malloc(void** ptr, size_t bytes);     /* pooled: freed blocks are reused by size   */
int         pvhip_free(void* ptr);                      /* returns the block to the pool             */
int         pvhip_pool_release(void);                   /* hipFree everything cached in the pool     */
int         pvhip_pool_stats(size_t* bytes_in_use, size_t* bytes_cached);
/* Allocation epochs, for forward passes that run on several streams or asynchronously (several requests in flight):
 * blocks allocated between _begin and _dispatched belong to the epoch; one of them that is freed while the epoch is
 * open waits until _end (called once the pass is known to have finished on the device) instead of being handed out
 * again at once; blocks of ended epochs -- the previous outputs a new pass replaces -- are reusable immediately. */
int         pvhip_pool_epoch_begin(int* epoch);
int         pvhip_pool_epoch_dispatched(void);          /* later allocations belong to no epoch */
int         pvhip_pool_epoch_end(int epoch);
int         pvhip_memcpy_h2d(void* dst, const void* src, size_t bytes);  /* Parameter.py:11-13, Const.py:11-13 upload; async w.r.t. device, host buffer reusable on return */
int         pvhip_memcpy_d2h(void* dst, const void* src, size_t bytes);  /* Result.py:17 read-back; SYNCHRONISES the stream */
void*       pvhip_host_alloc(size_t bytes);   /* ABI v16: page-locked host memory for read-backs (NULL when it cannot be had: use pageable memory) */
int         pvhip_host_free(void* p);
int         pvhip_memcpy_d2d(void* dst, const void* src, size_t bytes);
int         pvhip_memset(void* dst, int byte, size_t bytes);
int         pvhip_sync(void);                           /* host-side wait for every compute stream   */

/* Compute streams.  Every launch and copy goes to the CURRENT stream (stream 0 after pvhip_init).  The
 * scheduler may put independent branches of the graph (inference_engine.py:218-242 orders them serially)
 * on up to 8 streams (index 8 is for copies and the RCCL gather of requests in flight) and order them with untimed events; blocks freed while a stream other than 0 has been
 * used are handed out again only after the next full synchronisation (pvhip_sync, or pvhip_memcpy_d2h
 * issued on stream 0). */
#define PVHIP_MAX_STREAMS 9                             /* 8 compute streams + 1 for copies and the gather        */
int         pvhip_stream_select(int index);             /* make stream `index` current (created on first use) */
int         pvhip_stream_wait_event(void* ev);          /* current stream waits for a recorded event  */
int         pvhip_event_create_untimed(void** ev);      /* ordering-only event (no timestamps)        */

/* events on the compute stream: replace the per-node time.time() bracket of inference_engine.py:279-283 */
int         pvhip_event_create(void** ev);
int         pvhip_event_destroy(void* ev);
int         pvhip_event_record(void* ev);
int         pvhip_event_sync(void* ev);
int         pvhip_event_elapsed_ms(void* start, void* stop, float* ms);

/* hipGraph capture of one whole forward pass (the run_tasks loop, inference_engine.py:259-292): between _begin_capture and
 * _end_capture every launch on the current stream -- and on the streams that join it through events -- is recorded instead of
 * executed; pvhip_graph_launch replays the whole pass with one call.  While a capture is open pvhip_malloc must be served
 * by the pool (run the pass eagerly first) and freed blocks stay pinned until pvhip_graph_destroy: the captured kernels hold
 * their addresses.  No host-synchronising call (pvhip_sync, _memcpy_h2d / _d2h) may be made inside a capture.            */
int         pvhip_graph_begin_capture(void);
int         pvhip_graph_capture_status(int* status);    /* current stream: 0 not capturing, 1 capturing, 2 capture invalidated by an illegal call */
int         pvhip_graph_end_capture(void** graph_exec);
int         pvhip_graph_launch(void* graph_exec);
int         pvhip_graph_destroy(void* graph_exec);

/* ---------------------------------------------------------------- streaming elementwise ----- */
/* ReLU.py:9-12  kernel_ReLU_numpy: y = (x < 0) ? 0 : x   (NaN and -0.0 pass through)            */
int pvhip_relu_f32(const float* x, float* y, size_t n);
/* Clamp.py:9-12 kernel_Clamp_numpy: y = min(max(x, lo), hi), NaN propagates                     */
int pvhip_clamp_f32(const float* x, float* y, size_t n, float lo, float hi);
/* Sigmoid.py:10-13 kernel_Sigmoid_numpy: y = 1 / (1 + exp(-x))                                  */
int pvhip_sigmoid_f32(const float* x, float* y, size_t n);

/* Add.py:9-14 kernel_Add_numpy / Multiply.py:9-17 kernel_Multiply_numpy.
 * out = a (op) b with numpy broadcasting already resolved by the caller into element strides:
 * `shape` is the output shape (rank <= PVHIP_MAX_RANK), a_strides/b_strides are element strides of
 * the operands viewed at the output shape (0 on broadcast axes).  The library picks a float4
 * streaming kernel for the common cases (same shape; per-channel (1,C,1,1); trailing-row (1,F);
 * scalar) and a generic strided kernel otherwise.                                               */
#define PVHIP_MAX_RANK 6
int pvhip_add_f32(const float* a, const float* b, float* out, int rank,
                  const int64_t* shape, const int64_t* a_strides, const int64_t* b_strides);
int pvhip_mul_f32(const float* a, const float* b, float* out, int rank,
                  const int64_t* shape, const int64_t* a_strides, const int64_t* b_strides);

/* ---------------------------------------------------------------- pooling / normalisation --- */
/* MaxPool.py:41-72 kernel_MaxPool_numpy: max over the kh x kw window of the ZERO-padded input
 * (pad cells take part with value 0), window clipped at the padded extent.  (oh, ow) are computed
 * by the caller with the rule of MaxPool.py:10-38.                                               */
int pvhip_maxpool2d_f32(const float* x, float* y, int n, int c, int h, int w, int oh, int ow,
                        int kh, int kw, int sh, int sw, int pad_top, int pad_left,
                        int pad_bottom, int pad_right);
/* AvgPool.py:41-59 kernel_AvgPool_numpy: mean of x[y*sh : min(h-1, y*sh+kh), x*sw : min(w-1, x*sw+kw)]
 * -- no padding, window clipped at h-1 / w-1 (reference behaviour, kept).  An empty window yields NaN. */
int pvhip_avgpool2d_f32(const float* x, float* y, int n, int c, int h, int w, int oh, int ow,
                        int kh, int kw, int sh, int sw);
/* SoftMax.py:10-14 kernel_SoftMax_numpy applied per row: y[r,:] = exp(x[r,:]) / sum(exp(x[r,:])),
 * no max subtraction (identical to the reference at batch 1; rows are independent images).      */
int pvhip_softmax_rows_f32(const float* x, float* y, int rows, int cols);
/* LRN.py:10-22 kernel_LRN_numpy: y = x / (bias + alpha * sum_{c' in [c-size/2, c+size/2]} x^2)^beta,
 * window clipped to [0, C); alpha is NOT divided by size.  hw = H*W.                             */
int pvhip_lrn_f32(const float* x, float* y, int n, int c, int hw, int size,
                  float alpha, float beta, float bias);
/* LRN.py:10-22 followed by MaxPool.py:41-72 (3x3 window, stride 1 or 2) as ONE launch: y = maxpool(lrn(x)) with the
 * arithmetic and the pooling rules of the two entries above (bit-identical to calling them in turn); the LRN tensor is
 * never written.  x is (n, c, h, w), y is (n, c, oh, ow).  Covers size == 5, beta == 0.75, c % 8 == 0 and bands of
 * input rows that fit one workgroup; pvhip_lrn_maxpool_supported() (no device needed) tells whether a shape is covered,
 * the compute entry fails with PVHIP_EUNSUPPORTED otherwise.                                                        */
int pvhip_lrn_maxpool_supported(int n, int c, int h, int w, int size, float beta, float bias, int oh, int ow,
                                int kh, int kw, int sh, int sw, int pad_top, int pad_left, int pad_bottom, int pad_right);
int pvhip_lrn_maxpool_f32(const float* x, float* y, int n, int c, int h, int w, int size, float alpha, float beta,
                          float bias, int oh, int ow, int kh, int kw, int sh, int sw, int pad_top, int pad_left,
                          int pad_bottom, int pad_right);
/* The other order, MaxPool.py:41-72 (3x3 window, stride 1 or 2) followed by LRN.py:10-22, as ONE launch: y = lrn(maxpool(x)),
 * bit-identical to calling pvhip_maxpool2d_f32 then pvhip_lrn_f32; the pooled tensor is never written.  x is (n, c, h, w), y is
 * (n, c, oh, ow).  Same coverage as pvhip_lrn_maxpool_f32 (size == 5, beta == 0.75, c % 8 == 0, bands that fit one workgroup).   */
int pvhip_maxpool_lrn_supported(int n, int c, int h, int w, int oh, int ow, int kh, int kw, int sh, int sw, int pad_top,
                                int pad_left, int pad_bottom, int pad_right, int size, float beta, float bias);
int pvhip_maxpool_lrn_f32(const float* x, float* y, int n, int c, int h, int w, int oh, int ow, int kh, int kw, int sh, int sw,
                          int pad_top, int pad_left, int pad_bottom, int pad_right, int size, float alpha, float beta, float bias);
/* ... and with the 1x1 / stride 1 / unpadded convolution behind the LRN folded in as well (ABI v15; MaxPool.py:41-72, LRN.py:10-22,
 * Convolution.py:57-87 + the fused bias / activation of pvhip_conv2d_f32; GoogLeNet's pool1/3x3_s2 -> pool1/norm1 -> conv2/3x3_reduce): the
 * normalised value a lane would have stored is its column of a k = 1 outer-product MFMA with that channel's weights -- neither the pooled nor
 * the normalised tensor exists.  Rows of a multiple of four pixels, bands of at most 256 pooled outputs, c <= 64, k_out <= 64; w_oihw: the
 * (k_out, c, 1, 1) weights as they are; y: (n, k_out, oh, ow).  The reduction runs over the input channels in ascending order.                */
int pvhip_maxpool_lrn_conv1x1_supported(int n, int c, int h, int w, int oh, int ow, int kh, int kw, int sh, int sw, int pad_top,
                                        int pad_left, int pad_bottom, int pad_right, int size, float beta, float bias, int k_out);
int pvhip_maxpool_lrn_conv1x1_f32(const float* x, const float* w_oihw, float* y, int n, int c, int h, int w, int oh, int ow, int kh, int kw,
                                  int sh, int sw, int pad_top, int pad_left, int pad_bottom, int pad_right, int size, float alpha,
                                  float beta, float bias, int k_out, const float* conv_bias, int act, float act_lo, float act_hi);

/* ---------------------------------------------------------------- data movement ------------- */
/* Concat.py:9-13 kernel_Concat_numpy: srcs[i] is viewed as [outer][inner[i]] and copied to
 * dst[outer][sum(inner)] at its running offset.  srcs / inner are HOST arrays of n_src entries.  */
#define PVHIP_MAX_CONCAT 16
int pvhip_concat_f32(int n_src, const float* const* srcs, const int64_t* inner, float* dst, int64_t outer);
/* The zero-padded image Convolution.py:64-66 builds before it slides its window, as a tensor: y (n, c, h + pad_top + pad_bottom,
 * w + pad_left + pad_right) = x (n, c, h, w) with zeros around; channel_add (c floats, may be NULL) is added to every element of x on
 * the way (Add.py:9-14 of a per-channel constant feeding the convolution: one pass instead of two).  The Convolution plugin pads the
 * input of a layer whose channel count is not a multiple of 16 (GoogLeNet conv1) and convolves the result without padding: the
 * kernel's gather then needs no window test (pvhip_conv2d_f32 picks that form whenever no window leaves the tensor).             */
int pvhip_pad2d_f32(const float* x, float* y, int n, int c, int h, int w, int pad_top, int pad_left, int pad_bottom, int pad_right,
                    const float* channel_add);
/* Transpose.py:9-13 kernel_Transpose_numpy, materialised: y = x.transpose(perm), y contiguous.   */
int pvhip_transpose_f32(const float* x, float* y, int rank, const int64_t* in_shape, const int64_t* perm);

/* ---------------------------------------------------------------- MFMA kernels -------------- */
/* MatMul.py:9-17 kernel_MatMul_numpy: C[M,N] = op(A) . op(B); A is stored [M,K] (or [K,M] when
 * trans_a), B is stored [K,N] (or [N,K] when trans_b); fp32 MFMA (v_mfma_f32_32x32x2_f32).        */
int pvhip_matmul_f32(const float* a, const float* b, float* c, int m, int n, int k,
                     int trans_a, int trans_b);


/* Convolution.py:57-87 im2col + kernel_Convolution_im2col ("special"), as an implicit GEMM:
 *   y[n,k,oy,ox] = sum_{c,r,s} xpad[n,c,oy*sh+r,ox*sw+s] * w[k,c,r,s]      (dilation ignored, as :72-87 does)
 * Step 1 (once per weight tensor and input extent h x w): repack OIHW weights to the K-major panel the
 *   kernel streams and build the per-reduction-row gather table (byte offset of (c, r, s) in an h x w
 *   image).  wpack must hold pvhip_conv2d_pack_elems(k_out, c, kh, kw) floats.
 * Step 2: the convolution proper.  (oh, ow) computed by the caller per Convolution.py:21-49.
 *   bias (optional, may be NULL): per-output-channel value added in the epilogue; relu == 1 applies the
 *   ReLU.py:11 rule, relu == 2 the Clamp.py:11 rule with [act_lo, act_hi] (fused Convolution->Add->ReLU/Clamp).
 *   out_channels_total > 0: y points at a tensor [n, out_channels_total, oh, ow] and this convolution
 *   writes channels [out_channel_offset, out_channel_offset + k_out) of it -- the Concat.py:9-13 copy of an
 *   inception output done by the producer; 0 = y is the dense [n, k_out, oh, ow] result.            */
size_t pvhip_conv2d_pack_elems(int k_out, int c, int kh, int kw);
int    pvhip_conv2d_pack_f32(const float* w_oihw, float* wpack, int k_out, int c, int kh, int kw, int h, int w);
int    pvhip_conv2d_f32(const float* x, const float* wpack, float* y,
                        int n, int c, int h, int w, int k_out, int kh, int kw, int oh, int ow,
                        int sh, int sw, int pad_top, int pad_left,
                        const float* bias, int relu,
                        int out_channel_offset, int out_channels_total,
                        float act_lo, float act_hi);
/* Which kernel family pvhip_conv2d_f32 dispatches this geometry to (no device needed; honours the PVHIP_* switches): the
 * benchmark's roofline needs it, because the Winograd families execute a fraction of the algorithmic multiplies on the
 * matrix cores -- F(2x2,3x3) 16/36, F(4x4,3x3) 36/144, F(2x2,5x5) 36/100 -- and the direct families all of them.         */
#define PVHIP_CONV_KIND_IGEMM        0   /* implicit GEMM, LDS-DMA tiles (pvhip_conv.hip)                    */
#define PVHIP_CONV_KIND_POINTWISE    1   /* 1x1 / stride 1: fragment-ordered weights (pvhip_pw.hip)          */
#define PVHIP_CONV_KIND_WINO_F2_3X3  2
#define PVHIP_CONV_KIND_WINO_F4_3X3  3
#define PVHIP_CONV_KIND_WINO_F2_5X5  4
#define PVHIP_CONV_KIND_STEM         5   /* ABI v15: 7x7 / 2 over three channels from row spans (pvhip_stem.hip)  */
#define PVHIP_CONV_KIND_STEM_WINO    6   /* ABI v16: the same layer as Winograd F(3x3,4x4) on the space-to-depth image  */
int    pvhip_conv2d_kernel_kind(int n, int c, int h, int w, int k_out, int kh, int kw, int oh, int ow,
                                int sh, int sw, int pad_top, int pad_left);
/* MaxPool.py:41-72 (3x3 window, stride 1, pad 1 all round: output extent = input extent) followed by a 1x1 / stride 1 / unpadded
 * Convolution.py:57-87, as one launch: y = conv1x1(maxpool(x)); the pooled tensor is never written.  Bit-identical to
 * pvhip_maxpool2d_f32 followed by pvhip_conv2d_f32.  x is (n, c, h, w); wpack the pvhip_conv2d_pack_f32 panel of the (k_out, c, 1, 1)
 * weights; bias / act / out_channel_offset / out_channels_total / act_lo / act_hi as for pvhip_conv2d_f32.  Covers c % 16 == 0,
 * even w, k_out <= 128 (pvhip_conv2d_pooled_supported; no device needed); PVHIP_EUNSUPPORTED otherwise.                           */
int    pvhip_conv2d_pooled_supported(int n, int c, int h, int w, int k_out);
int    pvhip_conv2d_pooled_f32(const float* x, const float* wpack, float* y, int n, int c, int h, int w, int k_out,
                               const float* bias, int act, int out_channel_offset, int out_channels_total,
                               float act_lo, float act_hi);
/* The same pair for an FP16 IR read with fp16_as_fp32=False (ABI v13): the window maximum in fp32, then both operands rounded to fp16 as
 * they are read from LDS, fp32 accumulation on v_mfma_f32_32x32x16_f16 -- what pvhip_maxpool2d_f32 followed by pvhip_conv2d_f16_dma does. */
int    pvhip_conv2d_pooled_f16(const float* x, const float* wpack, float* y, int n, int c, int h, int w, int k_out,
                               const float* bias, int act, int out_channel_offset, int out_channels_total,
                               float act_lo, float act_hi);
/* Several Convolution.py:149-176 calls that share their input (the 1x1, 3x3_reduce and 5x5_reduce arms of an inception
 * module) as ONE launch: the input is read once and the small arms ride in the big one's grid.  Only 1x1 / stride 1 /
 * unpadded convolutions with c % 16 == 0 (pvhip_conv2d_multi_supported; no device needed).  wpack is the panel of
 * pvhip_conv2d_pack_f32 for the (k_panel, c, 1, 1) weight tensor that holds the convolutions' OIHW weights one after
 * the other, each padded with zero rows to a multiple of 32 output channels (k_panel = sum of the padded counts); bias
 * (k_panel values, optional) and act / act_lo / act_hi as for pvhip_conv2d_f32, shared by all.  dests[i] says where
 * convolution i stores: y (an (n, k, oh, ow) tensor, or with channels_total > 0 the (n, channels_total, oh, ow) tensor
 * whose channels [channel_offset, channel_offset + k) it fills).  Every output carries the bits of its own
 * pvhip_conv2d_f32 call.                                                                                            */
#define PVHIP_MAX_CONV_DESTS 6
typedef struct pvhip_conv_dest {
    float* y;           /* (layout 1: an fp16 tensor behind a float pointer) */
    int    k;
    int    channel_offset;
    int    channels_total;
    int    layout;          /* ABI v14.  0: fp32 NCHW, as above.  1 (pvhip_conv2d_multi_f16_dma only; channels_total = 0; act none or ReLU):
                             * y is an fp16 tensor with the channels blocked by eight, [n][ceil16(k) / 8][oh * ow][8 halves]
                             * (pvhip_c8_f16_elems floats), the input format of pvhip_conv2d_f16_c8: what the reference's float16
                             * tensor of this node holds (common_def.py:13-17), channels past k up to a whole 16 are zeros.           */
} pvhip_conv_dest;
int    pvhip_conv2d_multi_supported(int c, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int n_dest);
int    pvhip_conv2d_multi_f32(const float* x, const float* wpack, int n, int c, int h, int w, int kh, int kw,
                              int oh, int ow, int sh, int sw, int pad_top, int pad_left,
                              const float* bias, int act, float act_lo, float act_hi,
                              int n_dest, const pvhip_conv_dest* dests);
/* The same launch for an FP16 IR read with fp16_as_fp32=False (ABI v13): the f16 form of the LDS-DMA kernel (pvhip_conv2d_f16_dma) over
 * the members' panel -- the module input is read once, operands rounded to fp16 as they are read from LDS, fp32 accumulation.          */
int    pvhip_conv2d_multi_f16_dma(const float* x, const float* wpack, int n, int c, int h, int w, int kh, int kw,
                                  int oh, int ow, int sh, int sw, int pad_top, int pad_left,
                                  const float* bias, int act, float act_lo, float act_hi,
                                  int n_dest, const pvhip_conv_dest* dests);

/* ---- FP16 IRs.  The reference runs an FP16 IR in numpy float16 (common_def.py:13-17; Convolution.py:57-87 and
 * MatMul.py:9-17 then multiply AND accumulate in float16).  These entries take the same fp32 device tensors as their _f32
 * twins, round both operands to fp16 (round to nearest even; the constants of an FP16 IR are fp16 values already) and
 * accumulate in fp32 on the f16 matrix-core instructions (v_mfma_f32_32x32x16_f16, 16x the fp32 MFMA rate).  The engine
 * selects them only for an FP16 IR read with fp16_as_fp32=False.  pvhip_conv2d_f16_pack_elems counts FLOATS of wpack.    */
size_t pvhip_conv2d_f16_pack_elems(int k_out, int c, int kh, int kw);
int    pvhip_conv2d_f16_pack(const float* w_oihw, float* wpack, int k_out, int c, int kh, int kw, int h, int w);
int    pvhip_conv2d_f16(const float* x, const float* wpack, float* y,
                        int n, int c, int h, int w, int k_out, int kh, int kw, int oh, int ow,
                        int sh, int sw, int pad_top, int pad_left,
                        const float* bias, int act,
                        int out_channel_offset, int out_channels_total,
                        float act_lo, float act_hi);
/* The second f16 kernel (ABI v13): layers whose channel count is a multiple of 16 (every GoogLeNet layer but conv1) on the LDS-DMA
 * kernel of pvhip_conv2d_f32 -- the SAME fp32 tiles reach LDS by the same copies, wpack is the fp32 panel of pvhip_conv2d_pack_f32 --
 * with a stage of 16 channels as ONE v_mfma_f32_32x32x16_f16 per 32-channel tile (operands rounded to fp16 as they are read from LDS,
 * fp32 accumulation): 1/16 of the matrix-core cycles, no register-staged gather.  Same arithmetic as pvhip_conv2d_f16 in another
 * summation order ((r,s)-major).  _supported: C % 16 == 0 and a window of fewer than 64 taps.                                    */
int    pvhip_conv2d_f16_dma_supported(int c, int kh, int kw);
int    pvhip_conv2d_f16_dma(const float* x, const float* wpack, float* y,
                            int n, int c, int h, int w, int k_out, int kh, int kw, int oh, int ow,
                            int sh, int sw, int pad_top, int pad_left,
                            const float* bias, int act,
                            int out_channel_offset, int out_channels_total,
                            float act_lo, float act_hi);
/* The third f16 kernel (ABI v13): stride-1 "same" windows (1x1; 3x3 / pad 1; 5x5 / pad 2) with C % 16 == 0 and rows short enough that
 * a 128-pixel tile and its halo are one 1-KiB span per channel (128 + 2 * pad * (w + 1) + 3 <= 256 floats: every GoogLeNet layer but
 * conv1).  A workgroup copies ONE span per channel and stage into LDS and serves every tap and up to 128 output channels from it (no
 * copy per tap, none per channel tile); the weights are fp16 MFMA fragments packed once by _span_pack (wf: _span_pack_elems FLOATS).
 * Operands rounded to fp16 as they are read, fp32 accumulation; arguments as pvhip_conv2d_f32.                                     */
int    pvhip_conv2d_f16_span_supported(int c, int h, int w, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int oh, int ow);
size_t pvhip_conv2d_f16_span_pack_elems(int k_out, int c, int kh, int kw);
int    pvhip_conv2d_f16_span_pack(const float* w_oihw, float* wf, int k_out, int c, int kh, int kw);
int    pvhip_conv2d_f16_span(const float* x, const float* wf, float* y,
                             int n, int c, int h, int w, int k_out, int kh, int kw, int oh, int ow,
                             int sh, int sw, int pad_top, int pad_left,
                             const float* bias, int act,
                             int out_channel_offset, int out_channels_total,
                             float act_lo, float act_hi);
int    pvhip_matmul_f16(const float* a, const float* b, float* c, int m, int n, int k,
                        int trans_a, int trans_b);
/* The fourth f16 kernel (ABI v14) and the first fp16 TENSORS in HBM: the 3x3_reduce / 5x5_reduce convolutions of an FP16 IR hand
 * their output to the 3x3 / 5x5 convolution behind them as fp16 with the channels blocked by eight ("c8": [n][ceil16(c) / 8][h * w][8],
 * the eight channels of a pixel = one 16-byte MFMA operand; written by pvhip_conv2d_multi_f16_dma through pvhip_conv_dest.layout = 1).
 * pvhip_conv2d_f16_c8 reads it: stride-1 "same" windows (1x1; 3x3 / pad 1; 5x5 / pad 2) over rows of at most 64 - 2 pad pixels, any c
 * (padded to whole 16-channel stages with zero weights), fp32 NCHW output with bias / activation / channel offset as pvhip_conv2d_f32.
 * A producer wave copies whole input rows into LDS (the zero padding is the out-of-range rule of the copy), four consumer waves own a
 * 32-channel tile each.  wf: fragments packed by _c8_pack (_c8_pack_elems FLOATS).  _from_f32 / _to_f32 convert between NCHW fp32 and
 * c8 fp16 (round to nearest even): the boundary of the layout for any other reader, and the tests.                                    */
size_t pvhip_c8_f16_elems(int n, int c, int h, int w);
int    pvhip_c8_f16_from_f32(const float* x, void* xb, int n, int c, int h, int w);
int    pvhip_c8_f16_to_f32(const void* xb, float* x, int n, int c, int h, int w);
int    pvhip_conv2d_f16_c8_supported(int c, int h, int w, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int oh, int ow);
size_t pvhip_conv2d_f16_c8_pack_elems(int k_out, int c, int kh, int kw);
int    pvhip_conv2d_f16_c8_pack(const float* w_oihw, float* wf, int k_out, int c, int kh, int kw);
/* The module form (ABI v14): one or several 1x1 convolutions of the same c8 input as one launch (n_dest members; wf: _c8_pack of the
 * members' weights laid one after the other, each padded with zero rows to a multiple of 32 output channels, bias likewise), a 1x1
 * convolution behind a 3x3 / stride 1 / pad 1 MaxPool (pool = 1: MaxPool.py:41-72 then Convolution.py:57-87, the pooled tensor never
 * exists), or one 3x3 / 5x5 convolution.  dests[i].layout: 0 = fp32 NCHW (y, channel_offset, channels_total as pvhip_conv2d_multi_f32),
 * 1 = fp16 c8: a tensor of its own (channels_total = 0: [n][ceil16(k) / 8][h * w][8], zeros past k) or channels [channel_offset,
 * channel_offset + k) of a c8 tensor of channels_total channels (the module's Concat buffer: k and the offset multiples of 8, the
 * total a multiple of 16).  pvhip_maxpool3x3_c8: MaxPool.py:41-72 for a 3x3 window on a c8 tensor (any stride, zero padding, the
 * window clipped at the padded edge; a NaN wins), c8 output.                                                                     */
int    pvhip_conv2d_f16_c8_multi_supported(int c, int h, int w, int kh, int kw, int pool, int n_dest);
int    pvhip_conv2d_f16_c8_multi(const void* xb, const float* wf, int n, int c, int h, int w, int kh, int kw, int pool,
                                 const float* bias, int act, float act_lo, float act_hi,
                                 int n_dest, const pvhip_conv_dest* dests);
int    pvhip_maxpool3x3_c8(const void* x, void* y, int n, int c, int h, int w, int oh, int ow, int sh, int sw,
                           int pad_top, int pad_left, int pad_bottom, int pad_right);
/* The stem of an FP16 IR on blocked fp16 tensors (ABI v14): pvhip_conv2d_f16_dma_c8 = pvhip_conv2d_f16_dma with the output stored as fp16 c8
 * ([n][ceil16(k_out) / 8][oh * ow][8]; act none or ReLU); pvhip_maxpool3x3_lrn_c8 = MaxPool 3x3 (MaxPool.py:41-72) followed by LRN over
 * five channels (LRN.py:10-22) on a c8 tensor as one launch, c8 output.                                                            */
int    pvhip_conv2d_f16_dma_c8(const float* x, const float* wpack, void* yb,
                               int n, int c, int h, int w, int k_out, int kh, int kw, int oh, int ow,
                               int sh, int sw, int pad_top, int pad_left, const float* bias, int act);
int    pvhip_maxpool3x3_lrn_c8(const void* x, void* y, int n, int c, int h, int w, int oh, int ow, int sh, int sw,
                               int pad_top, int pad_left, int pad_bottom, int pad_right,
                               int size, float alpha, float beta, float bias);
/* ... with the 1x1 / stride 1 / unpadded convolution behind the LRN in the same launch (ABI v15; the FP16-IR twin of pvhip_maxpool_lrn_conv1x1_f32:
 * fp16 operands on v_mfma_f32_32x32x4_2b_f16, fp32 accumulation).  x: c8 (n, c, h, w), c <= 64; w_oihw: the (k_out, c, 1, 1) fp32 weights, rounded to
 * fp16 as they are staged; k_out <= 64; y: c8 (n, k_out, oh, ow); conv_bias may be NULL; act: none or ReLU.                                   */
int    pvhip_maxpool3x3_lrn_conv1x1_c8_supported(int c, int k_out, int size);
int    pvhip_maxpool3x3_lrn_conv1x1_c8(const void* x, const float* w_oihw, void* y, int n, int c, int h, int w, int oh, int ow, int sh, int sw,
                                       int pad_top, int pad_left, int pad_bottom, int pad_right, int size, float alpha, float beta, float bias,
                                       int k_out, const float* conv_bias, int act);
/* The first convolution of an image network as an FP16 layer, from row spans (ABI v14): 7x7 / stride 2 / pad 3 over three channels, at most 64
 * output channels (GoogLeNet's conv1).  _supported: 0, or the floats per row the padded input must have (w + 3 rounded up so that every
 * tap of the last output column exists, whole 16-byte pieces); xp: that padded input (n, 3, hp, wp), e.g. from pvhip_pad2d_f32 with
 * pad_top = pad_left = 3, pad_bottom = 3, pad_right = wp - w - 3 (and the per-channel constant of a folded Add); wf: _stem_pack of the
 * (k_out, 3, 7, 7) weights (_stem_pack_elems FLOATS); yb: fp16 c8 output (n, k_out, oh, ow); act: none or ReLU.                        */
int    pvhip_conv2d_f16_stem_supported(int c, int h, int w, int k_out, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int oh, int ow);
size_t pvhip_conv2d_f16_stem_pack_elems(int k_out);
int    pvhip_conv2d_f16_stem_pack(const float* w_oihw, float* wf, int k_out);
int    pvhip_conv2d_f16_stem(const float* xp, const float* wf, void* yb, int n, int hp, int wp, int k_out, int oh, int ow,
                             const float* bias, int act);
/* ... and straight from the UNPADDED image (ABI v15; as pvhip_conv2d_stem_direct_f32: w % 4 == 0, w <= 248; no padding pass; pre_add: the per-channel
 * constant of an Add in front of the layer, or NULL); wf from _direct_pack (the k slots of a filter row start one column in front of the window). */
int    pvhip_conv2d_f16_stem_direct_supported(int c, int h, int w, int k_out, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int oh, int ow);
int    pvhip_conv2d_f16_stem_direct_pack(const float* w_oihw, float* wf, int k_out);
int    pvhip_conv2d_f16_stem_direct(const float* x, const float* wf, void* yb, int n, int h, int w, int k_out, int oh, int ow,
                                    const float* pre_add, const float* bias, int act);
/* The same first convolution in fp32 (ABI v15; Convolution.py:57-87: 7x7 / stride 2 / pad 3 over three channels, at most 64 output channels, output
 * rows of at most 112 pixels and a multiple of four -- GoogLeNet's conv1): from row spans of the padded image, the whole weight tensor resident in
 * registers, no vector instruction in the reduction loop; the reduction runs over the taps in the reference's own (c, r, s) order.  _supported: 0,
 * or the floats per row the padded input must have (as pvhip_conv2d_f16_stem_supported); xp: that padded input (n, 3, hp, wp) with
 * hp >= 2 (oh - 1) + 7, from pvhip_pad2d_f32; wf: _pack of the (k_out, 3, 7, 7) weights (_pack_elems floats); y: (n, k_out, oh, ow) fp32;
 * bias / act / act_lo / act_hi as pvhip_conv2d_f32.  pvhip_conv2d_kernel_kind answers PVHIP_CONV_KIND_STEM for such a layer
 * (PVHIP_CONV_STEM=0: never).                                                                                                       */
int    pvhip_conv2d_stem_f32_supported(int c, int h, int w, int k_out, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int oh, int ow);
size_t pvhip_conv2d_stem_f32_pack_elems(int k_out);
int    pvhip_conv2d_stem_f32_pack(const float* w_oihw, float* wf, int k_out);
int    pvhip_conv2d_stem_f32(const float* xp, const float* wf, float* y, int n, int hp, int wp, int k_out, int oh, int ow,
                             const float* bias, int act, float act_lo, float act_hi);
/* ... and straight from the UNPADDED image x (n, 3, h, w) where w % 4 == 0 and w <= 248 (_direct_supported: 1 / 0): no padding pass at all -- the
 * zero padding is where the copies land in LDS plus out-of-range lanes and rows; pre_add: one fp32 constant per input channel added to the image
 * (not to its padding) on the way, or NULL: the Add of a per-channel Const in front of the layer (Add.py:9-14; GoogLeNet's data/mean).           */
int    pvhip_conv2d_stem_direct_supported(int c, int h, int w, int k_out, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int oh, int ow);
int    pvhip_conv2d_stem_direct_f32(const float* x, const float* wf, float* y, int n, int h, int w, int k_out, int oh, int ow,
                                    const float* pre_add, const float* bias, int act, float act_lo, float act_hi);
/* The same first convolution as WINOGRAD F(3x3, 4x4) on the space-to-depth image (ABI v16; Convolution.py:57-87 for a 7x7 / stride 2 / pad 3 layer
 * over three channels): x'(c; py, px; i, j) = xpad(c, 2 i + py, 2 j + px) turns it into a 4x4 / stride 1 convolution over 12 channels, which
 * 6x6-point tiles (the interpolation points of the F(4x4,3x3) / F(2x2,5x5) kernels) compute with 0.34 of the multiplies.  x: the UNPADDED
 * (n, 3, h, w) fp32 image, h even, w % 4 == 0, w <= 224; oh = h / 2, ow = w / 2; k_out <= 64 and a multiple of 16; u: pvhip_conv2d_stem_wino_pack's transformed weights
 * (pvhip_conv2d_stem_wino_pack_elems floats); pre_add / bias / act as pvhip_conv2d_stem_direct_f32.  NOT the bits of pvhip_conv2d_f32 (another
 * order of summation): the tolerance of the other Winograd forms.  OPT-IN: pvhip_conv2d_kernel_kind answers PVHIP_CONV_KIND_STEM_WINO only with
 * PVHIP_CONV_STEM_WINO=1 (measured at batch 256: 0.50 ms against the row-span kernel's 0.56, at 0.26 of the fp32 MFMA peak on executed flops).    */
int    pvhip_conv2d_stem_wino_supported(int c, int h, int w, int k_out, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int oh, int ow);
long   pvhip_conv2d_stem_wino_pack_elems(void);
int    pvhip_conv2d_stem_wino_pack(const float* w_oihw, float* u, int k_out);
int    pvhip_conv2d_stem_wino_f32(const float* x, const float* u, float* y, int n, int h, int w, int k_out, int oh, int ow,
                                  const float* pre_add, const float* bias, int act, float act_lo, float act_hi);
/* AvgPool.py:41-59 on a c8 tensor (the window rule of pvhip_avgpool2d_f32); the output is fp32 NCHW holding fp16 values (the mean in fp32,
 * rounded once: the reference's AvgPool of a float16 tensor returns float16). */
int    pvhip_avgpool_c8(const void* x, float* y, int n, int c, int h, int w, int oh, int ow, int kh, int kw, int sh, int sw);
/* ... and the other order: LRN over five channels followed by MaxPool 3x3 on a c8 tensor as one launch (LRN.py:10-22 then MaxPool.py:41-72;
 * the LRN tensor never exists).  _supported: pooled rows per workgroup (0: outside the kernel).                                        */
int    pvhip_lrn_maxpool3x3_c8_supported(int h, int w, int oh, int ow, int sh, int sw, int pad_top, int pad_left, int size);
int    pvhip_lrn_maxpool3x3_c8(const void* x, void* y, int n, int c, int h, int w, int size, float alpha, float beta, float bias,
                               int oh, int ow, int sh, int sw, int pad_top, int pad_left, int pad_bottom, int pad_right);
int    pvhip_conv2d_f16_c8(const void* xb, const float* wf, float* y,
                           int n, int c, int h, int w, int k_out, int kh, int kw,
                           const float* bias, int act,
                           int out_channel_offset, int out_channels_total,
                           float act_lo, float act_hi);

/* GroupConvolution.py:53-79 kernel_GroupConvolution_numpy, depthwise case only (weights
 * [G,1,1,kh,kw], one input and one output channel per group), applied to every image.  bias / act /
 * act_lo / act_hi: optional fused Add(per-channel Const) and ReLU (1) or Clamp (2), as for pvhip_conv2d_f32. */
int pvhip_dwconv2d_f32(const float* x, const float* w, float* y, int n, int g, int h, int wdt,
                       int kh, int kw, int oh, int ow, int sh, int sw, int pad_top, int pad_left,
                       const float* bias, int act, float act_lo, float act_hi);

/* ---------------------------------------------------------------- SSD head ------------------ */
/* DetectionOutput.py:163-259 kernel_DetectionOutput_naive with its helpers iou (:12-34), nms (:38-63),
 * screen_out_prior_boxes (:69-97), decode_bboxes (:100-150), clip_bounding_boxes (:153-158); share_location and
 * normalized boxes, as the reference asserts.  loc [n][P*4], conf [n][P*C], priors [1][2][P*4] (boxes, variances),
 * out [n*records][7] = [rank, class, score, xmin, ymin, xmax, ymax] in descending score order per image, a
 * [-1,0,..] terminator after the last record, zeros after it.  The reference handles n == 1 only; images are
 * independent here.  code_type_center_size: 1 = caffe.PriorBoxParameter.CENTER_SIZE, 0 = CORNER. */
int pvhip_detection_output_f32(const float* loc, const float* conf, const float* priors, float* out, int n,
                               int num_priors, int num_classes, int records_per_image, float confidence_threshold,
                               float nms_threshold, int code_type_center_size, int variance_encoded_in_target,
                               int clip_before_nms, int clip_after_nms);

/* ---------------------------------------------------------------- multi-GPU gather ---------- */
/* No reference counterpart (the reference is single-process).  Batch shards are independent; the only
 * exchange is an all-gather of the Result tensor over RCCL/xGMI.  unique_id is a 128-byte buffer. */
#define PVHIP_UNIQUE_ID_BYTES 128
int pvhip_comm_unique_id(void* unique_id_out);
int pvhip_comm_init(const void* unique_id, int rank, int world);
int pvhip_comm_allgather_f32(const float* send, float* recv, size_t count_per_rank);   /* every rank the SAME count (ncclAllGather) */
int pvhip_comm_ranks(int* count);                 /* ncclCommCount of the communicator: how many ranks RCCL itself sees */
int pvhip_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* PVHIP_H */
