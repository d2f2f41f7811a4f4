// DetectionOutput (SSD head) on the device: replaces kernel_DetectionOutput_naive
// (reference op_plugins/DetectionOutput.py:163-259) and its helpers iou (:12-34), nms (:38-63),
// screen_out_prior_boxes (:69-97), decode_bboxes (:100-150), clip_bounding_boxes (:153-158).
//
// One workgroup per image, everything after the class-score read lives in LDS:
//   1. per prior: best class and its score (ties -> the later class, what a stable ascending sort reversed gives;
//      the reference's np.argsort is unstable, so equal scores have no defined order there)
//   2. priors with score > threshold and class != 0 are kept IN PRIOR ORDER (block scan)
//   3. their boxes are decoded from the prior boxes / variances: float32 arithmetic in the reference's operation
//      order (-ffp-contract=off), exp evaluated in double and rounded once, as math.exp on a numpy float32 does
//   4. the reference's suppression is order-independent: for every pair (i < j) with IoU > threshold the box with the
//      lower score is dropped (the later one on equal scores) whether or not either was dropped before, so every
//      candidate can test itself against all others in parallel
//   5. survivors are clipped, ranked by descending score (equal scores: later candidate first) and written as
//      [rank, class, score, xmin, ymin, xmax, ymax]; a [-1, 0, ...] terminator follows when there are fewer
//      survivors than records; the rest of the image's records are zero.
#include <cmath>

#include "pvhip_common.h"

using namespace pvhip;

namespace {

struct DetArgs {
    const float* loc;      // [N][P*4]
    const float* conf;     // [N][P*C]
    const float* priors;   // [1][2][P*4]: boxes, then variances
    float*       out;      // [N*records][7]
    int   P, C, records;
    float conf_thr, nms_thr;
    int   center_size, var_encoded, clip_before, clip_after;
};

__device__ __forceinline__ float clip01(float v) { return fmaxf(0.0f, fminf(1.0f, v)); }

// DetectionOutput.py:12-34 (symmetric in its arguments: only commutative operations differ between the orders)
__device__ __forceinline__ float box_iou(const float4 a, const float4 b) {
    const float area_a = (a.z - a.x) * (a.w - a.y);
    const float area_b = (b.z - b.x) * (b.w - b.y);
    const float w = fminf(a.z, b.z) - fmaxf(a.x, b.x);
    const float h = fminf(a.w, b.w) - fmaxf(a.y, b.y);
    if (w < 0.0f || h < 0.0f) return 0.0f;
    const float inter = w * h;
    return inter / (area_a + area_b - inter);
}

// Order-preserving compaction of `flag[0..n)` into `list`: returns the number of set flags.  All threads call it.
__device__ int compact(const unsigned char* flag, int n, int* list, int* scratch /* kBlock + 1 ints */) {
    const int tid   = threadIdx.x;
    const int chunk = (n + kBlock - 1) / kBlock;
    const int lo = min(n, tid * chunk), hi = min(n, lo + chunk);
    int cnt = 0;
    for (int i = lo; i < hi; ++i) cnt += flag[i] ? 1 : 0;
    scratch[tid] = cnt;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < kBlock; ++t) {
            const int c = scratch[t];
            scratch[t]  = run;
            run += c;
        }
        scratch[kBlock] = run;
    }
    __syncthreads();
    int pos = scratch[tid];
    for (int i = lo; i < hi; ++i)
        if (flag[i]) list[pos++] = i;
    const int total = scratch[kBlock];
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(kBlock) void detection_output_kernel(DetArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int P = a.P;
    float4*        c_box   = reinterpret_cast<float4*>(lds);                 // [P] candidate boxes
    float*         p_score = reinterpret_cast<float*>(c_box + P);            // [P] best score of each prior
    int*           p_cls   = reinterpret_cast<int*>(p_score + P);            // [P] its class
    int*           sel     = p_cls + P;                                      // [P] candidate -> prior
    float*         c_score = reinterpret_cast<float*>(sel + P);              // [P]
    int*           c_cls   = reinterpret_cast<int*>(c_score + P);            // [P]
    int*           kept    = c_cls + P;                                      // [P] survivor -> candidate
    int*           scratch = kept + P;                                       // [kBlock + 1]
    unsigned char* flag    = reinterpret_cast<unsigned char*>(scratch + kBlock + 1);   // [P]

    const int tid = threadIdx.x;
    const int img = blockIdx.x;
    const float* __restrict__ conf = a.conf + (size_t)img * P * a.C;
    const float* __restrict__ loc  = a.loc + (size_t)img * P * 4;
    float* __restrict__ out        = a.out + (size_t)img * a.records * 7;

    for (int i = tid; i < a.records * 7; i += kBlock) out[i] = 0.0f;

    // 1. best class per prior
    for (int p = tid; p < P; p += kBlock) {
        const float* row = conf + (size_t)p * a.C;
        float best = row[0];
        int   bc   = 0;
        for (int c = 1; c < a.C; ++c) {
            const float v = row[c];
            if (v >= best) { best = v; bc = c; }
        }
        p_score[p] = best;
        p_cls[p]   = bc;
        flag[p]    = (best > a.conf_thr && bc != 0) ? 1 : 0;
    }
    __syncthreads();
    // 2. candidates in prior order
    const int M = compact(flag, P, sel, scratch);

    // 3. decode
    const float* __restrict__ pbox = a.priors;
    const float* __restrict__ pvar = a.priors + (size_t)P * 4;
    for (int i = tid; i < M; i += kBlock) {
        const int   p  = sel[i];
        const float x0 = pbox[p * 4 + 0], y0 = pbox[p * 4 + 1], x1 = pbox[p * 4 + 2], y1 = pbox[p * 4 + 3];
        const float v0 = pvar[p * 4 + 0], v1 = pvar[p * 4 + 1], v2 = pvar[p * 4 + 2], v3 = pvar[p * 4 + 3];
        const float l0 = loc[p * 4 + 0], l1 = loc[p * 4 + 1], l2 = loc[p * 4 + 2], l3 = loc[p * 4 + 3];
        float4 box;
        if (!a.center_size) {
            if (a.var_encoded) box = make_float4(x0 + l0, y0 + l1, x1 + l2, y1 + l3);
            else box = make_float4(x0 + v0 * l0, y0 + v1 * l1, x1 + v2 * l2, y1 + v3 * l3);
        } else {
            const float pw = x1 - x0, ph = y1 - y0;
            const float pcx = (x0 + x1) / 2.0f, pcy = (y0 + y1) / 2.0f;
            float cx, cy, w, h;
            if (a.var_encoded) {
                cx = l0 * pw + pcx;
                cy = l1 * ph + pcy;
                w  = (float)exp((double)l2) * pw;
                h  = (float)exp((double)l3) * ph;
            } else {
                cx = v0 * l0 * pw + pcx;
                cy = v1 * l1 * ph + pcy;
                w  = (float)exp((double)(v2 * l2)) * pw;
                h  = (float)exp((double)(v3 * l3)) * ph;
            }
            box = make_float4(cx - w / 2.0f, cy - h / 2.0f, cx + w / 2.0f, cy + h / 2.0f);
        }
        if (a.clip_before) box = make_float4(clip01(box.x), clip01(box.y), clip01(box.z), clip01(box.w));
        c_box[i]   = box;
        c_score[i] = p_score[p];
        c_cls[i]   = p_cls[p];
    }
    __syncthreads();

    // 4. all-pairs suppression
    for (int k = tid; k < M; k += kBlock) {
        const float4 bk = c_box[k];
        const float  sk = c_score[k];
        bool alive = true;
        for (int j = 0; j < M; ++j) {
            if (j == k) continue;
            const float sj    = c_score[j];
            const bool  loses = (k < j) ? (sk < sj) : !(sj < sk);   // pair (min, max): the first loses only if strictly lower
            if (loses && box_iou(bk, c_box[j]) > a.nms_thr) { alive = false; break; }
        }
        flag[k] = alive ? 1 : 0;
    }
    __syncthreads();
    const int K = compact(flag, M, kept, scratch);

    // 5. rank by descending score and write
    for (int q = tid; q < K; q += kBlock) {
        const int   cq = kept[q];
        const float sq = c_score[cq];
        int rank = 0;
        for (int r = 0; r < K; ++r) {
            const float sr = c_score[kept[r]];
            rank += (sr > sq || (sr == sq && r > q)) ? 1 : 0;
        }
        if (rank < a.records) {
            float4 box = c_box[cq];
            if (a.clip_after) box = make_float4(clip01(box.x), clip01(box.y), clip01(box.z), clip01(box.w));
            float* rec = out + (size_t)rank * 7;
            rec[0] = (float)rank;
            rec[1] = (float)c_cls[cq];
            rec[2] = sq;
            rec[3] = box.x;
            rec[4] = box.y;
            rec[5] = box.z;
            rec[6] = box.w;
        }
    }
    if (tid == 0 && K < a.records) out[(size_t)K * 7] = -1.0f;
}

}  // namespace

extern "C" {

int pvhip_detection_output_f32(const float* loc, const float* conf, const float* priors, float* out, int n,
                               int num_priors, int num_classes, int records_per_image, float confidence_threshold,
                               float nms_threshold, int code_type_center_size, int variance_encoded_in_target,
                               int clip_before_nms, int clip_after_nms) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && num_priors > 0 && num_classes > 0 && records_per_image > 0);
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(loc != nullptr && conf != nullptr && priors != nullptr && out != nullptr);
    const size_t lds = (size_t)num_priors * (sizeof(float4) + 6 * sizeof(int) + 1) + (kBlock + 1) * sizeof(int) + 16;
    if (lds > 150 * 1024)
        return fail(PVHIP_EUNSUPPORTED, "pvhip_detection_output_f32: %d priors need %zu bytes of LDS (limit 150 KB)", num_priors, lds);
    DetArgs a;
    a.loc = loc; a.conf = conf; a.priors = priors; a.out = out;
    a.P = num_priors; a.C = num_classes; a.records = records_per_image;
    a.conf_thr = confidence_threshold; a.nms_thr = nms_threshold;
    a.center_size = code_type_center_size; a.var_encoded = variance_encoded_in_target;
    a.clip_before = clip_before_nms; a.clip_after = clip_after_nms;
    if (lds > 64 * 1024)
        PVHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&detection_output_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(detection_output_kernel, dim3(n), dim3(kBlock), lds, state().stream, a);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // extern "C"
