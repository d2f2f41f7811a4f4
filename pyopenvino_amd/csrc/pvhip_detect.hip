// DetectionOutput (SSD head) on the device: replaces kernel_DetectionOutput_naive
// (reference op_plugins/DetectionOutput.py:163-259) and its helpers iou (:12-34), nms (:38-63),
// screen_out_prior_boxes (:69-97), decode_bboxes (:100-150), clip_bounding_boxes (:153-158).
//
// Three launches (a global workspace holds the candidates of every image between them):
//   candidates (one workgroup per image)
//   1. per prior: best class and its score (ties -> the later class, what a stable ascending sort reversed gives;
//      the reference's np.argsort is unstable, so equal scores have no defined order there)
//   2. priors with score > threshold and class != 0 are kept IN PRIOR ORDER (block scan)
//   3. their boxes are decoded from the prior boxes / variances: float32 arithmetic in the reference's operation
//      order (-ffp-contract=off), exp evaluated in double and rounded once, as math.exp on a numpy float32 does
//   suppression (256 candidates per workgroup, all images side by side)
//   4. the reference's suppression is order-independent: for every pair (i < j) with IoU > threshold the box with the
//      lower score is dropped (the later one on equal scores) whether or not either was dropped before, so every
//      candidate tests itself against all others of its image, which pass through LDS in tiles of 256
//   records (one workgroup per image)
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
    // workspace, per image: candidate boxes / scores / classes, survivor flags, candidate count
    float4*        c_box;     // [N][P]
    float*         c_score;   // [N][P]
    int*           c_cls;     // [N][P]
    unsigned char* alive;     // [N][P]
    int*           count;     // [N]
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
template <int kThreads>
__device__ int compact(const unsigned char* flag, int n, int* list, int* scratch /* kThreads + 1 ints */) {
    const int tid   = threadIdx.x;
    const int chunk = (n + kThreads - 1) / kThreads;
    const int lo = min(n, tid * chunk), hi = min(n, lo + chunk);
    int cnt = 0;
    for (int i = lo; i < hi; ++i) cnt += flag[i] ? 1 : 0;
    scratch[tid] = cnt;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < kThreads; ++t) {
            const int c = scratch[t];
            scratch[t]  = run;
            run += c;
        }
        scratch[kThreads] = run;
    }
    __syncthreads();
    int pos = scratch[tid];
    for (int i = lo; i < hi; ++i)
        if (flag[i]) list[pos++] = i;
    const int total = scratch[kThreads];
    __syncthreads();
    return total;
}

constexpr int kDetBlock = 1024;   // candidates / records kernels: one workgroup per image, 16 waves

__global__ __launch_bounds__(kDetBlock) void detect_candidates_kernel(DetArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int P = a.P;
    float*         p_score = reinterpret_cast<float*>(lds);                  // [P] best score of each prior
    int*           p_cls   = reinterpret_cast<int*>(p_score + P);            // [P] its class
    int*           sel     = p_cls + P;                                      // [P] candidate -> prior
    int*           scratch = sel + P;                                        // [kDetBlock + 1]
    unsigned char* flag    = reinterpret_cast<unsigned char*>(scratch + kDetBlock + 1);   // [P]

    const int tid = threadIdx.x;
    const int img = blockIdx.x;
    const float* __restrict__ conf = a.conf + (size_t)img * P * a.C;
    const float* __restrict__ loc  = a.loc + (size_t)img * P * 4;

    // 1. best class per prior
    for (int p = tid; p < P; p += kDetBlock) {
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
    const int M = compact<kDetBlock>(flag, P, sel, scratch);
    if (tid == 0) a.count[img] = M;

    // 3. decode
    const float* __restrict__ pbox = a.priors;
    const float* __restrict__ pvar = a.priors + (size_t)P * 4;
    float4* __restrict__ c_box   = a.c_box + (size_t)img * P;
    float* __restrict__  c_score = a.c_score + (size_t)img * P;
    int* __restrict__    c_cls   = a.c_cls + (size_t)img * P;
    for (int i = tid; i < M; i += kDetBlock) {
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
}

// 4. all-pairs suppression: blockIdx.y = image, blockIdx.x = a group of kBlock candidates of it
__global__ __launch_bounds__(kBlock) void detect_suppress_kernel(DetArgs a) {
    __shared__ float4 t_box[kBlock];
    __shared__ float  t_score[kBlock];
    const int img = blockIdx.y;
    const int M   = a.count[img];
    if ((int)blockIdx.x * kBlock >= M) return;      // whole group past the candidates (uniform)
    const float4* __restrict__ c_box   = a.c_box + (size_t)img * a.P;
    const float* __restrict__  c_score = a.c_score + (size_t)img * a.P;
    const int  k      = blockIdx.x * kBlock + threadIdx.x;
    const bool active = k < M;
    const float4 bk = active ? c_box[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float  sk = active ? c_score[k] : 0.0f;
    bool alive = true;
    for (int j0 = 0; j0 < M; j0 += kBlock) {
        const int j = j0 + threadIdx.x;
        if (j < M) { t_box[threadIdx.x] = c_box[j]; t_score[threadIdx.x] = c_score[j]; }
        __syncthreads();
        const int n = min(kBlock, M - j0);
        if (active && alive) {
            for (int t = 0; t < n; ++t) {
                const int jj = j0 + t;
                if (jj == k) continue;
                const float sj    = t_score[t];
                const bool  loses = (k < jj) ? (sk < sj) : !(sj < sk);   // pair (min, max): the first loses only if strictly lower
                if (loses && box_iou(bk, t_box[t]) > a.nms_thr) { alive = false; break; }
            }
        }
        __syncthreads();
    }
    if (active) a.alive[(size_t)img * a.P + k] = alive ? 1 : 0;
}

// 5. rank by descending score and write
__global__ __launch_bounds__(kDetBlock) void detect_records_kernel(DetArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int P = a.P;
    float* k_score = reinterpret_cast<float*>(lds);              // [P] scores of the survivors, in candidate order
    int*   kept    = reinterpret_cast<int*>(k_score + P);        // [P] survivor -> candidate
    int*   scratch = kept + P;                                   // [kDetBlock + 1]
    const int tid = threadIdx.x;
    const int img = blockIdx.x;
    const int M   = a.count[img];
    float* __restrict__ out = a.out + (size_t)img * a.records * 7;
    for (int i = tid; i < a.records * 7; i += kDetBlock) out[i] = 0.0f;
    const int K = compact<kDetBlock>(a.alive + (size_t)img * P, M, kept, scratch);
    const float4* __restrict__ c_box   = a.c_box + (size_t)img * P;
    const float* __restrict__  c_score = a.c_score + (size_t)img * P;
    const int* __restrict__    c_cls   = a.c_cls + (size_t)img * P;
    for (int q = tid; q < K; q += kDetBlock) k_score[q] = c_score[kept[q]];
    __syncthreads();
    for (int q = tid; q < K; q += kDetBlock) {
        const float sq = k_score[q];
        int rank = 0;
        for (int r = 0; r < K; ++r) {
            const float sr = k_score[r];
            rank += (sr > sq || (sr == sq && r > q)) ? 1 : 0;
        }
        if (rank < a.records) {
            const int cq = kept[q];
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
    const size_t P = (size_t)num_priors;
    const size_t lds_cand = P * (3 * sizeof(int) + 1) + (kDetBlock + 1) * sizeof(int) + 16;
    const size_t lds_rec  = P * 2 * sizeof(int) + (kDetBlock + 1) * sizeof(int) + 16;
    if (lds_cand > 150 * 1024)
        return fail(PVHIP_EUNSUPPORTED, "pvhip_detection_output_f32: %d priors need %zu bytes of LDS (limit 150 KB)", num_priors, lds_cand);
    // workspace: [boxes | scores | classes | counts | survivor flags]
    const size_t per_image = P * (sizeof(float4) + sizeof(float) + sizeof(int));
    const size_t ws_bytes  = (size_t)n * per_image + (size_t)n * sizeof(int) + (size_t)n * P;
    void* ws = nullptr;
    if (int rc = pvhip_malloc(&ws, ws_bytes)) return rc;
    DetArgs a;
    a.loc = loc; a.conf = conf; a.priors = priors; a.out = out;
    a.c_box   = static_cast<float4*>(ws);
    a.c_score = reinterpret_cast<float*>(a.c_box + (size_t)n * P);
    a.c_cls   = reinterpret_cast<int*>(a.c_score + (size_t)n * P);
    a.count   = a.c_cls + (size_t)n * P;
    a.alive   = reinterpret_cast<unsigned char*>(a.count + n);
    a.P = num_priors; a.C = num_classes; a.records = records_per_image;
    a.conf_thr = confidence_threshold; a.nms_thr = nms_threshold;
    a.center_size = code_type_center_size; a.var_encoded = variance_encoded_in_target;
    a.clip_before = clip_before_nms; a.clip_after = clip_after_nms;
    if (lds_cand > 64 * 1024)
        PVHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&detect_candidates_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cand));
    if (lds_rec > 64 * 1024)
        PVHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&detect_records_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rec));
    hipLaunchKernelGGL(detect_candidates_kernel, dim3(n), dim3(kDetBlock), lds_cand, state().stream, a);
    hipLaunchKernelGGL(detect_suppress_kernel, dim3((num_priors + kBlock - 1) / kBlock, n), dim3(kBlock), 0, state().stream, a);
    hipLaunchKernelGGL(detect_records_kernel, dim3(n), dim3(kDetBlock), lds_rec, state().stream, a);
    (void)pvhip_free(ws);   // stream-ordered (deferred while streams are forked)
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // extern "C"
