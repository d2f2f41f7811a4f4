// SoftMax (row-wise, wave-shuffle reductions) and cross-channel LRN (register sliding window).
#include "pvhip_common.h"

using namespace pvhip;

namespace {

// Global accesses of the LRN / pooling streams are NONTEMPORAL: every element is read once and written once, and on this chip a 16-byte
// stream with nt loads and stores runs at 6.0-6.4 TB/s against 5.1-5.4 plain (profiles/r03_stream_sweep.md).  One switch for A/B builds.
constexpr bool kStreamNT = true;
template <class T>
__device__ __forceinline__ T ldnt(const T* p) { return kStreamNT ? __builtin_nontemporal_load(p) : *p; }
template <class T>
__device__ __forceinline__ void stnt(T* p, T v) { if (kStreamNT) __builtin_nontemporal_store(v, p); else *p = v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// One workgroup per row: exp once into registers (cols <= kBlock * kPerThread) or recompute.
// y = exp(x) / sum(exp(x)) with no max shift, exactly the reference expression (SoftMax.py:12-13).
constexpr int kSoftmaxPerThread = 8;

__global__ __launch_bounds__(kBlock) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               int rows, int cols) {
    __shared__ float wave_part[kBlock / kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wid  = threadIdx.x / kWave;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const float* __restrict__ xr = x + (size_t)r * cols;
        float* __restrict__       yr = y + (size_t)r * cols;
        float e[kSoftmaxPerThread];
        float part = 0.0f;
        const bool in_regs = cols <= kBlock * kSoftmaxPerThread;
        if (in_regs) {
#pragma unroll
            for (int j = 0; j < kSoftmaxPerThread; ++j) {
                const int c = threadIdx.x + j * kBlock;
                e[j]        = (c < cols) ? expf(xr[c]) : 0.0f;
                part += e[j];
            }
        } else {
            for (int c = threadIdx.x; c < cols; c += kBlock) part += expf(xr[c]);
        }
        part = wave_sum(part);
        __syncthreads();  // wave_part free from the previous row
        if (lane == 0) wave_part[wid] = part;
        __syncthreads();
        float total = 0.0f;
#pragma unroll
        for (int wv = 0; wv < kBlock / kWave; ++wv) total += wave_part[wv];
        if (in_regs) {
#pragma unroll
            for (int j = 0; j < kSoftmaxPerThread; ++j) {
                const int c = threadIdx.x + j * kBlock;
                if (c < cols) yr[c] = e[j] / total;
            }
        } else {
            for (int c = threadIdx.x; c < cols; c += kBlock) yr[c] = expf(xr[c]) / total;
        }
    }
}

// One lane owns VEC adjacent pixels of one image and walks the channel axis in chunks of T = 8 channels
// (C must be a multiple of 8; other channel counts take the generic kernel).  The T loads of chunk k+1 are
// issued before the outputs that chunk k completes are computed, so T independent 16-byte loads per lane are
// in flight; every element is read once and written once.  After chunk k has landed the computable outputs
// are channels [T*k - HALF, T*k + T - HALF).  The loop body is branch-free: the first chunk, the last chunk and
// the trailing HALF outputs are peeled.  Window for channel c is [c - SIZE/2, c + SIZE/2] clipped to [0, C);
// squares are summed in ascending channel order, the order np.sum(axis=1) uses (LRN.py:19); channels outside
// the tensor contribute an exact 0.
template <int SIZE, int VEC, int BETA_MODE>
__global__ __launch_bounds__(kBlock) void lrn_window_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                                             int c, int hw, float alpha, float beta, float bias) {
    constexpr int beta_mode = BETA_MODE;   // compile-time: keeps the per-element path branch-free
    constexpr int HALF = SIZE / 2;
    constexpr int T    = 8;
    constexpr int E    = 2 * HALF + T;      // ext[]: channels [T*k - 2*HALF, T*k + T)
    static_assert(2 * HALF <= T, "window halo must fit in one chunk");
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int      cols_per_img = hw / VEC;
    const unsigned total        = (unsigned)n * (unsigned)cols_per_img;
    const unsigned stride       = gridDim.x * blockDim.x;
    const int      n_chunks     = c / T;
    const size_t   cstride      = (size_t)hw / VEC;   // channel stride in vec_t units

    // one output: centre ext[j + HALF], window ext[j .. j + SIZE)
#define PV_LRN_OUT(j_, ch_)                                                                    \
    {                                                                                          \
        vec_t s_ = ext[(j_)] * ext[(j_)];                                                      \
        _Pragma("unroll") for (int q = 1; q < SIZE; ++q) s_ = s_ + ext[(j_) + q] * ext[(j_) + q]; \
        vec_t o_;                                                                              \
        _Pragma("unroll") for (int v = 0; v < VEC; ++v) {                                      \
            const float d_ = bias + alpha * s_[v];                                             \
            o_[v]          = lrn_div(ext[(j_) + HALF][v], d_, beta, beta_mode);                \
        }                                                                                      \
        stnt(yv + (size_t)(ch_) * cstride, o_);                                                      \
    }

    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const unsigned img  = t / (unsigned)cols_per_img;
        const unsigned col  = t - img * (unsigned)cols_per_img;
        const size_t   base = (size_t)img * c * hw + (size_t)col * VEC;
        const vec_t* __restrict__ xv = reinterpret_cast<const vec_t*>(x + base);
        vec_t* __restrict__       yv = reinterpret_cast<vec_t*>(y + base);
        vec_t ext[E], nxt[T];
#pragma unroll
        for (int j = 0; j < 2 * HALF; ++j) ext[j] = (vec_t)(0.0f);
#pragma unroll
        for (int j = 0; j < T; ++j) ext[2 * HALF + j] = ldnt(xv + (size_t)j * cstride);
        // chunks 0 .. n_chunks-2: prefetch the next chunk, emit what this one completes
        for (int k = 0; k + 1 < n_chunks; ++k) {
            const vec_t* __restrict__ xn = xv + (size_t)(k + 1) * T * cstride;
#pragma unroll
            for (int j = 0; j < T; ++j) nxt[j] = ldnt(xn + (size_t)j * cstride);
            if (k == 0) {
#pragma unroll
                for (int j = HALF; j < T; ++j) PV_LRN_OUT(j, j - HALF)
            } else {
                const int ch0 = k * T - HALF;
#pragma unroll
                for (int j = 0; j < T; ++j) PV_LRN_OUT(j, ch0 + j)
            }
#pragma unroll
            for (int j = 0; j < 2 * HALF; ++j) ext[j] = ext[T + j];
#pragma unroll
            for (int j = 0; j < T; ++j) ext[2 * HALF + j] = nxt[j];
        }
        // last chunk (no prefetch), then the trailing HALF channels whose windows run past the tensor
        {
            const int k = n_chunks - 1, ch0 = k * T - HALF;
            if (k == 0) {
#pragma unroll
                for (int j = HALF; j < T; ++j) PV_LRN_OUT(j, j - HALF)
            } else {
#pragma unroll
                for (int j = 0; j < T; ++j) PV_LRN_OUT(j, ch0 + j)
            }
#pragma unroll
            for (int j = 0; j < 2 * HALF; ++j) ext[j] = ext[T + j];
#pragma unroll
            for (int j = 0; j < T; ++j) ext[2 * HALF + j] = (vec_t)(0.0f);
#pragma unroll
            for (int j = 0; j < HALF; ++j) PV_LRN_OUT(j, c - HALF + j)
        }
    }
#undef PV_LRN_OUT
}

// ---------------------------------------------------------------------------------------------------------------
// LRN followed by a 3x3 MaxPool (GoogLeNet's conv2/norm2 -> pool2/3x3_s2) in ONE pass over the input: the LRN
// tensor (1.2 GB of traffic at batch 256: written once, read once) never exists.
//
// A workgroup owns one image and a band of pooled rows; its input band (the rows those windows touch, full width)
// is one dense run per channel plane.  It walks the channel axis exactly like lrn_window_kernel -- one lane owns
// VEC adjacent pixels of the band, chunks of T = 8 channels, the loads of chunk k+1 in flight while chunk k is
// normalised, squares summed in ascending channel order -- but the normalised values of a chunk go to LDS ([T][band])
// instead of HBM, and after a barrier the workgroup pools those T planes (9 LDS reads per output; taps outside the
// tensor clamped onto a tap inside the same window, max(m, 0) where the window touches zero-pad cells, NaN wins: the
// rules of maxpool3x3_cols_kernel) and stores T x band_rows x ow outputs.  The LRN arithmetic is lrn_window_kernel's,
// so the result carries the bits of the two separate launches.
struct LrnPoolArgs {
    const float* x;
    float*       y;
    int   n, c, h, w, oh, ow;
    int   pt, pl, hp, wp;
    int   band_rows, n_bands;
    int   plane_l;               // floats per normalised plane in LDS
    float alpha, beta, bias;
    int   pool4;                 // lrn_maxpool3x3_kernel: a lane pools FOUR adjacent outputs of a row (stride 2, no padding on top / left, ow % 4 == 0), the planes dealt over the waves
    int   abl;                   // diagnostic build (PVHIP_CONV_ABLATE bits, wrong results): 1 no LRN arithmetic, 2 no pooling, 4 no stores, 8 loads hit L2
};

template <int SIZE, int VEC, int BETA_MODE, int ST, int NI>     // NI: pooled outputs of a band and plane per lane (ceil(band_rows * ow / 256))
__global__ __launch_bounds__(kBlock) void lrn_maxpool3x3_kernel(LrnPoolArgs a, FastDiv d_bands, FastDiv d_ow) {
    constexpr int beta_mode = BETA_MODE;
    constexpr int HALF = SIZE / 2;
    constexpr int T    = 8;
    constexpr int E    = 2 * HALF + T;
    static_assert(2 * HALF <= T, "window halo must fit in one chunk");
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    extern __shared__ __attribute__((aligned(16))) float planes[];   // [T][plane_l]
#ifdef PVHIP_DIAG
    const int abl = a.abl;          // scripts/time_lrnpool_abl.py: parts switched off
#else
    constexpr int abl = 0;
#endif

    const int tid = threadIdx.x;
    const int img = (int)fdiv(blockIdx.x, d_bands), band = (int)blockIdx.x - img * a.n_bands;
    const int oy0 = band * a.band_rows, oy1 = min(a.oh, oy0 + a.band_rows);
    const int rows_t = oy1 - oy0;
    const int iy_lo = max(0, oy0 * ST - a.pt), iy_hi = min(a.h, (oy1 - 1) * ST + 3 - a.pt);
    const int band_px = (iy_hi - iy_lo) * a.w;
    const int hw = a.h * a.w, ohw = a.oh * a.ow;
    const bool   active  = tid * VEC < band_px;
    const size_t cstride = (size_t)hw / VEC;
    const vec_t* __restrict__ xv =
        reinterpret_cast<const vec_t*>(a.x + (size_t)img * a.c * hw + (size_t)iy_lo * a.w + (active ? tid * VEC : 0));
    float* const mine = planes + tid * VEC;
    const int    n_chunks = a.c / T;
    const int    out_pp   = a.band_rows * a.ow;            // pooled outputs per plane and full band

    // normalise ext[j_ + HALF] with the window ext[j_ .. j_ + SIZE) into LDS plane slot_
#define PV_LRN_TO_LDS(j_, slot_)                                                               \
    {                                                                                          \
        vec_t s_ = ext[(j_)] * ext[(j_)];                                                      \
        _Pragma("unroll") for (int q = 1; q < SIZE; ++q) s_ = s_ + ext[(j_) + q] * ext[(j_) + q]; \
        vec_t o_;                                                                              \
        _Pragma("unroll") for (int v = 0; v < VEC; ++v) {                                      \
            const float d_ = a.bias + a.alpha * s_[v];                                         \
            o_[v]          = (abl & 1) ? ext[(j_) + HALF][v] : lrn_div(ext[(j_) + HALF][v], d_, a.beta, beta_mode); \
        }                                                                                      \
        if (active) *reinterpret_cast<vec_t*>(mine + (slot_) * a.plane_l) = o_;                \
    }

    // ---- pooling geometry, ONCE per lane: a lane owns the same NI output pixels of the band in every plane of every chunk (the
    // first version redid two divisions, six clamps and the pad tests per output, plane and chunk: ~60 vector instructions per
    // output against ~15 of LRN arithmetic per four inputs -- the kernel ran at 3.6 TB/s, VALU-bound).  Nine LDS byte offsets per
    // owned pixel (taps outside the tensor clamped onto a tap inside the same window), the output offset, and whether the window
    // touches zero-pad cells (then max(m, 0)).
    unsigned tap[NI][9];
    unsigned outo[NI];
    bool     live[NI], zpad[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const unsigned rem = (unsigned)(tid + i * kBlock);
        const unsigned oyl = fdiv(rem, d_ow), ox = rem - oyl * (unsigned)a.ow;
        live[i] = (int)rem < out_pp && (int)oyl < rows_t;
        const int oy = oy0 + (int)oyl;
        const int py0 = oy * ST - a.pt, px0 = (int)ox * ST - a.pl;
        const int c0 = min(max(px0, 0), a.w - 1), c1 = min(max(px0 + 1, 0), a.w - 1), c2 = min(max(px0 + 2, 0), a.w - 1);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int r = (min(max(py0 + k, 0), a.h - 1) - iy_lo) * a.w;
            tap[i][3 * k + 0] = live[i] ? (unsigned)(r + c0) * 4u : 0u;
            tap[i][3 * k + 1] = live[i] ? (unsigned)(r + c1) * 4u : 0u;
            tap[i][3 * k + 2] = live[i] ? (unsigned)(r + c2) * 4u : 0u;
        }
        const bool zc = (px0 < 0) || (min((int)ox * ST + 2, a.wp - 1) - a.pl >= a.w);
        const bool zr = (py0 < 0) || (min(oy * ST + 2, a.hp - 1) - a.pt >= a.h);
        zpad[i] = zc || zr;
        outo[i] = (unsigned)(oy * a.ow) + ox;
    }
    float* const yimg = a.y + (size_t)img * a.c * ohw;
    const char* const planes_b = reinterpret_cast<const char*>(planes);
    const unsigned plane_bytes = (unsigned)a.plane_l * 4u;

    // ---- the same pooling with FOUR adjacent outputs of a row per lane (round 5; a.pool4): their windows are nine columns of three rows -- two 16-byte
    // and one 4-byte LDS read per row instead of 36 4-byte ones, sixteen v_maximum3_f32 as before, ONE 16-byte store instead of four 4-byte ones; the
    // quads of a band fit one wave's lanes (rows_t x ow / 4 <= 64), so the planes of a chunk are dealt over the four waves (plane p: wave p % 4).
    unsigned p4_row[3] = {0u, 0u, 0u}, p4_c8 = 0u, p4_out = 0u, p4_z = 0u;
    bool     p4_live = false;
    const int wv4 = tid >> 6;
    if constexpr (VEC == 4 && ST == 2) {
        if (a.pool4 != 0) {
            const int ql = tid & 63, qpr = a.ow >> 2;
            const int oyl = ql / qpr, oxq = ql - oyl * qpr;
            p4_live = oyl < rows_t;
            const int oy = oy0 + (p4_live ? oyl : 0);
            const int py0 = oy * ST, c0 = 8 * oxq;                    // (pt = pl = 0)
#pragma unroll
            for (int k = 0; k < 3; ++k) p4_row[k] = (unsigned)((min(py0 + k, a.h - 1) - iy_lo) * a.w + c0) * 4u;
            p4_c8 = (unsigned)(min(c0 + 8, a.w - 1) - c0) * 4u;        // the ninth column, clamped into the window (a duplicate does not change a maximum)
            const bool zr = min(oy * ST + 2, a.hp - 1) >= a.h;
#pragma unroll
            for (int j = 0; j < 4; ++j) p4_z |= ((zr || min((4 * oxq + j) * ST + 2, a.wp - 1) >= a.w) ? 1u : 0u) << j;
            p4_out = (unsigned)(oy * a.ow + 4 * oxq);
        }
    }
    auto pool4 = [&](int n_pl, int ch0) {
        __syncthreads();
        if (p4_live && !(abl & 2)) {
            typedef float f4_t __attribute__((ext_vector_type(4)));
            for (int p = wv4; p < n_pl; p += 4) {
                const char* const pb = reinterpret_cast<const char*>(planes) + (unsigned)p * ((unsigned)a.plane_l * 4u);
                float hm[3][4];                  // row maxima of the four windows: a NaN IS the maximum (max3_nan), as for np.max
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const f4_t lo = *reinterpret_cast<const f4_t*>(pb + p4_row[k]), hi = *reinterpret_cast<const f4_t*>(pb + p4_row[k] + 16);
                    const float c8 = *reinterpret_cast<const float*>(pb + p4_row[k] + p4_c8);
                    hm[k][0] = max3_nan(lo[0], lo[1], lo[2]);
                    hm[k][1] = max3_nan(lo[2], lo[3], hi[0]);
                    hm[k][2] = max3_nan(hi[0], hi[1], hi[2]);
                    hm[k][3] = max3_nan(hi[2], hi[3], c8);
                }
                f4_t m;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float mj = max3_nan(hm[0][j], hm[1][j], hm[2][j]);
                    m[j] = ((p4_z >> j) & 1u) ? max3_nan(mj, 0.0f, 0.0f) : mj;
                }
                if (!(abl & 4)) *reinterpret_cast<f4_t*>(a.y + (size_t)img * a.c * (a.oh * a.ow) + (size_t)(ch0 + p) * (a.oh * a.ow) + p4_out) = m;
            }
        }
        __syncthreads();
    };
    // pool the first n_pl planes of LDS into channels [ch0, ch0 + n_pl)
    auto pool = [&](int n_pl, int ch0) {
        if constexpr (VEC == 4 && ST == 2) {
            if (a.pool4 != 0) { pool4(n_pl, ch0); return; }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (!live[i] || (abl & 2)) continue;
            float* yo = yimg + (size_t)ch0 * ohw + outo[i];
            unsigned pb = 0u;
            for (int p = 0; p < n_pl; ++p, pb += plane_bytes, yo += ohw) {
                float h[3];                      // row maxima: a NaN IS the maximum (max3_nan), as for np.max
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float v0 = *reinterpret_cast<const float*>(planes_b + pb + tap[i][3 * k + 0]);
                    const float v1 = *reinterpret_cast<const float*>(planes_b + pb + tap[i][3 * k + 1]);
                    const float v2 = *reinterpret_cast<const float*>(planes_b + pb + tap[i][3 * k + 2]);
                    h[k] = max3_nan(v0, v1, v2);
                }
                float m = max3_nan(h[0], h[1], h[2]);
                if (zpad[i]) m = max3_nan(m, 0.0f, 0.0f);
                // PLAIN stores: a lane writes one float per plane, 896-byte runs per plane and band -- nontemporal stores of such pieces
                // cost 4 % here and 9 % in maxpool3x3_lrn_kernel (the dense 16-byte runs of the MaxPool kernel are the opposite case)
                if (!(abl & 4)) *yo = m;
            }
        }
        __syncthreads();
    };

    vec_t ext[E], nxt[T];
#pragma unroll
    for (int j = 0; j < 2 * HALF; ++j) ext[j] = (vec_t)(0.0f);
#pragma unroll
    for (int j = 0; j < T; ++j) ext[2 * HALF + j] = ldnt(xv + (size_t)j * cstride);
    for (int k = 0; k + 1 < n_chunks; ++k) {
        const vec_t* __restrict__ xn = xv + (size_t)((abl & 8) ? 0 : (k + 1) * T) * cstride;        // abl 8: the first chunk again (L2 hits)
#pragma unroll
        for (int j = 0; j < T; ++j) nxt[j] = ldnt(xn + (size_t)j * cstride);
        if (k == 0) {
#pragma unroll
            for (int j = HALF; j < T; ++j) PV_LRN_TO_LDS(j, j - HALF)
            pool(T - HALF, 0);
        } else {
#pragma unroll
            for (int j = 0; j < T; ++j) PV_LRN_TO_LDS(j, j)
            pool(T, k * T - HALF);
        }
#pragma unroll
        for (int j = 0; j < 2 * HALF; ++j) ext[j] = ext[T + j];
#pragma unroll
        for (int j = 0; j < T; ++j) ext[2 * HALF + j] = nxt[j];
    }
    {
        const int k = n_chunks - 1;
        if (k == 0) {
#pragma unroll
            for (int j = HALF; j < T; ++j) PV_LRN_TO_LDS(j, j - HALF)
            pool(T - HALF, 0);
        } else {
#pragma unroll
            for (int j = 0; j < T; ++j) PV_LRN_TO_LDS(j, j)
            pool(T, k * T - HALF);
        }
#pragma unroll
        for (int j = 0; j < 2 * HALF; ++j) ext[j] = ext[T + j];
#pragma unroll
        for (int j = 0; j < T; ++j) ext[2 * HALF + j] = (vec_t)(0.0f);
#pragma unroll
        for (int j = 0; j < HALF; ++j) PV_LRN_TO_LDS(j, j)
        pool(HALF, a.c - HALF);
    }
#undef PV_LRN_TO_LDS
}

// ---------------------------------------------------------------------------------------------------------------
// The other order: a 3x3 MaxPool followed by LRN (GoogLeNet's pool1/3x3_s2 -> pool1/norm1) in ONE pass: the pooled
// tensor (411 MB at batch 256: written once, read once) never exists.  Same bands, same lane <-> input pixels for the
// loads; the RAW planes of a chunk of 8 channels go to LDS, every lane pools its own NI output pixels out of them (the
// geometry of lrn_maxpool3x3_kernel, computed once) and the pooled values of consecutive channels ARE the stream the LRN
// window slides over: five of them in registers per owned pixel, squares summed in ascending channel order, the
// arithmetic of lrn_window_kernel -- the bits of the two separate launches.
template <int SIZE, int VEC, int BETA_MODE, int ST, int NI>
__global__ __launch_bounds__(kBlock) void maxpool3x3_lrn_kernel(LrnPoolArgs a, FastDiv d_bands, FastDiv d_ow) {
    constexpr int beta_mode = BETA_MODE;
    constexpr int HALF = SIZE / 2;
    constexpr int T    = 8;
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    extern __shared__ __attribute__((aligned(16))) float planes[];   // [T][plane_l]

    const int tid = threadIdx.x;
    const int img = (int)fdiv(blockIdx.x, d_bands), band = (int)blockIdx.x - img * a.n_bands;
    const int oy0 = band * a.band_rows, oy1 = min(a.oh, oy0 + a.band_rows);
    const int rows_t = oy1 - oy0;
    const int iy_lo = max(0, oy0 * ST - a.pt), iy_hi = min(a.h, (oy1 - 1) * ST + 3 - a.pt);
    const int band_px = (iy_hi - iy_lo) * a.w;
    const int hw = a.h * a.w, ohw = a.oh * a.ow;
    const bool   active  = tid * VEC < band_px;
    const size_t cstride = (size_t)hw / VEC;
    const vec_t* __restrict__ xv =
        reinterpret_cast<const vec_t*>(a.x + (size_t)img * a.c * hw + (size_t)iy_lo * a.w + (active ? tid * VEC : 0));
    float* const mine = planes + tid * VEC;
    const int    n_chunks = a.c / T;
    const int    out_pp   = a.band_rows * a.ow;

    unsigned tap[NI][9];
    unsigned outo[NI];
    bool     live[NI], zpad[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const unsigned rem = (unsigned)(tid + i * kBlock);
        const unsigned oyl = fdiv(rem, d_ow), ox = rem - oyl * (unsigned)a.ow;
        live[i] = (int)rem < out_pp && (int)oyl < rows_t;
        const int oy = oy0 + (int)oyl;
        const int py0 = oy * ST - a.pt, px0 = (int)ox * ST - a.pl;
        const int c0 = min(max(px0, 0), a.w - 1), c1 = min(max(px0 + 1, 0), a.w - 1), c2 = min(max(px0 + 2, 0), a.w - 1);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int r = (min(max(py0 + k, 0), a.h - 1) - iy_lo) * a.w;
            tap[i][3 * k + 0] = live[i] ? (unsigned)(r + c0) * 4u : 0u;
            tap[i][3 * k + 1] = live[i] ? (unsigned)(r + c1) * 4u : 0u;
            tap[i][3 * k + 2] = live[i] ? (unsigned)(r + c2) * 4u : 0u;
        }
        const bool zc = (px0 < 0) || (min((int)ox * ST + 2, a.wp - 1) - a.pl >= a.w);
        const bool zr = (py0 < 0) || (min(oy * ST + 2, a.hp - 1) - a.pt >= a.h);
        zpad[i] = zc || zr;
        outo[i] = (unsigned)(oy * a.ow) + ox;
    }
    float* const yimg = a.y + (size_t)img * a.c * ohw;
    const char* const planes_b = reinterpret_cast<const char*>(planes);
    const unsigned plane_bytes = (unsigned)a.plane_l * 4u;

    float win[NI][SIZE];                   // pooled values of channels ch - SIZE + 1 .. ch of the owned pixels
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int q = 0; q < SIZE; ++q) win[i][q] = 0.0f;

    // a new pooled value v_ of channel ch_ enters the window; the window is then centred on channel ch_ - HALF
#define PV_PUSH_EMIT(i_, v_, ch_)                                                              \
    {                                                                                          \
        _Pragma("unroll") for (int q = 0; q + 1 < SIZE; ++q) win[i_][q] = win[i_][q + 1];      \
        win[i_][SIZE - 1] = (v_);                                                              \
        if ((ch_) >= HALF) {                                                                   \
            float s_ = win[i_][0] * win[i_][0];                                                \
            _Pragma("unroll") for (int q = 1; q < SIZE; ++q) s_ = s_ + win[i_][q] * win[i_][q]; \
            const float d_ = a.bias + a.alpha * s_;                                            \
            yimg[(size_t)((ch_) - HALF) * ohw + outo[i_]] = lrn_div(win[i_][HALF], d_, a.beta, beta_mode);   /* plain store: see lrn_maxpool3x3_kernel */ \
        }                                                                                      \
    }

    vec_t cur[T], nxt[T];
#pragma unroll
    for (int j = 0; j < T; ++j) cur[j] = ldnt(xv + (size_t)j * cstride);
    for (int k = 0; k < n_chunks; ++k) {
        if (k + 1 < n_chunks) {
            const vec_t* __restrict__ xn = xv + (size_t)(k + 1) * T * cstride;
#pragma unroll
            for (int j = 0; j < T; ++j) nxt[j] = ldnt(xn + (size_t)j * cstride);
        }
        if (active) {
#pragma unroll
            for (int j = 0; j < T; ++j) *reinterpret_cast<vec_t*>(mine + j * a.plane_l) = cur[j];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (!live[i]) continue;
            unsigned pb = 0u;
#pragma unroll
            for (int p = 0; p < T; ++p, pb += plane_bytes) {
                float h[3];                      // row maxima: a NaN IS the maximum (max3_nan), as for np.max
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float v0 = *reinterpret_cast<const float*>(planes_b + pb + tap[i][3 * q + 0]);
                    const float v1 = *reinterpret_cast<const float*>(planes_b + pb + tap[i][3 * q + 1]);
                    const float v2 = *reinterpret_cast<const float*>(planes_b + pb + tap[i][3 * q + 2]);
                    h[q] = max3_nan(v0, v1, v2);
                }
                float m = max3_nan(h[0], h[1], h[2]);
                if (zpad[i]) m = max3_nan(m, 0.0f, 0.0f);
                const float pooled = m;
                PV_PUSH_EMIT(i, pooled, k * T + p)
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < T; ++j) cur[j] = nxt[j];
    }
    // the trailing HALF channels: their windows run past the tensor (zeros)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (!live[i]) continue;
#pragma unroll
        for (int j = 0; j < HALF; ++j) PV_PUSH_EMIT(i, 0.0f, a.c + j)
    }
#undef PV_PUSH_EMIT
}

// ---------------------------------------------------------------------------------------------------------------
// MaxPool 3x3 -> LRN -> 1x1 convolution (GoogLeNet's pool1/3x3_s2 -> pool1/norm1 -> conv2/3x3_reduce, 64 -> 64 channels, + bias + ReLU) in ONE
// pass (round 5): the LRN tensor (205 MB at batch 256: written once, read once by a convolution that runs at its byte bound, 0.098 ms) never
// exists either.  maxpool3x3_lrn_kernel already holds, lane by lane, the normalised value of ITS pixel for one channel after the other -- which is
// exactly the B operand of the k = 1 matrix instruction: v_mfma_f32_32x32x1_2b_f32 adds the outer product A (32 x 1) x B (1 x 32) of each of its two
// blocks, and block b's B is one value per lane of lanes 32 b .. 32 b + 31.  So a wave does, per input channel c and per 32 output channels,
// ONE MFMA with A = W[k][c] (a broadcast LDS read of the transposed weights, 16 KB) and B = what it would have stored: D[k][pixel] += W[k][c] *
// lrn[c][pixel], channel after channel in ascending order -- the reduction order of the pointwise kernel (the bits of the two launches).  No
// transposition, no staging of an operand tile, no barrier beyond the two the pooling has.  Epilogue: register v of block b at lane (j, h) is
// output channel 8 (v / 4) + 4 h + v % 4 of the pixel lane 32 b + j owns: that lane's output offset comes over once by a wave shuffle.
typedef float floatx32 __attribute__((ext_vector_type(32)));

struct PoolLrnConvArgs {
    LrnPoolArgs  p;           // the pooling / LRN geometry (p.y unused)
    const float* cw;          // (k_out, c, 1, 1)
    const float* cbias;       // k_out, or null
    float*       y;           // (n, k_out, oh, ow)
    int   k_out, act;
    float act_lo, act_hi;
};

template <int BETA_MODE, int ST, int KT>        // KT: 32-channel tiles of the convolution's output (k_out <= 32 KT)
__global__ __launch_bounds__(kBlock) void maxpool3x3_lrn_conv1x1_kernel(PoolLrnConvArgs ca, FastDiv d_bands, FastDiv d_ow) {
    constexpr int beta_mode = BETA_MODE;
    constexpr int SIZE = 5, HALF = SIZE / 2, T = 8, VEC = 4;
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    extern __shared__ __attribute__((aligned(16))) float planes[];   // [T][plane_l], then the weights [c][32 KT]
    const LrnPoolArgs& a = ca.p;
    float* const wl = planes + T * a.plane_l;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int img = (int)fdiv(blockIdx.x, d_bands), band = (int)blockIdx.x - img * a.n_bands;
    const int oy0 = band * a.band_rows, oy1 = min(a.oh, oy0 + a.band_rows);
    const int rows_t = oy1 - oy0;
    const int iy_lo = max(0, oy0 * ST - a.pt), iy_hi = min(a.h, (oy1 - 1) * ST + 3 - a.pt);
    const int band_px = (iy_hi - iy_lo) * a.w;
    const int hw = a.h * a.w, ohw = a.oh * a.ow;
    const bool   active  = tid * VEC < band_px;
    const size_t cstride = (size_t)hw / VEC;
    const vec_t* __restrict__ xv =
        reinterpret_cast<const vec_t*>(a.x + (size_t)img * a.c * hw + (size_t)iy_lo * a.w + (active ? tid * VEC : 0));
    float* const mine = planes + tid * VEC;
    const int    n_chunks = a.c / T;
    const int    out_pp   = a.band_rows * a.ow;

    // the transposed weights: wl[c][k] = W[k][c], zeros past k_out
    for (int e = tid; e < a.c * 32 * KT; e += kBlock) {
        const int c = e / (32 * KT), k = e - c * (32 * KT);
        wl[e] = k < ca.k_out ? ca.cw[(size_t)k * a.c + c] : 0.0f;
    }

    unsigned tap[9];
    unsigned outo;
    bool     live, zpad;
    {
        const unsigned rem = (unsigned)tid;
        const unsigned oyl = fdiv(rem, d_ow), ox = rem - oyl * (unsigned)a.ow;
        live = (int)rem < out_pp && (int)oyl < rows_t;
        const int oy = oy0 + (int)oyl;
        const int py0 = oy * ST - a.pt, px0 = (int)ox * ST - a.pl;
        const int c0 = min(max(px0, 0), a.w - 1), c1 = min(max(px0 + 1, 0), a.w - 1), c2 = min(max(px0 + 2, 0), a.w - 1);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int r = (min(max(py0 + k, 0), a.h - 1) - iy_lo) * a.w;
            tap[3 * k + 0] = live ? (unsigned)(r + c0) * 4u : 0u;
            tap[3 * k + 1] = live ? (unsigned)(r + c1) * 4u : 0u;
            tap[3 * k + 2] = live ? (unsigned)(r + c2) * 4u : 0u;
        }
        const bool zc = (px0 < 0) || (min((int)ox * ST + 2, a.wp - 1) - a.pl >= a.w);
        const bool zr = (py0 < 0) || (min(oy * ST + 2, a.hp - 1) - a.pt >= a.h);
        zpad = zc || zr;
        outo = live ? (unsigned)(oy * a.ow) + ox : 0xffffffffu;
    }
    const char* const planes_b = reinterpret_cast<const char*>(planes);
    const unsigned plane_bytes = (unsigned)a.plane_l * 4u;
    const float* const wl_lane = wl + (lane & 31);

    float win[SIZE];
#pragma unroll
    for (int q = 0; q < SIZE; ++q) win[q] = 0.0f;
    floatx32 acc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 32; ++r) acc[kt][r] = 0.0f;

    // a new pooled value v_ of channel ch_ enters the window; the window is then centred on channel ch_ - HALF, whose normalised value -- what
    // maxpool3x3_lrn_kernel stores -- is this lane's column of the outer product with that channel's weights (every lane takes part: no branch)
#define PV_PUSH_MFMA(v_, ch_)                                                                  \
    {                                                                                          \
        _Pragma("unroll") for (int q = 0; q + 1 < SIZE; ++q) win[q] = win[q + 1];              \
        win[SIZE - 1] = (v_);                                                                  \
        if ((ch_) >= HALF) {                                                                   \
            float s_ = win[0] * win[0];                                                        \
            _Pragma("unroll") for (int q = 1; q < SIZE; ++q) s_ = s_ + win[q] * win[q];        \
            const float d_ = a.bias + a.alpha * s_;                                            \
            const float o_ = live ? lrn_div(win[HALF], d_, a.beta, beta_mode) : 0.0f;          \
            const float* const wc_ = wl_lane + ((ch_) - HALF) * (32 * KT);                     \
            _Pragma("unroll") for (int kt = 0; kt < KT; ++kt)                                  \
                acc[kt] = __builtin_amdgcn_mfma_f32_32x32x1f32(wc_[32 * kt], o_, acc[kt], 0, 0, 0); \
        }                                                                                      \
    }

    vec_t cur[T], nxt[T];
#pragma unroll
    for (int j = 0; j < T; ++j) cur[j] = ldnt(xv + (size_t)j * cstride);
    for (int k = 0; k < n_chunks; ++k) {
        if (k + 1 < n_chunks) {
            const vec_t* __restrict__ xn = xv + (size_t)(k + 1) * T * cstride;
#pragma unroll
            for (int j = 0; j < T; ++j) nxt[j] = ldnt(xn + (size_t)j * cstride);
        }
        if (active) {
#pragma unroll
            for (int j = 0; j < T; ++j) *reinterpret_cast<vec_t*>(mine + j * a.plane_l) = cur[j];
        }
        __syncthreads();
        {
            unsigned pb = 0u;
#pragma unroll
            for (int p = 0; p < T; ++p, pb += plane_bytes) {
                float h[3];                      // row maxima: a NaN IS the maximum (max3_nan), as for np.max
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float v0 = *reinterpret_cast<const float*>(planes_b + pb + tap[3 * q + 0]);
                    const float v1 = *reinterpret_cast<const float*>(planes_b + pb + tap[3 * q + 1]);
                    const float v2 = *reinterpret_cast<const float*>(planes_b + pb + tap[3 * q + 2]);
                    h[q] = max3_nan(v0, v1, v2);
                }
                float m = max3_nan(h[0], h[1], h[2]);
                if (zpad) m = max3_nan(m, 0.0f, 0.0f);
                PV_PUSH_MFMA(m, k * T + p)
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < T; ++j) cur[j] = nxt[j];
    }
    // the trailing HALF channels: their windows run past the tensor (zeros)
#pragma unroll
    for (int j = 0; j < HALF; ++j) PV_PUSH_MFMA(0.0f, a.c + j)
#undef PV_PUSH_MFMA

    // ---- epilogue: acc[kt][16 b + v] at lane (j = lane & 31, h = lane >> 5) = output channel 32 kt + 8 (v / 4) + 4 h + v % 4 of the pixel of lane 32 b + j
    const int h2 = lane >> 5;
    unsigned outo_b[2];
    outo_b[0] = (unsigned)__shfl((int)outo, lane & 31, kWave);
    outo_b[1] = (unsigned)__shfl((int)outo, 32 + (lane & 31), kWave);
    float* const yimg = ca.y + (size_t)img * ca.k_out * ohw;
    const ActBounds ab = act_bounds(ca.act, ca.act_lo, ca.act_hi);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        float bv[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int kk = 32 * kt + 8 * (v >> 2) + 4 * h2 + (v & 3);
            bv[v] = (ca.cbias != nullptr && kk < ca.k_out) ? ca.cbias[kk] : 0.0f;
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float vv[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) vv[v] = acc[kt][16 * b + v];
            bias_act_n<16>(vv, bv, ca.cbias != nullptr, ca.act, ab);
            if (outo_b[b] != 0xffffffffu) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int kk = 32 * kt + 8 * (v >> 2) + 4 * h2 + (v & 3);
                    if (kk < ca.k_out) yimg[(size_t)kk * ohw + outo_b[b]] = vv[v];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LRN + MaxPool 3x3 / stride 2 / no padding, WITHOUT A BARRIER (round 4; GoogLeNet's conv2/norm2 -> pool2/3x3_s2).  The workgroup form
// above spends a third of its time in its pooling phase and 0.076 of its 0.184 ms in the loop and its two barriers per eight channels
// (scripts/time_lrnpool_abl.py); its loads alone take 0.112 ms.  Here a WAVE is the unit: it owns four pooled rows of one image and a
// group of channels, i.e. nine input rows of W pixels as W / 8 lanes per row with eight adjacent pixels each (W = 56: 63 lanes), walks
// the channel axis in steps of four (the five-channel window in registers, squares summed in ascending channel order: the arithmetic of
// lrn_window_kernel -- the same bits), leaves four normalised planes in ITS OWN 8 KB of LDS, and pools them itself: a lane owns two
// adjacent outputs of a row, whose windows are five columns of three rows = three 16-byte and three 4-byte LDS reads instead of
// eighteen.  LDS executes a wave's instructions in order: no barrier, no counter, the other waves of the workgroup are strangers.
// The channel groups overlap by the window's reach (two channels each side: 52 loads per 48 channels).
// MEASURED, NOT THE DEFAULT (PVHIP_LRNPOOL_WAVE=1): 0.180 ms against 0.171 for the workgroup form on the same box, the same bits (scripts/
// time_lrnpool.py) -- 140 registers (three waves per SIMD) and ~20 issue slots of LRN arithmetic per element: the barriers were not it.
struct LrnPoolWaveArgs {
    const float* x;
    float*       y;
    int   n, c, h, w, oh, ow;
    int   R, n_bands, rows_in;     // pooled rows per wave, bands per image, input rows of a band (2 R + 1)
    int   cg, n_groups;            // channels per wave (a multiple of four), groups per image
    int   lpr, op;                 // lanes per input row (w / 8); output pairs per pooled row (ceil(ow / 2))
    long  n_items;
    float alpha, beta, bias;
};

template <int BETA_MODE>
__global__ __launch_bounds__(kBlock) void lrn_maxpool3x3_wave_kernel(LrnPoolWaveArgs a) {
    constexpr int T = 4;
    typedef float vec4 __attribute__((ext_vector_type(4)));
    struct V8 { vec4 lo, hi; };
    extern __shared__ __attribute__((aligned(16))) float wave_planes[];              // [4 waves][T][rows_in][w]
    const int lane = threadIdx.x & (kWave - 1);
    const int wid  = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const long item = (long)blockIdx.x * (kBlock / kWave) + wid;
    if (item >= a.n_items) return;
    // item -> (image, band, channel group): the groups of a band back to back (they share the input rows' neighbourhood in L2)
    const int g    = (int)(item % a.n_groups);
    const long ib  = item / a.n_groups;
    const int band = (int)(ib % a.n_bands), img = (int)(ib / a.n_bands);
    const int oy0  = band * a.R;
    const int iy0  = 2 * oy0;
    const int c0   = g * a.cg;
    const int hw = a.h * a.w, ohw = a.oh * a.ow;
    const int plane_l = a.rows_in * a.w;
    float* const lds = wave_planes + (size_t)wid * T * plane_l;

    // ---- load side: lane -> (input row of the band, eight columns)
    const int  lrow = lane / a.lpr, lcol = (lane - lrow * a.lpr) * 8;
    const bool lact = lrow < a.rows_in && iy0 + lrow < a.h;               // a row past the image: zeros (never pooled: the windows are clipped)
    const float* const xl = a.x + (size_t)img * a.c * hw + (size_t)(lact ? (iy0 + lrow) * a.w + lcol : 0);
    float* const lmine = lds + lrow * a.w + lcol;
    const bool lwrite = lrow < a.rows_in;

    // ---- pool side: lane -> (pooled row of the band, two adjacent outputs)
    const int  prow = lane / a.op, pj = lane - prow * a.op;
    const int  oy = oy0 + prow, ox = 2 * pj;
    const bool pact = prow < a.R && oy < a.oh && ox < a.ow;
    const bool second = ox + 1 < a.ow;
    // window rows 2 prow + k (clipped at the image: a row past it repeats the first), columns 4 pj .. 4 pj + 4 (the fifth may be past the row)
    unsigned roff[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int r = 2 * prow + k;
        roff[k] = (unsigned)(((pact && iy0 + r < a.h) ? r : 2 * (pact ? prow : 0)) * a.w + 4 * (pact ? pj : 0));
    }
    const bool col4 = 4 * pj + 4 < a.w;                                   // the fifth column exists
    float* const yout = a.y + ((size_t)img * a.c) * ohw + (size_t)(pact ? oy * a.ow + ox : 0);

    auto load8 = [&](int ch) -> V8 {
        V8 v;
        if (lact && ch >= 0 && ch < a.c) {
            const vec4* p = reinterpret_cast<const vec4*>(xl + (size_t)ch * hw);
            v.lo = ldnt(p); v.hi = ldnt(p + 1);
        } else {
            v.lo = (vec4)(0.0f); v.hi = (vec4)(0.0f);
        }
        return v;
    };
    V8 win[5], nxt[T];
#pragma unroll
    for (int q = 0; q < 5; ++q) { win[q].lo = (vec4)(0.0f); win[q].hi = (vec4)(0.0f); }
    const int n_chunks = a.cg / T + 1;                                    // incoming chunks of four channels: c0 - 2 + 4 i ..; chunk i >= 1 completes channels c0 + 4 (i - 1) ..
#pragma unroll
    for (int j = 0; j < T; ++j) nxt[j] = load8(c0 - 2 + j);
    for (int i = 0; i < n_chunks; ++i) {
        V8 cur[T];
#pragma unroll
        for (int j = 0; j < T; ++j) cur[j] = nxt[j];
        if (i + 1 < n_chunks) {
#pragma unroll
            for (int j = 0; j < T; ++j) nxt[j] = load8(c0 - 2 + T * (i + 1) + j);
        }
#pragma unroll
        for (int j = 0; j < T; ++j) {
            // channel c0 - 2 + 4 i + j enters the window; the window is then centred on the channel two below it
#pragma unroll
            for (int q = 0; q < 4; ++q) win[q] = win[q + 1];
            win[4] = cur[j];
            if (i >= 1) {
                vec4 slo = win[0].lo * win[0].lo, shi = win[0].hi * win[0].hi;
#pragma unroll
                for (int q = 1; q < 5; ++q) { slo = slo + win[q].lo * win[q].lo; shi = shi + win[q].hi * win[q].hi; }
                vec4 olo, ohi;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    olo[v] = lrn_div(win[2].lo[v], a.bias + a.alpha * slo[v], a.beta, BETA_MODE);
                    ohi[v] = lrn_div(win[2].hi[v], a.bias + a.alpha * shi[v], a.beta, BETA_MODE);
                }
                if (lwrite) {
                    *reinterpret_cast<vec4*>(lmine + j * plane_l) = olo;
                    *reinterpret_cast<vec4*>(lmine + j * plane_l + 4) = ohi;
                }
            }
        }
        if (i == 0) continue;
        __builtin_amdgcn_wave_barrier();                                  // (a scheduling fence: LDS runs this wave's writes before its reads)
        const int ch0 = c0 + T * (i - 1);
        if (pact) {
#pragma unroll
            for (int p = 0; p < T; ++p) {
                const float* const pl = lds + p * plane_l;
                float m0[3], m1[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const vec4  v  = *reinterpret_cast<const vec4*>(pl + roff[k]);
                    const float v4 = col4 ? pl[roff[k] + 4] : v[3];
                    m0[k] = max3_nan(v[0], v[1], v[2]);
                    m1[k] = max3_nan(v[2], v[3], v4);
                }
                const float r0 = max3_nan(m0[0], m0[1], m0[2]), r1 = max3_nan(m1[0], m1[1], m1[2]);
                float* const yo = yout + (size_t)(ch0 + p) * ohw;
                if (second && (a.ow & 1) == 0) *reinterpret_cast<float2*>(yo) = make_float2(r0, r1);
                else { yo[0] = r0; if (second) yo[1] = r1; }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Geometry of the wave form, or false when the pair is outside it (then the workgroup form above is asked).
bool plan_lrn_pool_wave(int n, int c, int h, int w, int size, float beta, float bias, int oh, int ow, int kh, int kw, int sh, int sw, int pt, int pl,
                        int pb, int pr, LrnPoolWaveArgs& a, int& bm) {
    if (n <= 0 || c < 8 || c % 4 != 0 || h < 3 || w < 8 || w % 8 != 0 || oh <= 0 || ow <= 0) return false;
    if (size != 5 || kh != 3 || kw != 3 || sh != 2 || sw != 2 || pt != 0 || pl != 0 || pb != 0 || pr != 0) return false;
    if (2 * (oh - 1) > h - 1 || 2 * (ow - 1) > w - 1 || 2 * (ow - 1) + 2 > w) return false;      // the first cell of every window is in the image; at most the third column / row is clipped
    if ((unsigned long long)n * c * h * w >= (1ull << 31)) return false;
    bm = lrn_beta_mode(beta, bias);
    if (bm != 4 && bm != 1) return false;
    const int lpr = w / 8;
    const int R = (kWave / lpr - 1) / 2;                  // rows_in = 2 R + 1 input rows in 64 lanes
    const int op = (ow + 1) / 2;
    if (R < 1 || R * op > kWave) return false;
    int cg = 0;
    for (int cand = 64; cand >= 16; cand -= 4)            // the largest group of at most 64 channels that divides C (48 for C = 192)
        if (c % cand == 0) { cg = cand; break; }
    if (cg == 0) return false;
    a.n = n; a.c = c; a.h = h; a.w = w; a.oh = oh; a.ow = ow;
    a.R = R; a.n_bands = (oh + R - 1) / R; a.rows_in = 2 * R + 1;
    a.cg = cg; a.n_groups = c / cg; a.lpr = lpr; a.op = op;
    a.n_items = (long)n * a.n_bands * a.n_groups;
    return (size_t)4 * 4 * a.rows_in * w * sizeof(float) <= 64 * 1024;
}

// Geometry of the fused launch, or false when the pair is outside what lrn_maxpool3x3_kernel covers.
bool plan_lrn_pool(int n, int c, int h, int w, int size, float beta, float bias, int oh, int ow, int kh, int kw, int sh, int sw,
                   int pt, int pl, int pb, int pr, LrnPoolArgs& a, int& vec, int& bm, size_t& lds) {
    if (n <= 0 || c < 8 || c % 8 != 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0) return false;
    if (size != 5 || kh != 3 || kw != 3 || sh != sw || (sh != 1 && sh != 2)) return false;
    if (pt < 0 || pl < 0 || pb < 0 || pr < 0 || pt > 2 || pl > 2) return false;
    if ((oh - 1) * sh > pt + h - 1 || (ow - 1) * sw > pl + w - 1) return false;       // clamped taps stay inside their window
    if ((unsigned long long)n * c * h * w >= (1ull << 31)) return false;
    bm = lrn_beta_mode(beta, bias);
    if (bm != 4 && bm != 1) return false;
    vec = (w % 4 == 0) ? 4 : 1;
    int rows = (kBlock * vec / w - 3) / sh + 1;                 // largest band whose input rows fit the 256 lanes
    if (kBlock * vec / w < 3 && h >= 3) return false;
    if (rows < 1) return false;
    if (rows > oh) rows = oh;
    int bands = (oh + rows - 1) / rows;
    rows  = (oh + bands - 1) / bands;
    bands = (oh + rows - 1) / rows;
    int in_rows = (rows - 1) * sh + 3;
    if (in_rows > h) in_rows = h;
    if (in_rows * w > kBlock * vec) return false;
    a.n = n; a.c = c; a.h = h; a.w = w; a.oh = oh; a.ow = ow; a.pt = pt; a.pl = pl; a.hp = h + pt + pb; a.wp = w + pl + pr;
    a.band_rows = rows; a.n_bands = bands;
    // the kernels are instantiated for up to four pooled outputs per lane and plane; a pooled row wider than the input row (ow > w) with
    // a tall band needs more: the QUERY must say no there, so that the caller falls back to two launches instead of failing at run time
    if ((rows * ow + kBlock - 1) / kBlock > 4) return false;
    a.plane_l   = (in_rows * w + 3) & ~3;
    lds         = (size_t)8 * a.plane_l * sizeof(float);
    return lds <= 64 * 1024 && (long long)n * bands < (1ll << 31);
}

// Fallback for window sizes without a register-window instantiation: every lane re-reads its window.
__global__ __launch_bounds__(kBlock) void lrn_generic_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                                              int c, int hw, int size, float alpha, float beta,
                                                              float bias, int beta_mode) {
    const size_t total  = (size_t)n * c * hw;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int    half   = size / 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t p   = e % hw;
        const size_t q   = e / hw;
        const int    ch  = (int)(q % c);
        const size_t img = q / c;
        int lo = ch - half, hi = ch + half + 1;
        if (lo < 0) lo = 0;
        if (hi > c) hi = c;
        float s = 0.0f;
        for (int k = lo; k < hi; ++k) {
            const float v = x[(img * c + k) * hw + p];
            s             = (k == lo) ? v * v : s + v * v;
        }
        const float d = bias + alpha * s;
        y[e]          = lrn_div(x[e], d, beta, beta_mode);
    }
}

template <int SIZE, int BETA_MODE>
void launch_lrn_window_b(const float* x, float* y, int n, int c, int hw, float alpha, float beta, float bias) {
    if (hw % 4 == 0) {
        const size_t cols = (size_t)n * (hw / 4);
        hipLaunchKernelGGL((lrn_window_kernel<SIZE, 4, BETA_MODE>), dim3(grid_for(cols)), dim3(kBlock), 0, state().stream, x, y,
                           n, c, hw, alpha, beta, bias);
    } else {
        const size_t cols = (size_t)n * hw;
        hipLaunchKernelGGL((lrn_window_kernel<SIZE, 1, BETA_MODE>), dim3(grid_for(cols)), dim3(kBlock), 0, state().stream, x, y,
                           n, c, hw, alpha, beta, bias);
    }
}

template <int SIZE>
void launch_lrn_window(const float* x, float* y, int n, int c, int hw, float alpha, float beta, float bias, int bm) {
    switch (bm) {
        case 1: launch_lrn_window_b<SIZE, 1>(x, y, n, c, hw, alpha, beta, bias); break;
        case 2: launch_lrn_window_b<SIZE, 2>(x, y, n, c, hw, alpha, beta, bias); break;
        case 3: launch_lrn_window_b<SIZE, 3>(x, y, n, c, hw, alpha, beta, bias); break;
        case 4: launch_lrn_window_b<SIZE, 4>(x, y, n, c, hw, alpha, beta, bias); break;
        default: launch_lrn_window_b<SIZE, 0>(x, y, n, c, hw, alpha, beta, bias); break;
    }
}

}  // namespace

extern "C" {

int pvhip_softmax_rows_f32(const float* x, float* y, int rows, int cols) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(rows >= 0 && cols >= 0);
    if (rows == 0 || cols == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr);
    const int g = rows < kMaxBlocks ? rows : kMaxBlocks;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(g), dim3(kBlock), 0, state().stream, x, y, rows, cols);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_lrn_f32(const float* x, float* y, int n, int c, int hw, int size, float alpha, float beta, float bias) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && c >= 0 && hw >= 0 && size >= 1);
    if ((size_t)n * c * hw == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr);
    if ((unsigned long long)n * c * hw >= (1ull << 31))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_lrn_f32: tensor exceeds 2^31 elements");
    const int bm = lrn_beta_mode(beta, bias);
    switch ((c % 8 == 0 && c >= 8) ? size : 0) {
        case 3: launch_lrn_window<3>(x, y, n, c, hw, alpha, beta, bias, bm); break;
        case 5: launch_lrn_window<5>(x, y, n, c, hw, alpha, beta, bias, bm); break;
        case 7: launch_lrn_window<7>(x, y, n, c, hw, alpha, beta, bias, bm); break;
        default:
            hipLaunchKernelGGL(lrn_generic_kernel, dim3(grid_for((size_t)n * c * hw)), dim3(kBlock), 0, state().stream, x, y,
                               n, c, hw, size, alpha, beta, bias, bm);
    }
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_lrn_maxpool_supported(int n, int c, int h, int w, int size, float beta, float bias, int oh, int ow, int kh, int kw,
                                int sh, int sw, int pad_top, int pad_left, int pad_bottom, int pad_right) {
    LrnPoolArgs a{};
    int vec = 0, bm = 0;
    size_t lds = 0;
    return plan_lrn_pool(n, c, h, w, size, beta, bias, oh, ow, kh, kw, sh, sw, pad_top, pad_left, pad_bottom, pad_right, a, vec, bm, lds) ? 1 : 0;
}

int pvhip_lrn_maxpool_f32(const float* x, float* y, int n, int c, int h, int w, int size, float alpha, float beta, float bias,
                          int oh, int ow, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int pad_bottom,
                          int pad_right) {
    PVHIP_REQUIRE_INIT();
    {
        LrnPoolWaveArgs wa{};
        int wbm = 0;
        if (settings().lrnpool_wave && plan_lrn_pool_wave(n, c, h, w, size, beta, bias, oh, ow, kh, kw, sh, sw, pad_top, pad_left, pad_bottom, pad_right, wa, wbm)) {
            PVHIP_CHECK_ARG(x != nullptr && y != nullptr);
            wa.x = x; wa.y = y; wa.alpha = alpha; wa.beta = beta; wa.bias = bias;
            const long blocks = (wa.n_items + kBlock / kWave - 1) / (kBlock / kWave);
            if (blocks <= 0x7fffffffL) {
                const size_t wlds = (size_t)(kBlock / kWave) * 4 * wa.rows_in * w * sizeof(float);
                if (wbm == 4) hipLaunchKernelGGL((lrn_maxpool3x3_wave_kernel<4>), dim3((unsigned)blocks), dim3(kBlock), wlds, state().stream, wa);
                else          hipLaunchKernelGGL((lrn_maxpool3x3_wave_kernel<1>), dim3((unsigned)blocks), dim3(kBlock), wlds, state().stream, wa);
                PVHIP_LAUNCH_CHECK();
                return PVHIP_OK;
            }
        }
    }
    LrnPoolArgs a{};
    int vec = 0, bm = 0;
    size_t lds = 0;
    if (!plan_lrn_pool(n, c, h, w, size, beta, bias, oh, ow, kh, kw, sh, sw, pad_top, pad_left, pad_bottom, pad_right, a, vec, bm, lds))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_lrn_maxpool_f32: shape outside the fused kernel (ask pvhip_lrn_maxpool_supported first)");
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr);
    a.x = x; a.y = y; a.alpha = alpha; a.beta = beta; a.bias = bias;
    a.abl = 0;
#ifdef PVHIP_DIAG
    a.abl = settings().conv_ablate;
#endif
    // four pooled outputs per lane (pool4 in the kernel): PVHIP_TUNE6=1 keeps one per lane (A/B runs)
    a.pool4 = (vec == 4 && sh == 2 && pad_top == 0 && pad_left == 0 && ow % 4 == 0 && 2 * ow <= w && (ow / 4) * a.band_rows <= kWave && settings().tune[6] != 1) ? 1 : 0;      // (2 ow <= w: the eight columns a quad reads whole lie in the row; only the ninth is clamped)
    const dim3 grid((unsigned)(n * a.n_bands));
    const FastDiv d_bands = make_fastdiv((unsigned)a.n_bands), d_ow = make_fastdiv((unsigned)ow);
    const int ni = (a.band_rows * ow + kBlock - 1) / kBlock;          // 1 at stride 2 (a band holds <= 1024 input pixels), up to 4 at stride 1
#define PV_LP(VEC_, BM_, ST_, NI_) \
    hipLaunchKernelGGL((lrn_maxpool3x3_kernel<5, VEC_, BM_, ST_, NI_>), grid, dim3(kBlock), lds, state().stream, a, d_bands, d_ow)
#define PV_LP_NI(VEC_, BM_, ST_)                                  \
    {                                                             \
        if (ni <= 1) PV_LP(VEC_, BM_, ST_, 1);                    \
        else if (ni <= 2) PV_LP(VEC_, BM_, ST_, 2);               \
        else PV_LP(VEC_, BM_, ST_, 4);                            \
    }
    if (ni > 4) return fail(PVHIP_EUNSUPPORTED, "pvhip_lrn_maxpool_f32: more than four pooled outputs per lane and plane");
    if (vec == 4) {
        if (bm == 4) { if (sh == 1) PV_LP_NI(4, 4, 1) else PV_LP_NI(4, 4, 2) }
        else         { if (sh == 1) PV_LP_NI(4, 1, 1) else PV_LP_NI(4, 1, 2) }
    } else {
        if (bm == 4) { if (sh == 1) PV_LP_NI(1, 4, 1) else PV_LP_NI(1, 4, 2) }
        else         { if (sh == 1) PV_LP_NI(1, 1, 1) else PV_LP_NI(1, 1, 2) }
    }
#undef PV_LP_NI
#undef PV_LP
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

// MaxPool (3x3) then LRN as one launch: same coverage rules as the other order (the geometry is the pooling's either way).
int pvhip_maxpool_lrn_supported(int n, int c, int h, int w, int oh, int ow, int kh, int kw, int sh, int sw, int pad_top, int pad_left,
                                int pad_bottom, int pad_right, int size, float beta, float bias) {
    LrnPoolArgs a{};
    int vec = 0, bm = 0;
    size_t lds = 0;
    return plan_lrn_pool(n, c, h, w, size, beta, bias, oh, ow, kh, kw, sh, sw, pad_top, pad_left, pad_bottom, pad_right, a, vec, bm, lds) ? 1 : 0;
}

int pvhip_maxpool_lrn_f32(const float* x, float* y, int n, int c, int h, int w, int oh, int ow, int kh, int kw, int sh, int sw,
                          int pad_top, int pad_left, int pad_bottom, int pad_right, int size, float alpha, float beta, float bias) {
    PVHIP_REQUIRE_INIT();
    LrnPoolArgs a{};
    int vec = 0, bm = 0;
    size_t lds = 0;
    if (!plan_lrn_pool(n, c, h, w, size, beta, bias, oh, ow, kh, kw, sh, sw, pad_top, pad_left, pad_bottom, pad_right, a, vec, bm, lds))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_maxpool_lrn_f32: shape outside the fused kernel (ask pvhip_maxpool_lrn_supported first)");
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr);
    a.x = x; a.y = y; a.alpha = alpha; a.beta = beta; a.bias = bias;
    const dim3 grid((unsigned)(n * a.n_bands));
    const FastDiv d_bands = make_fastdiv((unsigned)a.n_bands), d_ow = make_fastdiv((unsigned)ow);
    const int ni = (a.band_rows * ow + kBlock - 1) / kBlock;
    if (ni > 4) return fail(PVHIP_EUNSUPPORTED, "pvhip_maxpool_lrn_f32: more than four pooled outputs per lane and plane");
#define PV_PL(VEC_, BM_, ST_, NI_) \
    hipLaunchKernelGGL((maxpool3x3_lrn_kernel<5, VEC_, BM_, ST_, NI_>), grid, dim3(kBlock), lds, state().stream, a, d_bands, d_ow)
#define PV_PL_NI(VEC_, BM_, ST_)                                  \
    {                                                             \
        if (ni <= 1) PV_PL(VEC_, BM_, ST_, 1);                    \
        else if (ni <= 2) PV_PL(VEC_, BM_, ST_, 2);               \
        else PV_PL(VEC_, BM_, ST_, 4);                            \
    }
    if (vec == 4) {
        if (bm == 4) { if (sh == 1) PV_PL_NI(4, 4, 1) else PV_PL_NI(4, 4, 2) }
        else         { if (sh == 1) PV_PL_NI(4, 1, 1) else PV_PL_NI(4, 1, 2) }
    } else {
        if (bm == 4) { if (sh == 1) PV_PL_NI(1, 4, 1) else PV_PL_NI(1, 4, 2) }
        else         { if (sh == 1) PV_PL_NI(1, 1, 1) else PV_PL_NI(1, 1, 2) }
    }
#undef PV_PL_NI
#undef PV_PL
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

// MaxPool (3x3) then LRN then a 1x1 / stride 1 / unpadded convolution (+ bias, activation) as ONE launch (round 5): the MaxPool + LRN geometry of
// pvhip_maxpool_lrn_f32 with one pooled output per lane (bands of at most 256 outputs), a window of five channels, rows of a multiple of four
// pixels; at most 64 input and 64 output channels (the transposed weights stay in LDS).
static bool plan_pool_lrn_conv(int n, int c, int h, int w, int oh, int ow, int kh, int kw, int sh, int sw, int pt, int pl, int pb, int pr, int size,
                               float beta, float bias, int k_out, LrnPoolArgs& a, int& bm, size_t& lds) {
    int vec = 0;
    if (!plan_lrn_pool(n, c, h, w, size, beta, bias, oh, ow, kh, kw, sh, sw, pt, pl, pb, pr, a, vec, bm, lds)) return false;
    if (vec != 4 || a.band_rows * ow > kBlock || k_out <= 0 || k_out > 64 || c > 64) return false;
    if ((unsigned long long)n * k_out * oh * ow >= (1ull << 31)) return false;
    lds += (size_t)c * 32 * ((k_out + 31) / 32) * sizeof(float);
    return lds <= 64 * 1024;
}

int pvhip_maxpool_lrn_conv1x1_supported(int n, int c, int h, int w, int oh, int ow, int kh, int kw, int sh, int sw, int pad_top, int pad_left,
                                        int pad_bottom, int pad_right, int size, float beta, float bias, int k_out) {
    LrnPoolArgs a{};
    int bm = 0;
    size_t lds = 0;
    return plan_pool_lrn_conv(n, c, h, w, oh, ow, kh, kw, sh, sw, pad_top, pad_left, pad_bottom, pad_right, size, beta, bias, k_out, a, bm, lds) ? 1 : 0;
}

int pvhip_maxpool_lrn_conv1x1_f32(const float* x, const float* w_oihw, float* y, int n, int c, int h, int w, int oh, int ow, int kh, int kw, int sh, int sw,
                                  int pad_top, int pad_left, int pad_bottom, int pad_right, int size, float alpha, float beta, float bias,
                                  int k_out, const float* conv_bias, int act, float act_lo, float act_hi) {
    PVHIP_REQUIRE_INIT();
    PoolLrnConvArgs ca{};
    int bm = 0;
    size_t lds = 0;
    if (!plan_pool_lrn_conv(n, c, h, w, oh, ow, kh, kw, sh, sw, pad_top, pad_left, pad_bottom, pad_right, size, beta, bias, k_out, ca.p, bm, lds))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_maxpool_lrn_conv1x1_f32: shape outside the fused kernel (ask pvhip_maxpool_lrn_conv1x1_supported first)");
    PVHIP_CHECK_ARG(x != nullptr && w_oihw != nullptr && y != nullptr && act >= 0 && act <= 2);
    ca.p.x = x; ca.p.y = nullptr; ca.p.alpha = alpha; ca.p.beta = beta; ca.p.bias = bias;
    ca.cw = w_oihw; ca.cbias = conv_bias; ca.y = y; ca.k_out = k_out; ca.act = act; ca.act_lo = act_lo; ca.act_hi = act_hi;
    const dim3 grid((unsigned)(n * ca.p.n_bands));
    const FastDiv d_bands = make_fastdiv((unsigned)ca.p.n_bands), d_ow = make_fastdiv((unsigned)ow);
    const int kt = (k_out + 31) / 32;
#define PV_PLC(BM_, ST_, KT_) hipLaunchKernelGGL((maxpool3x3_lrn_conv1x1_kernel<BM_, ST_, KT_>), grid, dim3(kBlock), lds, state().stream, ca, d_bands, d_ow)
    if (bm == 4) {
        if (sh == 1) { if (kt == 1) PV_PLC(4, 1, 1); else PV_PLC(4, 1, 2); }
        else         { if (kt == 1) PV_PLC(4, 2, 1); else PV_PLC(4, 2, 2); }
    } else {
        if (sh == 1) { if (kt == 1) PV_PLC(1, 1, 1); else PV_PLC(1, 1, 2); }
        else         { if (kt == 1) PV_PLC(1, 2, 1); else PV_PLC(1, 2, 2); }
    }
#undef PV_PLC
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}


}  // extern "C"
