// Pooling kernels (HBM-bound).  NCHW fp32; lanes run along the output row so the window loads of a
// wave are contiguous (stride sw) segments of input rows and the stores are fully coalesced.
#include <cstdlib>

#include "pvhip_common.h"

using namespace pvhip;

namespace {

struct PoolArgs {
    int n_planes;  // N*C
    int h, w, oh, ow;
    int kh, kw, sh, sw;
    int pt, pl;    // pad top / left
    int hp, wp;    // padded extents (h + pt + pb, w + pl + pr)
};

// MaxPool over the zero-padded input, window clipped at the padded extent (MaxPool.py:41-72):
// a pad cell is a candidate with value 0.0; np.max semantics (NaN wins).
__global__ __launch_bounds__(kBlock) void maxpool2d_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            PoolArgs a, unsigned total) {
    const unsigned stride = gridDim.x * blockDim.x;
    const unsigned ohw    = (unsigned)(a.oh * a.ow);
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const unsigned plane = e / ohw;
        const unsigned rem   = e - plane * ohw;
        const int      oy    = (int)(rem / (unsigned)a.ow);
        const int      ox    = (int)(rem - (unsigned)oy * (unsigned)a.ow);
        const float* __restrict__ xp = x + (size_t)plane * (size_t)(a.h * a.w);
        const int py0 = oy * a.sh, px0 = ox * a.sw;  // window origin in padded coordinates
        float     m   = -INFINITY;
        for (int ky = 0; ky < a.kh; ++ky) {
            const int py = py0 + ky;
            if (py >= a.hp) break;
            const int  iy    = py - a.pt;
            const bool row_in = (unsigned)iy < (unsigned)a.h;
            for (int kx = 0; kx < a.kw; ++kx) {
                const int px = px0 + kx;
                if (px >= a.wp) break;
                const int ix = px - a.pl;
                float     v  = 0.0f;
                if (row_in && (unsigned)ix < (unsigned)a.w) v = xp[iy * a.w + ix];
                m = (v > m || v != v) ? v : m;
            }
        }
        y[e] = m;
    }
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

struct PoolDivs {
    FastDiv hw, w, ohw, ow;
};

// LDS-staged MaxPool: a workgroup owns G consecutive (n, c) planes.  Planes are contiguous in NCHW, so
// the input of a group is ONE dense run of G*H*W floats: it is read from HBM exactly once with 16-byte
// loads and laid out in LDS as G zero-PADDED planes [hp][wp] (the pad cells hold the 0.0 that takes part
// in the reference's max), so a window is kh*kw unconditional LDS reads.  Lanes own consecutive outputs
// (conflict-free LDS reads at stride 1, 2-way at stride 2); the outputs of a group are again one dense
// run, written coalesced.  CLIP handles ceil-mode windows that overhang the padded extent (those cells
// are excluded from the max, MaxPool.py:69).  np.max semantics: a NaN in the window wins.
template <int KH, int KW, bool CLIP>   // KH == 0: run-time window extent
__global__ __launch_bounds__(kBlock) void maxpool2d_lds_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                PoolArgs a, int G, int band_rows, PoolDivs dv) {
    // blockIdx.x: group of G planes; blockIdx.y: band of `band_rows` output rows (bands > 1 only with G == 1,
    // for planes too large to stage whole).  The LDS image holds the PADDED rows [py_lo, py_hi) the band needs.
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int kh = KH ? KH : a.kh, kw = KW ? KW : a.kw;
    const int g0 = blockIdx.x * G;
    const int gn = min(G, a.n_planes - g0);
    const int oy0 = blockIdx.y * band_rows;
    const int oy1 = min(a.oh, oy0 + band_rows);
    const bool whole = (gridDim.y == 1);             // whole planes: the dense run spans planes, so stage every row
    const int py_lo = oy0 * a.sh;
    const int py_hi = whole ? a.hp : min(a.hp, (oy1 - 1) * a.sh + kh);
    const int iy_lo = max(0, py_lo - a.pt);
    const int iy_hi = min(a.h, py_hi - a.pt);
    const int hb    = iy_hi - iy_lo;                 // input rows of this band actually present in the tensor
    const int hwb = hb * a.w, obw = (oy1 - oy0) * a.ow;
    const int wl = a.wp, plane_l = (py_hi - py_lo) * a.wp;
    const int n_in = gn * hwb, n_out = gn * obw;
    const size_t in_off = (size_t)g0 * a.h * a.w + (size_t)iy_lo * a.w;
    const float* __restrict__ xin = x + in_off;
    float* __restrict__ yout      = y + (size_t)g0 * a.oh * a.ow + (size_t)oy0 * a.ow;
    const bool padded  = (plane_l != hwb);
    const bool aligned = ((in_off & 3) == 0);
    const int  row_shift = iy_lo + a.pt - py_lo;     // LDS row of input row iy_lo (0 unless the band starts in the top pad)

    if (!padded) {
        if (aligned) {
            const float4* __restrict__ x4 = reinterpret_cast<const float4*>(xin);
            float4* t4 = reinterpret_cast<float4*>(tile);
            const int n4 = n_in >> 2;
            for (int i = threadIdx.x; i < n4; i += kBlock) t4[i] = x4[i];
            for (int i = (n4 << 2) + threadIdx.x; i < n_in; i += kBlock) tile[i] = xin[i];
        } else {
            for (int i = threadIdx.x; i < n_in; i += kBlock) tile[i] = xin[i];
        }
    } else {
        {
            float4* t4 = reinterpret_cast<float4*>(tile);
            const int n4 = (gn * plane_l + 3) >> 2;   // the allocation is rounded up to 16 bytes
            for (int i = threadIdx.x; i < n4; i += kBlock) t4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        __syncthreads();
        const int n4 = aligned ? (n_in >> 2) : 0;
        const float4* __restrict__ x4 = reinterpret_cast<const float4*>(xin);
        for (int i = threadIdx.x; i < n4; i += kBlock) {
            const float4   v = x4[i];
            const unsigned e = (unsigned)i * 4u;
            unsigned g  = whole ? fdiv(e, dv.hw) : 0u;
            unsigned r  = e - g * (unsigned)hwb;
            unsigned iy = fdiv(r, dv.w);
            unsigned ix = r - iy * (unsigned)a.w;
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tile[g * plane_l + (iy + row_shift) * wl + ix + a.pl] = vv[q];
                if (++ix == (unsigned)a.w) { ix = 0; if (++iy == (unsigned)hb) { iy = 0; ++g; } }
            }
        }
        for (int e = (n4 << 2) + threadIdx.x; e < n_in; e += kBlock) {
            const unsigned g = whole ? fdiv((unsigned)e, dv.hw) : 0u, r = (unsigned)e - g * (unsigned)hwb;
            const unsigned iy = fdiv(r, dv.w), ix = r - iy * (unsigned)a.w;
            tile[g * plane_l + (iy + row_shift) * wl + ix + a.pl] = xin[e];
        }
    }
    __syncthreads();

    for (int o = threadIdx.x; o < n_out; o += kBlock) {
        const unsigned g   = whole ? fdiv((unsigned)o, dv.ohw) : 0u;
        const unsigned rem = (unsigned)o - g * (unsigned)obw;
        const unsigned oyl = fdiv(rem, dv.ow);
        const unsigned ox  = rem - oyl * (unsigned)a.ow;
        const int py0 = (int)(oy0 + oyl) * a.sh, px0 = (int)ox * a.sw;
        const float* __restrict__ tp = tile + g * plane_l + (py0 - py_lo) * wl + px0;
        float m      = -INFINITY;
        bool  anynan = false;
        if (KH != 0) {
#pragma unroll
            for (int ky = 0; ky < (KH ? KH : 1); ++ky) {
                const bool row_ok = !CLIP || (py0 + ky < a.hp);
#pragma unroll
                for (int kx = 0; kx < (KW ? KW : 1); ++kx) {
                    if (row_ok && (!CLIP || (px0 + kx < a.wp))) {
                        const float v = tp[ky * wl + kx];
                        anynan |= (v != v);
                        m = fmaxf(m, v);
                    }
                }
            }
        } else {
            for (int ky = 0; ky < kh && py0 + ky < a.hp; ++ky)
                for (int kx = 0; kx < kw && px0 + kx < a.wp; ++kx) {
                    const float v = tp[ky * wl + kx];
                    anynan |= (v != v);
                    m = fmaxf(m, v);
                }
        }
        yout[o] = anynan ? NAN : m;
    }
}

// AvgPool with the reference's window rule (AvgPool.py:56): rows [oy*sh, min(h-1, oy*sh+kh)),
// cols [ox*sw, min(w-1, ox*sw+kw)), no padding; mean = sum / count in fp32; empty window -> NaN.
__global__ __launch_bounds__(kBlock) void avgpool2d_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            PoolArgs a, unsigned total) {
    const unsigned stride = gridDim.x * blockDim.x;
    const unsigned ohw    = (unsigned)(a.oh * a.ow);
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const unsigned plane = e / ohw;
        const unsigned rem   = e - plane * ohw;
        const int      oy    = (int)(rem / (unsigned)a.ow);
        const int      ox    = (int)(rem - (unsigned)oy * (unsigned)a.ow);
        const float* __restrict__ xp = x + (size_t)plane * (size_t)(a.h * a.w);
        const int y0 = oy * a.sh, x0 = ox * a.sw;
        int       y1 = y0 + a.kh, x1 = x0 + a.kw;
        if (y1 > a.h - 1) y1 = a.h - 1;
        if (x1 > a.w - 1) x1 = a.w - 1;
        float sum = 0.0f;
        int   cnt = 0;
        for (int iy = y0; iy < y1; ++iy)
            for (int ix = x0; ix < x1; ++ix) {
                sum += xp[iy * a.w + ix];
                ++cnt;
            }
        y[e] = (cnt > 0) ? sum / (float)cnt : NAN;
    }
}

// LDS-staged AvgPool for small planes (the 7x7 global pool of the classifiers): a workgroup streams G whole
// planes (one dense run) into LDS with 16-byte loads, then one lane per output averages its window from LDS.
// Same window rule as avgpool2d_kernel (AvgPool.py:56), same sequential fp32 sum.
__global__ __launch_bounds__(kBlock) void avgpool2d_lds_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                PoolArgs a, int G, PoolDivs dv) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int hw = a.h * a.w, ohw = a.oh * a.ow;
    const int g0 = blockIdx.x * G;
    const int gn = min(G, a.n_planes - g0);
    const int n_in = gn * hw, n_out = gn * ohw;
    const float* __restrict__ xin = x + (size_t)g0 * hw;
    if ((((size_t)g0 * hw) & 3) == 0) {
        const float4* __restrict__ x4 = reinterpret_cast<const float4*>(xin);
        float4* t4 = reinterpret_cast<float4*>(tile);
        const int n4 = n_in >> 2;
        for (int i = threadIdx.x; i < n4; i += kBlock) t4[i] = x4[i];
        for (int i = (n4 << 2) + threadIdx.x; i < n_in; i += kBlock) tile[i] = xin[i];
    } else {
        for (int i = threadIdx.x; i < n_in; i += kBlock) tile[i] = xin[i];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < n_out; o += kBlock) {
        const unsigned g   = fdiv((unsigned)o, dv.ohw);
        const unsigned rem = (unsigned)o - g * (unsigned)ohw;
        const unsigned oy  = fdiv(rem, dv.ow);
        const unsigned ox  = rem - oy * (unsigned)a.ow;
        const float* __restrict__ tp = tile + g * hw;
        const int y0 = (int)oy * a.sh, x0 = (int)ox * a.sw;
        int       y1 = y0 + a.kh, x1 = x0 + a.kw;
        if (y1 > a.h - 1) y1 = a.h - 1;
        if (x1 > a.w - 1) x1 = a.w - 1;
        float sum = 0.0f;
        int   cnt = 0;
        for (int iy = y0; iy < y1; ++iy)
            for (int ix = x0; ix < x1; ++ix) {
                sum += tp[iy * a.w + ix];
                ++cnt;
            }
        y[(size_t)g0 * ohw + o] = (cnt > 0) ? sum / (float)cnt : NAN;
    }
}

// Depthwise convolution (GroupConvolution.py:53-79, one input and one output channel per group), LDS-staged
// like MaxPool: a workgroup owns G consecutive (n, channel) planes (or a band of output rows of one plane), the
// zero-PADDED image in LDS makes every tap an unconditional LDS read, the kh*kw weights of the group's planes sit
// in LDS too.  Optional fused epilogue: + bias[channel], then ReLU or clamp (the Add / Clamp nodes that follow
// every depthwise layer of the MobileNet backbone).
struct DwEpilogue {
    const float* bias;   // [channels] or nullptr
    int          act;    // 0 none, 1 ReLU, 2 clamp
    float        lo, hi;
};

template <int KH, int KW>   // 0 = run-time extent
__global__ __launch_bounds__(kBlock) void dwconv2d_lds_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               float* __restrict__ y, PoolArgs a, int channels, int G,
                                                               int band_rows, PoolDivs dv, DwEpilogue ep) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int kh = KH ? KH : a.kh, kw = KW ? KW : a.kw;
    const int g0 = blockIdx.x * G;
    const int gn = min(G, a.n_planes - g0);
    const int oy0 = blockIdx.y * band_rows;
    const int oy1 = min(a.oh, oy0 + band_rows);
    const bool whole = (gridDim.y == 1);
    const int py_lo = oy0 * a.sh;
    const int py_hi = whole ? a.hp : min(a.hp, (oy1 - 1) * a.sh + kh);
    const int iy_lo = max(0, py_lo - a.pt);
    const int iy_hi = min(a.h, py_hi - a.pt);
    const int hb    = iy_hi - iy_lo;
    const int hwb = hb * a.w, obw = (oy1 - oy0) * a.ow;
    const int wl = a.wp, plane_l = (py_hi - py_lo) * a.wp;
    const int n_in = gn * hwb, n_out = gn * obw;
    const size_t in_off = (size_t)g0 * a.h * a.w + (size_t)iy_lo * a.w;
    const float* __restrict__ xin = x + in_off;
    float* __restrict__ yout      = y + (size_t)g0 * a.oh * a.ow + (size_t)oy0 * a.ow;
    const bool aligned   = ((in_off & 3) == 0);
    const int  row_shift = iy_lo + a.pt - py_lo;
    float* wts = tile + ((gn * plane_l + 3) & ~3);           // [gn][kh*kw] after the images

    {   // zero the padded images, stage the weights of this group's planes
        float4* t4 = reinterpret_cast<float4*>(tile);
        const int n4 = (gn * plane_l + 3) >> 2;
        for (int i = threadIdx.x; i < n4; i += kBlock) t4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const int kk = kh * kw;
        for (int i = threadIdx.x; i < gn * kk; i += kBlock) {
            const int g = i / kk, t = i - g * kk;
            wts[i] = w[(size_t)((g0 + g) % channels) * kk + t];
        }
    }
    __syncthreads();
    {
        const int n4 = aligned ? (n_in >> 2) : 0;
        const float4* __restrict__ x4 = reinterpret_cast<const float4*>(xin);
        for (int i = threadIdx.x; i < n4; i += kBlock) {
            const float4   v = x4[i];
            const unsigned e = (unsigned)i * 4u;
            unsigned g  = whole ? fdiv(e, dv.hw) : 0u;
            unsigned r  = e - g * (unsigned)hwb;
            unsigned iy = fdiv(r, dv.w);
            unsigned ix = r - iy * (unsigned)a.w;
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tile[g * plane_l + (iy + row_shift) * wl + ix + a.pl] = vv[q];
                if (++ix == (unsigned)a.w) { ix = 0; if (++iy == (unsigned)hb) { iy = 0; ++g; } }
            }
        }
        for (int e = (n4 << 2) + threadIdx.x; e < n_in; e += kBlock) {
            const unsigned g = whole ? fdiv((unsigned)e, dv.hw) : 0u, r = (unsigned)e - g * (unsigned)hwb;
            const unsigned iy = fdiv(r, dv.w), ix = r - iy * (unsigned)a.w;
            tile[g * plane_l + (iy + row_shift) * wl + ix + a.pl] = xin[e];
        }
    }
    __syncthreads();

    for (int o = threadIdx.x; o < n_out; o += kBlock) {
        const unsigned g   = whole ? fdiv((unsigned)o, dv.ohw) : 0u;
        const unsigned rem = (unsigned)o - g * (unsigned)obw;
        const unsigned oyl = fdiv(rem, dv.ow);
        const unsigned ox  = rem - oyl * (unsigned)a.ow;
        const int py0 = (int)(oy0 + oyl) * a.sh, px0 = (int)ox * a.sw;
        const float* __restrict__ tp = tile + g * plane_l + (py0 - py_lo) * wl + px0;
        const float* __restrict__ wg = wts + g * (kh * kw);
        float sum = 0.0f;
        if (KH != 0) {
#pragma unroll
            for (int ky = 0; ky < (KH ? KH : 1); ++ky)
#pragma unroll
                for (int kx = 0; kx < (KW ? KW : 1); ++kx) sum += tp[ky * wl + kx] * wg[ky * (KW ? KW : 1) + kx];
        } else {
            for (int ky = 0; ky < kh; ++ky)
                for (int kx = 0; kx < kw; ++kx) sum += tp[ky * wl + kx] * wg[ky * kw + kx];
        }
        if (ep.bias != nullptr) sum = sum + ep.bias[(g0 + (int)g) % channels];
        if (ep.act == 1) sum = (sum < 0.0f) ? 0.0f : sum;
        else if (ep.act == 2) { sum = (sum < ep.lo) ? ep.lo : sum; sum = (sum > ep.hi) ? ep.hi : sum; }
        yout[o] = sum;
    }
}

// Direct fallback (planes whose bands do not fit LDS): one lane per output.
struct DwArgs {
    int G, H, W, OH, OW, kh, kw, sh, sw, pt, pl;
};
__global__ __launch_bounds__(kBlock) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ y, DwArgs a, unsigned total, DwEpilogue ep) {
    const unsigned stride = gridDim.x * blockDim.x;
    const unsigned ohw    = (unsigned)(a.OH * a.OW);
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const unsigned plane = e / ohw;  // n*G + g
        const unsigned rem   = e - plane * ohw;
        const int      oy    = (int)(rem / (unsigned)a.OW);
        const int      ox    = (int)(rem - (unsigned)oy * (unsigned)a.OW);
        const int      g     = (int)(plane % (unsigned)a.G);
        const float* __restrict__ xp = x + (size_t)plane * (size_t)(a.H * a.W);
        const float* __restrict__ wg = w + (size_t)g * (a.kh * a.kw);
        const int iy0 = oy * a.sh - a.pt, ix0 = ox * a.sw - a.pl;
        float     sum = 0.0f;
        for (int r = 0; r < a.kh; ++r) {
            const int iy = iy0 + r;
            if ((unsigned)iy >= (unsigned)a.H) continue;
            for (int s = 0; s < a.kw; ++s) {
                const int ix = ix0 + s;
                if ((unsigned)ix >= (unsigned)a.W) continue;
                sum += xp[iy * a.W + ix] * wg[r * a.kw + s];
            }
        }
        if (ep.bias != nullptr) sum = sum + ep.bias[g];
        if (ep.act == 1) sum = (sum < 0.0f) ? 0.0f : sum;
        else if (ep.act == 2) { sum = (sum < ep.lo) ? ep.lo : sum; sum = (sum > ep.hi) ? ep.hi : sum; }
        y[e] = sum;
    }
}

int check_pool_dims(const char* who, int n, int c, int h, int w, int oh, int ow, int kh, int kw, int sh, int sw) {
    if (n < 0 || c < 0 || h <= 0 || w <= 0 || oh < 0 || ow < 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0)
        return fail(PVHIP_EINVAL, "%s: bad dims n=%d c=%d h=%d w=%d oh=%d ow=%d k=%dx%d s=%dx%d", who, n, c, h, w, oh, ow,
                    kh, kw, sh, sw);
    const unsigned long long in_e  = (unsigned long long)n * c * h * w;
    const unsigned long long out_e = (unsigned long long)n * c * oh * ow;
    if (in_e >= (1ull << 31) || out_e >= (1ull << 31))
        return fail(PVHIP_EUNSUPPORTED, "%s: tensor exceeds 2^31 elements", who);
    return PVHIP_OK;
}

}  // namespace

extern "C" {

int pvhip_maxpool2d_f32(const float* x, float* y, int n, int c, int h, int w, int oh, int ow, int kh, int kw, int sh,
                        int sw, int pad_top, int pad_left, int pad_bottom, int pad_right) {
    PVHIP_REQUIRE_INIT();
    int rc = check_pool_dims("pvhip_maxpool2d_f32", n, c, h, w, oh, ow, kh, kw, sh, sw);
    if (rc) return rc;
    PVHIP_CHECK_ARG(pad_top >= 0 && pad_left >= 0 && pad_bottom >= 0 && pad_right >= 0);
    const unsigned total = (unsigned)n * c * oh * ow;
    if (total == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr);
    PoolArgs a{n * c, h, w, oh, ow, kh, kw, sh, sw, pad_top, pad_left, h + pad_top + pad_bottom, w + pad_left + pad_right};
    // every window must start inside the padded extent (numpy would raise on an empty np.max)
    if ((oh - 1) * sh >= a.hp || (ow - 1) * sw >= a.wp)
        return fail(PVHIP_EINVAL, "pvhip_maxpool2d_f32: window starts outside the padded input");
    // LDS-staged path.  ~16 KB of LDS per workgroup (8-10 workgroups per CU overlap each other's load and
    // compute phases; measured best on the GoogLeNet shapes): several whole planes per workgroup when planes
    // are small, bands of output rows of one plane when a plane is larger than the budget.
    size_t group_bytes = 16 * 1024;
    if (const char* e = getenv("PVHIP_POOL_LDS_KB")) group_bytes = (size_t)atoi(e) * 1024;   // tuning runs only
    const size_t row_bytes   = (size_t)a.wp * sizeof(float);
    const size_t plane_bytes = (size_t)a.hp * row_bytes;
    const size_t min_band    = (size_t)(kh + sh) * row_bytes;       // at least two output rows per band
    if (min_band <= 60 * 1024) {
        const int planes = n * c;
        int G = 1, band_rows = oh, n_bands = 1;
        if (plane_bytes <= group_bytes) {
            G = (int)(group_bytes / plane_bytes);
            while (G > 4 && (planes + G - 1) / G < 8 * kNumCU) G >>= 1;
            if ((h * w) % 4 != 0 && G >= 4) G &= ~3;     // every group starts 16-byte aligned in HBM
            if (G > planes) G = planes;
        } else if (plane_bytes > 60 * 1024 || plane_bytes > group_bytes) {
            const size_t budget = plane_bytes <= group_bytes ? plane_bytes : (group_bytes > min_band ? group_bytes : min_band);
            int rows_in = (int)(budget / row_bytes);                 // padded input rows per band
            band_rows   = (rows_in - kh) / sh + 1;
            if (band_rows < 1) band_rows = 1;
            if (band_rows > oh) band_rows = oh;
            n_bands = (oh + band_rows - 1) / band_rows;
            band_rows = (oh + n_bands - 1) / n_bands;                // even out the bands
            n_bands = (oh + band_rows - 1) / band_rows;
        }
        const int    rows_l = (n_bands == 1) ? a.hp : ((band_rows - 1) * sh + kh < a.hp ? (band_rows - 1) * sh + kh : a.hp);
        const size_t lds    = (((size_t)G * rows_l * row_bytes) + 15) & ~(size_t)15;
        if (lds > 64 * 1024 || n_bands > 65535) {
            hipLaunchKernelGGL(maxpool2d_kernel, dim3(grid_for(total)), dim3(kBlock), 0, state().stream, x, y, a, total);
            PVHIP_LAUNCH_CHECK();
            return PVHIP_OK;
        }
        const dim3 grid((planes + G - 1) / G, n_bands);
        const bool clip = ((oh - 1) * sh + kh > a.hp) || ((ow - 1) * sw + kw > a.wp);
        PoolDivs dv{make_fastdiv((unsigned)(h * w)), make_fastdiv((unsigned)w), make_fastdiv((unsigned)(oh * ow)),
                    make_fastdiv((unsigned)ow)};
#define PV_POOL_LAUNCH(KH_, KW_)                                                                              \
    do {                                                                                                      \
        if (clip)                                                                                             \
            hipLaunchKernelGGL((maxpool2d_lds_kernel<KH_, KW_, true>), grid, dim3(kBlock), lds, state().stream, x, y, a, G, band_rows, dv);  \
        else                                                                                                  \
            hipLaunchKernelGGL((maxpool2d_lds_kernel<KH_, KW_, false>), grid, dim3(kBlock), lds, state().stream, x, y, a, G, band_rows, dv); \
    } while (0)
        if (kh == 3 && kw == 3) PV_POOL_LAUNCH(3, 3);
        else if (kh == 2 && kw == 2) PV_POOL_LAUNCH(2, 2);
        else PV_POOL_LAUNCH(0, 0);
#undef PV_POOL_LAUNCH
    } else {
        hipLaunchKernelGGL(maxpool2d_kernel, dim3(grid_for(total)), dim3(kBlock), 0, state().stream, x, y, a, total);
    }
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_avgpool2d_f32(const float* x, float* y, int n, int c, int h, int w, int oh, int ow, int kh, int kw, int sh,
                        int sw) {
    PVHIP_REQUIRE_INIT();
    int rc = check_pool_dims("pvhip_avgpool2d_f32", n, c, h, w, oh, ow, kh, kw, sh, sw);
    if (rc) return rc;
    const unsigned total = (unsigned)n * c * oh * ow;
    if (total == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr);
    PoolArgs a{n * c, h, w, oh, ow, kh, kw, sh, sw, 0, 0, h, w};
    const size_t plane_bytes = (size_t)h * w * sizeof(float);
    if (plane_bytes <= 16 * 1024) {
        const int planes = n * c;
        int       G      = (int)(16 * 1024 / plane_bytes);
        while (G > 4 && (planes + G - 1) / G < 8 * kNumCU) G >>= 1;
        if ((h * w) % 4 != 0 && G >= 4) G &= ~3;
        if (G > planes) G = planes;
        const size_t lds = (((size_t)G * plane_bytes) + 15) & ~(size_t)15;
        PoolDivs dv{make_fastdiv((unsigned)(h * w)), make_fastdiv((unsigned)w), make_fastdiv((unsigned)(oh * ow)),
                    make_fastdiv((unsigned)ow)};
        hipLaunchKernelGGL(avgpool2d_lds_kernel, dim3((planes + G - 1) / G), dim3(kBlock), lds, state().stream, x, y, a, G, dv);
    } else {
        hipLaunchKernelGGL(avgpool2d_kernel, dim3(grid_for(total)), dim3(kBlock), 0, state().stream, x, y, a, total);
    }
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_dwconv2d_f32(const float* x, const float* w, float* y, int n, int g, int h, int wdt, int kh, int kw, int oh,
                       int ow, int sh, int sw, int pad_top, int pad_left, const float* bias, int act, float act_lo,
                       float act_hi) {
    PVHIP_REQUIRE_INIT();
    int rc = check_pool_dims("pvhip_dwconv2d_f32", n, g, h, wdt, oh, ow, kh, kw, sh, sw);
    if (rc) return rc;
    PVHIP_CHECK_ARG(pad_top >= 0 && pad_left >= 0 && act >= 0 && act <= 2);
    const unsigned total = (unsigned)n * g * oh * ow;
    if (total == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && w != nullptr && y != nullptr);
    DwEpilogue ep{bias, act, act_lo, act_hi};
    // padded extents: everything the windows touch (cells past the tensor are zeros by definition)
    int hp = (oh - 1) * sh + kh, wp = (ow - 1) * sw + kw;
    if (hp < h + pad_top) hp = h + pad_top;
    if (wp < wdt + pad_left) wp = wdt + pad_left;
    PoolArgs a{n * g, h, wdt, oh, ow, kh, kw, sh, sw, pad_top, pad_left, hp, wp};
    const size_t group_bytes = 16 * 1024;
    const size_t row_bytes   = (size_t)wp * sizeof(float);
    const size_t plane_bytes = (size_t)hp * row_bytes;
    const size_t min_band    = (size_t)(kh + sh) * row_bytes;
    bool         lds_ok      = min_band <= 48 * 1024;
    int          G = 1, band_rows = oh, n_bands = 1;
    if (lds_ok) {
        const int planes = n * g;
        if (plane_bytes <= group_bytes) {
            G = (int)(group_bytes / plane_bytes);
            while (G > 4 && (planes + G - 1) / G < 8 * kNumCU) G >>= 1;
            if ((h * wdt) % 4 != 0 && G >= 4) G &= ~3;
            if (G > planes) G = planes;
        } else {
            const size_t budget = group_bytes > min_band ? group_bytes : min_band;
            const int    rows_in = (int)(budget / row_bytes);
            band_rows = (rows_in - kh) / sh + 1;
            if (band_rows < 1) band_rows = 1;
            if (band_rows > oh) band_rows = oh;
            n_bands   = (oh + band_rows - 1) / band_rows;
            band_rows = (oh + n_bands - 1) / n_bands;
            n_bands   = (oh + band_rows - 1) / band_rows;
        }
        const int    rows_l = (n_bands == 1) ? hp : ((band_rows - 1) * sh + kh < hp ? (band_rows - 1) * sh + kh : hp);
        const size_t lds    = ((((size_t)G * rows_l * wp + 3) & ~(size_t)3) + (size_t)G * kh * kw) * sizeof(float);
        if (lds > 64 * 1024 || n_bands > 65535) lds_ok = false;
        if (lds_ok) {
            const dim3 grid((planes + G - 1) / G, n_bands);
            PoolDivs dv{make_fastdiv((unsigned)(h * wdt)), make_fastdiv((unsigned)wdt), make_fastdiv((unsigned)(oh * ow)),
                        make_fastdiv((unsigned)ow)};
            if (kh == 3 && kw == 3)
                hipLaunchKernelGGL((dwconv2d_lds_kernel<3, 3>), grid, dim3(kBlock), lds, state().stream, x, w, y, a, g, G,
                                   band_rows, dv, ep);
            else
                hipLaunchKernelGGL((dwconv2d_lds_kernel<0, 0>), grid, dim3(kBlock), lds, state().stream, x, w, y, a, g, G,
                                   band_rows, dv, ep);
        }
    }
    if (!lds_ok) {
        DwArgs d{g, h, wdt, oh, ow, kh, kw, sh, sw, pad_top, pad_left};
        hipLaunchKernelGGL(dwconv_kernel, dim3(grid_for((size_t)total)), dim3(kBlock), 0, state().stream, x, w, y, d, total, ep);
    }
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // extern "C"
