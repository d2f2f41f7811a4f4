// Pooling kernels (HBM-bound).  NCHW fp32; lanes run along the output row so the window loads of a
// wave are contiguous (stride sw) segments of input rows and the stores are fully coalesced.
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
    hipLaunchKernelGGL(maxpool2d_kernel, dim3(grid_for(total)), dim3(kBlock), 0, state().stream, x, y, a, total);
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
    hipLaunchKernelGGL(avgpool2d_kernel, dim3(grid_for(total)), dim3(kBlock), 0, state().stream, x, y, a, total);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // extern "C"
