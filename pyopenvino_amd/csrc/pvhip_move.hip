// Data-movement kernels: Concat (strided device copy) and Transpose (materialised permutation).
#include "pvhip_common.h"

using namespace pvhip;

namespace {

struct ConcatArgs {
    int          n_src;
    const float* src[PVHIP_MAX_CONCAT];
    unsigned     inner[PVHIP_MAX_CONCAT];   // elements per outer step of source i (in VEC units)
    unsigned     offset[PVHIP_MAX_CONCAT];  // running offset inside one dst outer step (in VEC units)
    unsigned     total_inner;               // sum(inner) (in VEC units)
    unsigned     outer;
};

// grid.y = source index; each source is a dense [outer][inner_i] block copied to dst[outer][off_i ...].
template <typename T>
__global__ __launch_bounds__(kBlock) void concat_kernel(ConcatArgs a, T* __restrict__ dst) {
    const int      s      = blockIdx.y;
    const unsigned inner  = a.inner[s];
    const unsigned n      = inner * a.outer;
    const T* __restrict__ src = reinterpret_cast<const T*>(a.src[s]);
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const unsigned o = e / inner;
        const unsigned r = e - o * inner;
        dst[(size_t)o * a.total_inner + a.offset[s] + r] = src[e];
    }
}

// Zero padding as a pass of its own (Convolution.py:64-66 builds the padded image the same way): y[plane][pt + iy][pl + ix] =
// x[plane][iy][ix] (+ add[plane % c]), zeros around.  grid.y = plane; a thread walks the plane's OUTPUT elements, so the stores are
// dense; two divisions by launch constants per element (multiply-high).
__global__ __launch_bounds__(kBlock) void pad2d_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ add,
                                                       int c, int h, int w, int pt, int pl, unsigned hp, unsigned wp, FastDiv d_wp) {
    const unsigned plane = blockIdx.y;
    const float* __restrict__ xp = x + (size_t)plane * h * w;
    float* __restrict__       yp = y + (size_t)plane * hp * wp;
    const float b = add != nullptr ? add[plane % (unsigned)c] : -0.0f;          // v + -0.0 == v for every v
    const unsigned total = hp * wp, stride = gridDim.x * blockDim.x;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const unsigned oy = fdiv(e, d_wp), ox = e - oy * wp;
        const int      iy = (int)oy - pt, ix = (int)ox - pl;
        float v = 0.0f;
        if ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) v = xp[(size_t)iy * w + ix] + b;
        __builtin_nontemporal_store(v, yp + e);
    }
}

struct PermArgs {
    int      rank;
    unsigned out_shape[PVHIP_MAX_RANK];
    unsigned in_stride_for_out_axis[PVHIP_MAX_RANK];
};

__global__ __launch_bounds__(kBlock) void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, PermArgs p,
                                                            unsigned total) {
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        unsigned rem = e, src = 0;
        for (int d = p.rank - 1; d >= 0; --d) {
            const unsigned idx = rem % p.out_shape[d];
            rem /= p.out_shape[d];
            src += idx * p.in_stride_for_out_axis[d];
        }
        y[e] = x[src];
    }
}

}  // namespace

extern "C" {

int pvhip_concat_f32(int n_src, const float* const* srcs, const int64_t* inner, float* dst, int64_t outer) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n_src >= 1 && n_src <= PVHIP_MAX_CONCAT && srcs != nullptr && inner != nullptr && outer >= 0);
    int64_t total = 0, max_inner = 0;
    bool    vec4  = true;
    for (int i = 0; i < n_src; ++i) {
        PVHIP_CHECK_ARG(inner[i] >= 0);
        if (inner[i] % 4 != 0) vec4 = false;
        total += inner[i];
        if (inner[i] > max_inner) max_inner = inner[i];
    }
    if (total == 0 || outer == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(dst != nullptr);
    if ((unsigned long long)total * (unsigned long long)outer >= (1ull << 31))
        return fail(PVHIP_EUNSUPPORTED, "pvhip_concat_f32: tensor exceeds 2^31 elements");
    const unsigned div = vec4 ? 4u : 1u;
    ConcatArgs     a;
    a.n_src        = n_src;
    unsigned off   = 0;
    for (int i = 0; i < n_src; ++i) {
        PVHIP_CHECK_ARG(inner[i] == 0 || srcs[i] != nullptr);
        a.src[i]    = srcs[i];
        a.inner[i]  = (unsigned)(inner[i] / div);
        a.offset[i] = off;
        off += a.inner[i];
    }
    for (int i = n_src; i < PVHIP_MAX_CONCAT; ++i) {
        a.src[i] = nullptr;
        a.inner[i] = a.offset[i] = 0;
    }
    a.total_inner = off;
    a.outer       = (unsigned)outer;
    const size_t work = (size_t)(max_inner / div) * (size_t)outer;
    const int    gx   = grid_for(work);
    if (vec4)
        hipLaunchKernelGGL(concat_kernel<float4>, dim3(gx, n_src), dim3(kBlock), 0, state().stream, a,
                           reinterpret_cast<float4*>(dst));
    else
        hipLaunchKernelGGL(concat_kernel<float>, dim3(gx, n_src), dim3(kBlock), 0, state().stream, a, dst);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_pad2d_f32(const float* x, float* y, int n, int c, int h, int w, int pad_top, int pad_left, int pad_bottom, int pad_right,
                    const float* channel_add) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n >= 0 && c > 0 && h > 0 && w > 0 && pad_top >= 0 && pad_left >= 0 && pad_bottom >= 0 && pad_right >= 0);
    if (n == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr);
    const unsigned long long hp = (unsigned long long)h + pad_top + pad_bottom, wp = (unsigned long long)w + pad_left + pad_right;
    if (hp * wp >= (1ull << 31) || (unsigned long long)n * c > 65535ull)
        return fail(PVHIP_EUNSUPPORTED, "pvhip_pad2d_f32: plane of %llu x %llu elements or %llu planes: too large", hp, wp,
                    (unsigned long long)n * c);
    const int gx = (int)((hp * wp + (unsigned long long)kBlock * 4 - 1) / ((unsigned long long)kBlock * 4));      // four elements per thread
    hipLaunchKernelGGL(pad2d_kernel, dim3(gx > 0 ? gx : 1, n * c), dim3(kBlock), 0, state().stream, x, y, channel_add, c, h, w, pad_top,
                       pad_left, (unsigned)hp, (unsigned)wp, make_fastdiv((unsigned)wp));
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_transpose_f32(const float* x, float* y, int rank, const int64_t* in_shape, const int64_t* perm) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(rank >= 1 && rank <= PVHIP_MAX_RANK && in_shape != nullptr && perm != nullptr);
    int64_t in_stride[PVHIP_MAX_RANK];
    int64_t acc = 1;
    for (int d = rank - 1; d >= 0; --d) {
        PVHIP_CHECK_ARG(in_shape[d] >= 0);
        in_stride[d] = acc;
        acc *= in_shape[d];
    }
    if (acc == 0) return PVHIP_OK;
    if (acc >= (1ll << 31)) return fail(PVHIP_EUNSUPPORTED, "pvhip_transpose_f32: tensor exceeds 2^31 elements");
    PVHIP_CHECK_ARG(x != nullptr && y != nullptr);
    bool     seen[PVHIP_MAX_RANK] = {false, false, false, false, false, false};
    PermArgs p;
    p.rank = rank;
    for (int d = 0; d < rank; ++d) {
        const int64_t ax = perm[d];
        PVHIP_CHECK_ARG(ax >= 0 && ax < rank && !seen[ax]);
        seen[ax]                    = true;
        p.out_shape[d]              = (unsigned)in_shape[ax];
        p.in_stride_for_out_axis[d] = (unsigned)in_stride[ax];
    }
    for (int d = rank; d < PVHIP_MAX_RANK; ++d) {
        p.out_shape[d] = 1;
        p.in_stride_for_out_axis[d] = 0;
    }
    hipLaunchKernelGGL(transpose_kernel, dim3(grid_for((size_t)acc)), dim3(kBlock), 0, state().stream, x, y, p,
                       (unsigned)acc);
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // extern "C"
