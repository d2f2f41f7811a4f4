// Streaming elementwise kernels (HBM-bound): ReLU, Clamp, Sigmoid, broadcast Add / Multiply.
// One float4 (16 B) per lane per access, grid-stride over <= 2048 workgroups of 4 waves.  No fast-math: results are the IEEE results of the
// numpy expressions they replace.
#include "pvhip_common.h"

using namespace pvhip;

namespace {

struct ReluOp {
    __device__ __forceinline__ float operator()(float v) const { return (v < 0.0f) ? 0.0f : v; }
};
struct ClampOp {
    float lo, hi;
    // np.clip == minimum(maximum(x, lo), hi): NaN propagates
    __device__ __forceinline__ float operator()(float v) const {
        float t = (v < lo) ? lo : v;
        return (t > hi) ? hi : t;
    }
};
struct SigmoidOp {
    __device__ __forceinline__ float operator()(float v) const { return 1.0f / (1.0f + expf(-v)); }
};

// One 16-byte load + store per lane per iteration, consecutive lanes on consecutive 16-byte words, grid-stride
// over <= 2048 workgroups.  (A 4-way unrolled variant with four far-apart streams per lane measured 4.4-4.8 TB/s
// against 5.2-5.5 TB/s for this form on 0.4-3.3 GB tensors; hipMemcpyDtoD reaches 5.0-5.4 TB/s on the same box.)
template <class Op>
__global__ __launch_bounds__(kBlock) void unary_f4_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           size_t n4, size_t n, Op op) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    float4* __restrict__       y4 = reinterpret_cast<float4*>(y);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = x4[i];
        a.x = op(a.x); a.y = op(a.y); a.z = op(a.z); a.w = op(a.w);
        y4[i] = a;
    }
    // scalar tail (n not a multiple of 4)
    const size_t t = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] = op(x[t]);
}

template <class Op>
int launch_unary(const float* x, float* y, size_t n, Op op) {
    if (n == 0) return PVHIP_OK;
    const size_t n4 = n / 4;
    const int    g  = grid_for(n4 > 0 ? n4 : 1);
    hipLaunchKernelGGL(unary_f4_kernel<Op>, dim3(g), dim3(kBlock), 0, state().stream, x, y, n4, n, op);
    return PVHIP_OK;
}

struct AddOp {
    __device__ __forceinline__ float operator()(float a, float b) const { return a + b; }
};
struct MulOp {
    __device__ __forceinline__ float operator()(float a, float b) const { return a * b; }
};

// a and b have the output shape.
template <class Op>
__global__ __launch_bounds__(kBlock) void binary_same_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              float* __restrict__ out, size_t n4, size_t n, Op op) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t       i      = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float4* __restrict__ a4 = reinterpret_cast<const float4*>(a);
    const float4* __restrict__ b4 = reinterpret_cast<const float4*>(b);
    float4* __restrict__       o4 = reinterpret_cast<float4*>(out);
    for (; i + stride < n4; i += 2 * stride) {
        float4 p = a4[i], q = b4[i], r = a4[i + stride], s = b4[i + stride];
        p.x = op(p.x, q.x); p.y = op(p.y, q.y); p.z = op(p.z, q.z); p.w = op(p.w, q.w);
        r.x = op(r.x, s.x); r.y = op(r.y, s.y); r.z = op(r.z, s.z); r.w = op(r.w, s.w);
        o4[i] = p; o4[i + stride] = r;
    }
    for (; i < n4; i += stride) {
        float4 p = a4[i], q = b4[i];
        p.x = op(p.x, q.x); p.y = op(p.y, q.y); p.z = op(p.z, q.z); p.w = op(p.w, q.w);
        o4[i] = p;
    }
    const size_t t = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = op(a[t], b[t]);
}

// out viewed as [outer][C][inner]; b holds C values (per-channel bias / scale / FC bias row).
// C == 1 is the scalar broadcast.  Indices fit 32 bits (checked on the host).
template <class Op, bool kSwap>
__global__ __launch_bounds__(kBlock) void binary_channel_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                 float* __restrict__ out, unsigned n4, unsigned n,
                                                                 unsigned C, unsigned inner, Op op) {
    const unsigned stride = gridDim.x * blockDim.x;
    const float4* __restrict__ a4 = reinterpret_cast<const float4*>(a);
    float4* __restrict__       o4 = reinterpret_cast<float4*>(out);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4         v    = a4[i];
        const unsigned e    = i * 4u;
        const unsigned q    = e / inner;
        unsigned       r    = e - q * inner;
        unsigned       c    = q % C;
        float          bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bv[j] = b[c];
            if (++r >= inner) { r = 0; if (++c >= C) c = 0; }
        }
        if (kSwap) {
            v.x = op(bv[0], v.x); v.y = op(bv[1], v.y); v.z = op(bv[2], v.z); v.w = op(bv[3], v.w);
        } else {
            v.x = op(v.x, bv[0]); v.y = op(v.y, bv[1]); v.z = op(v.z, bv[2]); v.w = op(v.w, bv[3]);
        }
        o4[i] = v;
    }
    const unsigned t = n4 * 4u + blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        const float bvv = b[(t / inner) % C];
        out[t]          = kSwap ? op(bvv, a[t]) : op(a[t], bvv);
    }
}

// The same with inner % 4 == 0 (a float4 never straddles two channels: data/mean on (N,3,224,224), per-channel bias rows): ONE
// channel index per float4, by multiply-high instead of two runtime divisions and a four-step carry chain (that form ran the
// 308 MB data/mean Add of GoogLeNet at 2.1 TB/s, vector-ALU-bound).  n < 2^31.
template <class Op, bool kSwap>
__global__ __launch_bounds__(kBlock) void binary_channel4_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                  float* __restrict__ out, unsigned n4, unsigned C, FastDiv d_inner4,
                                                                  FastDiv d_c, Op op) {
    const unsigned stride = gridDim.x * blockDim.x;
    const float4* __restrict__ a4 = reinterpret_cast<const float4*>(a);
    float4* __restrict__       o4 = reinterpret_cast<float4*>(out);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4         v  = a4[i];
        const unsigned q  = fdiv(i, d_inner4);               // i / (inner / 4)
        const float    bv = b[q - fdiv(q, d_c) * C];         // q % C
        if (kSwap) {
            v.x = op(bv, v.x); v.y = op(bv, v.y); v.z = op(bv, v.z); v.w = op(bv, v.w);
        } else {
            v.x = op(v.x, bv); v.y = op(v.y, bv); v.z = op(v.z, bv); v.w = op(v.w, bv);
        }
        o4[i] = v;
    }
}

struct StridedArgs {
    int     rank;
    int64_t shape[PVHIP_MAX_RANK];
    int64_t as[PVHIP_MAX_RANK];
    int64_t bs[PVHIP_MAX_RANK];
};

template <class Op>
__global__ __launch_bounds__(kBlock) void binary_strided_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                 float* __restrict__ out, size_t n, StridedArgs s, Op op) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        size_t  rem = i;
        int64_t ia = 0, ib = 0;
        for (int d = s.rank - 1; d >= 0; --d) {
            const size_t dim = (size_t)s.shape[d];
            const size_t idx = rem % dim;
            rem /= dim;
            ia += (int64_t)idx * s.as[d];
            ib += (int64_t)idx * s.bs[d];
        }
        out[i] = op(a[ia], b[ib]);
    }
}

// Is `st` the contiguous stride set of `shape`?  (axes of extent 1 are free)
bool is_contiguous(int rank, const int64_t* shape, const int64_t* st) {
    int64_t expect = 1;
    for (int d = rank - 1; d >= 0; --d) {
        if (shape[d] != 1 && st[d] != expect) return false;
        expect *= shape[d];
    }
    return true;
}

// Does `st` describe "a contiguous vector of C values laid along a run of adjacent axes, broadcast
// everywhere else"?  On success returns C and inner (product of the axes after the run).
bool channel_pattern(int rank, const int64_t* shape, const int64_t* st, int64_t* C, int64_t* inner) {
    int lo = -1, hi = -1;
    for (int d = 0; d < rank; ++d) {
        if (shape[d] != 1 && st[d] != 0) {
            if (lo < 0) lo = d;
            hi = d;
        }
    }
    if (lo < 0) {  // pure scalar broadcast
        *C = 1;
        *inner = 1;
        return true;
    }
    int64_t expect = 1;
    for (int d = hi; d >= lo; --d) {
        if (shape[d] == 1) continue;
        if (st[d] != expect) return false;
        expect *= shape[d];
    }
    *C        = expect;
    int64_t in = 1;
    for (int d = hi + 1; d < rank; ++d) in *= shape[d];
    *inner = in;
    return true;
}

template <class Op>
int launch_binary(const char* who, const float* a, const float* b, float* out, int rank, const int64_t* shape,
                  const int64_t* as, const int64_t* bs, bool commutative) {
    if (rank < 0 || rank > PVHIP_MAX_RANK) return fail(PVHIP_EINVAL, "%s: rank %d not in [0,%d]", who, rank, PVHIP_MAX_RANK);
    size_t n = 1;
    for (int d = 0; d < rank; ++d) {
        if (shape[d] < 0) return fail(PVHIP_EINVAL, "%s: negative extent", who);
        n *= (size_t)shape[d];
    }
    if (n == 0) return PVHIP_OK;
    const size_t n4     = n / 4;
    const bool   a_full = is_contiguous(rank, shape, as);
    const bool   b_full = is_contiguous(rank, shape, bs);
    Op           op;
    if (a_full && b_full) {
        const int g = grid_for(n4 > 0 ? (n4 + 1) / 2 : 1);
        hipLaunchKernelGGL(binary_same_kernel<Op>, dim3(g), dim3(kBlock), 0, state().stream, a, b, out, n4, n, op);
        return PVHIP_OK;
    }
    int64_t C = 0, inner = 0;
    if (n < (1ull << 32)) {
        if (a_full && channel_pattern(rank, shape, bs, &C, &inner)) {
            const int g = grid_for(n4 > 0 ? n4 : 1);
            if (inner % 4 == 0 && n % 4 == 0 && n < (1ull << 31)) {
                hipLaunchKernelGGL((binary_channel4_kernel<Op, false>), dim3(g), dim3(kBlock), 0, state().stream, a, b, out, (unsigned)n4,
                                   (unsigned)C, make_fastdiv((unsigned)(inner / 4)), make_fastdiv((unsigned)C), op);
                return PVHIP_OK;
            }
            hipLaunchKernelGGL((binary_channel_kernel<Op, false>), dim3(g), dim3(kBlock), 0, state().stream, a, b, out,
                               (unsigned)n4, (unsigned)n, (unsigned)C, (unsigned)inner, op);
            return PVHIP_OK;
        }
        if (b_full && channel_pattern(rank, shape, as, &C, &inner)) {
            // the broadcast operand is `a`: stream b, keep operand order for non-commutative ops
            const int g = grid_for(n4 > 0 ? n4 : 1);
            if (inner % 4 == 0 && n % 4 == 0 && n < (1ull << 31)) {
                if (commutative)
                    hipLaunchKernelGGL((binary_channel4_kernel<Op, false>), dim3(g), dim3(kBlock), 0, state().stream, b, a, out,
                                       (unsigned)n4, (unsigned)C, make_fastdiv((unsigned)(inner / 4)), make_fastdiv((unsigned)C), op);
                else
                    hipLaunchKernelGGL((binary_channel4_kernel<Op, true>), dim3(g), dim3(kBlock), 0, state().stream, b, a, out,
                                       (unsigned)n4, (unsigned)C, make_fastdiv((unsigned)(inner / 4)), make_fastdiv((unsigned)C), op);
                return PVHIP_OK;
            }
            if (commutative)
                hipLaunchKernelGGL((binary_channel_kernel<Op, false>), dim3(g), dim3(kBlock), 0, state().stream, b, a, out,
                                   (unsigned)n4, (unsigned)n, (unsigned)C, (unsigned)inner, op);
            else
                hipLaunchKernelGGL((binary_channel_kernel<Op, true>), dim3(g), dim3(kBlock), 0, state().stream, b, a, out,
                                   (unsigned)n4, (unsigned)n, (unsigned)C, (unsigned)inner, op);
            return PVHIP_OK;
        }
    }
    StridedArgs s;
    s.rank = rank;
    for (int d = 0; d < PVHIP_MAX_RANK; ++d) {
        s.shape[d] = d < rank ? shape[d] : 1;
        s.as[d]    = d < rank ? as[d] : 0;
        s.bs[d]    = d < rank ? bs[d] : 0;
    }
    hipLaunchKernelGGL(binary_strided_kernel<Op>, dim3(grid_for(n)), dim3(kBlock), 0, state().stream, a, b, out, n, s, op);
    return PVHIP_OK;
}

}  // namespace

extern "C" {

int pvhip_relu_f32(const float* x, float* y, size_t n) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n == 0 || (x != nullptr && y != nullptr));
    int rc = launch_unary(x, y, n, ReluOp{});
    if (rc) return rc;
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_clamp_f32(const float* x, float* y, size_t n, float lo, float hi) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n == 0 || (x != nullptr && y != nullptr));
    int rc = launch_unary(x, y, n, ClampOp{lo, hi});
    if (rc) return rc;
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_sigmoid_f32(const float* x, float* y, size_t n) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(n == 0 || (x != nullptr && y != nullptr));
    int rc = launch_unary(x, y, n, SigmoidOp{});
    if (rc) return rc;
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_add_f32(const float* a, const float* b, float* out, int rank, const int64_t* shape,
                  const int64_t* a_strides, const int64_t* b_strides) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(a && b && out && (rank == 0 || (shape && a_strides && b_strides)));
    int rc = launch_binary<AddOp>("pvhip_add_f32", a, b, out, rank, shape, a_strides, b_strides, true);
    if (rc) return rc;
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

int pvhip_mul_f32(const float* a, const float* b, float* out, int rank, const int64_t* shape,
                  const int64_t* a_strides, const int64_t* b_strides) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(a && b && out && (rank == 0 || (shape && a_strides && b_strides)));
    int rc = launch_binary<MulOp>("pvhip_mul_f32", a, b, out, rank, shape, a_strides, b_strides, true);
    if (rc) return rc;
    PVHIP_LAUNCH_CHECK();
    return PVHIP_OK;
}

}  // extern "C"
