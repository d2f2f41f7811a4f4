// Streaming elementwise kernels (HBM-bound): ReLU, Clamp, Sigmoid, broadcast Add / Multiply.
// One float4 (16 B) per lane per access, grid-stride over <= 2048 workgroups of 4 waves.  No fast-math: results are the IEEE results of the
// numpy expressions they replace.
#include "pvhip_common.h"

using namespace pvhip;

namespace {

// 16-byte accesses of the streaming kernels.  NT = nontemporal: on this chip a float4 stream with `nt` loads AND stores runs at
// 6.0-6.4 TB/s where the plain forms (and hipMemcpyDtoD) stop at 5.1-5.4 (profiles/r03_stream_sweep.md, scripts/sweep_stream.py:
// same box, same process) -- each element of these kernels is touched exactly once, so there is nothing for the caches to keep.
typedef float f4v __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ldg4(const float4* p) {
    const f4v v = NT ? __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p)) : *reinterpret_cast<const f4v*>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
template <bool NT>
__device__ __forceinline__ void stg4(float4* p, const float4& a) {
    f4v v;
    v.x = a.x; v.y = a.y; v.z = a.z; v.w = a.w;
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(p));
    else *reinterpret_cast<f4v*>(p) = v;
}
// Nontemporal accesses from the size on where a tensor cannot stay in the 32 MiB of L2 for its consumer anyway (PVHIP_STREAM_NT=0
// never, =2 always); up to PVHIP_STREAM_WG (16) workgroups per CU -- the sweep's best -- instead of 8.
inline bool stream_nt(size_t bytes_moved) {
    const int m = settings().stream_nt;
    return m == 2 || (m == 1 && bytes_moved >= ((size_t)64 << 20));
}
inline int stream_grid(size_t work_items) {
    size_t b = (work_items + kBlock - 1) / kBlock;
    const size_t cap = (size_t)kNumCU * (size_t)settings().stream_wg;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

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

// A workgroup owns U adjacent 4 KiB pieces per iteration (one contiguous run of U x 256 float4): the U 16-byte loads of a lane are
// issued before its first store; whole runs first, then the float4 that are left, then the scalar tail.  profiles/r03_stream_sweep.md
// (scripts/sweep_stream.py, two boxes): with nontemporal accesses and up to 32 workgroups per CU this shape is the best or within
// 2 % of the best for 1.6-4.9 GB moved (5.8-6.3 TB/s; one load per lane and iteration, the form of rounds 1-2: 5.3-5.8, plain
// accesses 5.1-5.4 = hipMemcpyDtoD); with far-apart streams per lane (a whole grid between a lane's loads) U > 1 loses.
template <class Op, bool NT, int U>
__global__ __launch_bounds__(kBlock) void unary_f4_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           size_t n4, size_t n, Op op) {
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    float4* __restrict__       y4 = reinterpret_cast<float4*>(y);
    const size_t nruns = n4 / (size_t)(U * kBlock);
    for (size_t r = blockIdx.x; r < nruns; r += gridDim.x) {
        const size_t at = r * (size_t)(U * kBlock) + threadIdx.x;
        float4 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = ldg4<NT>(x4 + at + u * kBlock);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u].x = op(a[u].x); a[u].y = op(a[u].y); a[u].z = op(a[u].z); a[u].w = op(a[u].w);
            stg4<NT>(y4 + at + u * kBlock, a[u]);
        }
    }
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = nruns * (size_t)(U * kBlock) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = ldg4<NT>(x4 + i);
        a.x = op(a.x); a.y = op(a.y); a.z = op(a.z); a.w = op(a.w);
        stg4<NT>(y4 + i, a);
    }
    // scalar tail (n not a multiple of 4)
    const size_t t = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] = op(x[t]);
}

template <class Op>
int launch_unary(const float* x, float* y, size_t n, Op op) {
    if (n == 0) return PVHIP_OK;
    const size_t n4 = n / 4;
    hipStream_t  st = state().stream;
    if (stream_nt(n * 8)) {          // large: runs of four pieces per workgroup, nontemporal, up to 2 x PVHIP_STREAM_WG workgroups per CU
        const size_t cap = (size_t)kNumCU * 2 * (size_t)settings().stream_wg, runs = (n4 + 4 * kBlock - 1) / (4 * kBlock);
        hipLaunchKernelGGL((unary_f4_kernel<Op, true, 4>), dim3((unsigned)(runs < cap ? runs : cap)), dim3(kBlock), 0, st, x, y, n4, n, op);
    } else {
        hipLaunchKernelGGL((unary_f4_kernel<Op, false, 1>), dim3(stream_grid(n4 > 0 ? n4 : 1)), dim3(kBlock), 0, st, x, y, n4, n, op);
    }
    return PVHIP_OK;
}

struct AddOp {
    __device__ __forceinline__ float operator()(float a, float b) const { return a + b; }
};
struct MulOp {
    __device__ __forceinline__ float operator()(float a, float b) const { return a * b; }
};

// a and b have the output shape.
template <class Op, bool NT>
__global__ __launch_bounds__(kBlock) void binary_same_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              float* __restrict__ out, size_t n4, size_t n, Op op) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t       i      = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float4* __restrict__ a4 = reinterpret_cast<const float4*>(a);
    const float4* __restrict__ b4 = reinterpret_cast<const float4*>(b);
    float4* __restrict__       o4 = reinterpret_cast<float4*>(out);
    for (; i + stride < n4; i += 2 * stride) {
        float4 p = ldg4<NT>(a4 + i), q = ldg4<NT>(b4 + i), r = ldg4<NT>(a4 + i + stride), s = ldg4<NT>(b4 + i + stride);
        p.x = op(p.x, q.x); p.y = op(p.y, q.y); p.z = op(p.z, q.z); p.w = op(p.w, q.w);
        r.x = op(r.x, s.x); r.y = op(r.y, s.y); r.z = op(r.z, s.z); r.w = op(r.w, s.w);
        stg4<NT>(o4 + i, p); stg4<NT>(o4 + i + stride, r);
    }
    for (; i < n4; i += stride) {
        float4 p = ldg4<NT>(a4 + i), q = ldg4<NT>(b4 + i);
        p.x = op(p.x, q.x); p.y = op(p.y, q.y); p.z = op(p.z, q.z); p.w = op(p.w, q.w);
        stg4<NT>(o4 + i, p);
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
template <class Op, bool kSwap, bool NT, int U>
__global__ __launch_bounds__(kBlock) void binary_channel4_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                  float* __restrict__ out, unsigned n4, unsigned C, FastDiv d_inner4,
                                                                  FastDiv d_c, Op op) {
    const float4* __restrict__ a4 = reinterpret_cast<const float4*>(a);
    float4* __restrict__       o4 = reinterpret_cast<float4*>(out);
    auto one = [&](unsigned i, float4 v) {
        const unsigned q  = fdiv(i, d_inner4);               // i / (inner / 4)
        const float    bv = b[q - fdiv(q, d_c) * C];         // q % C
        if (kSwap) {
            v.x = op(bv, v.x); v.y = op(bv, v.y); v.z = op(bv, v.z); v.w = op(bv, v.w);
        } else {
            v.x = op(v.x, bv); v.y = op(v.y, bv); v.z = op(v.z, bv); v.w = op(v.w, bv);
        }
        stg4<NT>(o4 + i, v);
    };
    // whole runs of U adjacent 4 KiB pieces per workgroup (the U loads of a lane before its first store: see unary_f4_kernel), then the rest
    const unsigned nruns = n4 / (unsigned)(U * kBlock);
    for (unsigned r = blockIdx.x; r < nruns; r += gridDim.x) {
        const unsigned at = r * (unsigned)(U * kBlock) + threadIdx.x;
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ldg4<NT>(a4 + at + u * kBlock);
#pragma unroll
        for (int u = 0; u < U; ++u) one(at + (unsigned)(u * kBlock), v[u]);
    }
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned i = nruns * (unsigned)(U * kBlock) + blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) one(i, ldg4<NT>(a4 + i));
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
        const int g = stream_grid(n4 > 0 ? (n4 + 1) / 2 : 1);
        if (stream_nt(n * 12)) hipLaunchKernelGGL((binary_same_kernel<Op, true>), dim3(g), dim3(kBlock), 0, state().stream, a, b, out, n4, n, op);
        else                   hipLaunchKernelGGL((binary_same_kernel<Op, false>), dim3(g), dim3(kBlock), 0, state().stream, a, b, out, n4, n, op);
        return PVHIP_OK;
    }
    int64_t C = 0, inner = 0;
    if (n < (1ull << 32)) {
        if (a_full && channel_pattern(rank, shape, bs, &C, &inner)) {
            const int g = grid_for(n4 > 0 ? n4 : 1);
            if (inner % 4 == 0 && n % 4 == 0 && n < (1ull << 31)) {
                const bool     nt   = stream_nt(n * 8);
                const unsigned runs = (unsigned)((n4 + 4 * kBlock - 1) / (4 * kBlock)), cap = (unsigned)(kNumCU * 2 * settings().stream_wg);
                const int      gs   = nt ? (int)(runs < cap ? runs : cap) : stream_grid(n4);
                if (nt)
                    hipLaunchKernelGGL((binary_channel4_kernel<Op, false, true, 4>), dim3(gs), dim3(kBlock), 0, state().stream, a, b, out, (unsigned)n4,
                                       (unsigned)C, make_fastdiv((unsigned)(inner / 4)), make_fastdiv((unsigned)C), op);
                else
                    hipLaunchKernelGGL((binary_channel4_kernel<Op, false, false, 1>), dim3(gs), dim3(kBlock), 0, state().stream, a, b, out, (unsigned)n4,
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
                const bool     nt   = stream_nt(n * 8);
                const unsigned runs = (unsigned)((n4 + 4 * kBlock - 1) / (4 * kBlock)), cap = (unsigned)(kNumCU * 2 * settings().stream_wg);
                const int      gs   = nt ? (int)(runs < cap ? runs : cap) : stream_grid(n4);
#define PV_CH4(SWAP_, NT_, U_)                                                                                                  \
    hipLaunchKernelGGL((binary_channel4_kernel<Op, SWAP_, NT_, U_>), dim3(gs), dim3(kBlock), 0, state().stream, b, a, out, (unsigned)n4, \
                       (unsigned)C, make_fastdiv((unsigned)(inner / 4)), make_fastdiv((unsigned)C), op)
                if (commutative) { if (nt) PV_CH4(false, true, 4); else PV_CH4(false, false, 1); }
                else             { if (nt) PV_CH4(true, true, 4); else PV_CH4(true, false, 1); }
#undef PV_CH4
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
