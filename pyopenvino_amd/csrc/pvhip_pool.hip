// Pooling kernels (HBM-bound).  NCHW fp32; lanes run along the output row so the window loads of a
// wave are contiguous (stride sw) segments of input rows and the stores are fully coalesced.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

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
        float m = -INFINITY;             // max3_nan: a NaN in the window IS its maximum
        if (KH != 0) {
#pragma unroll
            for (int ky = 0; ky < (KH ? KH : 1); ++ky) {
                const bool row_ok = !CLIP || (py0 + ky < a.hp);
#pragma unroll
                for (int kx = 0; kx < (KW ? KW : 1); ++kx) {
                    if (row_ok && (!CLIP || (px0 + kx < a.wp))) {
                        const float v = tp[ky * wl + kx];
                        m = max3_nan(m, v, v);
                    }
                }
            }
        } else {
            for (int ky = 0; ky < kh && py0 + ky < a.hp; ++ky)
                for (int kx = 0; kx < kw && px0 + kx < a.wp; ++kx) {
                    const float v = tp[ky * wl + kx];
                    m = max3_nan(m, v, v);
                }
        }
        yout[o] = m;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 MaxPool, stride 1 or 2, as a persistent double-buffered pipeline (the GoogLeNet pools: 3x3/s2 ceil and
// 3x3/s1/p1).  Same semantics as maxpool2d_lds_kernel (MaxPool.py:41-72: zero pad cells take part in the max,
// windows are clipped at the padded extent, NaN wins).
//
// A tile is G planes x a band of output rows.  Its input (whole planes: one dense run; bands: one dense run per
// plane) goes global -> LDS by LDS-DMA (16 bytes per lane, no staging registers), and the DMA of tile t+1 is in
// flight while tile t is computed and stored.  The LDS image is the DENSE input (no pad cells): a lane owns one
// output COLUMN of a segment of rows and slides down it, so every input row costs it 3 LDS reads and one
// horizontal max, and an output is the max of three of those row values (3 instead of 9 LDS reads per output at
// stride 1, 6 at stride 2).  Taps outside the tensor are clamped onto a tap of the same window that is inside (a
// duplicate changes no max); a window that touches pad cells inside the padded extent gets max(m, 0) -- the value
// those cells hold in the reference.  The outputs of a tile are collected in LDS and leave as dense 16-byte runs.
struct Pool3Args {
    const float* x;
    float*       y;
    unsigned long long x_bytes;
    int n_planes, h, w, oh, ow;
    int pt, pl, hp, wp;
    int G, S, band_rows, n_bands;   // tile = G planes x band_rows output rows, split over S row segments per column
    int dense;                      // 1: bands == 1, a tile's input is ONE dense run of G*h*w floats
    int plane_l;                    // floats between plane images in LDS (dense: h*w; bands: a multiple of 256)
    int in_floats;                  // floats per input buffer (whole 1-KiB pieces)
    int out_plane_l;                // floats between plane images of the output stage
    int n_tiles;
    int vec_out;                    // 16-byte stores are aligned for every tile
};

typedef __attribute__((address_space(3))) void* pool_lds_ptr_t;

// NT: nontemporal -- the input of a pooling launch is read once (halo rows twice) and its output written once; on this chip a 16-byte
// stream with nt loads and stores runs at 6.0-6.4 TB/s against 5.1-5.4 plain (profiles/r03_stream_sweep.md).
template <bool NT>
__device__ __forceinline__ void pool_dma_b128(__amdgpu_buffer_rsrc_t r, float* dst, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lds = (unsigned)(unsigned long)(pool_lds_ptr_t)dst;
    if (NT)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds"
                     :: "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
#endif
}
typedef float pool_f4v __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void pool_st4(float* dst, const float* src_lds) {
    const pool_f4v v = *reinterpret_cast<const pool_f4v*>(src_lds);
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<pool_f4v*>(dst));
    else *reinterpret_cast<pool_f4v*>(dst) = v;
}
template <bool NT>
__device__ __forceinline__ void pool_st1(float* dst, float v) {
    if (NT) __builtin_nontemporal_store(v, dst);
    else *dst = v;
}
__device__ __forceinline__ void pool_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct Pool3Divs {
    FastDiv bands, sow, ow;
};

template <int ST, bool STAGE, bool NT>
__global__ __launch_bounds__(kBlock) void maxpool3x3_cols_kernel(Pool3Args a, Pool3Divs dv) {
    extern __shared__ __attribute__((aligned(1024))) float lds3[];
    float* const outb = lds3 + 2 * a.in_floats;
    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)(tid & 63) * 16u;
    const int hw = a.h * a.w, ohw = a.oh * a.ow;

    // input of tile t -> buffer `dst`
    auto issue = [&](int t, float* dst) {
        const int pg = (int)fdiv((unsigned)t, dv.bands), b = t - pg * a.n_bands;
        const int g0 = pg * a.G, gn = min(a.G, a.n_planes - g0);
        const int oy0 = b * a.band_rows, oy1 = min(a.oh, oy0 + a.band_rows);
        const int iy_lo = max(0, oy0 * ST - a.pt), iy_hi = min(a.h, (oy1 - 1) * ST + 3 - a.pt);
        const size_t src = (size_t)g0 * hw + (size_t)iy_lo * a.w;            // floats
        const unsigned long long left = a.x_bytes - (unsigned long long)src * 4ull;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x + src), 0, (int)(left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)left), 0x00020000);
        if (a.dense) {
            const int pieces = (gn * hw + 255) >> 8;
            // (the whole offset rides in the VECTOR offset: the scalar one is not range-checked, and the last pieces of the last tile reach
            // past the tensor -- they must read zeros, not the bytes behind the allocation)
            for (int q = wave; q < pieces; q += 4) pool_dma_b128<NT>(r, dst + q * 256, lane16 + (unsigned)q * 1024u, 0u);
        } else {
            const int ppp = ((iy_hi - iy_lo) * a.w + 255) >> 8;               // pieces per plane
            for (int p = 0; p < gn; ++p)
                for (int q = wave; q < ppp; q += 4)
                    pool_dma_b128<NT>(r, dst + p * a.plane_l + q * 256, lane16 + (unsigned)(p * hw) * 4u + (unsigned)q * 1024u, 0u);
        }
    };

    int t = blockIdx.x, cur = 0;
    if (t < a.n_tiles) issue(t, lds3);
    for (; t < a.n_tiles; t += gridDim.x) {
        pool_dma_wait();
        __syncthreads();          // tile t has landed; every wave is done with the other buffer and with the output stage
        {
            const int tn = t + (int)gridDim.x;
            if (tn < a.n_tiles) issue(tn, lds3 + (cur ^ 1) * a.in_floats);
        }
        const float* const in = lds3 + cur * a.in_floats;
        const int pg = (int)fdiv((unsigned)t, dv.bands), b = t - pg * a.n_bands;
        const int g0 = pg * a.G, gn = min(a.G, a.n_planes - g0);
        const int oy0 = b * a.band_rows, oy1 = min(a.oh, oy0 + a.band_rows);
        const int iy_lo = max(0, oy0 * ST - a.pt);
        const int rows_t = oy1 - oy0;
        const int seg_rows = (rows_t + a.S - 1) / a.S;
        const int n_items = gn * a.S * a.ow;
        const int sow = a.S * a.ow;
        for (int it = tid; it < n_items; it += kBlock) {
            const unsigned p = fdiv((unsigned)it, dv.sow), rem = (unsigned)it - p * (unsigned)sow;
            const unsigned seg = fdiv(rem, dv.ow), ox = rem - seg * (unsigned)a.ow;
            const int ys = oy0 + (int)seg * seg_rows, ye = min(oy1, ys + seg_rows);
            if (ys >= oy1) continue;                                          // trailing segment of a short band
            const int px0 = (int)ox * ST - a.pl;
            const int c0 = min(max(px0, 0), a.w - 1), c1 = min(max(px0 + 1, 0), a.w - 1), c2 = min(max(px0 + 2, 0), a.w - 1);
            const bool  zc    = (px0 < 0) || (min((int)ox * ST + 2, a.wp - 1) - a.pl >= a.w);
            const float flr_c = zc ? 0.0f : -INFINITY;
            const float* const L = in + (int)p * a.plane_l - iy_lo * a.w;      // input row iy of this plane at L + iy*w
#define PV_HROW(IY, HV)                                                                       \
    do {                                                                                      \
        const int    r_ = min(max((IY), 0), a.h - 1);                                         \
        const float* q_ = L + r_ * a.w;                                                       \
        const float  v0_ = q_[c0], v1_ = q_[c1], v2_ = q_[c2];                                \
        HV = max3_nan(v0_, v1_, v2_);           /* a NaN in the row IS its maximum, as for np.max */ \
    } while (0)
            float hA, hB, hC;
            PV_HROW(ys * ST - a.pt, hA);
            if (ST == 1) PV_HROW(ys - a.pt + 1, hB);
            float* const yo = STAGE ? outb + (int)p * a.out_plane_l + (ys - oy0) * a.ow + (int)ox
                                    : a.y + (size_t)(g0 + (int)p) * ohw + (size_t)ys * a.ow + ox;
            for (int oy = ys; oy < ye; ++oy) {
                if (ST == 2) PV_HROW(oy * 2 - a.pt + 1, hB);
                PV_HROW(oy * ST - a.pt + 2, hC);
                const float m  = max3_nan(hA, hB, hC);
                const bool  zr = (oy * ST < a.pt) || (min(oy * ST + 2, a.hp - 1) - a.pt >= a.h);
                const float fl = zr ? 0.0f : flr_c;
                yo[(oy - ys) * a.ow] = max3_nan(m, fl, fl);
                if (ST == 1) { hA = hB; hB = hC; }
                else         { hA = hC; }
            }
#undef PV_HROW
        }
        if (STAGE) {
            __syncthreads();
            if (a.dense) {
                const int n_out = gn * ohw;
                float* const yd = a.y + (size_t)g0 * ohw;
                if (a.vec_out) {
                    const int n4 = n_out >> 2;
                    for (int i = tid; i < n4; i += kBlock) pool_st4<NT>(yd + 4 * i, outb + 4 * i);
                    for (int i = (n4 << 2) + tid; i < n_out; i += kBlock) pool_st1<NT>(yd + i, outb[i]);
                } else {
                    for (int i = tid; i < n_out; i += kBlock) pool_st1<NT>(yd + i, outb[i]);
                }
            } else {
                const int run = rows_t * a.ow;
                for (int p = 0; p < gn; ++p) {
                    float* const       yd = a.y + (size_t)(g0 + p) * ohw + (size_t)oy0 * a.ow;
                    const float* const so = outb + p * a.out_plane_l;
                    if (a.vec_out) {
                        const int n4 = run >> 2;
                        for (int i = tid; i < n4; i += kBlock) pool_st4<NT>(yd + 4 * i, so + 4 * i);
                        for (int i = (n4 << 2) + tid; i < run; i += kBlock) pool_st1<NT>(yd + i, so[i]);
                    } else {
                        for (int i = tid; i < run; i += kBlock) pool_st1<NT>(yd + i, so[i]);
                    }
                }
            }
        }
        cur ^= 1;
    }
}

// Tile geometry for maxpool3x3_cols_kernel, or false when the shape is not its (the caller falls back).
// Env overrides for tuning runs: PVHIP_POOL3=0 disables, PVHIP_POOL3_CFG="G,S,band_rows", PVHIP_POOL3_KB (LDS budget
// of one input buffer), PVHIP_POOL3_STAGE=0 (direct stores), PVHIP_POOL3_WG (workgroups per CU of the persistent grid).
struct Pool3Plan {
    Pool3Args a;
    size_t    lds;
    int       grid;
    bool      stage;
};

// any_align: the caller's kernel copies from the 16-byte boundary below a band / group start and reads its LDS image shifted by the
// remainder (dwconv3x3_cols_kernel), so starts need not be 16-byte aligned (odd widths, 150-wide rows)
bool plan_pool3_search(const float* x, float* y, int planes, int h, int w, int oh, int ow, int st, int pt, int pl, int hp, int wp,
                Pool3Plan& out, bool any_align = false, int stage_override = -1) {
    if (pt > 2 || pl > 2 || (oh - 1) * st > pt + h - 1 || (ow - 1) * st > pl + w - 1) return false;   // clamped taps stay in their window
    if (ow > kBlock || hp < pt + h || wp < pl + w) return false;
    const int hw = h * w, ohw = oh * ow;
    const Settings& cfg = settings();                   // PVHIP_POOL3_KB / _STAGE / _CFG / _WG: tuning runs only
    const size_t budget = (size_t)cfg.pool3_kb * 1024;
    const bool stage = stage_override >= 0 ? stage_override != 0 : cfg.pool3_stage != 0;
    int G = cfg.pool3_g, S = cfg.pool3_s, band = cfg.pool3_band;
    const bool forced = G > 0 && S > 0 && band > 0;
    const bool need4 = !any_align && ((hw % 4 != 0) || (ohw % 4 != 0));      // group starts must stay 16-byte aligned
    if (!forced) {
        double best = -1.0;
        const size_t row_b = (size_t)w * 4;
        for (int nb = 1; nb <= oh; ++nb) {                    // number of bands
            const int br = (oh + nb - 1) / nb;
            if ((oh + br - 1) / br != nb) continue;
            const int rows_in = (nb == 1) ? h : min(h, (br - 1) * st + 3);
            if (nb > 1 && (w % 4 != 0) && !any_align) break;  // band starts must be 16-byte aligned
            const size_t plane_b = (size_t)rows_in * row_b;
            if (plane_b > 2 * budget) continue;
            const double halo_eff = (nb == 1) ? 1.0 : (double)(br * st) / (double)((br - 1) * st + 3);
            for (int g = 1; g <= 64 && (size_t)g * plane_b <= budget + budget / 2; ++g) {
                if (g > planes) break;
                if (need4 && nb == 1 && (g % 4 != 0) && g != planes) continue;
                if (need4 && nb > 1) continue;
                // any_align (depthwise): the plan must fit as it is (two input buffers + the output stage in 64 KB: a stride-1 layer's
                // output is as large as its input), and tiles whose outputs start on 16-byte boundaries are preferred
                bool   fits = true;
                double vec_eff = 1.0;
                if (any_align) {
                    const size_t in_f  = (nb == 1) ? (((size_t)g * hw + 3 + 255) & ~(size_t)255) : (size_t)g * (((size_t)rows_in * w + 3 + 255) & ~(size_t)255);
                    const size_t out_f = (nb == 1) ? (size_t)g * ohw : (size_t)g * br * ow;
                    fits = in_f * 8 + (stage ? ((out_f + 3) & ~(size_t)3) * 4 : 0) <= 64 * 1024;
                    const bool vec = (nb == 1) ? ((size_t)g * ohw % 4 == 0 || g == planes) : (ohw % 4 == 0 && (br * ow) % 4 == 0);
                    vec_eff = vec ? 1.0 : 0.8;
                    if (nb > 1 && (w & 1)) vec_eff *= 0.6;       // bands of odd-width planes start on no boundary at all (measured: slow)
                }
                if (!fits) break;
                for (int s = 1; s <= 8 && s <= br; ++s) {
                    const int items = g * s * ow;
                    const int sr = (br + s - 1) / s;
                    const double util = (double)items / (double)(((items + kBlock - 1) / kBlock) * kBlock);
                    const double seg_eff = (double)br / (double)(s * sr) * ((double)(sr * st) / (double)(sr * st + 3 - st));
                    const double size_eff = (double)((size_t)g * plane_b) / (double)((size_t)g * plane_b + 2048);   // per-tile overhead
                    const double sc = util * seg_eff * halo_eff * size_eff * vec_eff;
                    if (sc > best) { best = sc; G = g; S = s; band = br; }
                }
            }
            if (nb == 1 && best > 0.0 && (size_t)h * row_b <= budget) break;   // whole planes fit: no bands
        }
        if (best < 0.0) return false;
    }
    Pool3Args a{};
    a.x = x; a.y = y; a.n_planes = planes; a.h = h; a.w = w; a.oh = oh; a.ow = ow; a.pt = pt; a.pl = pl; a.hp = hp; a.wp = wp;
    a.x_bytes = (unsigned long long)planes * hw * 4ull;
    if (G > planes) G = planes;
    a.G = G; a.S = S; a.band_rows = band; a.n_bands = (oh + band - 1) / band;
    a.dense = (a.n_bands == 1);
    if (!a.dense && !any_align && (w % 4 != 0 || need4)) return false;
    if (a.dense && need4 && (G % 4 != 0) && G != planes) return false;
    const int rows_in = a.dense ? h : min(h, (band - 1) * st + 3);
    const int slack = any_align ? 3 : 0;               // the copy starts up to three floats below the first one
    a.plane_l     = a.dense ? hw : ((rows_in * w + slack + 255) & ~255);
    a.in_floats   = a.dense ? ((G * hw + slack + 255) & ~255) : G * a.plane_l;
    a.out_plane_l = a.dense ? ohw : band * ow;
    a.n_tiles     = ((planes + G - 1) / G) * a.n_bands;
    a.vec_out     = a.dense ? (((size_t)G * ohw % 4 == 0 || G == planes) ? 1 : 0) : ((ohw % 4 == 0) && ((band * ow) % 4 == 0));
    const size_t out_b = stage ? (size_t)((G * a.out_plane_l + 3) & ~3) * 4 : 0;
    out.lds = (size_t)a.in_floats * 8 + out_b;
    if (out.lds > 64 * 1024 || G * S * ow <= 0) return false;
    int per_cu = (int)((size_t)(160 * 1024) / out.lds);
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    if (cfg.pool3_wg > 0) per_cu = cfg.pool3_wg;
    out.grid  = min(a.n_tiles, per_cu * kNumCU);
    out.a     = a;
    out.stage = stage;
    return true;
}

// The search above costs tens of microseconds; a forward pass asks for the same few shapes over and over.
bool plan_pool3(const float* x, float* y, int planes, int h, int w, int oh, int ow, int st, int pt, int pl, int hp, int wp,
                Pool3Plan& out, bool any_align = false, int stage_override = -1) {
    if (!settings().pool3) return false;
    struct Entry { int key[12]; bool ok; Pool3Plan plan; };
    static std::vector<Entry> cache;
    const int key[12] = {planes, h, w, oh, ow, st, pt, pl, hp, wp, settings().generation, (any_align ? 1 : 0) + 2 * (stage_override + 1)};
    for (const Entry& e : cache)
        if (memcmp(e.key, key, sizeof key) == 0) {
            if (!e.ok) return false;
            out = e.plan; out.a.x = x; out.a.y = y;
            return true;
        }
    Entry e;
    memcpy(e.key, key, sizeof key);
    e.ok = plan_pool3_search(x, y, planes, h, w, oh, ow, st, pt, pl, hp, wp, e.plan, any_align, stage_override);
    if (settings().pool3_verbose)
        fprintf(stderr, "pool3 planes=%d %dx%d->%dx%d s%d: %s G=%d S=%d band=%d bands=%d tiles=%d lds=%zu grid=%d\n", planes, h, w, oh, ow, st,
                e.ok ? "ok" : "fallback", e.plan.a.G, e.plan.a.S, e.plan.a.band_rows, e.plan.a.n_bands, e.plan.a.n_tiles, e.plan.lds, e.plan.grid);
    if (cache.size() >= 256) cache.clear();
    cache.push_back(e);
    if (!e.ok) return false;
    out = e.plan;
    return true;
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

// Depthwise 3x3 (stride 1 or 2) on the STRUCTURE of maxpool3x3_cols_kernel (same planner, same tiles): persistent grid, the dense input
// of tile t + 1 arrives by LDS-DMA (nontemporal, 16 bytes per lane) while tile t is computed; a lane owns one output COLUMN of a row
// segment and slides down it with the last rows' three taps in registers (stride 1: three LDS reads per output instead of nine, and
// no LDS reads for the weights: the nine weights of the lane's plane sit in registers; a tap in the padding reads as 0.0);
// outputs meet in LDS and leave as dense 16-byte nontemporal runs.  The one-shot kernel above zeroes a padded LDS image, fills it
// through registers with three divisions per 16 bytes, reads 18 LDS words per output and stores 4 bytes per lane: 2.3-3.5 TB/s on the
// MobileNet layers.  Same products in the same order (ky, then kx; a padded tap is 0 * w): the same bits.
// STAGE = false (stride-1 layers whose planes fill LDS twice over as it is: MobileNet's 75x75): a lane stores its outputs itself
template <int ST, bool NT, bool STAGE = true>
__global__ __launch_bounds__(kBlock) void dwconv3x3_cols_kernel(Pool3Args a, Pool3Divs dv, const float* __restrict__ wts, int channels,
                                                                 DwEpilogue ep) {
    extern __shared__ __attribute__((aligned(1024))) float lds3[];
    float* const outb = lds3 + 2 * a.in_floats;
    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)(tid & 63) * 16u;
    const int hw = a.h * a.w, ohw = a.oh * a.ow;

    // Input of tile t -> buffer `dst`.  A copy starts at the 16-byte boundary at or below its first float (a 150- or 75-wide row, a
    // 5625-float plane do not start on one); the image in LDS is then shifted by the remainder, which the reads add back.
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, (int)(a.x_bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)a.x_bytes), 0x00020000);
    auto issue = [&](int t, float* dst) {
        const int pg = (int)fdiv((unsigned)t, dv.bands), b = t - pg * a.n_bands;
        const int g0 = pg * a.G, gn = min(a.G, a.n_planes - g0);
        const int oy0 = b * a.band_rows, oy1 = min(a.oh, oy0 + a.band_rows);
        const int iy_lo = max(0, oy0 * ST - a.pt), iy_hi = min(a.h, (oy1 - 1) * ST + 3 - a.pt);
        if (a.dense) {
            const size_t src = (size_t)g0 * hw;                              // floats
            const int    sh  = (int)(src & 3);
            const int pieces = (gn * hw + sh + 255) >> 8;
            for (int q = wave; q < pieces; q += 4) pool_dma_b128<NT>(xr, dst + q * 256, lane16 + (unsigned)((src - sh) * 4) + (unsigned)q * 1024u, 0u);      // whole offset in the vector offset: range-checked
        } else {
            for (int p = 0; p < gn; ++p) {
                const size_t src = (size_t)(g0 + p) * hw + (size_t)iy_lo * a.w;
                const int    sh  = (int)(src & 3);
                const int    ppp = ((iy_hi - iy_lo) * a.w + sh + 255) >> 8;   // pieces of this plane's band
                for (int q = wave; q < ppp; q += 4)
                    pool_dma_b128<NT>(xr, dst + p * a.plane_l + q * 256, lane16 + (unsigned)((src - sh) * 4) + (unsigned)q * 1024u, 0u);
            }
        }
    };

    int t = blockIdx.x, cur = 0;
    if (t < a.n_tiles) issue(t, lds3);
    for (; t < a.n_tiles; t += gridDim.x) {
        pool_dma_wait();
        __syncthreads();          // tile t has landed; every wave is done with the other buffer and with the output stage
        {
            const int tn = t + (int)gridDim.x;
            if (tn < a.n_tiles) issue(tn, lds3 + (cur ^ 1) * a.in_floats);
        }
        const float* const in = lds3 + cur * a.in_floats;
        const int pg = (int)fdiv((unsigned)t, dv.bands), b = t - pg * a.n_bands;
        const int g0 = pg * a.G, gn = min(a.G, a.n_planes - g0);
        const int oy0 = b * a.band_rows, oy1 = min(a.oh, oy0 + a.band_rows);
        const int iy_lo = max(0, oy0 * ST - a.pt);
        const int rows_t = oy1 - oy0;
        const int seg_rows = (rows_t + a.S - 1) / a.S;
        const int n_items = gn * a.S * a.ow;
        const int sow = a.S * a.ow;
        for (int it = tid; it < n_items; it += kBlock) {
            const unsigned p = fdiv((unsigned)it, dv.sow), rem = (unsigned)it - p * (unsigned)sow;
            const unsigned seg = fdiv(rem, dv.ow), ox = rem - seg * (unsigned)a.ow;
            const int ys = oy0 + (int)seg * seg_rows, ye = min(oy1, ys + seg_rows);
            if (ys >= oy1) continue;                                          // trailing segment of a short band
            const int px0 = (int)ox * ST - a.pl;
            const int c0 = min(max(px0, 0), a.w - 1), c1 = min(max(px0 + 1, 0), a.w - 1), c2 = min(max(px0 + 2, 0), a.w - 1);
            // the plane's nine weights in registers; a tap in the padding reads as 0.0 (the VALUE is masked, as in the reference's padded
            // image: 0 * w -- masking the weight instead would turn an infinite neighbour into NaN)
            const int ch = (g0 + (int)p) % channels;
            const bool m0 = (unsigned)px0 < (unsigned)a.w, m1 = (unsigned)(px0 + 1) < (unsigned)a.w, m2 = (unsigned)(px0 + 2) < (unsigned)a.w;
            float wk[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) wk[k] = wts[ch * 9 + k];
            const float bv = ep.bias != nullptr ? ep.bias[ch] : -0.0f;
            // input row iy of this plane at L + iy * w (the copy started `shift` floats below the band)
            const int shift = a.dense ? (int)(((size_t)g0 * hw) & 3) : (int)(((size_t)(g0 + (int)p) * hw + (size_t)iy_lo * a.w) & 3);
            const float* const L = in + (int)p * a.plane_l + shift - iy_lo * a.w;
            // the three taps of input row IY (zeros for a row in the padding)
#define PV_DROW(IY, V0, V1, V2)                                                               \
    do {                                                                                      \
        const int    iy_ = (IY);                                                              \
        const bool   ok_ = (unsigned)iy_ < (unsigned)a.h;                                     \
        const float* q_  = L + min(max(iy_, 0), a.h - 1) * a.w;                               \
        V0 = (ok_ && m0) ? q_[c0] : 0.0f; V1 = (ok_ && m1) ? q_[c1] : 0.0f; V2 = (ok_ && m2) ? q_[c2] : 0.0f; \
    } while (0)
            float a0, a1, a2, b0, b1, b2, c0v, c1v, c2v;
            PV_DROW(ys * ST - a.pt, a0, a1, a2);
            if (ST == 1) PV_DROW(ys - a.pt + 1, b0, b1, b2);
            float* const yo = STAGE ? outb + (int)p * a.out_plane_l + (ys - oy0) * a.ow + (int)ox
                                    : a.y + (size_t)(g0 + (int)p) * ohw + (size_t)ys * a.ow + ox;
            for (int oy = ys; oy < ye; ++oy) {
                if (ST == 2) PV_DROW(oy * 2 - a.pt + 1, b0, b1, b2);
                PV_DROW(oy * ST - a.pt + 2, c0v, c1v, c2v);
                float sum = 0.0f;
                sum += a0 * wk[0]; sum += a1 * wk[1]; sum += a2 * wk[2];
                sum += b0 * wk[3]; sum += b1 * wk[4]; sum += b2 * wk[5];
                sum += c0v * wk[6]; sum += c1v * wk[7]; sum += c2v * wk[8];
                sum = sum + bv;
                if (ep.act == 1) sum = (sum < 0.0f) ? 0.0f : sum;
                else if (ep.act == 2) { sum = (sum < ep.lo) ? ep.lo : sum; sum = (sum > ep.hi) ? ep.hi : sum; }
                yo[(oy - ys) * a.ow] = sum;
                if (ST == 1) { a0 = b0; a1 = b1; a2 = b2; b0 = c0v; b1 = c1v; b2 = c2v; }
                else         { a0 = c0v; a1 = c1v; a2 = c2v; }
            }
#undef PV_DROW
        }
        if (STAGE) {
        __syncthreads();
        if (a.dense) {
            const int n_out = gn * ohw;
            float* const yd = a.y + (size_t)g0 * ohw;
            if (a.vec_out) {
                const int n4 = n_out >> 2;
                for (int i = tid; i < n4; i += kBlock) pool_st4<NT>(yd + 4 * i, outb + 4 * i);
                for (int i = (n4 << 2) + tid; i < n_out; i += kBlock) pool_st1<NT>(yd + i, outb[i]);
            } else {
                for (int i = tid; i < n_out; i += kBlock) pool_st1<NT>(yd + i, outb[i]);
            }
        } else {
            const int run = rows_t * a.ow;
            for (int p = 0; p < gn; ++p) {
                float* const       yd = a.y + (size_t)(g0 + p) * ohw + (size_t)oy0 * a.ow;
                const float* const so = outb + p * a.out_plane_l;
                if (a.vec_out) {
                    const int n4 = run >> 2;
                    for (int i = tid; i < n4; i += kBlock) pool_st4<NT>(yd + 4 * i, so + 4 * i);
                    for (int i = (n4 << 2) + tid; i < run; i += kBlock) pool_st1<NT>(yd + i, so[i]);
                } else {
                    for (int i = tid; i < run; i += kBlock) pool_st1<NT>(yd + i, so[i]);
                }
            }
        }
        }
        cur ^= 1;
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
    if (kh == 3 && kw == 3 && sh == sw && (sh == 1 || sh == 2)) {
        Pool3Plan pl3;
        if (plan_pool3(x, y, n * c, h, w, oh, ow, sh, pad_top, pad_left, a.hp, a.wp, pl3)) {
            Pool3Divs dv3{make_fastdiv((unsigned)pl3.a.n_bands), make_fastdiv((unsigned)(pl3.a.S * ow)), make_fastdiv((unsigned)ow)};
            const dim3 g3(pl3.grid), b3(kBlock);
            // nontemporal accesses from the size on where the tensors cannot stay in L2 for the consumer anyway (PVHIP_STREAM_NT)
            const int  ntm = settings().stream_nt;
            const bool nt  = ntm == 2 || (ntm == 1 && (size_t)n * c * ((size_t)h * w + (size_t)oh * ow) * 4 >= ((size_t)64 << 20));
#define PV_P3(ST_, STAGE_)                                                                                            \
    {                                                                                                                 \
        if (nt) hipLaunchKernelGGL((maxpool3x3_cols_kernel<ST_, STAGE_, true>), g3, b3, pl3.lds, state().stream, pl3.a, dv3);  \
        else    hipLaunchKernelGGL((maxpool3x3_cols_kernel<ST_, STAGE_, false>), g3, b3, pl3.lds, state().stream, pl3.a, dv3); \
    }
            if (sh == 1) { if (pl3.stage) PV_P3(1, true) else PV_P3(1, false) }
            else         { if (pl3.stage) PV_P3(2, true) else PV_P3(2, false) }
#undef PV_P3
            PVHIP_LAUNCH_CHECK();
            return PVHIP_OK;
        }
    }
    // LDS-staged path.  ~16 KB of LDS per workgroup (8-10 workgroups per CU overlap each other's load and
    // compute phases; measured best on the GoogLeNet shapes): several whole planes per workgroup when planes
    // are small, bands of output rows of one plane when a plane is larger than the budget.
    const size_t group_bytes = (size_t)settings().pool_lds_kb * 1024;       // PVHIP_POOL_LDS_KB: tuning runs only
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
    // 3x3, stride 1 or 2: the pipelined kernel on the MaxPool kernel's tiles (PVHIP_DWCONV_COLS=0: the one-shot LDS kernel)
    if (kh == 3 && kw == 3 && sh == sw && (sh == 1 || sh == 2) && settings().dwconv_cols) {
        Pool3Plan plan;
        // The lanes store their outputs themselves (PVHIP_DWCONV_COLS=2: through the MaxPool kernel's output stage in LDS, measured
        // slower on every MobileNet layer: 1.32 against 1.16 ms over the 13 -- the stage costs LDS, i.e. halo rows and resident tiles).
        // (bands of odd-width planes start on no boundary at all and ran 20 % slower than the one-shot kernel; MobileNet's 75x75
        // planes fit whole)
        const bool ok = plan_pool3(x, y, n * g, h, wdt, oh, ow, sh, pad_top, pad_left, hp, wp, plan, true, settings().dwconv_cols == 2 ? 1 : 0) &&
                        !(plan.a.n_bands > 1 && (wdt & 1));
        if (ok) {
            const Pool3Divs dv3{make_fastdiv((unsigned)plan.a.n_bands), make_fastdiv((unsigned)(plan.a.S * ow)), make_fastdiv((unsigned)ow)};
            const int  ntm = settings().stream_nt;
            const bool nt  = ntm == 2 || (ntm == 1 && (size_t)n * g * ((size_t)h * wdt + (size_t)oh * ow) * 4 >= ((size_t)64 << 20));
            const dim3 g3(plan.grid), b3(kBlock);
#define PV_DW(ST_, NT_)                                                                                                       \
    {                                                                                                                         \
        if (plan.stage) hipLaunchKernelGGL((dwconv3x3_cols_kernel<ST_, NT_, true>), g3, b3, plan.lds, state().stream, plan.a, dv3, w, g, ep);  \
        else            hipLaunchKernelGGL((dwconv3x3_cols_kernel<ST_, NT_, false>), g3, b3, plan.lds, state().stream, plan.a, dv3, w, g, ep); \
    }
            if (sh == 1) { if (nt) PV_DW(1, true) else PV_DW(1, false) }
            else         { if (nt) PV_DW(2, true) else PV_DW(2, false) }
#undef PV_DW
            PVHIP_LAUNCH_CHECK();
            return PVHIP_OK;
        }
    }
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
