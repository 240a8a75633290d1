// 3x3 average / max pooling (stride 1 or 2, pad 1) and bilinear x2 resampling, NHWC fp32, gfx950.
// All HBM-bound gathers: one thread = V channels of one output pixel, 16-byte accesses along C.
// Backward passes are written in gather form too (deterministic, no atomics).
#include "common.h"

namespace senas {

struct PoolGeom {
    int n, h, w, c, ho, wo, stride;
};

template <int V>
__device__ __forceinline__ void decode(long idx, int cv, int wo, int ho, int& c, int& ox, int& oy, int& n) {
    c = (int)(idx % cv) * V;
    long p = idx / cv;
    ox = (int)(p % wo);
    p /= wo;
    oy = (int)(p % ho);
    n = (int)(p / ho);
}

// forward kernels: block b owns flat elements [b*256*P, (b+1)*256*P) as P chunks of 256; `P > 0` doubles as "uniform":
// the block lies inside one image (decided by the launcher), so statistics are flushed once per block
#define SENAS_FWD_LOOP_BEGIN(total_, P_)                                                        \
    Stats4 acc_st;                                                                              \
    stats_init4(acc_st);                                                                        \
    const bool uniform = (P_) > 0;                                                              \
    const int chunks = uniform ? (P_) : 1;                                                      \
    int n_blk = 0, ch_thr = 0;                                                                  \
    for (int kk = 0; kk < chunks; ++kk) {                                                       \
        long idx = ((long)xcd_block().x * chunks + kk) * 256 + threadIdx.x;                        \
        const bool active = idx < (total_);                                                     \
        if (!active) idx = (total_) - 1;
#define SENAS_FWD_LOOP_END(stats_, c_)                                                          \
    }                                                                                           \
    if constexpr (V == 4) stats_flush4(acc_st, stats_, uniform, n_blk, c_, ch_thr);

template <int V>
__device__ __forceinline__ void fwd_stats(Stats4& a, double* stats, bool uniform, int n, int c, int ch, const float (&v)[V], bool active) {
    if constexpr (V == 4) {
        stats_accumulate4(a, stats, uniform, n, c, ch, v, active);
    } else {
        if (stats == nullptr || !active) return;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            double* st = stats + ((size_t)n * c + ch + j) * 2;
            atomicAdd(st, (double)v[j]);
            atomicAdd(st + 1, (double)v[j] * v[j]);
        }
    }
}

// ---------------------------------------------------------------- average pool, count_include_pad=False
template <int V>
__global__ __launch_bounds__(256) void avgpool3_fwd_kernel(PoolGeom g, const float* __restrict__ x, int in_relu,
                                                           float* __restrict__ y, double* __restrict__ stats, long total, int P) {
    SENAS_FWD_LOOP_BEGIN(total, P)
    int ch, ox, oy, n;
    decode<V>(idx, g.c / V, g.wo, g.ho, ch, ox, oy, n);
    n_blk = n; ch_thr = ch;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    int cnt = 0;
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * g.stride - 1 + ky;
        if (iy < 0 || iy >= g.h) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * g.stride - 1 + kx;
            if (ix < 0 || ix >= g.w) continue;
            float v[V];
            ldv<V>(x + ((size_t)(n * g.h + iy) * g.w + ix) * g.c + ch, v);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += in_relu ? fmaxf(v[j], 0.f) : v[j];
            ++cnt;
        }
    }
    // torch divides the window sum by the number of in-bounds taps
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = acc[j] / (float)cnt;
    if (active) stv<V>(y + ((size_t)(n * g.ho + oy) * g.wo + ox) * g.c + ch, acc);
    fwd_stats<V>(acc_st, stats, uniform, n, g.c, ch, acc, active);
    SENAS_FWD_LOOP_END(stats, g.c)
}

__device__ __forceinline__ int window_count(int o, int stride, int lim) {
    int lo = o * stride - 1, hi = lo + 2;
    if (lo < 0) lo = 0;
    if (hi > lim - 1) hi = lim - 1;
    return hi - lo + 1;
}

// dx[iy,ix] = sum over windows covering (iy,ix) of dy[oy,ox] / count(oy,ox)
template <int V>
__global__ __launch_bounds__(256) void avgpool3_bwd_kernel(PoolGeom g, const float* __restrict__ dy, int in_relu,
                                                           const float* __restrict__ x, float* __restrict__ dx, long total) {
    const long idx = (long)xcd_block().x * 256 + threadIdx.x;     // (XCD-aware order: common.h)
    if (idx >= total) return;
    int ch, ix, iy, n;
    decode<V>(idx, g.c / V, g.w, g.h, ch, ix, iy, n);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    // windows (oy, ky) with oy*s - 1 + ky == iy
    for (int ky = 0; ky < 3; ++ky) {
        const int t = iy + 1 - ky;
        if (t < 0 || t % g.stride != 0) continue;
        const int oy = t / g.stride;
        if (oy >= g.ho) continue;
        const int cy = window_count(oy, g.stride, g.h);
        for (int kx = 0; kx < 3; ++kx) {
            const int u = ix + 1 - kx;
            if (u < 0 || u % g.stride != 0) continue;
            const int ox = u / g.stride;
            if (ox >= g.wo) continue;
            const float inv = 1.f / (float)(cy * window_count(ox, g.stride, g.w));
            float v[V];
            ldv<V>(dy + ((size_t)(n * g.ho + oy) * g.wo + ox) * g.c + ch, v);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += v[j] * inv;
        }
    }
    const size_t o = ((size_t)(n * g.h + iy) * g.w + ix) * g.c + ch;
    if (in_relu) {
        float xv[V];
        ldv<V>(x + o, xv);
#pragma unroll
        for (int j = 0; j < V; ++j) if (!(xv[j] > 0.f)) acc[j] = 0.f;
    }
    stv<V>(dx + o, acc);
}

// ---------------------------------------------------------------- max pool (first maximum wins)
template <int V>
__global__ __launch_bounds__(256) void maxpool3_fwd_kernel(PoolGeom g, const float* __restrict__ x, int in_relu,
                                                           float* __restrict__ y, uint8_t* __restrict__ amax,
                                                           double* __restrict__ stats, long total, int P) {
    SENAS_FWD_LOOP_BEGIN(total, P)
    int ch, ox, oy, n;
    decode<V>(idx, g.c / V, g.wo, g.ho, ch, ox, oy, n);
    n_blk = n; ch_thr = ch;
    float best[V];
    int arg[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { best[j] = -INFINITY; arg[j] = -1; }
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * g.stride - 1 + ky;
        if (iy < 0 || iy >= g.h) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * g.stride - 1 + kx;
            if (ix < 0 || ix >= g.w) continue;
            float v[V];
            ldv<V>(x + ((size_t)(n * g.h + iy) * g.w + ix) * g.c + ch, v);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float t = in_relu ? fmaxf(v[j], 0.f) : v[j];
                if (arg[j] < 0 || t > best[j] || t != t) { best[j] = t; arg[j] = ky * 3 + kx; }
            }
        }
    }
    const size_t o = ((size_t)(n * g.ho + oy) * g.wo + ox) * g.c + ch;
    if (active) {
        stv<V>(y + o, best);
#pragma unroll
        for (int j = 0; j < V; ++j) amax[o + j] = (uint8_t)arg[j];
    }
    fwd_stats<V>(acc_st, stats, uniform, n, g.c, ch, best, active);
    SENAS_FWD_LOOP_END(stats, g.c)
}

template <int V>
__global__ __launch_bounds__(256) void maxpool3_bwd_kernel(PoolGeom g, const float* __restrict__ dy,
                                                           const uint8_t* __restrict__ amax, int in_relu,
                                                           const float* __restrict__ x, float* __restrict__ dx, long total) {
    const long idx = (long)xcd_block().x * 256 + threadIdx.x;     // (XCD-aware order: common.h)
    if (idx >= total) return;
    int ch, ix, iy, n;
    decode<V>(idx, g.c / V, g.w, g.h, ch, ix, iy, n);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int t = iy + 1 - ky;             // oy*stride
        if (t < 0 || t % g.stride != 0) continue;
        const int oy = t / g.stride;
        if (oy >= g.ho) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int u = ix + 1 - kx;
            if (u < 0 || u % g.stride != 0) continue;
            const int ox = u / g.stride;
            if (ox >= g.wo) continue;
            const size_t o = ((size_t)(n * g.ho + oy) * g.wo + ox) * g.c + ch;
            float v[V];
            ldv<V>(dy + o, v);
#pragma unroll
            for (int j = 0; j < V; ++j) if (amax[o + j] == ky * 3 + kx) acc[j] += v[j];
        }
    }
    const size_t o = ((size_t)(n * g.h + iy) * g.w + ix) * g.c + ch;
    if (in_relu) {
        float xv[V];
        ldv<V>(x + o, xv);
#pragma unroll
        for (int j = 0; j < V; ++j) if (!(xv[j] > 0.f)) acc[j] = 0.f;
    }
    stv<V>(dx + o, acc);
}

// ---------------------------------------------------------------- bilinear x2, align_corners=False
// source index for destination d: max(0, 0.5*(d+0.5) - 0.5); i1 = min(i0+1, lim-1)
__device__ __forceinline__ void bilinear_src(int d, int lim, int& i0, int& i1, float& l0, float& l1) {
    float s = 0.5f * ((float)d + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    i1 = i0 + (i0 < lim - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

template <int V>
__global__ __launch_bounds__(256) void bilinear2x_fwd_kernel(PoolGeom g, const float* __restrict__ x,
                                                             float* __restrict__ y, double* __restrict__ stats, long total, int P) {
    SENAS_FWD_LOOP_BEGIN(total, P)
    int ch, ox, oy, n;
    decode<V>(idx, g.c / V, g.wo, g.ho, ch, ox, oy, n);
    n_blk = n; ch_thr = ch;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    bilinear_src(oy, g.h, y0, y1, ly0, ly1);
    bilinear_src(ox, g.w, x0, x1, lx0, lx1);
    float a[V], b[V], c[V], d[V], r[V];
    ldv<V>(x + ((size_t)(n * g.h + y0) * g.w + x0) * g.c + ch, a);
    ldv<V>(x + ((size_t)(n * g.h + y0) * g.w + x1) * g.c + ch, b);
    ldv<V>(x + ((size_t)(n * g.h + y1) * g.w + x0) * g.c + ch, c);
    ldv<V>(x + ((size_t)(n * g.h + y1) * g.w + x1) * g.c + ch, d);
#pragma unroll
    for (int j = 0; j < V; ++j) r[j] = ly0 * (lx0 * a[j] + lx1 * b[j]) + ly1 * (lx0 * c[j] + lx1 * d[j]);
    if (active) stv<V>(y + ((size_t)(n * g.ho + oy) * g.wo + ox) * g.c + ch, r);
    fwd_stats<V>(acc_st, stats, uniform, n, g.c, ch, r, active);
    SENAS_FWD_LOOP_END(stats, g.c)
}

// weight with which destination row/col d reads source index i
__device__ __forceinline__ float bilinear_w(int d, int lim_src, int i) {
    int i0, i1;
    float l0, l1;
    bilinear_src(d, lim_src, i0, i1, l0, l1);
    return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

template <int V>
__global__ __launch_bounds__(256) void bilinear2x_bwd_kernel(PoolGeom g, const float* __restrict__ dy,
                                                             float* __restrict__ dx, long total) {
    const long idx = (long)xcd_block().x * 256 + threadIdx.x;     // (XCD-aware order: common.h)
    if (idx >= total) return;
    int ch, ix, iy, n;
    decode<V>(idx, g.c / V, g.w, g.h, ch, ix, iy, n);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int oy = 2 * iy - 2; oy <= 2 * iy + 2; ++oy) {
        if (oy < 0 || oy >= g.ho) continue;
        const float wy = bilinear_w(oy, g.h, iy);
        if (wy == 0.f) continue;
        for (int ox = 2 * ix - 2; ox <= 2 * ix + 2; ++ox) {
            if (ox < 0 || ox >= g.wo) continue;
            const float wgt = wy * bilinear_w(ox, g.w, ix);
            if (wgt == 0.f) continue;
            float v[V];
            ldv<V>(dy + ((size_t)(n * g.ho + oy) * g.wo + ox) * g.c + ch, v);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] = fmaf(wgt, v[j], acc[j]);
        }
    }
    stv<V>(dx + ((size_t)(n * g.h + iy) * g.w + ix) * g.c + ch, acc);
}

static bool pool_geom(int n, int h, int w, int c, int stride, PoolGeom& g) {
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || (stride != 1 && stride != 2)) return false;
    g = PoolGeom{n, h, w, c, (h + 2 - 3) / stride + 1, (w + 2 - 3) / stride + 1, stride};
    return true;
}

}  // namespace senas

using namespace senas;

#define SENAS_LAUNCH_V(kernel, total_expr, ...)                                                                 \
    do {                                                                                                        \
        const int V_ = (g.c % 4 == 0) ? 4 : 1;                                                                  \
        const long total = (total_expr) * (long)(g.c / V_);                                                     \
        dim3 grid((unsigned)((total + 255) / 256));                                                             \
        if (V_ == 4) hipLaunchKernelGGL((kernel<4>), grid, dim3(256), 0, as_stream(stream), g, __VA_ARGS__, total); \
        else hipLaunchKernelGGL((kernel<1>), grid, dim3(256), 0, as_stream(stream), g, __VA_ARGS__, total);     \
    } while (0)

// forward kernels: P chunks of 256 elements per block when that keeps every block inside one image (statistics are
// then flushed once per block), else one chunk per block with per-element atomics
#define SENAS_LAUNCH_FWD(kernel, per_img_expr, c_out, ...)                                                      \
    do {                                                                                                        \
        const int V_ = (g.c % 4 == 0) ? 4 : 1;                                                                  \
        const long per_img = (per_img_expr) * (long)(g.c / V_);                                                 \
        const long total = per_img * g.n;                                                                       \
        const int P_ = V_ == 4 ? stats_chunks_per_block(per_img, c_out, total) : 0;                                    \
        dim3 grid((unsigned)((total + 256L * (P_ > 0 ? P_ : 1) - 1) / (256L * (P_ > 0 ? P_ : 1))));             \
        if (V_ == 4) hipLaunchKernelGGL((kernel<4>), grid, dim3(256), 0, as_stream(stream), g, __VA_ARGS__, total, P_); \
        else hipLaunchKernelGGL((kernel<1>), grid, dim3(256), 0, as_stream(stream), g, __VA_ARGS__, total, P_); \
    } while (0)

extern "C" int senas_avgpool3_fwd(int n, int h, int w, int c, int stride, const float* x, int in_relu, float* y,
                                  double* stats, void* stream) {
    PoolGeom g;
    SENAS_REQUIRE(pool_geom(n, h, w, c, stride, g) && x && y, "avgpool3_fwd: bad argument");
    SENAS_LAUNCH_FWD(avgpool3_fwd_kernel, (long)g.ho * g.wo, g.c, x, in_relu, y, stats);
    return launch_status("avgpool3_fwd");
}

extern "C" int senas_avgpool3_bwd(int n, int h, int w, int c, int stride, const float* dy, int in_relu, const float* x,
                                  float* dx, void* stream) {
    PoolGeom g;
    SENAS_REQUIRE(pool_geom(n, h, w, c, stride, g) && dy && dx && (!in_relu || x), "avgpool3_bwd: bad argument");
    SENAS_LAUNCH_V(avgpool3_bwd_kernel, (long)n * g.h * g.w, dy, in_relu, x, dx);
    return launch_status("avgpool3_bwd");
}

extern "C" int senas_maxpool3_fwd(int n, int h, int w, int c, int stride, const float* x, int in_relu, float* y,
                                  uint8_t* argmax, double* stats, void* stream) {
    PoolGeom g;
    SENAS_REQUIRE(pool_geom(n, h, w, c, stride, g) && x && y && argmax, "maxpool3_fwd: bad argument");
    SENAS_LAUNCH_FWD(maxpool3_fwd_kernel, (long)g.ho * g.wo, g.c, x, in_relu, y, argmax, stats);
    return launch_status("maxpool3_fwd");
}

extern "C" int senas_maxpool3_bwd(int n, int h, int w, int c, int stride, const float* dy, const uint8_t* argmax,
                                  int in_relu, const float* x, float* dx, void* stream) {
    PoolGeom g;
    SENAS_REQUIRE(pool_geom(n, h, w, c, stride, g) && dy && dx && argmax && (!in_relu || x), "maxpool3_bwd: bad argument");
    SENAS_LAUNCH_V(maxpool3_bwd_kernel, (long)n * g.h * g.w, dy, argmax, in_relu, x, dx);
    return launch_status("maxpool3_bwd");
}

extern "C" int senas_bilinear2x_fwd(int n, int h, int w, int c, const float* x, float* y, double* stats, void* stream) {
    SENAS_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && x && y, "bilinear2x_fwd: bad argument");
    PoolGeom g{n, h, w, c, 2 * h, 2 * w, 2};
    SENAS_LAUNCH_FWD(bilinear2x_fwd_kernel, (long)g.ho * g.wo, g.c, x, y, stats);
    return launch_status("bilinear2x_fwd");
}

// ---------------------------------------------------------------- channel un-stacking (search cell: stacked candidates)
// src [n][hw][k*c] -> dst_e [n][hw][c] for e < k, with the producer-side batch-norm statistics of every part.
// grid.y = part; the element loop and the statistics are those of the other forward kernels.
struct UnstackParts {
    float* dst[SENAS_MAX_STACK];
    double* stats[SENAS_MAX_STACK];
};

template <int V>
__global__ __launch_bounds__(256) void unstack_kernel(PoolGeom g, const float* __restrict__ src, UnstackParts parts, int k,
                                                      long total, int P) {
    const int e = (int)xcd_block().y;                      // (with the loop's xcd_block().x: one decode of the grid)
    float* __restrict__ dst = parts.dst[e];
    double* __restrict__ stats = parts.stats[e];
    if (dst == nullptr) return;                            // a padding part of the stack: nobody reads it
    SENAS_FWD_LOOP_BEGIN(total, P)
    int ch, ox, oy, n;
    decode<V>(idx, g.c / V, g.wo, g.ho, ch, ox, oy, n);
    n_blk = n; ch_thr = ch;
    const size_t pix = (size_t)(n * g.ho + oy) * g.wo + ox;
    float v[V];
    ldv<V>(src + pix * ((size_t)k * g.c) + (size_t)e * g.c + ch, v);
    if (active) stv<V>(dst + pix * g.c + ch, v);
    fwd_stats<V>(acc_st, stats, uniform, n, g.c, ch, v, active);
    SENAS_FWD_LOOP_END(stats, g.c)
}

extern "C" int senas_unstack_fwd(int n, int64_t hw, int c, int k, const float* src, float* const* dst, double* const* stats,
                                 void* stream) {
    SENAS_REQUIRE(n > 0 && hw > 0 && c > 0 && k >= 1 && k <= SENAS_MAX_STACK && src && dst, "unstack_fwd: bad argument");
    UnstackParts parts;
    for (int e = 0; e < SENAS_MAX_STACK; ++e) {
        parts.dst[e] = e < k ? dst[e] : nullptr;
        parts.stats[e] = (e < k && stats != nullptr) ? stats[e] : nullptr;
        SENAS_REQUIRE(e != 0 || parts.dst[e] != nullptr, "unstack_fwd: null first part");
    }
    PoolGeom g{n, 1, (int)hw, c, 1, (int)hw, 1};                     // one row of hw pixels per image
    SENAS_REQUIRE(hw < 0x7fffffffL, "unstack_fwd: map too large");
    const int V_ = (c % 4 == 0) ? 4 : 1;
    const long per_img = hw * (long)(c / V_);
    const long total = per_img * n;
    const bool want = stats != nullptr;
    const int P_ = (V_ == 4 && want) ? stats_chunks_per_block(per_img, c, total) : 0;
    dim3 grid((unsigned)((total + 256L * (P_ > 0 ? P_ : 1) - 1) / (256L * (P_ > 0 ? P_ : 1))), (unsigned)k);
    if (V_ == 4) hipLaunchKernelGGL((unstack_kernel<4>), grid, dim3(256), 0, as_stream(stream), g, src, parts, k, total, P_);
    else hipLaunchKernelGGL((unstack_kernel<1>), grid, dim3(256), 0, as_stream(stream), g, src, parts, k, total, P_);
    return launch_status("unstack_fwd");
}

extern "C" int senas_bilinear2x_bwd(int n, int h, int w, int c, const float* dy, float* dx, void* stream) {
    SENAS_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && dy && dx, "bilinear2x_bwd: bad argument");
    PoolGeom g{n, h, w, c, 2 * h, 2 * w, 2};
    SENAS_LAUNCH_V(bilinear2x_bwd_kernel, (long)n * g.h * g.w, dy, dx);
    return launch_status("bilinear2x_bwd");
}
