// Convolution kernels (dense + depthwise), NHWC fp32, gfx950.
//
// "direct" family: one lane = one output pixel, the weight tile of the block is wave-uniform so it
// is fetched through the scalar cache (s_load) and every FMA has an SGPR operand; activations are
// read as 16-byte vectors along the channel axis.  Handles every shape on the path (any channel
// count, stride 1/2, dilation, transposed) and is the fallback for the MFMA kernels in conv_mfma.hip.
#include "common.h"

namespace senas {

// ---------------------------------------------------------------------------------------------
// weight repack: torch layout src[d0][d1][taps] -> dst[tap][A][B]
//   swap == 0: A = d0, B = d1        swap == 1: A = d1, B = d0
__global__ void pack_weights_kernel(const float* __restrict__ src, float* __restrict__ dst, int d0, int d1, int taps,
                                    int swap) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int total = d0 * d1 * taps;
    if (i >= total) return;
    // iterate in dst order for coalesced stores
    int A = swap ? d1 : d0, B = swap ? d0 : d1;
    int b = i % B, a = (i / B) % A, t = i / (A * B);
    int s0 = swap ? b : a, s1 = swap ? a : b;
    dst[i] = src[(s0 * d1 + s1) * taps + t];
}

// ---------------------------------------------------------------------------------------------
// dense direct conv.  wp: [tap][cin][cout].  grid = (pixel tiles, n, cout tiles)
template <int COT, bool TG, bool VEC4>
__global__ __launch_bounds__(256) void conv_direct_kernel(GatherGeom g, const float* __restrict__ in,
                                                          const float* __restrict__ wp, float* __restrict__ out,
                                                          int in_relu, const float* __restrict__ mask,
                                                          double* __restrict__ stats) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    const int cob = blockIdx.z * COT;
    const bool live = pix < g.hout * g.wout;
    const int oy = live ? pix / g.wout : 0, ox = live ? pix % g.wout : 0;
    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = 0.f;

    for (int ky = 0; ky < g.kh; ++ky) {
        int iy;
        const bool oky = tap_src<TG>(g, oy, ky, g.hin, iy);
        for (int kx = 0; kx < g.kw; ++kx) {
            int ix;
            const bool ok = live && oky && tap_src<TG>(g, ox, kx, g.win, ix);
            const float* wt = wp + (size_t)(ky * g.kw + kx) * g.cin * g.cout + cob;
            if (ok) {
                const float* ip = in + ((size_t)(n * g.hin + iy) * g.win + ix) * g.cin;
                if (VEC4) {
                    for (int ci = 0; ci < g.cin; ci += 4) {
                        float4 v = *reinterpret_cast<const float4*>(ip + ci);
                        if (in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        const float* w0 = wt + (size_t)ci * g.cout;
#pragma unroll
                        for (int j = 0; j < COT; ++j) {
                            if (cob + j < g.cout) {
                                acc[j] = fmaf(v.x, w0[j], acc[j]);
                                acc[j] = fmaf(v.y, w0[g.cout + j], acc[j]);
                                acc[j] = fmaf(v.z, w0[2 * g.cout + j], acc[j]);
                                acc[j] = fmaf(v.w, w0[3 * g.cout + j], acc[j]);
                            }
                        }
                    }
                } else {
                    for (int ci = 0; ci < g.cin; ++ci) {
                        float v = ip[ci];
                        if (in_relu) v = fmaxf(v, 0.f);
                        const float* w0 = wt + (size_t)ci * g.cout;
#pragma unroll
                        for (int j = 0; j < COT; ++j)
                            if (cob + j < g.cout) acc[j] = fmaf(v, w0[j], acc[j]);
                    }
                }
            }
        }
    }
    const size_t obase = ((size_t)n * g.hout * g.wout + pix) * g.cout + cob;
    if (live) {
#pragma unroll
        for (int j = 0; j < COT; ++j) {
            if (cob + j < g.cout) {
                float v = acc[j];
                if (mask != nullptr && !(mask[obase + j] > 0.f)) v = 0.f;
                acc[j] = v;
                out[obase + j] = v;
            }
        }
    }
    if (stats != nullptr) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int j = 0; j < COT; ++j) {
            if (cob + j < g.cout) {
                double v = live ? (double)acc[j] : 0.0;
                double s = wave_sum(v), q = wave_sum(v * v);
                if (lane == 0) {
                    double* st = stats + ((size_t)n * g.cout + cob + j) * 2;
                    atomicAdd(st, s);
                    atomicAdd(st + 1, q);
                }
            }
        }
    }
}

template <bool TG>
static int launch_direct(const GatherGeom& g, const float* in, const float* wp, float* out, int in_relu,
                         const float* mask, double* stats, hipStream_t st) {
    const int tiles = (g.hout * g.wout + 255) / 256;
    const bool v4 = (g.cin % 4) == 0;
#define SENAS_GO(COT)                                                                                       \
    do {                                                                                                    \
        dim3 grid(tiles, g.n, (g.cout + COT - 1) / COT);                                                    \
        if (v4) hipLaunchKernelGGL((conv_direct_kernel<COT, TG, true>), grid, dim3(256), 0, st, g, in, wp, out, in_relu, mask, stats); \
        else hipLaunchKernelGGL((conv_direct_kernel<COT, TG, false>), grid, dim3(256), 0, st, g, in, wp, out, in_relu, mask, stats);   \
    } while (0)
    if (g.cout >= 16) SENAS_GO(16);
    else if (g.cout > 4) SENAS_GO(8);
    else SENAS_GO(4);
#undef SENAS_GO
    return launch_status("conv_direct");
}

// ---------------------------------------------------------------------------------------------
// dense weight gradient.  dW[b][a][tap] = sum_{n,p} I[n, p*s - pad + k*d][a] * G[n,p][b]
// G lives on the coarse grid (hg x wg, channels B), I on the fine grid (hi x wi, channels A).
// grid = (pixel chunks, taps, a-tiles * b-tiles); block 256 = 32 b-lanes x 8 a-groups of 4.

__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradGeom g, const float* __restrict__ I,
                                                         const float* __restrict__ G, float* __restrict__ dw,
                                                         int i_relu, int g_relu) {
    const int tap = blockIdx.y, ky = tap / g.kw, kx = tap % g.kw;
    const int btiles = (g.B + 31) / 32;
    const int b = (blockIdx.z % btiles) * 32 + (threadIdx.x & 31);
    const int a0 = (blockIdx.z / btiles) * 32 + (threadIdx.x >> 5) * 4;
    const int per_img = g.hg * g.wg;
    const long total = (long)g.n * per_img;
    long p0 = (long)blockIdx.x * g.chunk, p1 = p0 + g.chunk;
    if (p1 > total) p1 = total;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const bool bok = b < g.B;
    for (long p = p0; p < p1; ++p) {
        const int n = (int)(p / per_img), r = (int)(p % per_img);
        const int gy = r / g.wg, gx = r % g.wg;
        const int iy = gy * g.stride - g.pad + ky * g.dil, ix = gx * g.stride - g.pad + kx * g.dil;
        if (iy < 0 || iy >= g.hi || ix < 0 || ix >= g.wi) continue;   // block-uniform
        float gv = bok ? G[(size_t)p * g.B + b] : 0.f;
        if (g_relu) gv = fmaxf(gv, 0.f);
        const float* ip = I + ((size_t)(n * g.hi + iy) * g.wi + ix) * g.A;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (a0 + j < g.A) {
                float iv = ip[a0 + j];
                if (i_relu) iv = fmaxf(iv, 0.f);
                acc[j] = fmaf(iv, gv, acc[j]);
            }
        }
    }
    if (bok) {
        const int taps = g.kh * g.kw;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (a0 + j < g.A) atomicAdd(&dw[((size_t)b * g.A + a0 + j) * taps + tap], acc[j]);
    }
}

// ---------------------------------------------------------------------------------------------
// depthwise.  w: torch layout [c][1][kh][kw] (same indexing for Conv2d and ConvTranspose2d).
// one thread = 4 channels of one output pixel (c % 4 == 0) or 1 channel.
template <bool TG, int V, bool EPI = false>
__device__ __forceinline__ void dwconv_body(const GatherGeom& g, const float* __restrict__ in,
                                            const float* __restrict__ w, float* __restrict__ out,
                                            int in_relu, const float* __restrict__ mask,
                                            double* __restrict__ stats, long total, int P, const Epi& epi) {
    // block b owns P chunks of 256 flat elements; P > 0 means the launcher guarantees the block lies inside one image,
    // so the batch-norm statistics are kept in registers and flushed once per block (common.h)
    extern __shared__ __attribute__((aligned(16))) float wl[];      // weights as [tap][C]: one 16-byte LDS read per tap
    const int taps = g.kh * g.kw;
    for (int i = threadIdx.x; i < taps * g.cout; i += 256) {
        const int t = i / g.cout, cc = i - t * g.cout;
        wl[i] = w[cc * taps + t];
    }
    __syncthreads();
    Stats4 acc_st;
    stats_init4(acc_st);
    const bool uniform = P > 0;
    const int chunks = uniform ? P : 1;
    const int cv = g.cout / V;
    int n_blk = 0, c_thr = 0;
    for (int kk = 0; kk < chunks; ++kk) {
        long idx = ((long)xcd_block().x * chunks + kk) * 256 + threadIdx.x;
        const bool active = idx < total;
        if (!active) idx = total - 1;
        const int c = (int)(idx % cv) * V;
        long pix = idx / cv;
        const int ox = (int)(pix % g.wout);
        pix /= g.wout;
        const int oy = (int)(pix % g.hout), n = (int)(pix / g.hout);
        n_blk = n; c_thr = c;
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        for (int ky = 0; ky < g.kh; ++ky) {
            int iy;
            if (!tap_src<TG>(g, oy, ky, g.hin, iy)) continue;
            for (int kx = 0; kx < g.kw; ++kx) {
                int ix;
                if (!tap_src<TG>(g, ox, kx, g.win, ix)) continue;
                const float* ip = in + ((size_t)(n * g.hin + iy) * g.win + ix) * g.cin + c;
                float v[V], wt[V];
                ldv<V>(ip, v);
                ldv<V>(wl + (ky * g.kw + kx) * g.cout + c, wt);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    float x = in_relu ? fmaxf(v[j], 0.f) : v[j];
                    acc[j] = fmaf(x, wt[j], acc[j]);
                }
            }
        }
        const size_t o = (((size_t)n * g.hout + oy) * g.wout + ox) * g.cout + c;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            if (mask != nullptr && !(mask[o + j] > 0.f)) acc[j] = 0.f;
        }
        if constexpr (EPI) {                       // eval-mode batch-norm (+ ReLU) of DepSepConv's depthwise half
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const int pc = n * g.cout + c + j;
                acc[j] = fmaf(acc[j], epi.scale[pc], epi.bias[pc]);
                if (epi.relu) acc[j] = fmaxf(acc[j], 0.f);
            }
        }
        if (active) stv<V>(out + o, acc);
        if constexpr (V == 4) {
            stats_accumulate4(acc_st, stats, uniform, n, g.cout, c, acc, active);
        } else {
            if (stats != nullptr && active) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    double* st = stats + ((size_t)n * g.cout + c + j) * 2;
                    atomicAdd(st, (double)acc[j]);
                    atomicAdd(st + 1, (double)acc[j] * acc[j]);
                }
            }
        }
    }
    if constexpr (V == 4) stats_flush4(acc_st, stats, uniform, n_blk, g.cout, c_thr);
}

// Plain-gather depthwise convolution, dilation 1, 4 channels x FOUR consecutive output columns per thread: a row of the
// window is loaded once (4*S + KS - S float4 loads: 8 for 5x5 stride 1 instead of 20) and serves the four outputs --
// the one-pixel-per-thread form above is bound by L1 requests (25 per output), not by HBM.
// flip: use the taps mirrored (the data gradient of a stride-1 depthwise convolution is the same gather with the
// kernel turned by 180 degrees).  Requires wout % 4 == 0.  Thread order: channel quad fastest, then column group.
template <int KS, int S>
__device__ __forceinline__ void dwconv_x4_body(const GatherGeom& g, const float* __restrict__ in,
                                               const float* __restrict__ w, float* __restrict__ out, int flip,
                                               double* __restrict__ stats, long total, int P) {
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [tap][C]
    constexpr int TAPS = KS * KS, COLS = 3 * S + KS;
    const int C = g.cout;
    SENAS_PHASE(16);
    for (int i = threadIdx.x; i < TAPS * C; i += 256) {
        const int t = i / C, cc = i - t * C;
        wl[i] = w[cc * TAPS + (flip ? TAPS - 1 - t : t)];
    }
    __syncthreads();
    SENAS_PHASE(17);
    Stats4 acc_st;
    stats_init4(acc_st);
    const bool uniform = P > 0;
    const int chunks = uniform ? P : 1;
    const int cv = C >> 2, wq = g.wout >> 2;
    int n_blk = 0, c_thr = 0;
    for (int kk = 0; kk < chunks; ++kk) {
        long idx = ((long)xcd_block().x * chunks + kk) * 256 + threadIdx.x;
        const bool active = idx < total;
        if (!active) idx = total - 1;
        const int c = (int)(idx % cv) * 4;
        long r = idx / cv;
        const int ox0 = (int)(r % wq) * 4;
        r /= wq;
        const int oy = (int)(r % g.hout), n = (int)(r / g.hout);
        n_blk = n; c_thr = c;
        float acc[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[j][q] = 0.f;
        const int ix0 = ox0 * S - g.pad;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int iy = oy * S - g.pad + ky;
            if (iy < 0 || iy >= g.hin) continue;
            const float* row = in + ((size_t)(n * g.hin + iy) * g.win) * g.cin + c;
            float4 col[COLS];
#pragma unroll
            for (int x = 0; x < COLS; ++x) {
                const int ix = ix0 + x;
                col[x] = (ix >= 0 && ix < g.win) ? *reinterpret_cast<const float4*>(row + (size_t)ix * g.cin) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const float4 wt = *reinterpret_cast<const float4*>(wl + (ky * KS + kx) * C + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 v = col[j * S + kx];
                    acc[j][0] = fmaf(v.x, wt.x, acc[j][0]); acc[j][1] = fmaf(v.y, wt.y, acc[j][1]);
                    acc[j][2] = fmaf(v.z, wt.z, acc[j][2]); acc[j][3] = fmaf(v.w, wt.w, acc[j][3]);
                }
            }
        }
        SENAS_PHASE(18);
        const size_t o = (((size_t)n * g.hout + oy) * g.wout + ox0) * C + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (active) stv<4>(out + o + (size_t)j * C, acc[j]);
            stats_accumulate4(acc_st, stats, uniform, n, C, c, acc[j], active);
        }
    }
    SENAS_PHASE(19);
    stats_flush4(acc_st, stats, uniform, n_blk, C, c_thr);
    SENAS_PHASE(20);
}

template <int KS, int S>
__global__ __launch_bounds__(256) void dwconv_x4_kernel(GatherGeom g, const float* __restrict__ in, const float* __restrict__ w,
                                                        float* __restrict__ out, int flip, double* __restrict__ stats, long total, int P) {
    dwconv_x4_body<KS, S>(g, in, w, out, flip, stats, total, P);
}

template <bool TG, int V, bool EPI = false>
__global__ __launch_bounds__(256) void dwconv_kernel(GatherGeom g, const float* __restrict__ in,
                                                     const float* __restrict__ w, float* __restrict__ out,
                                                     int in_relu, const float* __restrict__ mask,
                                                     double* __restrict__ stats, long total, int P, Epi epi = Epi{}) {
    dwconv_body<TG, V, EPI>(g, in, w, out, in_relu, mask, stats, total, P, epi);
}

// k depthwise convolutions of ONE input with k weight sets (the same-named DepSepConv candidates of the edges leaving
// a search-cell state): blockIdx.y = problem
struct DwTab {
    const float* a[SENAS_MAX_DWMULTI];      // per-problem operand (forward: the problem's own input or NULL; data gradient: dy_p)
    const float* w[SENAS_MAX_DWMULTI];
    float* out[SENAS_MAX_DWMULTI];
    double* stats[SENAS_MAX_DWMULTI];
};

// Two groups of problems may share a launch: problems [0, ka) have geometry ga (3x3 in the mixed launches), the rest gb
// (5x5) -- dep_sep_conv_3 and dep_sep_conv_5 of the same edges differ in nothing but the kernel size (and its padding).
template <bool TG>
__global__ __launch_bounds__(256) void dwconv_multi_fwd_kernel(GatherGeom ga, GatherGeom gb, int ka, const float* __restrict__ in, DwTab tab,
                                                               long total, int P) {
    const int p = (int)xcd_block().y;
    dwconv_body<TG, 4, false>(p < ka ? ga : gb, tab.a[p] ? tab.a[p] : in, tab.w[p], tab.out[p], 0, nullptr, tab.stats[p], total, P, Epi{});
}

template <int KSA, int KSB, int S>
__global__ __launch_bounds__(256) void dwconv_multi_fwd_x4_kernel(GatherGeom ga, GatherGeom gb, int ka, const float* __restrict__ in, DwTab tab,
                                                                  long total, int P) {
    const int p = (int)xcd_block().y;
    const float* src = tab.a[p] ? tab.a[p] : in;          // a problem may bring its own input (the two input states of a search cell)
    if (KSA == KSB || p < ka) dwconv_x4_body<KSA, S>(ga, src, tab.w[p], tab.out[p], 0, tab.stats[p], total, P);
    else dwconv_x4_body<KSB, S>(gb, src, tab.w[p], tab.out[p], 0, tab.stats[p], total, P);
}

// ConvTranspose2d(k, stride 2, padding k / 2, output_padding 1), depthwise: the UP candidates' dep_sep_conv_3 / _5
// (utils/operations.py:58-60,107-115).  One thread = the 2 x 2 output quad of one input-grid position, 4 channels: output
// (2 qy + py, 2 qx + px) takes the taps with ky = (py + pad) mod 2 (mod 2) from input row qy + (py + pad - ky) / 2, so the quad
// reads a 3 x 3 (5x5) or 2 x 2 (3x3) neighbourhood ONCE and every weight once -- the one-output-per-thread gather walked
// all k*k taps per output and threw three quarters of them away on the parity test.
template <int KS>
__device__ __forceinline__ void dwconv_t2_quad_body(const GatherGeom& g, const float* __restrict__ in, const float* __restrict__ w,
                                                    float* __restrict__ out, double* __restrict__ stats, long total, int P) {
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [tap][C]
    constexpr int TAPS = KS * KS, PAD = KS / 2;
    constexpr int LO = KS == 3 ? 0 : -1;                            // first neighbourhood row / column relative to (qy, qx)
    const int C = g.cout;
    for (int i = threadIdx.x; i < TAPS * C; i += 256) {
        const int t = i / C, cc = i - t * C;
        wl[i] = w[cc * TAPS + t];
    }
    __syncthreads();
    Stats4 acc_st;
    stats_init4(acc_st);
    const bool uniform = P > 0;
    const int chunks = uniform ? P : 1;
    const int cv = C >> 2;
    int n_blk = 0, c_thr = 0;
    for (int kk = 0; kk < chunks; ++kk) {
        long idx = ((long)xcd_block().x * chunks + kk) * 256 + threadIdx.x;
        const bool active = idx < total;
        if (!active) idx = total - 1;
        const int c = (int)(idx % cv) * 4;
        long r = idx / cv;
        const int qx = (int)(r % g.win);
        r /= g.win;
        const int qy = (int)(r % g.hin), n = (int)(r / g.hin);
        n_blk = n; c_thr = c;
        float4 v[2 - LO][2 - LO];
#pragma unroll
        for (int dy = LO; dy <= 1; ++dy)
#pragma unroll
            for (int dx = LO; dx <= 1; ++dx) {
                const int iy = qy + dy, ix = qx + dx;
                const bool ok = iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win;
                v[dy - LO][dx - LO] = ok ? *reinterpret_cast<const float4*>(in + ((size_t)(n * g.hin + iy) * g.win + ix) * C + c)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) {
                    if (((py + PAD - ky) & 1) != 0) continue;        // (compile-time)
                    const int dy = (py + PAD - ky) / 2;              // -1, 0, 1 (exact: the numerator is even)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        if (((px + PAD - kx) & 1) != 0) continue;
                        const int dx = (px + PAD - kx) / 2;
                        const float4 a = v[dy - LO][dx - LO];
                        const float4 wt = *reinterpret_cast<const float4*>(wl + (ky * KS + kx) * C + c);
                        acc[0] = fmaf(a.x, wt.x, acc[0]); acc[1] = fmaf(a.y, wt.y, acc[1]);
                        acc[2] = fmaf(a.z, wt.z, acc[2]); acc[3] = fmaf(a.w, wt.w, acc[3]);
                    }
                }
                if (active) stv<4>(out + (((size_t)n * g.hout + 2 * qy + py) * g.wout + 2 * qx + px) * C + c, acc);
                stats_accumulate4(acc_st, stats, uniform, n, C, c, acc, active);
            }
    }
    stats_flush4(acc_st, stats, uniform, n_blk, C, c_thr);
}

template <int KSA, int KSB>
__global__ __launch_bounds__(256) void dwconv_multi_fwd_t2_kernel(GatherGeom ga, GatherGeom gb, int ka, const float* __restrict__ in, DwTab tab,
                                                                  long total, int P) {
    const int p = (int)xcd_block().y;
    const float* src = tab.a[p] ? tab.a[p] : in;
    if (KSA == KSB || p < ka) dwconv_t2_quad_body<KSA>(ga, src, tab.w[p], tab.out[p], tab.stats[p], total, P);
    else dwconv_t2_quad_body<KSB>(gb, src, tab.w[p], tab.out[p], tab.stats[p], total, P);
}

static bool dw_t2_quad_ok(const GatherGeom& gg, int transposed) {
    return transposed && gg.stride == 2 && gg.dil == 1 && gg.kh == gg.kw && (gg.kh == 3 || gg.kh == 5) && gg.pad == gg.kh / 2 &&
           gg.cout % 4 == 0 && gg.cin == gg.cout && gg.hout == 2 * gg.hin && gg.wout == 2 * gg.win;
}

// data gradient of the same: dx = sum over problems of the (transposed / plain) gather of dy_p with w_p, one pass
template <bool TG>
__global__ __launch_bounds__(256) void dwconv_multi_dgrad_kernel(GatherGeom ga, GatherGeom gb, int ka, DwTab tab, int k, float* __restrict__ out,
                                                                 long total) {
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [problem][tap][C], group a first
    const int C = ga.cout, tapsa = ga.kh * ga.kw, tapsb = gb.kh * gb.kw;
    const int na = ka * tapsa * C, nb = (k - ka) * tapsb * C;
    for (int i = threadIdx.x; i < na; i += 256) {
        const int p = i / (tapsa * C), r = i - p * tapsa * C, t = r / C, cc = r - t * C;
        wl[i] = tab.w[p][cc * tapsa + t];
    }
    for (int i = threadIdx.x; i < nb; i += 256) {
        const int p = i / (tapsb * C), r = i - p * tapsb * C, t = r / C, cc = r - t * C;
        wl[na + i] = tab.w[ka + p][cc * tapsb + t];
    }
    __syncthreads();
    const long idx = (long)xcd_block().x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cv = C / 4;
    const int c = (int)(idx % cv) * 4;
    long pix = idx / cv;
    const int ox = (int)(pix % ga.wout);
    pix /= ga.wout;
    const int oy = (int)(pix % ga.hout), n = (int)(pix / ga.hout);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int grp = 0; grp < 2; ++grp) {
        const GatherGeom& g = grp == 0 ? ga : gb;
        const int p0 = grp == 0 ? 0 : ka, p1 = grp == 0 ? ka : k, taps = g.kh * g.kw;
        const float* wg = wl + (grp == 0 ? 0 : na);
        if (p0 == p1) continue;
        for (int ky = 0; ky < g.kh; ++ky) {
            int iy;
            if (!tap_src<TG>(g, oy, ky, g.hin, iy)) continue;
            for (int kx = 0; kx < g.kw; ++kx) {
                int ix;
                if (!tap_src<TG>(g, ox, kx, g.win, ix)) continue;
                const size_t off = ((size_t)(n * g.hin + iy) * g.win + ix) * g.cin + c;
                for (int p = p0; p < p1; ++p) {
                    float v[4], wt[4];
                    ldv<4>(tab.a[p] + off, v);
                    ldv<4>(wg + ((size_t)(p - p0) * taps + ky * g.kw + kx) * C + c, wt);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(v[j], wt[j], acc[j]);
                }
            }
        }
    }
    stv<4>(out + (((size_t)n * ga.hout + oy) * ga.wout + ox) * C + c, acc);
}

// The stride-2 transposed gather of the above (data gradient of the DOWN candidates' depthwise half) for SMALL launches.  Output
// (oy, ox) takes the taps with ky = (oy + pad) mod 2 (mod 2): at most SL = (KS + 1) / 2 per dimension.  The general kernel walks
// all k x k taps with a branch per tap and one problem per trip -- 25 dependent round trips for 3 + 3 problems, 21 us on an
// 8 x 8 map; here the SL x SL slots of three problems are requested together (clamped addresses, predicated FMAs).
template <int KS>
__device__ __forceinline__ void dw_dgrad_s2_accumulate(const GatherGeom& g, const DwTab& tab, int p0, int p1, const float* wg, int C,
                                                       int c, int n, int oy, int ox, float (&acc)[4]) {
    constexpr int SL = (KS + 1) / 2, PB = 3;
    int iy[SL], ix[SL], kys[SL], kxs[SL];
    bool oky[SL], okx[SL];
#pragma unroll
    for (int i = 0; i < SL; ++i) {
        const int ky = ((oy + g.pad) & 1) + 2 * i, ty = oy + g.pad - ky;
        const int kx = ((ox + g.pad) & 1) + 2 * i, tx = ox + g.pad - kx;
        oky[i] = ky < KS && ty >= 0 && (ty >> 1) < g.hin;
        okx[i] = kx < KS && tx >= 0 && (tx >> 1) < g.win;
        iy[i] = oky[i] ? ty >> 1 : 0; kys[i] = oky[i] ? ky : 0;
        ix[i] = okx[i] ? tx >> 1 : 0; kxs[i] = okx[i] ? kx : 0;
    }
    for (int pb = p0; pb < p1; pb += PB) {
        float v[PB][SL][SL][4];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const float* src = tab.a[pb + u < p1 ? pb + u : p1 - 1] + c;
#pragma unroll
            for (int i = 0; i < SL; ++i)
#pragma unroll
                for (int j = 0; j < SL; ++j) ldv<4>(src + ((size_t)(n * g.hin + iy[i]) * g.win + ix[j]) * g.cin, v[u][i][j]);
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const bool live = pb + u < p1;
            const float* wp = wg + (size_t)(live ? pb + u - p0 : 0) * KS * KS * C + c;
#pragma unroll
            for (int i = 0; i < SL; ++i)
#pragma unroll
                for (int j = 0; j < SL; ++j) {
                    float wt[4];
                    ldv<4>(wp + (size_t)(kys[i] * KS + kxs[j]) * C, wt);
                    const bool ok = live && oky[i] && okx[j];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = ok ? fmaf(v[u][i][j][q], wt[q], acc[q]) : acc[q];
                }
        }
    }
}

template <int KSA, int KSB>
__global__ __launch_bounds__(256) void dwconv_multi_dgrad_s2_kernel(GatherGeom ga, GatherGeom gb, int ka, DwTab tab, int k,
                                                                    float* __restrict__ out, long total) {
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [problem][tap][C], group a first
    const int C = ga.cout;
    const int na = ka * KSA * KSA * C, nb = (k - ka) * KSB * KSB * C;
    for (int i = threadIdx.x; i < na; i += 256) {
        const int p = i / (KSA * KSA * C), r = i - p * KSA * KSA * C, t = r / C, cc = r - t * C;
        wl[i] = tab.w[p][cc * KSA * KSA + t];
    }
    for (int i = threadIdx.x; i < nb; i += 256) {
        const int p = i / (KSB * KSB * C), r = i - p * KSB * KSB * C, t = r / C, cc = r - t * C;
        wl[na + i] = tab.w[ka + p][cc * KSB * KSB + t];
    }
    __syncthreads();
    const long idx = (long)xcd_block().x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cv = C / 4;
    const int c = (int)(idx % cv) * 4;
    long pix = idx / cv;
    const int ox = (int)(pix % ga.wout);
    pix /= ga.wout;
    const int oy = (int)(pix % ga.hout), n = (int)(pix / ga.hout);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    dw_dgrad_s2_accumulate<KSA>(ga, tab, 0, ka, wl, C, c, n, oy, ox, acc);
    if (k > ka) dw_dgrad_s2_accumulate<KSB>(gb, tab, ka, k, wl + na, C, c, n, oy, ox, acc);
    stv<4>(out + (((size_t)n * ga.hout + oy) * ga.wout + ox) * C + c, acc);
}

// depthwise weight gradient: dW[c][tap] = sum_{n,p} I[n, p*s-pad+k*d][c] * G[n,p][c]
// grid = pixel chunks; block = rows x C lanes (C <= 256); every thread keeps one partial per tap, so
// G is read once and the taps' I reads hit L1.
template <int KS>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(WgradGeom g, const float* __restrict__ I,
                                                           const float* __restrict__ G, float* __restrict__ dw,
                                                           int i_relu, int g_relu) {
    __shared__ float red[256];
    constexpr int TAPS = KS * KS;
    const int C = g.A;
    const int rows = 256 / C;
    const int c = threadIdx.x % C, row = threadIdx.x / C;
    const int per_img = g.hg * g.wg;
    const long total = (long)g.n * per_img;
    long p0 = (long)blockIdx.x * g.chunk, p1 = p0 + g.chunk;
    if (p1 > total) p1 = total;
    float acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = 0.f;
    if (row < rows) {
        for (long p = p0 + row; p < p1; p += rows) {
            const int n = (int)(p / per_img), r = (int)(p % per_img);
            const int gy = r / g.wg, gx = r % g.wg;
            float gv = G[(size_t)p * C + c];
            if (g_relu) gv = fmaxf(gv, 0.f);
            const float* In = I + (size_t)n * g.hi * g.wi * C + c;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int iy = gy * g.stride - g.pad + ky * g.dil;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int ix = gx * g.stride - g.pad + kx * g.dil;
                    const bool ok = iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi;
                    float iv = In[ok ? (size_t)(iy * g.wi + ix) * C : 0];
                    if (!ok) iv = 0.f;
                    if (i_relu) iv = fmaxf(iv, 0.f);
                    acc[ky * KS + kx] = fmaf(iv, gv, acc[ky * KS + kx]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        __syncthreads();
        red[threadIdx.x] = row < rows ? acc[t] : 0.f;
        __syncthreads();
        if (row == 0) {
            float v = acc[t];
            for (int r = 1; r < rows; ++r) v += red[r * C + c];
            atomicAdd(&dw[c * TAPS + t], v);
        }
    }
}

// Two-stage, atomic-free form for C % 4 == 0 (every depthwise conv on the real path):
// stage 1: thread = (pixel lane, 4-channel group); per-tap partial sums in registers, reduced over the
//          pixel lanes of the wave by shuffles and over the 4 waves through LDS -> part[block][c][tap]
// stage 2: dw[c][tap] = sum over blocks (fixed order: bitwise reproducible)
// the same sum over problems as a 4-columns-per-thread PLAIN gather (dilation 1): the data gradient of k transposed
// depthwise convolutions (gather over dy_p at the convolution's stride) or of k stride-1 ones (flip: kernels turned by 180 degrees)
// the taps of problems [p0, p1) (KS x KS, weights at wl: [problem][tap][C]) added to the 4 x 4 outputs of this thread
template <int KS, int S>
__device__ __forceinline__ void dw_dgrad_x4_accumulate(const GatherGeom& g, const DwTab& tab, int p0, int p1, const float* wl, int C, int c,
                                                       int n, int oy, int ox0, float (&acc)[4][4], int first = 0, int step = 1) {
    constexpr int TAPS = KS * KS, COLS = 3 * S + KS;
    const int ix0 = ox0 * S - g.pad;
    for (int p = p0 + first; p < p1; p += step) {
        const float* __restrict__ in = tab.a[p];
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int iy = oy * S - g.pad + ky;
            if (iy < 0 || iy >= g.hin) continue;
            const float* row = in + ((size_t)(n * g.hin + iy) * g.win) * g.cin + c;
            float4 col[COLS];
#pragma unroll
            for (int x = 0; x < COLS; ++x) {
                const int ix = ix0 + x;
                col[x] = (ix >= 0 && ix < g.win) ? *reinterpret_cast<const float4*>(row + (size_t)ix * g.cin) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const float4 wt = *reinterpret_cast<const float4*>(wl + ((size_t)(p - p0) * TAPS + ky * KS + kx) * C + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 v = col[j * S + kx];
                    acc[j][0] = fmaf(v.x, wt.x, acc[j][0]); acc[j][1] = fmaf(v.y, wt.y, acc[j][1]);
                    acc[j][2] = fmaf(v.z, wt.z, acc[j][2]); acc[j][3] = fmaf(v.w, wt.w, acc[j][3]);
                }
            }
        }
    }
}

// PS = 4 (small maps: the launch is a handful of blocks and each thread's walk over all problems and taps IS the launch
// time): four adjacent lanes share one output and deal the problems among themselves -- lane q takes problems q, q + 4, ..
// of the 3x3 group and, shifted by ka, of the 5x5 group -- then add up over the quad.
template <int KSA, int KSB, int S, int PS>
__global__ __launch_bounds__(256) void dwconv_multi_dgrad_x4_kernel(GatherGeom ga, GatherGeom gb, int ka, DwTab tab, int k, int flip,
                                                                    float* __restrict__ out, long total) {
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [problem][tap][C], group a first
    constexpr int TA = KSA * KSA, TB = KSB * KSB;
    const int C = ga.cout;
    const int na = ka * TA * C, nb = (k - ka) * TB * C;
    for (int i = threadIdx.x; i < na; i += 256) {
        const int p = i / (TA * C), r = i - p * TA * C, t = r / C, cc = r - t * C;
        wl[i] = tab.w[p][cc * TA + (flip ? TA - 1 - t : t)];
    }
    for (int i = threadIdx.x; i < nb; i += 256) {
        const int p = i / (TB * C), r = i - p * TB * C, t = r / C, cc = r - t * C;
        wl[na + i] = tab.w[ka + p][cc * TB + (flip ? TB - 1 - t : t)];
    }
    __syncthreads();
    const long idx = ((long)xcd_block().x * 256 + threadIdx.x) / PS;
    const int sp = PS > 1 ? (int)(threadIdx.x % PS) : 0;
    if (idx >= total) return;                                        // (the PS lanes of an output leave together)
    const int cv = C >> 2, wq = ga.wout >> 2;
    const int c = (int)(idx % cv) * 4;
    long r = idx / cv;
    const int ox0 = (int)(r % wq) * 4;
    r /= wq;
    const int oy = (int)(r % ga.hout), n = (int)(r / ga.hout);
    float acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][q] = 0.f;
    dw_dgrad_x4_accumulate<KSA, S>(ga, tab, 0, ka, wl, C, c, n, oy, ox0, acc, sp, PS);
    if (k > ka) dw_dgrad_x4_accumulate<KSB, S>(gb, tab, ka, k, wl + na, C, c, n, oy, ox0, acc, PS > 1 ? (sp + ka) % PS : 0, PS);
    if (PS > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = acc[j][q];
#pragma unroll
                for (int m = 1; m < PS; m <<= 1) v += __shfl_xor(v, m, 64);
                acc[j][q] = v;
            }
        if (sp != 0) return;
    }
    const size_t o = (((size_t)n * ga.hout + oy) * ga.wout + ox0) * C + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) stv<4>(out + o + (size_t)j * C, acc[j]);
}

template <int KS>
__device__ __forceinline__ void dwconv_wgrad_part_body(const WgradGeom& g, const float* __restrict__ I,
                                                       const float* __restrict__ G, float* __restrict__ part,
                                                       int i_relu, int g_relu) {
    constexpr int TAPS = KS * KS;
    extern __shared__ __attribute__((aligned(16))) float red[];     // [4 waves x row slots][c4 groups][TAPS][4]
    const int C = g.A, C4 = C >> 2;
    const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;         // consecutive threads -> consecutive 16-byte pieces
    const int lanes = 256 / C4;                                     // pixel lanes per block
    const int per_img = g.hg * g.wg;
    const long total = (long)g.n * per_img;
    const unsigned vbx = xcd_block().x;
    long p0 = (long)vbx * g.chunk, p1 = p0 + g.chunk;
    if (p1 > total) p1 = total;
    float acc[TAPS][4];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
    SENAS_PHASE(0);
    if (pl < lanes) {
        for (long p = p0 + pl; p < p1; p += lanes) {
            const int n = (int)(p / per_img), r = (int)(p % per_img);
            const int gy = r / g.wg, gx = r % g.wg;
            float gv[4];
            ldv<4>(G + (size_t)p * C + c4 * 4, gv);
            if (g_relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) gv[j] = fmaxf(gv[j], 0.f);
            }
            const float* In = I + (size_t)n * g.hi * g.wi * C + c4 * 4;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int iy = gy * g.stride - g.pad + ky * g.dil;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int ix = gx * g.stride - g.pad + kx * g.dil;
                    const bool ok = iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi;
                    float iv[4];
                    ldv<4>(In + (ok ? (size_t)(iy * g.wi + ix) * C : 0), iv);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = ok ? iv[j] : 0.f;
                        if (i_relu) v = fmaxf(v, 0.f);
                        acc[ky * KS + kx][j] = fmaf(v, gv[j], acc[ky * KS + kx][j]);
                    }
                }
            }
        }
    }
    SENAS_PHASE(1);
    // fold the pixel lanes that share c4: inside a 16-lane row with DPP (VALU speed), the rows and the 4 waves through LDS
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = row_strided_sum(acc[t][j], C4);
    SENAS_PHASE(2);
    const int slots = wave_slots(C4), nparts = 4 * slots;
    if (lane_holds_partial(lane, C4)) {
        float* dst = red + (size_t)((wave * slots + lane_slot(lane, C4)) * C4 + c4) * TAPS * 4;
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[t * 4 + j] = acc[t][j];
    }
    __syncthreads();
    SENAS_PHASE(3);
    for (int i = threadIdx.x; i < C4 * TAPS * 4; i += 256) {
        const int j = i & 3, t = (i >> 2) % TAPS, cg = (i >> 2) / TAPS;
        float v = 0.f;
        for (int w = 0; w < nparts; ++w) v += red[((size_t)(w * C4 + cg) * TAPS + t) * 4 + j];
        part[((size_t)vbx * C + cg * 4 + j) * TAPS + t] = v;
    }
    SENAS_PHASE(4);
}

template <int KS>
__global__ __launch_bounds__(256) void dwconv_wgrad_part_kernel(WgradGeom g, const float* __restrict__ I,
                                                                const float* __restrict__ G, float* __restrict__ part,
                                                                int i_relu, int g_relu) {
    dwconv_wgrad_part_body<KS>(g, I, G, part, i_relu, g_relu);
}

// k problems that share one side (x) and differ in the other (dy_p): blockIdx.y = problem
struct DwWgradTab {
    const float* I[SENAS_MAX_DWMULTI];
    const float* G[SENAS_MAX_DWMULTI];
    float* part[SENAS_MAX_DWMULTI];
    float* dw[SENAS_MAX_DWMULTI];
};

template <int KSA, int KSB>
__global__ __launch_bounds__(256) void dwconv_wgrad_part_multi_kernel(WgradGeom ga, WgradGeom gb, int ka, DwWgradTab tab) {
    const int p = (int)xcd_block().y;
    if (KSA == KSB || p < ka) dwconv_wgrad_part_body<KSA>(ga, tab.I[p], tab.G[p], tab.part[p], 0, 0);
    else dwconv_wgrad_part_body<KSB>(gb, tab.I[p], tab.G[p], tab.part[p], 0, 0);
}

// one wave per output element: lanes stride over the blocks' partials, fixed-order shuffle tree at the end
__global__ __launch_bounds__(256) void dwconv_wgrad_sum_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                               int n_elem, int n_blocks) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n_elem) return;
    float v = 0.f;
    for (int b = lane; b < n_blocks; b += 64) v += part[(size_t)b * n_elem + i];
    v = wave_sum(v);
    if (lane == 0) dw[i] = v;
}

__global__ __launch_bounds__(256) void dwconv_wgrad_sum_multi_kernel(DwWgradTab tab, int n_elem, int n_blocks) {
    const float* __restrict__ part = tab.part[blockIdx.y];
    float* __restrict__ dw = tab.dw[blockIdx.y];
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n_elem) return;
    float v = 0.f;
    for (int b = lane; b < n_blocks; b += 64) v += part[(size_t)b * n_elem + i];
    v = wave_sum(v);
    if (lane == 0) dw[i] = v;
}

SENAS_PHASE_READER(conv)

static bool geom_ok(const senas_conv_geom* g) {
    if (!g || g->n <= 0 || g->ci <= 0 || g->co <= 0 || g->kh <= 0 || g->kw <= 0) return false;
    if (g->stride != 1 && g->stride != 2) return false;
    if (g->groups != 1 && !(g->groups == g->ci && g->ci == g->co)) return false;
    int ho, wo;
    if (!g->transposed) {
        ho = (g->hi + 2 * g->pad - g->dil * (g->kh - 1) - 1) / g->stride + 1;
        wo = (g->wi + 2 * g->pad - g->dil * (g->kw - 1) - 1) / g->stride + 1;
        return ho == g->ho && wo == g->wo;
    }
    // transposed: ho = (hi-1)*s - 2p + d(k-1) + output_padding + 1, output_padding in [0, s)
    int base = (g->hi - 1) * g->stride - 2 * g->pad + g->dil * (g->kh - 1) + 1;
    int basw = (g->wi - 1) * g->stride - 2 * g->pad + g->dil * (g->kw - 1) + 1;
    return g->ho >= base && g->ho < base + g->stride && g->wo >= basw && g->wo < basw + g->stride;
}

}  // namespace senas

using namespace senas;

extern "C" int64_t senas_conv2d_ws_bytes(const senas_conv_geom* g) {
    if (!g) return 0;
    if (g->groups != 1) return (int64_t)2048 * g->ci * g->kh * g->kw * sizeof(float) + 256;  // per-block wgrad partials
    // repacked weights: [n-tile][tap][reduction channels][32] for the MFMA kernels (either direction)
    const int64_t big = g->ci > g->co ? g->ci : g->co, small = g->ci > g->co ? g->co : g->ci;
    const int64_t cols = ((small + 31) / 32) * 32 > ((big + 31) / 32) * 32 ? ((small + 31) / 32) * 32 : ((big + 31) / 32) * 32;
    // (x 3/2: the three-plane bf16 image of the split-operand convolutions, conv_bf.hip, is 6 bytes per weight)
    return (int64_t)g->kh * g->kw * big * cols * 6 + 256;
}

// Workspace of the weight gradient: which path it takes decides the size and whether it must arrive zero-filled.
extern "C" int senas_conv2d_bwd_weight_ws(const senas_conv_geom* g, int64_t* bytes, int32_t* needs_zero) {
    SENAS_REQUIRE(geom_ok(g) && bytes && needs_zero, "conv2d_bwd_weight_ws: bad argument");
    *bytes = senas_conv2d_ws_bytes(g);
    *needs_zero = 0;
    if (g->groups != 1) return SENAS_OK;                           // per-block partials, overwritten
    WgradGeom wg = !g->transposed ? WgradGeom{g->n, g->ho, g->wo, g->co, g->hi, g->wi, g->ci, g->kh, g->kw, g->stride, g->pad, g->dil, 0}
                                  : WgradGeom{g->n, g->hi, g->wi, g->ci, g->ho, g->wo, g->co, g->kh, g->kw, g->stride, g->pad, g->dil, 0};
    if (!g->transposed && stem_wgrad_ok(wg)) { *bytes = stem_wgrad_ws_bytes(wg) + 256; return SENAS_OK; }
    if (thin_n_wgrad_ok(wg)) { *bytes = thin_n_wgrad_ws_bytes(wg) + 256; return SENAS_OK; }
    if (!g->transposed && c8_mfma_wgrad_ok(wg)) { *bytes = c8_mfma_wgrad_ws_bytes(wg) + 256; return SENAS_OK; }
    if (wgrad_c8_ok(wg)) { *bytes = wgrad_c8_ws_bytes(wg) + 256; return SENAS_OK; }
    if (lds_wgrad_ok(wg)) { *bytes = lds_wgrad_ws_bytes(wg) + 256; return SENAS_OK; }      // (a superset of the launcher's condition)
    if (mfma_wgrad_ok(wg)) *needs_zero = 1;                        // split-K image accumulated with atomics
    return SENAS_OK;
}

// the 4-columns-per-thread depthwise form: plain gather (Conv2d forward), dilation 1, 3x3 / 5x5, c % 4 == 0, output width % 4 == 0
static bool dw_x4_ok(const GatherGeom& gg, int transposed) {
    return !transposed && gg.dil == 1 && gg.kh == gg.kw && (gg.kh == 3 || gg.kh == 5) && gg.cout % 4 == 0 && gg.cin == gg.cout &&
           gg.wout % 4 == 0 && (gg.stride == 1 || gg.stride == 2);
}

// forward: Conv2d -> plain gather over x; ConvTranspose2d -> transposed gather over x
// plane != 0: the output in planar 8-channel groups (common.h GatherGeom::oplane) -- only where the geometry lands on a kernel that
// has that epilogue (the LDS-window stride-1 kernel, the stride-2 transposed one); SENAS_EUNSUPPORTED, nothing launched, otherwise
static int conv2d_fwd_impl(const senas_conv_geom* g, const float* x, const float* w, float* y, int64_t plane, int in_relu,
                           double* stats, void* ws, const float* packed, void* stream) {
    SENAS_REQUIRE(geom_ok(g), "conv2d_fwd: inconsistent geometry");
    SENAS_REQUIRE(x && w && y, "conv2d_fwd: null pointer");
    hipStream_t st = as_stream(stream);
    GatherGeom gg{g->n, g->hi, g->wi, g->ci, g->ho, g->wo, g->co, g->kh, g->kw, g->stride, g->pad, g->dil};
    if (plane != 0) {
        SENAS_REQUIRE(plane == (int64_t)g->n * g->ho * g->wo * 8 && g->co % 8 == 0, "conv2d_fwd_planar: the plane is n * h * w * 8 floats, c_out a multiple of 8");
        const bool lds = !g->transposed && lds_gather_ok(gg), t2 = g->transposed && !in_relu && t2_lds_ok(gg);
        const bool c8 = g->groups == 1 && !g->transposed && !(stem_mfma_ok(gg)) && !in_relu && c8_mfma_ok(gg);      // (the 8 -> 16 stacks of the inner edges)
        const bool elsewhere = g->groups != 1 || (!g->transposed && stem_mfma_ok(gg)) || (!c8 && thin_k_ok(gg)) ||
                               (!c8 && thin_n_ok(gg) && (gg.cout <= 4 || g->transposed || !lds_gather_ok(gg)));
        if (elsewhere || !(c8 || lds || t2)) return SENAS_EUNSUPPORTED;
        gg.oplane = (long)plane;
    }
    if (g->groups != 1) {
        const int V = (g->co % 4 == 0) ? 4 : 1;
        const size_t dw_lds = (size_t)g->kh * g->kw * g->co * sizeof(float);
        if (!in_relu && dw_x4_ok(gg, g->transposed)) {              // 4 output columns per thread (L1 requests / 2.5)
            const long per_img4 = (long)g->ho * (g->wo / 4) * (g->co / 4), total4 = per_img4 * g->n;
            const int P4 = stats != nullptr ? stats_chunks_per_block(per_img4, g->co, total4) : 0;
            dim3 grid4((unsigned)((total4 + 256L * (P4 > 0 ? P4 : 1) - 1) / (256L * (P4 > 0 ? P4 : 1))));
#define SENAS_X4(KS_, S_) hipLaunchKernelGGL((dwconv_x4_kernel<KS_, S_>), grid4, dim3(256), dw_lds, st, gg, x, w, y, 0, stats, total4, P4)
            if (g->kh == 3) { if (g->stride == 1) SENAS_X4(3, 1); else SENAS_X4(3, 2); }
            else { if (g->stride == 1) SENAS_X4(5, 1); else SENAS_X4(5, 2); }
#undef SENAS_X4
            return launch_status("dwconv_fwd (x4)");
        }
        const long per_img = (long)g->ho * g->wo * (g->co / V);
        long total = per_img * g->n;
        const int P = (V == 4 && stats != nullptr) ? stats_chunks_per_block(per_img, g->co, total) : 0;
        dim3 grid((unsigned)((total + 256L * (P > 0 ? P : 1) - 1) / (256L * (P > 0 ? P : 1))));
        if (g->transposed) {
            if (V == 4) hipLaunchKernelGGL((dwconv_kernel<true, 4>), grid, dim3(256), dw_lds, st, gg, x, w, y, in_relu, (const float*)nullptr, stats, total, P);
            else hipLaunchKernelGGL((dwconv_kernel<true, 1>), grid, dim3(256), dw_lds, st, gg, x, w, y, in_relu, (const float*)nullptr, stats, total, P);
        } else {
            if (V == 4) hipLaunchKernelGGL((dwconv_kernel<false, 4>), grid, dim3(256), dw_lds, st, gg, x, w, y, in_relu, (const float*)nullptr, stats, total, P);
            else hipLaunchKernelGGL((dwconv_kernel<false, 1>), grid, dim3(256), dw_lds, st, gg, x, w, y, in_relu, (const float*)nullptr, stats, total, P);
        }
        return launch_status("dwconv_fwd");
    }
    // the stem (1..4 input channels, stride 1): (tap, channel) pairs on the K axis of the fp32 MFMA
    if (!g->transposed && stem_mfma_ok(gg)) return launch_stem_mfma(gg, x, w, y, in_relu, stats, st);
    // the search cell's 8-channel inner edges (5x5 dilated, 8 -> 8 / 16): 16 x 16 x 4 MFMA tiles
    if (!g->transposed && !in_relu && c8_mfma_ok(gg)) return launch_c8_mfma(gg, x, w, g->ci, 1, 0, y, stats, st);
    // thin shapes (stem, head): single-pass HBM-bound kernels that read the torch-layout weights directly
    if (thin_k_ok(gg)) {
        if (!g->transposed && thin_k4_ok(gg)) return launch_thin_k4(gg, x, w, g->ci, 1, 0, y, in_relu, stats, st);
        if (!g->transposed) return launch_thin_k<false>(gg, x, w, g->ci, 1, y, in_relu, nullptr, stats, st);
        return launch_thin_k<true>(gg, x, w, g->co, 0, y, in_relu, nullptr, stats, st);
    }
    // 5..8 outputs: only where the alternative is the direct-global MFMA kernel (strided / transposed); the LDS window
    // kernel is faster on stride-1 shapes even at 1/4 tile occupancy (measured on the supernet step)
    if (thin_n_ok(gg) && (gg.cout <= 4 || g->transposed || !lds_gather_ok(gg))) {
        if (!g->transposed) return launch_thin_n<false>(gg, x, w, g->ci, 1, y, in_relu, stats, st);
        return launch_thin_n<true>(gg, x, w, g->co, 0, y, in_relu, stats, st);
    }
    SENAS_REQUIRE(ws, "conv2d_fwd: null workspace");
    float* wp = reinterpret_cast<float*>(ws);
    const int taps = g->kh * g->kw, total = taps * g->ci * g->co;
    // MFMA paths read the fragment image; `packed` (if given) is that image, refreshed by the caller
    // once per step (senas_pack_batched) instead of once per launch
    const bool use_lds = !g->transposed && lds_gather_ok(gg);
    const bool use_s2 = !g->transposed && lds_gather_s2_ok(gg);
    const bool use_t2 = g->transposed && !in_relu && t2_lds_ok(gg);          // ConvTranspose2d stride 2: the four output phases share one LDS window
    if (use_lds || use_s2 || use_t2 || mfma_gather_ok(gg, g->transposed != 0)) {
        const float* img = packed;
        if (img == nullptr) {
            if (!g->transposed) launch_pack_mfma(w, wp, g->co, g->ci, taps, 1, st);
            else launch_pack_mfma(w, wp, g->ci, g->co, taps, 0, st);
            img = wp;
        }
        if (use_lds) return launch_lds_gather<false>(gg, x, img, y, in_relu, nullptr, stats, st);
        if (use_s2) return launch_lds_gather_s2(gg, x, img, y, in_relu, nullptr, stats, st);
        if (use_t2) return launch_t2_lds(gg, x, img, y, stats, st);
        if (!g->transposed) return launch_mfma_gather<false>(gg, x, img, y, in_relu, nullptr, stats, st);
        return launch_mfma_gather<true>(gg, x, img, y, in_relu, nullptr, stats, st);
    }
    // Conv2d w[co][ci][tap] -> wp[tap][ci][co] (swap); ConvTranspose2d w[ci][co][tap] -> wp[tap][ci][co]
    if (!g->transposed) hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, wp, g->co, g->ci, taps, 1);
    else hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, wp, g->ci, g->co, taps, 0);
    if (!g->transposed) return launch_direct<false>(gg, x, wp, y, in_relu, nullptr, stats, st);
    return launch_direct<true>(gg, x, wp, y, in_relu, nullptr, stats, st);
}

extern "C" int senas_conv2d_fwd(const senas_conv_geom* g, const float* x, const float* w, float* y, int in_relu,
                                double* stats, void* ws, const float* packed, void* stream) {
    return conv2d_fwd_impl(g, x, w, y, 0, in_relu, stats, ws, packed, stream);
}

extern "C" int senas_conv2d_fwd_planar(const senas_conv_geom* g, const float* x, const float* w, float* y, int64_t y_plane, int in_relu,
                                       double* stats, void* ws, const float* packed, void* stream) {
    if (y_plane == 0) return SENAS_EUNSUPPORTED;
    return conv2d_fwd_impl(g, x, w, y, y_plane, in_relu, stats, ws, packed, stream);
}

// forward with the inference epilogue; SENAS_EUNSUPPORTED (nothing launched) when the geometry is not on a kernel
// that has one -- the caller then runs the plain forward and the fused node pass
extern "C" int senas_conv2d_fwd_epilogue(const senas_conv_geom* g, const float* x, const float* w, float* y, int in_relu,
                                         const senas_conv_epilogue* e, void* ws, const float* packed, void* stream) {
    SENAS_REQUIRE(geom_ok(g), "conv2d_fwd_epilogue: inconsistent geometry");
    SENAS_REQUIRE(x && w && y && e && e->scale && e->bias, "conv2d_fwd_epilogue: null pointer");
    hipStream_t st = as_stream(stream);
    GatherGeom gg{g->n, g->hi, g->wi, g->ci, g->ho, g->wo, g->co, g->kh, g->kw, g->stride, g->pad, g->dil};
    const Epi epi{e->scale, e->bias, e->addend, e->add_scale, e->relu};
    if (g->groups != 1) {
        if (g->co % 4 != 0 || e->addend != nullptr) return SENAS_EUNSUPPORTED;
        const size_t dw_lds = (size_t)g->kh * g->kw * g->co * sizeof(float);
        const long total = (long)g->n * g->ho * g->wo * (g->co / 4);
        dim3 grid((unsigned)((total + 255) / 256));
        if (g->transposed) hipLaunchKernelGGL((dwconv_kernel<true, 4, true>), grid, dim3(256), dw_lds, st, gg, x, w, y, in_relu, (const float*)nullptr, (double*)nullptr, total, 0, epi);
        else hipLaunchKernelGGL((dwconv_kernel<false, 4, true>), grid, dim3(256), dw_lds, st, gg, x, w, y, in_relu, (const float*)nullptr, (double*)nullptr, total, 0, epi);
        return launch_status("dwconv_fwd_epilogue");
    }
    if (g->transposed || thin_k_ok(gg) || !lds_gather_ok(gg)) return SENAS_EUNSUPPORTED;
    if (thin_n_ok(gg) && gg.cout <= 4) return SENAS_EUNSUPPORTED;
    const float* img = packed;
    if (img == nullptr) {
        SENAS_REQUIRE(ws, "conv2d_fwd_epilogue: null workspace");
        launch_pack_mfma(w, reinterpret_cast<float*>(ws), g->co, g->ci, g->kh * g->kw, 1, st);
        img = reinterpret_cast<float*>(ws);
    }
    return launch_lds_gather_epi(gg, x, img, y, in_relu, epi, st);
}

// data gradient: Conv2d -> transposed gather over dy; ConvTranspose2d -> plain gather over dy
extern "C" int senas_conv2d_bwd_data(const senas_conv_geom* g, const float* dy, const float* w, float* dx, int in_relu,
                                     const float* x, void* ws, const float* packed, void* stream) {
    SENAS_REQUIRE(geom_ok(g), "conv2d_bwd_data: inconsistent geometry");
    SENAS_REQUIRE(dy && w && dx, "conv2d_bwd_data: null pointer");
    SENAS_REQUIRE(!in_relu || x, "conv2d_bwd_data: in_relu needs x");
    hipStream_t st = as_stream(stream);
    const float* mask = in_relu ? x : nullptr;
    GatherGeom gg{g->n, g->ho, g->wo, g->co, g->hi, g->wi, g->ci, g->kh, g->kw, g->stride, g->pad, g->dil};
    if (g->groups != 1) {
        const int V = (g->ci % 4 == 0) ? 4 : 1;
        const size_t dw_lds = (size_t)g->kh * g->kw * g->ci * sizeof(float);
        // the 4-columns-per-thread plain gather where the data gradient is one: ConvTranspose2d (gather over dy at the conv's
        // stride), and a stride-1 "same" Conv2d (the same gather with the kernel turned by 180 degrees)
        const bool flip = !g->transposed && g->stride == 1 && g->hi == g->ho && g->wi == g->wo && g->pad == g->dil * (g->kh / 2);
        if (mask == nullptr && (g->transposed || flip) && dw_x4_ok(gg, 0)) {
            const long total4 = (long)g->n * g->hi * (g->wi / 4) * (g->ci / 4);
            dim3 grid4((unsigned)((total4 + 255) / 256));
#define SENAS_X4(KS_, S_) hipLaunchKernelGGL((dwconv_x4_kernel<KS_, S_>), grid4, dim3(256), dw_lds, st, gg, dy, w, dx, flip ? 1 : 0, (double*)nullptr, total4, 0)
            if (g->kh == 3) { if (gg.stride == 1) SENAS_X4(3, 1); else SENAS_X4(3, 2); }
            else { if (gg.stride == 1) SENAS_X4(5, 1); else SENAS_X4(5, 2); }
#undef SENAS_X4
            return launch_status("dwconv_bwd_data (x4)");
        }
        long total = (long)g->n * g->hi * g->wi * (g->ci / V);
        dim3 grid((unsigned)((total + 255) / 256));
        if (!g->transposed) {
            if (V == 4) hipLaunchKernelGGL((dwconv_kernel<true, 4>), grid, dim3(256), dw_lds, st, gg, dy, w, dx, 0, mask, (double*)nullptr, total, 0);
            else hipLaunchKernelGGL((dwconv_kernel<true, 1>), grid, dim3(256), dw_lds, st, gg, dy, w, dx, 0, mask, (double*)nullptr, total, 0);
        } else {
            if (V == 4) hipLaunchKernelGGL((dwconv_kernel<false, 4>), grid, dim3(256), dw_lds, st, gg, dy, w, dx, 0, mask, (double*)nullptr, total, 0);
            else hipLaunchKernelGGL((dwconv_kernel<false, 1>), grid, dim3(256), dw_lds, st, gg, dy, w, dx, 0, mask, (double*)nullptr, total, 0);
        }
        return launch_status("dwconv_bwd_data");
    }
    // the data gradient of an 8-channel inner-edge convolution (8 or 16 stacked outputs -> 8): the same gather, taps mirrored
    if (!g->transposed && mask == nullptr && c8_mfma_ok(gg)) return launch_c8_mfma(gg, dy, w, g->ci, 0, 1, dx, nullptr, st);
    if (thin_k_ok(gg)) {
        // a stride-1 "same" Conv2d: its data gradient is the plain gather over dy with the kernel turned by 180 degrees
        if (!g->transposed && mask == nullptr && thin_k4_ok(gg)) return launch_thin_k4(gg, dy, w, g->ci, 0, 1, dx, 0, nullptr, st);
        if (!g->transposed) return launch_thin_k<true>(gg, dy, w, g->ci, 0, dx, 0, mask, nullptr, st);
        return launch_thin_k<false>(gg, dy, w, g->co, 1, dx, 0, mask, nullptr, st);
    }
    SENAS_REQUIRE(ws, "conv2d_bwd_data: null workspace");
    float* wp = reinterpret_cast<float*>(ws);
    const int taps = g->kh * g->kw, total = taps * g->ci * g->co;
    const bool use_lds = !g->transposed && lds_gather_ok(gg);
    const bool use_s2 = g->transposed && lds_gather_s2_ok(gg);          // ConvTranspose2d: dx is a stride-2 plain gather over dy
    const bool use_t2 = !g->transposed && mask == nullptr && t2_lds_ok(gg);      // stride-2 Conv2d: dx is a transposed gather over dy
    if (use_lds || use_s2 || use_t2 || mfma_gather_ok(gg, g->transposed == 0)) {
        const float* img = packed;
        if (img == nullptr) {
            if (!g->transposed) launch_pack_mfma(w, wp, g->co, g->ci, taps, 0, st);
            else launch_pack_mfma(w, wp, g->ci, g->co, taps, 1, st);
            img = wp;
        }
        if (use_lds) return launch_lds_gather<true>(gg, dy, img, dx, 0, mask, nullptr, st);
        if (use_s2) return launch_lds_gather_s2(gg, dy, img, dx, 0, mask, nullptr, st);
        if (use_t2) return launch_t2_lds(gg, dy, img, dx, nullptr, st);
        if (!g->transposed) return launch_mfma_gather<true>(gg, dy, img, dx, 0, mask, nullptr, st);
        return launch_mfma_gather<false>(gg, dy, img, dx, 0, mask, nullptr, st);
    }
    // wp[tap][a = co][b = ci]
    if (!g->transposed) hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, wp, g->co, g->ci, taps, 0);
    else hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, wp, g->ci, g->co, taps, 1);
    if (!g->transposed) return launch_direct<true>(gg, dy, wp, dx, 0, mask, nullptr, st);
    return launch_direct<false>(gg, dy, wp, dx, 0, mask, nullptr, st);
}

// ---- two convolutions of one launch: same tensor shapes, kernel size and stride, different dilation (the dil_3_conv_5 and
// dil_2_conv_5 candidates of the same edges, utils/operations.py:69-72).  SENAS_EUNSUPPORTED (nothing launched) unless both
// take the same kernel of the conv_c8 / conv_lds family; the caller then makes the two single calls.
static bool pair_geoms_ok(const senas_conv_geom* a, const senas_conv_geom* b) {
    return geom_ok(a) && geom_ok(b) && a->n == b->n && a->hi == b->hi && a->wi == b->wi && a->ci == b->ci && a->ho == b->ho && a->wo == b->wo &&
           a->co == b->co && a->kh == b->kh && a->kw == b->kw && a->stride == b->stride && !a->transposed && !b->transposed &&
           a->groups == 1 && b->groups == 1;
}

static int conv2d_fwd_pair_impl(const senas_conv_geom* ga, const senas_conv_geom* gb, const float* x, const float* wa, const float* wb,
                                float* ya, float* yb, int64_t plane, int in_relu, double* stats_a, double* stats_b, void* ws_a, void* ws_b,
                                const float* packed_a, const float* packed_b, void* stream) {
    if (!ga || !gb || !pair_geoms_ok(ga, gb)) return SENAS_EUNSUPPORTED;
    SENAS_REQUIRE(x && wa && wb && ya && yb, "conv2d_fwd_pair: null pointer");
    hipStream_t st = as_stream(stream);
    GatherGeom g1{ga->n, ga->hi, ga->wi, ga->ci, ga->ho, ga->wo, ga->co, ga->kh, ga->kw, ga->stride, ga->pad, ga->dil};
    const GatherGeom g2{gb->n, gb->hi, gb->wi, gb->ci, gb->ho, gb->wo, gb->co, gb->kh, gb->kw, gb->stride, gb->pad, gb->dil};
    if (stem_mfma_ok(g1) || stem_mfma_ok(g2)) return SENAS_EUNSUPPORTED;
    const bool c8a = !in_relu && c8_mfma_ok(g1), c8b = !in_relu && c8_mfma_ok(g2);
    if (c8a != c8b) return SENAS_EUNSUPPORTED;
    if (plane != 0) {                                    // planar groups: the stride-1 LDS-window kernel only (see conv2d_fwd_impl)
        SENAS_REQUIRE(plane == (int64_t)ga->n * ga->ho * ga->wo * 8 && ga->co % 8 == 0, "conv2d_fwd_pair_planar: the plane is n * h * w * 8 floats, c_out a multiple of 8");
        if (!c8a && (thin_k_ok(g1) || thin_k_ok(g2) || !(lds_gather_ok(g1) && lds_gather_ok(g2)) || (thin_n_ok(g1) && g1.cout <= 4))) return SENAS_EUNSUPPORTED;
        g1.oplane = (long)plane;
    }
    if (c8a) return launch_c8_mfma(g1, x, wa, ga->ci, 1, 0, ya, stats_a, st, Pair2{x, wb, yb, nullptr, stats_b, g2.dil, g2.pad, 1});
    if (thin_k_ok(g1) || thin_k_ok(g2)) return SENAS_EUNSUPPORTED;
    const bool lds = lds_gather_ok(g1) && lds_gather_ok(g2), s2 = lds_gather_s2_ok(g1) && lds_gather_s2_ok(g2);
    if (!lds && !s2) return SENAS_EUNSUPPORTED;
    if (thin_n_ok(g1) && (g1.cout <= 4 || !lds_gather_ok(g1))) return SENAS_EUNSUPPORTED;
    const int taps = ga->kh * ga->kw;
    const float* ia = packed_a;
    const float* ib = packed_b;
    if (ia == nullptr) { SENAS_REQUIRE(ws_a, "conv2d_fwd_pair: null workspace"); launch_pack_mfma(wa, reinterpret_cast<float*>(ws_a), ga->co, ga->ci, taps, 1, st); ia = reinterpret_cast<float*>(ws_a); }
    if (ib == nullptr) { SENAS_REQUIRE(ws_b, "conv2d_fwd_pair: null workspace"); launch_pack_mfma(wb, reinterpret_cast<float*>(ws_b), gb->co, gb->ci, taps, 1, st); ib = reinterpret_cast<float*>(ws_b); }
    const Pair2 pr{x, ib, yb, nullptr, stats_b, g2.dil, g2.pad, 1};
    if (lds) return launch_lds_gather<false>(g1, x, ia, ya, in_relu, nullptr, stats_a, st, pr);
    return launch_lds_gather_s2(g1, x, ia, ya, in_relu, nullptr, stats_a, st, pr);
}

extern "C" int senas_conv2d_fwd_pair(const senas_conv_geom* ga, const senas_conv_geom* gb, const float* x, const float* wa, const float* wb,
                                     float* ya, float* yb, int in_relu, double* stats_a, double* stats_b, void* ws_a, void* ws_b,
                                     const float* packed_a, const float* packed_b, void* stream) {
    return conv2d_fwd_pair_impl(ga, gb, x, wa, wb, ya, yb, 0, in_relu, stats_a, stats_b, ws_a, ws_b, packed_a, packed_b, stream);
}

extern "C" int senas_conv2d_fwd_pair_planar(const senas_conv_geom* ga, const senas_conv_geom* gb, const float* x, const float* wa, const float* wb,
                                            float* ya, float* yb, int64_t y_plane, int in_relu, double* stats_a, double* stats_b, void* ws_a,
                                            void* ws_b, const float* packed_a, const float* packed_b, void* stream) {
    if (y_plane == 0) return SENAS_EUNSUPPORTED;
    return conv2d_fwd_pair_impl(ga, gb, x, wa, wb, ya, yb, y_plane, in_relu, stats_a, stats_b, ws_a, ws_b, packed_a, packed_b, stream);
}

extern "C" int senas_conv2d_bwd_data_pair(const senas_conv_geom* ga, const senas_conv_geom* gb, const float* dya, const float* dyb,
                                          const float* wa, const float* wb, float* dxa, float* dxb, int in_relu, const float* x,
                                          void* ws_a, void* ws_b, const float* packed_a, const float* packed_b, void* stream) {
    if (!ga || !gb || !pair_geoms_ok(ga, gb)) return SENAS_EUNSUPPORTED;
    SENAS_REQUIRE(dya && dyb && wa && wb && dxa && dxb && (!in_relu || x), "conv2d_bwd_data_pair: null pointer");
    hipStream_t st = as_stream(stream);
    const float* mask = in_relu ? x : nullptr;
    const GatherGeom g1{ga->n, ga->ho, ga->wo, ga->co, ga->hi, ga->wi, ga->ci, ga->kh, ga->kw, ga->stride, ga->pad, ga->dil};
    const GatherGeom g2{gb->n, gb->ho, gb->wo, gb->co, gb->hi, gb->wi, gb->ci, gb->kh, gb->kw, gb->stride, gb->pad, gb->dil};
    const bool c8a = mask == nullptr && c8_mfma_ok(g1), c8b = mask == nullptr && c8_mfma_ok(g2);
    if (c8a != c8b) return SENAS_EUNSUPPORTED;
    if (c8a) return launch_c8_mfma(g1, dya, wa, ga->ci, 0, 1, dxa, nullptr, st, Pair2{dyb, wb, dxb, nullptr, nullptr, g2.dil, g2.pad, 1});
    if (thin_k_ok(g1) || thin_k_ok(g2)) return SENAS_EUNSUPPORTED;
    if (!(lds_gather_ok(g1) && lds_gather_ok(g2))) return SENAS_EUNSUPPORTED;      // (stride 2: the transposed gather of conv_t2 has no pair form)
    const int taps = ga->kh * ga->kw;
    const float* ia = packed_a;
    const float* ib = packed_b;
    if (ia == nullptr) { SENAS_REQUIRE(ws_a, "conv2d_bwd_data_pair: null workspace"); launch_pack_mfma(wa, reinterpret_cast<float*>(ws_a), ga->co, ga->ci, taps, 0, st); ia = reinterpret_cast<float*>(ws_a); }
    if (ib == nullptr) { SENAS_REQUIRE(ws_b, "conv2d_bwd_data_pair: null workspace"); launch_pack_mfma(wb, reinterpret_cast<float*>(ws_b), gb->co, gb->ci, taps, 0, st); ib = reinterpret_cast<float*>(ws_b); }
    return launch_lds_gather<true>(g1, dya, ia, dxa, 0, mask, nullptr, st, Pair2{dyb, ib, dxb, mask, nullptr, g2.dil, g2.pad, 1});
}

// ---- split-operand / bf16 forms of the stride-1 "same" dense convolution (conv_bf.hip)
static inline bool lp_terms_ok(int terms) { return terms == 1 || terms == 3 || terms == 6; }

extern "C" int senas_conv2d_pack_layout_lp(const senas_conv_geom* g, int direction, int terms, int32_t* d0, int32_t* d1, int32_t* swap,
                                           int64_t* elems) {
    SENAS_REQUIRE(g && d0 && d1 && swap && elems && (direction == 0 || direction == 1) && lp_terms_ok(terms), "conv2d_pack_layout_lp: bad argument");
    *d0 = g->transposed ? g->ci : g->co;
    *d1 = g->transposed ? g->co : g->ci;
    const int reduce_is_d1 = g->transposed ? (direction == 1) : (direction == 0);
    const int A = reduce_is_d1 ? *d1 : *d0, B = reduce_is_d1 ? *d0 : *d1;
    *swap = reduce_is_d1 | (terms << 8);
    // only shapes conv_bf serves: a Conv2d (the data gradient of a transposed one is a strided gather), 3x3 / 5x5, full tiles
    const bool ok = g->groups == 1 && !g->transposed && g->kh == g->kw && (g->kh == 3 || g->kh == 5) && A % (terms == 1 ? 32 : 16) == 0 && B % 32 == 0;
    *elems = ok ? bf_image_bytes(A, B, g->kh * g->kw, terms) / 4 : 0;
    return SENAS_OK;
}

extern "C" int senas_pack_batched_lp(const senas_pack_item* items_dev, int n, int64_t max_elems, void* stream) {
    SENAS_REQUIRE(items_dev && n > 0 && max_elems > 0, "pack_batched_lp: bad argument");
    return launch_bf_pack_batched(items_dev, n, max_elems, as_stream(stream));
}

extern "C" int senas_conv2d_fwd_lp(const senas_conv_geom* g, const float* x, const float* w, float* y, int in_relu, double* stats,
                                   void* ws, const void* packed_lp, int terms, void* stream) {
    SENAS_REQUIRE(geom_ok(g) && lp_terms_ok(terms), "conv2d_fwd_lp: inconsistent geometry");
    SENAS_REQUIRE(x && w && y, "conv2d_fwd_lp: null pointer");
    if (g->groups != 1 || g->transposed) return SENAS_EUNSUPPORTED;
    GatherGeom gg{g->n, g->hi, g->wi, g->ci, g->ho, g->wo, g->co, g->kh, g->kw, g->stride, g->pad, g->dil};
    if (!bf_gather_ok(gg, terms)) return SENAS_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const void* img = packed_lp;
    if (img == nullptr) {
        SENAS_REQUIRE(ws, "conv2d_fwd_lp: null workspace");
        launch_bf_pack(w, ws, g->co, g->ci, g->kh * g->kw, 1, terms, st);
        img = ws;
    }
    return launch_bf_gather<false>(gg, terms, x, img, y, in_relu, nullptr, stats, st);
}

extern "C" int senas_conv2d_bwd_data_lp(const senas_conv_geom* g, const float* dy, const float* w, float* dx, int in_relu, const float* x,
                                        void* ws, const void* packed_lp, int terms, void* stream) {
    SENAS_REQUIRE(geom_ok(g) && lp_terms_ok(terms), "conv2d_bwd_data_lp: inconsistent geometry");
    SENAS_REQUIRE(dy && w && dx, "conv2d_bwd_data_lp: null pointer");
    SENAS_REQUIRE(!in_relu || x, "conv2d_bwd_data_lp: in_relu needs x");
    if (g->groups != 1 || g->transposed) return SENAS_EUNSUPPORTED;
    GatherGeom gg{g->n, g->ho, g->wo, g->co, g->hi, g->wi, g->ci, g->kh, g->kw, g->stride, g->pad, g->dil};
    if (!bf_gather_ok(gg, terms)) return SENAS_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const void* img = packed_lp;
    if (img == nullptr) {
        SENAS_REQUIRE(ws, "conv2d_bwd_data_lp: null workspace");
        launch_bf_pack(w, ws, g->co, g->ci, g->kh * g->kw, 0, terms, st);
        img = ws;
    }
    return launch_bf_gather<true>(gg, terms, dy, img, dx, 0, in_relu ? x : nullptr, nullptr, st);
}

extern "C" int senas_conv2d_bwd_weight_ws_lp(const senas_conv_geom* g, int terms, int64_t* bytes) {
    SENAS_REQUIRE(geom_ok(g) && bytes && lp_terms_ok(terms), "conv2d_bwd_weight_ws_lp: bad argument");
    *bytes = 0;
    if (g->groups != 1 || g->transposed) return SENAS_OK;
    const WgradGeom wg{g->n, g->ho, g->wo, g->co, g->hi, g->wi, g->ci, g->kh, g->kw, g->stride, g->pad, g->dil, 0};
    if (bf_wgrad_ok(wg, terms)) *bytes = bf_wgrad_ws_bytes(wg, terms) + 256;
    return SENAS_OK;
}

extern "C" int senas_conv2d_bwd_weight_lp(const senas_conv_geom* g, const float* x, int in_relu, const float* dy, float* dw, void* ws,
                                          int terms, senas_sum_item* defer, void* stream) {
    if (defer != nullptr) defer->kind = 0;
    SENAS_REQUIRE(geom_ok(g) && lp_terms_ok(terms), "conv2d_bwd_weight_lp: inconsistent geometry");
    SENAS_REQUIRE(x && dy && dw, "conv2d_bwd_weight_lp: null pointer");
    if (g->groups != 1 || g->transposed) return SENAS_EUNSUPPORTED;
    const WgradGeom wg{g->n, g->ho, g->wo, g->co, g->hi, g->wi, g->ci, g->kh, g->kw, g->stride, g->pad, g->dil, 0};
    if (!bf_wgrad_ok(wg, terms)) return SENAS_EUNSUPPORTED;
    SENAS_REQUIRE(ws, "conv2d_bwd_weight_lp: null workspace");
    senas_sum_item item;
    const int rc = launch_bf_wgrad(wg, terms, x, dy, reinterpret_cast<float*>(ws), dw, in_relu, &item, as_stream(stream));
    if (rc != SENAS_OK) return rc;
    if (defer != nullptr) { *defer = item; return SENAS_OK; }
    return senas_wgrad_sum_batched(&item, 1, stream);
}

// ---- "bf16s": the bf16-pipe convolutions with their OUTPUT (forward) and its GRADIENT (data / weight gradient operand) stored as
// bf16 tensors; x, dx and dw stay fp32, products are plain bf16 x bf16 into fp32 accumulators, statistics from the accumulators
extern "C" int senas_conv2d_fwd_bf16s(const senas_conv_geom* g, const float* x, const float* w, void* y_bf16, int in_relu, double* stats,
                                      void* ws, const void* packed_lp, void* stream) {
    SENAS_REQUIRE(geom_ok(g), "conv2d_fwd_bf16s: inconsistent geometry");
    SENAS_REQUIRE(x && w && y_bf16, "conv2d_fwd_bf16s: null pointer");
    if (g->groups != 1 || g->transposed) return SENAS_EUNSUPPORTED;
    GatherGeom gg{g->n, g->hi, g->wi, g->ci, g->ho, g->wo, g->co, g->kh, g->kw, g->stride, g->pad, g->dil};
    if (!bf_gather_ok(gg, 1)) return SENAS_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const void* img = packed_lp;
    if (img == nullptr) {
        SENAS_REQUIRE(ws, "conv2d_fwd_bf16s: null workspace");
        launch_bf_pack(w, ws, g->co, g->ci, g->kh * g->kw, 1, 1, st);
        img = ws;
    }
    return launch_bf_gather_stored(gg, false, x, img, y_bf16, in_relu, nullptr, stats, st);
}

extern "C" int senas_conv2d_bwd_data_bf16s(const senas_conv_geom* g, const void* dy_bf16, const float* w, float* dx, int in_relu,
                                           const float* x, void* ws, const void* packed_lp, void* stream) {
    SENAS_REQUIRE(geom_ok(g), "conv2d_bwd_data_bf16s: inconsistent geometry");
    SENAS_REQUIRE(dy_bf16 && w && dx, "conv2d_bwd_data_bf16s: null pointer");
    SENAS_REQUIRE(!in_relu || x, "conv2d_bwd_data_bf16s: in_relu needs x");
    if (g->groups != 1 || g->transposed) return SENAS_EUNSUPPORTED;
    GatherGeom gg{g->n, g->ho, g->wo, g->co, g->hi, g->wi, g->ci, g->kh, g->kw, g->stride, g->pad, g->dil};
    if (!bf_gather_ok(gg, 1)) return SENAS_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const void* img = packed_lp;
    if (img == nullptr) {
        SENAS_REQUIRE(ws, "conv2d_bwd_data_bf16s: null workspace");
        launch_bf_pack(w, ws, g->co, g->ci, g->kh * g->kw, 0, 1, st);
        img = ws;
    }
    return launch_bf_gather_stored(gg, true, dy_bf16, img, dx, 0, in_relu ? x : nullptr, nullptr, st);
}

extern "C" int senas_conv2d_bwd_weight_bf16s(const senas_conv_geom* g, const float* x, int in_relu, const void* dy_bf16, float* dw, void* ws,
                                             senas_sum_item* defer, void* stream) {
    if (defer != nullptr) defer->kind = 0;
    SENAS_REQUIRE(geom_ok(g), "conv2d_bwd_weight_bf16s: inconsistent geometry");
    SENAS_REQUIRE(x && dy_bf16 && dw, "conv2d_bwd_weight_bf16s: null pointer");
    if (g->groups != 1 || g->transposed) return SENAS_EUNSUPPORTED;
    const WgradGeom wg{g->n, g->ho, g->wo, g->co, g->hi, g->wi, g->ci, g->kh, g->kw, g->stride, g->pad, g->dil, 0};
    if (!bf_wgrad_ok(wg, 1)) return SENAS_EUNSUPPORTED;
    SENAS_REQUIRE(ws, "conv2d_bwd_weight_bf16s: null workspace");
    senas_sum_item item;
    const int rc = launch_bf_wgrad_stored(wg, x, dy_bf16, reinterpret_cast<float*>(ws), dw, in_relu, &item, as_stream(stream));
    if (rc != SENAS_OK) return rc;
    if (defer != nullptr) { *defer = item; return SENAS_OK; }
    return senas_wgrad_sum_batched(&item, 1, stream);
}

extern "C" const char* senas_conv2d_kernel_name_lp(const senas_conv_geom* g, int which, int terms) {
    if (!geom_ok(g) || which < 0 || which > 2 || !lp_terms_ok(terms) || g->groups != 1 || g->transposed) return "";
    if (which == 2) {
        const WgradGeom wg{g->n, g->ho, g->wo, g->co, g->hi, g->wi, g->ci, g->kh, g->kw, g->stride, g->pad, g->dil, 0};
        if (!bf_wgrad_ok(wg, terms)) return "";
        static char wbuf[8][48];
        static int wslot = 0;
        char* b = wbuf[wslot++ & 7];
        bf_wgrad_name(wg, terms, b, 48);
        return b;
    }
    GatherGeom gg = which == 0 ? GatherGeom{g->n, g->hi, g->wi, g->ci, g->ho, g->wo, g->co, g->kh, g->kw, g->stride, g->pad, g->dil}
                               : GatherGeom{g->n, g->ho, g->wo, g->co, g->hi, g->wi, g->ci, g->kh, g->kw, g->stride, g->pad, g->dil};
    if (!bf_gather_ok(gg, terms)) return "";
    static char buf[8][48];
    static int slot = 0;
    char* b = buf[slot++ & 7];
    bf_gather_name(gg, terms, which == 1, b, 48);
    return b;
}

// One launch for the second stage of up to SENAS_MAX_SUMS two-stage weight gradients: block b belongs to the item whose
// block range [first[i], first[i + 1]) contains it.
struct SumTab {
    senas_sum_item it[SENAS_MAX_SUMS];
    int first[SENAS_MAX_SUMS + 1];
    int n;
};

__global__ __launch_bounds__(256) void wgrad_sum_batched_kernel(SumTab tab) {
    __shared__ float red[8][32];
    int i = 0;
    while (i + 1 < tab.n && (int)blockIdx.x >= tab.first[i + 1]) ++i;               // block-uniform
    const senas_sum_item& it = tab.it[i];
    const int blk = blockIdx.x - tab.first[i];
    if (it.kind == 1) {             // flat partials [nblk][n_elem] -> dw[n_elem]: one wave per element, fixed-order tree
        const int e = blk * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (e >= it.n_elem) return;
        float v = 0.f;
        for (int b = lane; b < it.nblk; b += 64) v += it.part[(size_t)b * it.n_elem + e];
        v = wave_sum(v);
        if (lane == 0) it.dw[e] = v;
        return;
    }
    // kind 2: wgrad_lds partial images [nblk][unit][32 a][32 b] -> torch layout dw[b][a][tap] (as wgrad_lds_sum_kernel)
    const int b = threadIdx.x & 31, k = threadIdx.x >> 5;
    const int unit = blk >> 5, arow = blk & 31;
    const int a_tiles = it.A / 32;
    const size_t per_blk = (size_t)it.taps * a_tiles * 1024;
    const float* p = it.part + ((size_t)unit * 32 + arow) * 32 + b;
    float sacc = 0.f;
    int j = k;
    for (; j + 24 < it.nblk; j += 32) {
        const float v0 = p[(size_t)j * per_blk], v1 = p[(size_t)(j + 8) * per_blk], v2 = p[(size_t)(j + 16) * per_blk],
                    v3 = p[(size_t)(j + 24) * per_blk];
        sacc += v0; sacc += v1; sacc += v2; sacc += v3;
    }
    for (; j < it.nblk; j += 8) sacc += p[(size_t)j * per_blk];
    red[k][b] = sacc;
    __syncthreads();
    if (k == 0 && b < it.B) {
        float tot = red[0][b];
#pragma unroll
        for (int q = 1; q < 8; ++q) tot += red[q][b];
        const int tap = unit / a_tiles, a = (unit - tap * a_tiles) * 32 + arow;
        it.dw[((size_t)b * it.A + a) * it.taps + tap] = tot;
    }
}

extern "C" int senas_wgrad_sum_batched(const senas_sum_item* items, int n, void* stream) {
    SENAS_REQUIRE(items && n >= 1 && n <= SENAS_MAX_SUMS, "wgrad_sum_batched: 1..SENAS_MAX_SUMS items");
    SumTab tab{};
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const senas_sum_item& it = items[i];
        SENAS_REQUIRE(it.part && it.dw && it.nblk >= 1 && (it.kind == 1 || it.kind == 2), "wgrad_sum_batched: bad item");
        tab.it[i] = it;
        tab.first[i] = total;
        if (it.kind == 1) { SENAS_REQUIRE(it.n_elem >= 1, "wgrad_sum_batched: bad item"); total += (it.n_elem + 3) / 4; }
        else { SENAS_REQUIRE(it.A % 32 == 0 && it.A >= 32 && it.B >= 1 && it.B <= 32 && it.taps >= 1, "wgrad_sum_batched: bad item"); total += it.taps * (it.A / 32) * 32; }
    }
    tab.first[n] = total;
    tab.n = n;
    hipLaunchKernelGGL(wgrad_sum_batched_kernel, dim3((unsigned)total), dim3(256), 0, as_stream(stream), tab);
    return launch_status("wgrad_sum_batched");
}

static void flat_sum(const float* part, float* dw, int n_elem, int nblk, senas_sum_item* defer, hipStream_t st) {
    if (defer != nullptr) {
        *defer = senas_sum_item{part, dw, 1, 0, 0, 0, n_elem, nblk};
        return;
    }
    hipLaunchKernelGGL(dwconv_wgrad_sum_kernel, dim3((n_elem + 3) / 4), dim3(256), 0, st, part, dw, n_elem, nblk);
}

extern "C" int senas_conv2d_bwd_weight(const senas_conv_geom* g, const float* x, int in_relu, const float* dy, float* dw,
                                       void* ws, int ws_is_zero, void* stream) {
    return senas_conv2d_bwd_weight_deferred(g, x, in_relu, dy, dw, ws, ws_is_zero, nullptr, stream);
}

extern "C" int senas_conv2d_bwd_weight_deferred(const senas_conv_geom* g, const float* x, int in_relu, const float* dy, float* dw,
                                                void* ws, int ws_is_zero, senas_sum_item* defer, void* stream) {
    if (defer != nullptr) defer->kind = 0;
    SENAS_REQUIRE(geom_ok(g), "conv2d_bwd_weight: inconsistent geometry");
    SENAS_REQUIRE(x && dy && dw, "conv2d_bwd_weight: null pointer");
    hipStream_t st = as_stream(stream);
    const int taps = g->kh * g->kw;
    WgradGeom wg;
    const float *I, *G;
    int i_relu, g_relu;
    if (!g->transposed) {   // I = x (fine grid), G = dy (coarse grid)
        wg = WgradGeom{g->n, g->ho, g->wo, g->co, g->hi, g->wi, g->ci, g->kh, g->kw, g->stride, g->pad, g->dil, 0};
        I = x; G = dy; i_relu = in_relu; g_relu = 0;
    } else {                // I = dy (fine grid), G = x (coarse grid)
        wg = WgradGeom{g->n, g->hi, g->wi, g->ci, g->ho, g->wo, g->co, g->kh, g->kw, g->stride, g->pad, g->dil, 0};
        I = dy; G = x; i_relu = 0; g_relu = in_relu;
    }
    const long total = (long)wg.n * wg.hg * wg.wg;
    if (g->groups != 1) {
        SENAS_REQUIRE(g->ci <= 256, "depthwise wgrad: more than 256 channels");
        SENAS_REQUIRE(g->kh == g->kw && (g->kh == 3 || g->kh == 5), "depthwise wgrad: only 3x3 and 5x5 are on the path");
        const int c4 = g->ci / 4;
        if (g->ci % 4 == 0 && (c4 & (c4 - 1)) == 0 && c4 <= 64 && ws != nullptr &&
            (size_t)4 * (c4 <= 16 ? 4 : 64 / c4) * c4 * taps * 16 <= 64 * 1024) {                         // two-stage, atomic-free
            // short chunks: the per-block loop (pixel lanes x taps of dependent loads) is the critical path of this
            // launch on every map size; 2 passes of the 256/c4 pixel lanes per block, at most 2048 partial rows
            const long lanes = 256 / c4;
            long nblk = (total + 2 * lanes - 1) / (2 * lanes);
            if (nblk > 2048) nblk = 2048;
            wg.chunk = (int)((total + nblk - 1) / nblk);
            nblk = (total + wg.chunk - 1) / wg.chunk;
            float* part = reinterpret_cast<float*>(ws);
            const size_t lds = (size_t)4 * (c4 <= 16 ? 4 : 64 / c4) * c4 * taps * 4 * sizeof(float);     // [waves x row slots][c4][taps][4]
            if (g->kh == 3) hipLaunchKernelGGL((dwconv_wgrad_part_kernel<3>), dim3((unsigned)nblk), dim3(256), lds, st, wg, I, G, part, i_relu, g_relu);
            else hipLaunchKernelGGL((dwconv_wgrad_part_kernel<5>), dim3((unsigned)nblk), dim3(256), lds, st, wg, I, G, part, i_relu, g_relu);
            flat_sum(part, dw, g->ci * taps, (int)nblk, defer, st);
            return launch_status("dwconv_wgrad");
        }
        hipError_t e = hipMemsetAsync(dw, 0, (size_t)g->ci * taps * sizeof(float), st);
        if (e != hipSuccess) { set_error("memset dw", e); return SENAS_ELAUNCH; }
        long chunk = (total + 127) / 128;                 // few blocks: every block ends in atomics on the same addresses
        if (chunk < 64) chunk = 64;
        wg.chunk = (int)chunk;
        dim3 grid((unsigned)((total + wg.chunk - 1) / wg.chunk));
        if (g->kh == 3) hipLaunchKernelGGL((dwconv_wgrad_kernel<3>), grid, dim3(256), 0, st, wg, I, G, dw, i_relu, g_relu);
        else hipLaunchKernelGGL((dwconv_wgrad_kernel<5>), grid, dim3(256), 0, st, wg, I, G, dw, i_relu, g_relu);
        return launch_status("dwconv_wgrad");
    }
    if (!g->transposed && !g_relu && stem_wgrad_ok(wg)) {         // the stem: 1..4 input channels, x window + dy tile in LDS
        SENAS_REQUIRE(ws, "conv2d_bwd_weight: null workspace");
        int nblk = 0;
        const int rc = launch_stem_wgrad(wg, I, G, reinterpret_cast<float*>(ws), i_relu, &nblk, st);
        if (rc != SENAS_OK) return rc;
        flat_sum(reinterpret_cast<const float*>(ws), dw, g->ci * g->co * taps, nblk, defer, st);
        return launch_status("wgrad_stem sum");
    }
    if (thin_n_wgrad_ok(wg)) {
        SENAS_REQUIRE(ws, "conv2d_bwd_weight: null workspace");
        int nblk = 0;
        const int rc = launch_thin_n_wgrad(wg, I, G, reinterpret_cast<float*>(ws), i_relu, g_relu, &nblk, st);
        if (rc != SENAS_OK) return rc;
        flat_sum(reinterpret_cast<const float*>(ws), dw, g->ci * g->co * taps, nblk, defer, st);
        return launch_status("wgrad_thin_n sum");
    }
    if (!g->transposed && !i_relu && !g_relu && c8_mfma_wgrad_ok(wg)) {     // the search cell's 8-channel inner edges: 16 x 16 x 4 MFMA tiles
        SENAS_REQUIRE(ws, "conv2d_bwd_weight: null workspace");
        int nblk = 0;
        const int rc = launch_c8_mfma_wgrad(wg, I, G, reinterpret_cast<float*>(ws), &nblk, st);
        if (rc != SENAS_OK) return rc;
        flat_sum(reinterpret_cast<const float*>(ws), dw, g->ci * g->co * taps, nblk, defer, st);
        return launch_status("wgrad_c8_mfma sum");
    }
    if (wgrad_c8_ok(wg)) {
        SENAS_REQUIRE(ws, "conv2d_bwd_weight: null workspace");
        int nblk = 0;
        const int rc = launch_wgrad_c8(wg, I, G, reinterpret_cast<float*>(ws), i_relu, g_relu, &nblk, st);
        if (rc != SENAS_OK) return rc;
        flat_sum(reinterpret_cast<const float*>(ws), dw, g->ci * g->co * taps, nblk, defer, st);
        return launch_status("wgrad_c8 sum");
    }
    if (lds_wgrad_ok(wg) && !g_relu) {        // ConvTranspose2d: I = dy on the fine grid; a ReLU on the coarse operand is not in the kernel
        SENAS_REQUIRE(ws, "conv2d_bwd_weight: null workspace");
        return launch_lds_wgrad(wg, I, G, reinterpret_cast<float*>(ws), dw, i_relu, defer, st);      // ws need not be zero here
    }
    if (mfma_wgrad_ok(wg)) {
        SENAS_REQUIRE(ws, "conv2d_bwd_weight: null workspace");
        return launch_mfma_wgrad(wg, I, G, dw, reinterpret_cast<float*>(ws), i_relu, g_relu, ws_is_zero, st);
    }
    hipError_t e = hipMemsetAsync(dw, 0, (size_t)g->ci * g->co * taps * sizeof(float), st);
    if (e != hipSuccess) { set_error("memset dw", e); return SENAS_ELAUNCH; }
    wg.chunk = 1024;
    const int tiles = ((wg.A + 31) / 32) * ((wg.B + 31) / 32);
    dim3 grid((unsigned)((total + wg.chunk - 1) / wg.chunk), taps, tiles);
    hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), 0, st, wg, I, G, dw, i_relu, g_relu);
    return launch_status("conv_wgrad");
}

// Weight gradients of TWO Conv2d of one tensor that differ in dilation / padding only (senas_conv2d_fwd_pair's backward pass) as
// ONE first-stage launch: problem 2 on blockIdx.y.  SENAS_EUNSUPPORTED (nothing launched) unless both take the same kernel of
// the wgrad_c8_mfma / wgrad_lds family with one tile list; the caller then makes the two single calls.  defer_a / defer_b: both
// NULL (the second stages run here) or both given (senas_wgrad_sum_batched later).
extern "C" int senas_conv2d_bwd_weight_pair(const senas_conv_geom* ga, const senas_conv_geom* gb, const float* x, int in_relu, const float* dya,
                                            const float* dyb, float* dwa, float* dwb, void* ws_a, void* ws_b, senas_sum_item* defer_a,
                                            senas_sum_item* defer_b, void* stream) {
    if (defer_a != nullptr) defer_a->kind = 0;
    if (defer_b != nullptr) defer_b->kind = 0;
    if (!ga || !gb || !pair_geoms_ok(ga, gb) || (defer_a == nullptr) != (defer_b == nullptr)) return SENAS_EUNSUPPORTED;
    SENAS_REQUIRE(x && dya && dyb && dwa && dwb && ws_a && ws_b, "conv2d_bwd_weight_pair: null pointer");
    hipStream_t st = as_stream(stream);
    const WgradGeom w1{ga->n, ga->ho, ga->wo, ga->co, ga->hi, ga->wi, ga->ci, ga->kh, ga->kw, ga->stride, ga->pad, ga->dil, 0};
    const WgradGeom w2{gb->n, gb->ho, gb->wo, gb->co, gb->hi, gb->wi, gb->ci, gb->kh, gb->kw, gb->stride, gb->pad, gb->dil, 0};
    // (the single call's dispatch order: stem, thin-N, 8-channel MFMA, 8-channel VALU, LDS)
    if (stem_wgrad_ok(w1) || stem_wgrad_ok(w2) || thin_n_wgrad_ok(w1) || thin_n_wgrad_ok(w2)) return SENAS_EUNSUPPORTED;
    const int taps = ga->kh * ga->kw;
    float* pa = reinterpret_cast<float*>(ws_a);
    float* pb = reinterpret_cast<float*>(ws_b);
    const bool c8a = !in_relu && c8_mfma_wgrad_ok(w1), c8b = !in_relu && c8_mfma_wgrad_ok(w2);
    if (c8a != c8b) return SENAS_EUNSUPPORTED;
    if (c8a) {
        int nblk = 0;
        const int rc = launch_c8_mfma_wgrad(w1, x, dya, pa, &nblk, st, WPair2{dyb, pb, w2.dil, w2.pad, 1});
        if (rc != SENAS_OK) return rc;
        flat_sum(pa, dwa, ga->ci * ga->co * taps, nblk, defer_a, st);
        flat_sum(pb, dwb, gb->ci * gb->co * taps, nblk, defer_b, st);
        return launch_status("wgrad_c8_mfma pair sum");
    }
    if (wgrad_c8_ok(w1) || wgrad_c8_ok(w2) || !lds_wgrad_pair_ok(w1, w2)) return SENAS_EUNSUPPORTED;
    return launch_lds_wgrad_pair(w1, w2, x, dya, dyb, pa, pb, dwa, dwb, in_relu, defer_a, defer_b, st);
}

// Which kernel a convolution call dispatches to (same predicates as the launchers above), as the readable
// symbol rocprofv3 prints -- lets bench.py attribute HIP-event time to the kernel the profile shows.
// which: 0 forward, 1 data gradient, 2 weight gradient.
extern "C" const char* senas_conv2d_kernel_name(const senas_conv_geom* g, int which) {
    if (!geom_ok(g) || which < 0 || which > 2) return "invalid";
    const bool tr = g->transposed != 0;
    if (which == 2) {
        if (g->groups != 1) {
            const int c4 = g->ci / 4;
            const bool two_stage = g->ci % 4 == 0 && (c4 & (c4 - 1)) == 0 && c4 <= 64 &&
                                   (size_t)4 * (c4 <= 16 ? 4 : 64 / c4) * c4 * g->kh * g->kw * 16 <= 64 * 1024;
            if (two_stage) return g->kh == 3 ? "dwconv_wgrad_part_kernel<3>" : "dwconv_wgrad_part_kernel<5>";
            return g->kh == 3 ? "dwconv_wgrad_kernel<3>" : "dwconv_wgrad_kernel<5>";
        }
        WgradGeom wg = !tr ? WgradGeom{g->n, g->ho, g->wo, g->co, g->hi, g->wi, g->ci, g->kh, g->kw, g->stride, g->pad, g->dil, 0}
                           : WgradGeom{g->n, g->hi, g->wi, g->ci, g->ho, g->wo, g->co, g->kh, g->kw, g->stride, g->pad, g->dil, 0};
        if (!tr && stem_wgrad_ok(wg)) {
            static const char* names[8] = {"", "wgrad_stem_kernel<1>", "wgrad_stem_kernel<2>", "wgrad_stem_kernel<3>", "wgrad_stem_kernel<4>",
                                           "wgrad_stem_kernel<5>", "wgrad_stem_kernel<6>", "wgrad_stem_kernel<7>"};
            return names[(wg.A * wg.kh * wg.kw + 31) / 32];
        }
        if (thin_n_wgrad_ok(wg)) {
            if (g->kh == 3) return wg.B <= 2 ? "wgrad_thin_n_kernel<3, 2>" : "wgrad_thin_n_kernel<3, 4>";
            return wg.B <= 2 ? "wgrad_thin_n_kernel<1, 2>" : "wgrad_thin_n_kernel<1, 4>";
        }
        if (!tr && c8_mfma_wgrad_ok(wg)) return "wgrad_c8_mfma_kernel";        // (callers without a ReLU on load)
        if (wgrad_c8_ok(wg)) return wg.B <= 8 ? "wgrad_c8_kernel<8>" : "wgrad_c8_kernel<16>";
        if (lds_wgrad_ok(wg)) {
            static char buf[8][48];
            static int slot = 0;
            char* b = buf[slot++ & 7];
            lds_wgrad_name(wg, b, 48);
            return b;
        }
        if (mfma_wgrad_ok(wg)) return wg.A % 8 != 0 ? "wgrad_smallc_mfma_kernel<5>" : "wgrad_mfma_kernel<7>";
        return "conv_wgrad_kernel";
    }
    if (g->groups != 1) {
        const bool tg = (which == 0) == tr;
        const bool v4 = (which == 0 ? g->co : g->ci) % 4 == 0;
        return tg ? (v4 ? "dwconv_kernel<true, 4>" : "dwconv_kernel<true, 1>") : (v4 ? "dwconv_kernel<false, 4>" : "dwconv_kernel<false, 1>");
    }
    GatherGeom gg = which == 0 ? GatherGeom{g->n, g->hi, g->wi, g->ci, g->ho, g->wo, g->co, g->kh, g->kw, g->stride, g->pad, g->dil}
                               : GatherGeom{g->n, g->ho, g->wo, g->co, g->hi, g->wi, g->ci, g->kh, g->kw, g->stride, g->pad, g->dil};
    const bool tg = (which == 0) == tr;                 // transposed gather: ConvTranspose2d forward, Conv2d data gradient
    if (which == 0 && !tr && stem_mfma_ok(gg)) {
        static char buf[8][48];
        static int slot = 0;
        char* b = buf[slot++ & 7];
        snprintf(b, 48, "conv_stem_mfma_kernel<%d, %d>", gg.kh, gg.cin);
        return b;
    }
    if (!tr && c8_mfma_ok(gg)) {                                                                                 // (callers without a ReLU on load)
        static thread_local char c8name[48];
        snprintf(c8name, sizeof c8name, "conv_c8_mfma_kernel<%d, %d>", gg.cin == 8 ? 8 : 16, c8_mfma_tiles_per_wave(gg));
        return c8name;
    }
    if (thin_k_ok(gg) && thin_k3_ok(gg)) {
        if (gg.cin == 2) return tg ? "conv_thin_k3_kernel<2, true>" : "conv_thin_k3_kernel<2, false>";
        return tg ? "conv_thin_k3_kernel<4, true>" : "conv_thin_k3_kernel<4, false>";
    }
    if (thin_k_ok(gg)) return tg ? "conv_thin_k_kernel<true>" : "conv_thin_k_kernel<false>";
    if (which == 0 && thin_n_ok(gg) && (gg.cout <= 4 || tr || !lds_gather_ok(gg))) {
        static char buf[8][48];
        static int slot = 0;
        char* b = buf[slot++ & 7];
        snprintf(b, 48, "conv_thin_n%s_kernel<%d, %s>", thin_n3_ok(gg) ? "3" : "", gg.cout <= 2 ? 2 : (gg.cout <= 4 ? 4 : 8), tr ? "true" : "false");
        return b;
    }
    if (!tr && lds_gather_ok(gg)) {
        static char buf[8][64];
        static int slot = 0;
        char* b = buf[slot++ & 7];
        lds_gather_name(gg, tg, b, 64);
        return b;
    }
    if (!tg && lds_gather_s2_ok(gg)) {
        static char buf[8][64];
        static int slot = 0;
        char* b = buf[slot++ & 7];
        snprintf(b, 64, "conv_lds_kernel<false, 1, 4, %d, 0, 1, %d, 2, false>", gg.kh * gg.kw <= 9 ? 3 : 7, gg.wout >= 16 ? 16 : 8);
        return b;
    }
    if (tg && t2_lds_ok(gg)) return "conv_t2_lds_kernel";               // (callers without a ReLU on load / mask)
    if (mfma_gather_ok(gg, tg)) {
        const bool s2 = tg && gg.stride == 2;
        const long per_phase = (long)gg.n * (s2 ? (gg.hout / 2) * (gg.wout / 2) : gg.hout * gg.wout);
        const long px = per_phase * (s2 ? 4 : 1);
        if (px >= 256L * 512) return tg ? "conv_mfma_kernel<true, 2, 1>" : "conv_mfma_kernel<false, 2, 1>";
        if (px <= 32L * 1024) return tg ? "conv_mfma_kernel<true, 1, 4>" : "conv_mfma_kernel<false, 1, 4>";
        return tg ? "conv_mfma_kernel<true, 1, 1>" : "conv_mfma_kernel<false, 1, 1>";
    }
    return tg ? "conv_direct_kernel<TG>" : "conv_direct_kernel";
}

// ---- k depthwise convolutions of one input (the same-named DepSepConv candidates of the edges leaving a state) ------------
// ka problems of geometry ga followed by kb of geometry gb (gb may be NULL with kb == 0): the mixed launches run the 3x3
// and the 5x5 candidates of the same edges together.
namespace {
bool dw_geom_ok(const senas_conv_geom* g) {
    if (!senas::geom_ok(g) || g->groups == 1) return false;
    const int c4 = g->ci / 4, taps = g->kh * g->kw;
    if (g->ci % 4 != 0 || (c4 & (c4 - 1)) != 0 || c4 > 64) return false;
    if (g->kh != g->kw || (g->kh != 3 && g->kh != 5)) return false;
    return (size_t)4 * (c4 <= 16 ? 4 : 64 / c4) * c4 * taps * 16 <= 64 * 1024;
}

bool dw_pair_ok(const senas_conv_geom* ga, int ka, const senas_conv_geom* gb, int kb, bool all_weights_in_lds = true) {
    if (ka < 1 || kb < 0 || ka + kb > SENAS_MAX_DWMULTI || !dw_geom_ok(ga)) return false;
    size_t wbytes = (size_t)ka * ga->kh * ga->kw * ga->ci * sizeof(float);
    if (kb > 0) {
        if (gb == nullptr || !dw_geom_ok(gb)) return false;
        if (ga->n != gb->n || ga->hi != gb->hi || ga->wi != gb->wi || ga->ci != gb->ci || ga->ho != gb->ho || ga->wo != gb->wo ||
            ga->co != gb->co || ga->stride != gb->stride || ga->transposed != gb->transposed || ga->groups != gb->groups) return false;
        if (!(ga->kh == 3 && gb->kh == 5 && ga->dil == 1 && gb->dil == 1)) return false;     // the one mix there is: 3x3 then 5x5
        wbytes += (size_t)kb * gb->kh * gb->kw * gb->ci * sizeof(float);
    }
    return !all_weights_in_lds || wbytes <= 60 * 1024;     // (only the data gradient keeps every problem's weights in LDS)
}

long dw_multi_wgrad_blocks(const senas_conv_geom* g, int* chunk) {
    const long total = (long)g->n * (g->transposed ? (long)g->hi * g->wi : (long)g->ho * g->wo);
    const long lanes = 256 / (g->ci / 4);
    // every block ends in a fold of taps x 4 partial sums per thread (DPP + LDS) that costs several times the MACs of one
    // pixel: give a thread 16 or 8 pixels where the map is big enough to still fill the chip with 256 blocks per problem, 2 otherwise
    const long fine = (total + 2 * lanes - 1) / (2 * lanes);
    long nblk = (total + 16 * lanes - 1) / (16 * lanes);
    if (nblk < 256) nblk = (total + 8 * lanes - 1) / (8 * lanes);
    if (nblk < 256) nblk = fine < 256 ? fine : 256;
    if (nblk > 2048) nblk = 2048;
    *chunk = (int)((total + nblk - 1) / nblk);
    return (total + *chunk - 1) / *chunk;
}

senas::GatherGeom fwd_geom(const senas_conv_geom* g) {
    return senas::GatherGeom{g->n, g->hi, g->wi, g->ci, g->ho, g->wo, g->co, g->kh, g->kw, g->stride, g->pad, g->dil};
}
senas::GatherGeom bwd_geom(const senas_conv_geom* g) {
    return senas::GatherGeom{g->n, g->ho, g->wo, g->co, g->hi, g->wi, g->ci, g->kh, g->kw, g->stride, g->pad, g->dil};
}
senas::WgradGeom wgrad_geom(const senas_conv_geom* g) {
    return !g->transposed ? senas::WgradGeom{g->n, g->ho, g->wo, g->co, g->hi, g->wi, g->ci, g->kh, g->kw, g->stride, g->pad, g->dil, 0}
                          : senas::WgradGeom{g->n, g->hi, g->wi, g->ci, g->ho, g->wo, g->co, g->kh, g->kw, g->stride, g->pad, g->dil, 0};
}
}  // namespace

extern "C" int senas_dwconv_pair_fwd_xs(const senas_conv_geom* ga, int ka, const senas_conv_geom* gb, int kb, const float* x,
                                        const float* const* xs, const float* const* w, float* const* y, double* const* stats,
                                        void* stream) {
    using namespace senas;
    if (!dw_pair_ok(ga, ka, gb, kb, xs == nullptr)) return SENAS_EUNSUPPORTED;
    SENAS_REQUIRE((x || xs) && w && y, "dwconv_pair_fwd: null pointer");
    const int k = ka + kb;
    const senas_conv_geom* g = ga;
    const GatherGeom gga = fwd_geom(ga), ggb = kb > 0 ? fwd_geom(gb) : gga;
    DwTab tab{};
    bool want = stats != nullptr;
    for (int p = 0; p < k; ++p) {
        SENAS_REQUIRE(w[p] && y[p] && (xs == nullptr || xs[p]), "dwconv_pair_fwd: null pointer");
        tab.a[p] = xs != nullptr ? xs[p] : nullptr;
        tab.w[p] = w[p]; tab.out[p] = y[p]; tab.stats[p] = want ? stats[p] : nullptr;
        want = want && tab.stats[p] != nullptr;
    }
    const int kmax = kb > 0 ? gb->kh : ga->kh;
    const size_t lds = (size_t)kmax * kmax * g->co * sizeof(float);
    if (dw_x4_ok(gga, g->transposed) && dw_x4_ok(ggb, g->transposed)) {
        const long per_img4 = (long)g->ho * (g->wo / 4) * (g->co / 4), total4 = per_img4 * g->n;
        const int P4 = want ? stats_chunks_per_block(per_img4, g->co, total4) : 0;
        dim3 grid4((unsigned)((total4 + 256L * (P4 > 0 ? P4 : 1) - 1) / (256L * (P4 > 0 ? P4 : 1))), k);
#define SENAS_X4(KA_, KB_, S_) hipLaunchKernelGGL((dwconv_multi_fwd_x4_kernel<KA_, KB_, S_>), grid4, dim3(256), lds, as_stream(stream), gga, ggb, ka, x, tab, total4, P4)
        if (kb > 0) { if (g->stride == 1) SENAS_X4(3, 5, 1); else SENAS_X4(3, 5, 2); }
        else if (g->kh == 3) { if (g->stride == 1) SENAS_X4(3, 3, 1); else SENAS_X4(3, 3, 2); }
        else { if (g->stride == 1) SENAS_X4(5, 5, 1); else SENAS_X4(5, 5, 2); }
#undef SENAS_X4
        return launch_status("dwconv_pair_fwd (x4)");
    }
    if (dw_t2_quad_ok(gga, g->transposed) && dw_t2_quad_ok(ggb, g->transposed)) {         // stride-2 transposed: one 2 x 2 output quad per thread
        const long per_imgq = (long)g->hi * g->wi * (g->co / 4), totalq = per_imgq * g->n;
        const int Pq = want ? stats_chunks_per_block(per_imgq, g->co, totalq) : 0;
        dim3 gridq((unsigned)((totalq + 256L * (Pq > 0 ? Pq : 1) - 1) / (256L * (Pq > 0 ? Pq : 1))), k);
#define SENAS_T2(KA_, KB_) hipLaunchKernelGGL((dwconv_multi_fwd_t2_kernel<KA_, KB_>), gridq, dim3(256), lds, as_stream(stream), gga, ggb, ka, x, tab, totalq, Pq)
        if (kb > 0) SENAS_T2(3, 5);
        else if (g->kh == 3) SENAS_T2(3, 3);
        else SENAS_T2(5, 5);
#undef SENAS_T2
        return launch_status("dwconv_pair_fwd (transposed quads)");
    }
    const long per_img = (long)g->ho * g->wo * (g->co / 4);
    const long total = per_img * g->n;
    const int P = want ? stats_chunks_per_block(per_img, g->co, total) : 0;
    dim3 grid((unsigned)((total + 256L * (P > 0 ? P : 1) - 1) / (256L * (P > 0 ? P : 1))), k);
    if (g->transposed) hipLaunchKernelGGL((dwconv_multi_fwd_kernel<true>), grid, dim3(256), lds, as_stream(stream), gga, ggb, ka, x, tab, total, P);
    else hipLaunchKernelGGL((dwconv_multi_fwd_kernel<false>), grid, dim3(256), lds, as_stream(stream), gga, ggb, ka, x, tab, total, P);
    return launch_status("dwconv_pair_fwd");
}

extern "C" int senas_dwconv_pair_fwd(const senas_conv_geom* ga, int ka, const senas_conv_geom* gb, int kb, const float* x,
                                     const float* const* w, float* const* y, double* const* stats, void* stream) {
    return senas_dwconv_pair_fwd_xs(ga, ka, gb, kb, x, nullptr, w, y, stats, stream);
}

extern "C" int senas_dwconv_pair_bwd_data(const senas_conv_geom* ga, int ka, const senas_conv_geom* gb, int kb, const float* const* dy,
                                          const float* const* w, float* dx, void* stream) {
    using namespace senas;
    if (!dw_pair_ok(ga, ka, gb, kb)) return SENAS_EUNSUPPORTED;
    SENAS_REQUIRE(dy && w && dx, "dwconv_pair_bwd_data: null pointer");
    const int k = ka + kb;
    const senas_conv_geom* g = ga;
    const GatherGeom gga = bwd_geom(ga), ggb = kb > 0 ? bwd_geom(gb) : gga;
    DwTab tab{};
    for (int p = 0; p < k; ++p) {
        SENAS_REQUIRE(dy[p] && w[p], "dwconv_pair_bwd_data: null pointer");
        tab.a[p] = dy[p]; tab.w[p] = w[p];
    }
    const size_t lds = ((size_t)ka * ga->kh * ga->kw + (kb > 0 ? (size_t)kb * gb->kh * gb->kw : 0)) * g->ci * sizeof(float);
    auto flips = [](const senas_conv_geom* q) {
        return !q->transposed && q->stride == 1 && q->hi == q->ho && q->wi == q->wo && q->pad == q->dil * (q->kh / 2);
    };
    const bool flip = flips(ga) && (kb == 0 || flips(gb));
    if ((g->transposed || flip) && dw_x4_ok(gga, 0) && dw_x4_ok(ggb, 0)) {                  // a plain gather: four output columns per thread
        const long total4 = (long)g->n * g->hi * (g->wi / 4) * (g->ci / 4);
        const bool split = k >= 3 && total4 <= 128L * 256;              // at most 128 blocks: latency-bound, four lanes per output
        dim3 grid4((unsigned)(((split ? 4 : 1) * total4 + 255) / 256));
#define SENAS_X4(KA_, KB_, S_)                                                                                                          \
    do {                                                                                                                                 \
        if (split) hipLaunchKernelGGL((dwconv_multi_dgrad_x4_kernel<KA_, KB_, S_, 4>), grid4, dim3(256), lds, as_stream(stream), gga, ggb, ka, tab, k, flip ? 1 : 0, dx, total4); \
        else hipLaunchKernelGGL((dwconv_multi_dgrad_x4_kernel<KA_, KB_, S_, 1>), grid4, dim3(256), lds, as_stream(stream), gga, ggb, ka, tab, k, flip ? 1 : 0, dx, total4); \
    } while (0)
        if (kb > 0) { if (gga.stride == 1) SENAS_X4(3, 5, 1); else SENAS_X4(3, 5, 2); }
        else if (g->kh == 3) { if (gga.stride == 1) SENAS_X4(3, 3, 1); else SENAS_X4(3, 3, 2); }
        else { if (gga.stride == 1) SENAS_X4(5, 5, 1); else SENAS_X4(5, 5, 2); }
#undef SENAS_X4
        return launch_status("dwconv_pair_bwd_data (x4)");
    }
    const long total = (long)g->n * g->hi * g->wi * (g->ci / 4);
    dim3 grid((unsigned)((total + 255) / 256));
    auto s2 = [](const senas_conv_geom* q) { return !q->transposed && q->stride == 2 && q->dil == 1 && q->pad == q->kh / 2; };
    if (s2(ga) && (kb == 0 || s2(gb)) && total <= 512L * 256) {                               // small maps: latency-bound
        hipStream_t st = as_stream(stream);
        if (kb > 0) hipLaunchKernelGGL((dwconv_multi_dgrad_s2_kernel<3, 5>), grid, dim3(256), lds, st, gga, ggb, ka, tab, k, dx, total);
        else if (g->kh == 3) hipLaunchKernelGGL((dwconv_multi_dgrad_s2_kernel<3, 3>), grid, dim3(256), lds, st, gga, ggb, ka, tab, k, dx, total);
        else hipLaunchKernelGGL((dwconv_multi_dgrad_s2_kernel<5, 5>), grid, dim3(256), lds, st, gga, ggb, ka, tab, k, dx, total);
        return launch_status("dwconv_pair_bwd_data (stride 2)");
    }
    if (!g->transposed) hipLaunchKernelGGL((dwconv_multi_dgrad_kernel<true>), grid, dim3(256), lds, as_stream(stream), gga, ggb, ka, tab, k, dx, total);
    else hipLaunchKernelGGL((dwconv_multi_dgrad_kernel<false>), grid, dim3(256), lds, as_stream(stream), gga, ggb, ka, tab, k, dx, total);
    return launch_status("dwconv_pair_bwd_data");
}

extern "C" int64_t senas_dwconv_pair_ws_bytes(const senas_conv_geom* ga, int ka, const senas_conv_geom* gb, int kb) {
    if (!dw_pair_ok(ga, ka, gb, kb, ka + kb <= 8)) return 0;             // (more than 8 problems: the per-problem-input forms only)
    int chunk;
    const int64_t nblk = dw_multi_wgrad_blocks(ga, &chunk);
    return nblk * ga->ci * ((int64_t)ka * ga->kh * ga->kw + (kb > 0 ? (int64_t)kb * gb->kh * gb->kw : 0)) * sizeof(float) + 256;
}

extern "C" int senas_dwconv_pair_bwd_weight_xs(const senas_conv_geom* ga, int ka, const senas_conv_geom* gb, int kb, const float* x,
                                               const float* const* xs, const float* const* dy, float* const* dw, void* ws,
                                               senas_sum_item* defer, void* stream) {
    using namespace senas;
    if (!dw_pair_ok(ga, ka, gb, kb, xs == nullptr)) return SENAS_EUNSUPPORTED;
    SENAS_REQUIRE((x || xs) && dy && dw && ws, "dwconv_pair_bwd_weight: null pointer");
    const int k = ka + kb, c4 = ga->ci / 4;
    WgradGeom wga = wgrad_geom(ga), wgb = kb > 0 ? wgrad_geom(gb) : wga;
    int chunk;
    const long nblk = dw_multi_wgrad_blocks(ga, &chunk);
    wga.chunk = wgb.chunk = chunk;
    DwWgradTab tab{};
    int n_elem[SENAS_MAX_DWMULTI];
    float* part = reinterpret_cast<float*>(ws);
    for (int p = 0; p < k; ++p) {
        const senas_conv_geom* g = p < ka ? ga : gb;
        SENAS_REQUIRE(dy[p] && dw[p] && (xs == nullptr || xs[p]), "dwconv_pair_bwd_weight: null pointer");
        const float* xp = xs != nullptr ? xs[p] : x;       // a problem may bring its own input
        tab.I[p] = g->transposed ? dy[p] : xp;             // fine-grid operand
        tab.G[p] = g->transposed ? xp : dy[p];             // coarse-grid operand
        tab.part[p] = part;
        tab.dw[p] = dw[p];
        n_elem[p] = g->ci * g->kh * g->kw;
        part += (size_t)nblk * n_elem[p];
    }
    hipStream_t st = as_stream(stream);
    const int tmax = kb > 0 ? gb->kh * gb->kw : ga->kh * ga->kw;
    const size_t lds = (size_t)4 * (c4 <= 16 ? 4 : 64 / c4) * c4 * tmax * 4 * sizeof(float);
    dim3 grid((unsigned)nblk, k);
    if (kb > 0) hipLaunchKernelGGL((dwconv_wgrad_part_multi_kernel<3, 5>), grid, dim3(256), lds, st, wga, wgb, ka, tab);
    else if (ga->kh == 3) hipLaunchKernelGGL((dwconv_wgrad_part_multi_kernel<3, 3>), grid, dim3(256), lds, st, wga, wgb, ka, tab);
    else hipLaunchKernelGGL((dwconv_wgrad_part_multi_kernel<5, 5>), grid, dim3(256), lds, st, wga, wgb, ka, tab);
    for (int p = 0; p < k; ++p) {
        if (defer != nullptr) defer[p] = senas_sum_item{tab.part[p], tab.dw[p], 1, 0, 0, 0, n_elem[p], (int)nblk};
        else hipLaunchKernelGGL(dwconv_wgrad_sum_kernel, dim3((n_elem[p] + 3) / 4), dim3(256), 0, st, tab.part[p], tab.dw[p], n_elem[p], (int)nblk);
    }
    return launch_status("dwconv_pair_bwd_weight");
}

extern "C" int senas_dwconv_pair_bwd_weight(const senas_conv_geom* ga, int ka, const senas_conv_geom* gb, int kb, const float* x,
                                            const float* const* dy, float* const* dw, void* ws, senas_sum_item* defer, void* stream) {
    return senas_dwconv_pair_bwd_weight_xs(ga, ka, gb, kb, x, nullptr, dy, dw, ws, defer, stream);
}

// the single-geometry forms (include/senas_hip.h)
extern "C" int senas_dwconv_multi_fwd(const senas_conv_geom* g, int k, const float* x, const float* const* w, float* const* y,
                                      double* const* stats, void* stream) {
    return senas_dwconv_pair_fwd(g, k, nullptr, 0, x, w, y, stats, stream);
}
extern "C" int senas_dwconv_multi_bwd_data(const senas_conv_geom* g, int k, const float* const* dy, const float* const* w, float* dx,
                                           void* stream) {
    return senas_dwconv_pair_bwd_data(g, k, nullptr, 0, dy, w, dx, stream);
}
extern "C" int64_t senas_dwconv_multi_ws_bytes(const senas_conv_geom* g, int k) { return senas_dwconv_pair_ws_bytes(g, k, nullptr, 0); }
extern "C" int senas_dwconv_multi_bwd_weight(const senas_conv_geom* g, int k, const float* x, const float* const* dy, float* const* dw,
                                             void* ws, void* stream) {
    return senas_dwconv_pair_bwd_weight(g, k, nullptr, 0, x, dy, dw, ws, nullptr, stream);
}
extern "C" int senas_dwconv_multi_bwd_weight_deferred(const senas_conv_geom* g, int k, const float* x, const float* const* dy,
                                                      float* const* dw, void* ws, senas_sum_item* defer, void* stream) {
    return senas_dwconv_pair_bwd_weight(g, k, nullptr, 0, x, dy, dw, ws, defer, stream);
}
