// 8-channel search-cell convolutions on the matrix cores.
//
// The inner edges of a search cell (search/cell.py:81-89: every edge that starts at a node) work on c_part = 8 channels:
// dil_3_conv_5 / dil_2_conv_5 (utils/operations.py:69-72) are 8 -> 8 convolutions there, 8 -> 16 when the two edges that
// leave a node share one launch (senas_amd/cell.py, stacked candidates).  A 32-wide MFMA tile is 3/4 empty on them and
// the thin-K VALU gathers (conv_thin.hip) were L1-bound: ~29 us for 4 x 8 x 128 x 128 -> 16 (14 TFLOP/s).
//
// Here: v_mfma_f32_16x16x4_f32 with the OUTPUT CHANNELS on the 16 rows and 16 consecutive pixels of an image row on the
// 16 columns -- a 16 x 16 tile is exactly one 8 -> 16 problem, and a lane ends up holding 4 consecutive output channels of
// one pixel (one 16-byte store).  K = (tap, 4 input channels): the weight fragments (25 taps x c_in / 4 registers per lane)
// live in registers for the whole block, the input window is staged once in LDS (pixel stride c_in + 4 floats: the
// 64 lanes of a B-fragment read hit 64 different banks).  Forward (plain gather) and -- with the taps mirrored -- the
// data gradient of the same stride-1 "same" convolution (c_in 8 or 16 -> 8).
//
// Block = 256 threads = 4 waves; tile = 8 rows x 32 columns; wave w owns rows 2w, 2w + 1 (four 16-pixel MFMA tiles).
#include "common.h"

namespace senas {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float c8_weight(const float* __restrict__ src, int d1, int taps, int swap, int t, int a, int b) {
    const int s0 = swap ? b : a, s1 = swap ? a : b;               // as conv_thin.hip: [tap][a = input channel][b = output channel]
    return src[((size_t)s0 * d1 + s1) * taps + t];
}

constexpr int TH = 8, TW = 32;

// NT: 16-pixel MFMA tiles per wave -- 4 (tile 8 x 32, wave w owns rows 2w, 2w + 1) or, on maps that would leave most CUs
// without a block, 2 (tile 4 x 32, wave w owns row w: twice the blocks, half the serial MFMA chain of a wave)
template <int CIN, int NT>
__global__ __launch_bounds__(256) void conv_c8_mfma_kernel(GatherGeom g, const float* __restrict__ in, const float* __restrict__ w,
                                                           int d1, int swap, int flip, float* __restrict__ out,
                                                           double* __restrict__ stats, Pair2 pr) {
    constexpr int PS = CIN + 4, CG = CIN / 4, KS = 5, TAPS = 25, NA = TAPS * CG;
    constexpr int TH = 2 * NT, RW = NT / 2;                         // tile rows, rows per wave   (shadows the 8-row default)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 15, lk = lane >> 4;                      // pixel within a tile / k within a K-step (= channel quad of D)
    int n = blockIdx.z;
    if (pr.nz != 0 && n >= pr.nz) {                                // the launch's second problem (block-uniform)
        n -= pr.nz;
        in = pr.in; w = pr.w; out = pr.out; stats = pr.stats;
        g.dil = pr.dil; g.pad = pr.pad;
    }
    const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
    const int halo = g.pad, WW = TW + 2 * halo, WH = TH + 2 * halo;
    const int cout = g.cout;

    // ---- weights -> LDS (coalesced over the torch layout), then this lane's A fragments -> registers
    float* wl = lds;                                               // [tap][ci][16 co] (zero beyond cout)
    for (int i = threadIdx.x; i < TAPS * CIN * 16; i += 256) {
        const int co = i & 15, ci = (i >> 4) % CIN, t = i / (16 * CIN);
        wl[i] = co < cout ? c8_weight(w, d1, TAPS, swap, flip ? TAPS - 1 - t : t, ci, co) : 0.f;
    }
    __syncthreads();
    float areg[NA];
#pragma unroll
    for (int s = 0; s < NA; ++s) {
        const int t = s / CG, cg = s - t * CG;
        areg[s] = wl[(t * CIN + 4 * cg + lk) * 16 + lp];           // A[row = co = lp][k = lk]
    }
    __syncthreads();

    // ---- input window -> LDS, zero outside the image
    float* win = lds;                                              // [WH][WW][PS]
    const float* src = in + (size_t)n * g.hin * g.win * CIN;
    for (int i = threadIdx.x; i < WH * WW * CG; i += 256) {
        const int q = i % CG, px = i / CG;
        const int wy = px / WW, wx = px - wy * WW;
        const int iy = oy0 - halo + wy, ix = ox0 - halo + wx;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win) v = *reinterpret_cast<const float4*>(src + ((size_t)iy * g.win + ix) * CIN + 4 * q);
        *reinterpret_cast<float4*>(win + (size_t)px * PS + 4 * q) = v;
    }
    __syncthreads();

    // ---- NT tiles per wave: tile j = row RW * wave + (j >> 1), columns 16 * (j & 1) ..
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int base[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) base[j] = ((RW * wave + (j >> 1)) * WW + 16 * (j & 1) + lp) * PS + lk;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const int toff = (ky * g.dil * WW + kx * g.dil) * PS;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
                const float a = areg[(ky * KS + kx) * CG + cg];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float b = win[base[j] + toff + 4 * cg];      // B[k = lk][col = pixel lp]
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: lane holds output channels 4 * lk .. + 3 of pixel lp of each tile
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    const bool ch_ok = 4 * lk < cout;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int oy = oy0 + RW * wave + (j >> 1), ox = ox0 + 16 * (j & 1) + lp;
        if (ch_ok && oy < g.hout && ox < g.wout) {
            float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
            stv<4>(out + out_offset(g, ((size_t)n * g.hout + oy) * g.wout + ox, 4 * lk), v);
#pragma unroll
            for (int q = 0; q < 4; ++q) { s1[q] += (double)v[q]; s2[q] += (double)v[q] * (double)v[q]; }
        }
    }
    if (stats == nullptr) return;                                  // block-uniform
    // per-image channel sums: the 16 pixel lanes of a row group with DPP, the 4 waves through LDS, one fp64 atomic pair
    __syncthreads();                                               // (the window is dead: its LDS is reused)
    double* red = reinterpret_cast<double*>(lds);                  // [wave][16 channels][2]
#pragma unroll
    for (int q = 0; q < 4; ++q) { s1[q] = group_sum(s1[q], 16); s2[q] = group_sum(s2[q], 16); }
    if (lp == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { red[(wave * 16 + 4 * lk + q) * 2] = s1[q]; red[(wave * 16 + 4 * lk + q) * 2 + 1] = s2[q]; }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * cout) {
        const int ch = threadIdx.x >> 1, which = threadIdx.x & 1;
        const double tot = red[(0 * 16 + ch) * 2 + which] + red[(1 * 16 + ch) * 2 + which] + red[(2 * 16 + ch) * 2 + which] +
                           red[(3 * 16 + ch) * 2 + which];
        atomicAdd(stats + ((size_t)n * cout + ch) * 2 + which, tot);
    }
}

template <int CIN>
size_t c8_lds_bytes(const GatherGeom& g, int th = TH) {
    const size_t window = (size_t)(th + 2 * g.pad) * (TW + 2 * g.pad) * (CIN + 4) * sizeof(float);
    const size_t weights = (size_t)25 * CIN * 16 * sizeof(float);
    return window > weights ? window : weights;
}

}  // namespace

// stride-1 "same" 5x5 (dilation 1..3) gather with 8 or 16 input channels and 8 or 16 output channels
bool c8_mfma_ok(const GatherGeom& g) {
    if (!(g.stride == 1 && g.kh == 5 && g.kw == 5 && g.dil >= 1 && g.dil <= 3 && g.pad == 2 * g.dil && g.hout == g.hin && g.wout == g.win))
        return false;
    if (!((g.cin == 8 || g.cin == 16) && (g.cout == 8 || g.cout == 16) && !(g.cin == 16 && g.cout == 16))) return false;
    return g.n >= 1 && g.n <= 65535 && (long)g.n * g.hout * g.wout * 16 < 0x7fffffffL;
}

// 16-pixel MFMA tiles per wave of the launch: 4 (8 x 32 tiles), or 2 (4 x 32) when that leaves CUs with fewer than two blocks
int c8_mfma_tiles_per_wave(const GatherGeom& g) {
    const long blocks8 = (long)((g.wout + TW - 1) / TW) * ((g.hout + TH - 1) / TH) * g.n;
    return blocks8 < (g.cin == 16 ? 512 : 256) && g.hout > 4 ? 2 : 4;       // (16 input channels: twice the chain per tile)
}

// w: torch layout viewed as [tap][a = input channel][b = output channel] through (d1, swap) as in conv_thin.hip
int launch_c8_mfma(const GatherGeom& g, const float* in, const float* w, int d1, int swap, int flip, float* out, double* stats,
                   hipStream_t st, const Pair2& pr0) {
    // fewer than one 8 x 32 tile per CU: 4 x 32 tiles (the launch is as long as one wave's serial chain of 25 x c_in / 4 MFMAs
    // per 16-pixel tile -- half the tiles per wave, half the chain)
    GatherGeom gmax = g;                                           // the wider halo of the two problems sizes the window
    Pair2 pr = pr0;
    if (pr.nz != 0) { pr.nz = g.n; if (pr.pad > gmax.pad) gmax.pad = pr.pad; }
    const long both = pr.nz != 0 ? 2 : 1;
    GatherGeom gt = g;
    gt.n = (int)(g.n * both);                                      // (tile choice by the blocks of the whole launch)
    const int th = 2 * c8_mfma_tiles_per_wave(gt);
    dim3 grid((g.wout + TW - 1) / TW, (g.hout + th - 1) / th, (unsigned)(g.n * both));
    if (g.cin == 8) {
        if (th == 4) hipLaunchKernelGGL((conv_c8_mfma_kernel<8, 2>), grid, dim3(256), c8_lds_bytes<8>(gmax, 4), st, g, in, w, d1, swap, flip, out, stats, pr);
        else hipLaunchKernelGGL((conv_c8_mfma_kernel<8, 4>), grid, dim3(256), c8_lds_bytes<8>(gmax), st, g, in, w, d1, swap, flip, out, stats, pr);
    } else {
        const size_t bytes = c8_lds_bytes<16>(gmax, th);
        const void* fn = th == 4 ? reinterpret_cast<const void*>(&conv_c8_mfma_kernel<16, 2>) : reinterpret_cast<const void*>(&conv_c8_mfma_kernel<16, 4>);
        if (int rc = raise_lds_limit(fn, 96 * 1024, "conv_c8_mfma: raising the dynamic LDS limit")) return rc;
        if (th == 4) hipLaunchKernelGGL((conv_c8_mfma_kernel<16, 2>), grid, dim3(256), bytes, st, g, in, w, d1, swap, flip, out, stats, pr);
        else hipLaunchKernelGGL((conv_c8_mfma_kernel<16, 4>), grid, dim3(256), bytes, st, g, in, w, d1, swap, flip, out, stats, pr);
    }
    return launch_status("conv_c8_mfma");
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same convolutions: dW[b][a][t] = sum_{n,p} x[n, p + (t - 12) * d][a] * dy[n, p][b] with a = 8 input
// channels, b = 8 or 16 (stacked) gradient channels.  MFMA view: rows = b (16), columns = (tap of a pair, a) -- two taps
// fill the 16 columns -- and K = 4 consecutive pixels of an image row; 13 accumulators (tap pairs) per wave.  The x window
// sits in LDS as in the forward kernel, dy is read from global memory (64 consecutive floats per K-step when b = 16).
// Blocks loop over tiles (at most 256 blocks), fold their 4 waves through LDS and leave ONE partial row
// part[block][(b * 8 + a) * 25 + t]; the sum over the blocks is the caller's deferred second stage (wgrad_sum_batched).
namespace {

constexpr int PAIRS = 13;

__global__ __launch_bounds__(256) void wgrad_c8_mfma_kernel(WgradGeom g, const float* __restrict__ X, const float* __restrict__ G1,
                                                            float* __restrict__ part1, int tiles_x, int tiles_y, int ntiles, WPair2 second) {
    constexpr int CIN = 8, PS = CIN + 4, KS = 5;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const float* __restrict__ G = G1;
    float* __restrict__ part = part1;
    if (blockIdx.y != 0) { G = second.G; part = second.part; g.dil = second.dil; g.pad = second.pad; }     // second problem of a pair launch
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lj = lane & 15, lk = lane >> 4;                      // A: row b = lj, pixel k = lk;  B: pixel k = lk, column lj
    const int halo = g.pad, WW = TW + 2 * halo, WH = TH + 2 * halo;
    const int a = lj & 7, s = lj >> 3;                             // column = (tap of the pair, input channel)
    f32x4 acc[PAIRS];
#pragma unroll
    for (int u = 0; u < PAIRS; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    // LDS offsets of this lane's B element for each tap pair (relative to the pixel of K-step 0 of a row)
    int toff[PAIRS];
#pragma unroll
    for (int u = 0; u < PAIRS; ++u) {
        const int t = 2 * u + s < KS * KS ? 2 * u + s : 0;         // (the 26th "tap" does not exist: its column is dropped below)
        toff[u] = ((t / KS) * g.dil * WW + (t % KS) * g.dil) * PS + a;
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int oy0 = ty * TH, ox0 = tx * TW;
        __syncthreads();                                           // the previous tile's window is no longer read
        const float* src = X + (size_t)n * g.hi * g.wi * CIN;
        for (int i = threadIdx.x; i < WH * WW * 2; i += 256) {
            const int q = i & 1, px = i >> 1;
            const int wy = px / WW, wx = px - wy * WW;
            const int iy = oy0 - halo + wy, ix = ox0 - halo + wx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi) v = *reinterpret_cast<const float4*>(src + ((size_t)iy * g.wi + ix) * CIN + 4 * q);
            *reinterpret_cast<float4*>(lds + (size_t)px * PS + 4 * q) = v;
        }
        __syncthreads();
        // wave w: rows 2w, 2w + 1; 8 K-steps of 4 pixels per row
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int oy = oy0 + 2 * wave + r;
            float av[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int ox = ox0 + 4 * ks + lk;
                av[ks] = (lj < g.B && oy < g.hg && ox < g.wg) ? G[(((size_t)n * g.hg + oy) * g.wg + ox) * g.B + lj] : 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int pbase = ((2 * wave + r) * WW + 4 * ks + lk) * PS;
#pragma unroll
                for (int u = 0; u < PAIRS; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], lds[pbase + toff[u]], acc[u], 0, 0, 0);
            }
        }
    }
    // ---- fold the 4 waves: red[wave][pair][lane][4]
    __syncthreads();
    float* red = lds;
#pragma unroll
    for (int u = 0; u < PAIRS; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) red[((wave * PAIRS + u) * 64 + lane) * 4 + v] = acc[u][v];
    __syncthreads();
    const int n_elem = g.B * CIN * KS * KS;
    float* dst = part + (size_t)blockIdx.x * n_elem;
    for (int e = threadIdx.x; e < n_elem; e += 256) {              // e = (b * 8 + a) * 25 + t
        const int t = e % 25, aa = (e / 25) & 7, b = e / 200;
        const int u = t >> 1, col = (t & 1) * 8 + aa, ln = (b >> 2) * 16 + col, v = b & 3;
        dst[e] = red[((0 * PAIRS + u) * 64 + ln) * 4 + v] + red[((1 * PAIRS + u) * 64 + ln) * 4 + v] +
                 red[((2 * PAIRS + u) * 64 + ln) * 4 + v] + red[((3 * PAIRS + u) * 64 + ln) * 4 + v];
    }
}

long c8_wgrad_tiles(const WgradGeom& g, int& tiles_x, int& tiles_y) {
    tiles_x = (g.wg + TW - 1) / TW;
    tiles_y = (g.hg + TH - 1) / TH;
    return (long)tiles_x * tiles_y * g.n;
}

}  // namespace

bool c8_mfma_wgrad_ok(const WgradGeom& g) {
    return g.A == 8 && (g.B == 8 || g.B == 16) && g.stride == 1 && g.kh == 5 && g.kw == 5 && g.dil >= 1 && g.dil <= 3 &&
           g.pad == 2 * g.dil && g.hg == g.hi && g.wg == g.wi && g.n >= 1;
}

int c8_mfma_wgrad_blocks(const WgradGeom& g) {
    int tx, ty;
    const long nt = c8_wgrad_tiles(g, tx, ty);
    return (int)(nt < 256 ? nt : 256);
}

int64_t c8_mfma_wgrad_ws_bytes(const WgradGeom& g) {
    return (int64_t)c8_mfma_wgrad_blocks(g) * g.A * g.B * 25 * sizeof(float);
}

int launch_c8_mfma_wgrad(const WgradGeom& g, const float* X, const float* G, float* part, int* nblk_out, hipStream_t st, const WPair2& second) {
    int tx, ty;
    const long nt = c8_wgrad_tiles(g, tx, ty);
    const int nblk = c8_mfma_wgrad_blocks(g);
    const int pad = second.on && second.pad > g.pad ? second.pad : g.pad;
    const size_t window = (size_t)(TH + 2 * pad) * (TW + 2 * pad) * 12 * sizeof(float);
    const size_t fold = (size_t)4 * PAIRS * 64 * 4 * sizeof(float);
    hipLaunchKernelGGL(wgrad_c8_mfma_kernel, dim3((unsigned)nblk, second.on ? 2 : 1), dim3(256), window > fold ? window : fold, st, g, X, G, part, tx, ty,
                       (int)nt, second);
    *nblk_out = nblk;
    return launch_status("wgrad_c8_mfma");
}

}  // namespace senas
