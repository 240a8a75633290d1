// Stem convolution forward (ConvBn(in_channels, c, kernel_size=7), utils/operations.py:89-95 as used at
// search/senas_search.py:29 and models/senas_model.py:96): c_in = 1..4 input channels, stride 1, "same" padding.
//
// The thin-K gather kernel does this on the VALU (49 FMAs per output, ~7 TF/s: 0.23 ms at 8x1x256x256).  Here the
// (tap, channel) pairs are the K axis of an implicit GEMM on the fp32 MFMA: M = 32 output pixels of one image row,
// N = 32 output channels, K = taps * c_in (49 -> 25 steps of v_mfma_f32_32x32x2_f32).  The input window of a block
// (8 rows x 32 columns + halo, a few KB) sits in LDS; a lane's A operand of step s is ONE ds_read_b32 at a
// compile-time (tap, channel) offset from its pixel; the B operands (weights, K x 32) live in registers for the whole
// block.  The launch is bound by writing the output (HBM), not by arithmetic.
#include "common.h"

namespace senas {

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

constexpr int TH = 8, TWD = 32;            // output tile: 8 rows x 32 columns, wave w owns rows 2w, 2w + 1

// grid = (tiles_x, tiles_y, n * cout/32); block = 256; dynamic LDS = window floats (+ statistics scratch)
template <int KS, int CIN>
__global__ __launch_bounds__(256) void conv_stem_mfma_kernel(GatherGeom g, const float* __restrict__ in,
                                                             const float* __restrict__ w, float* __restrict__ out,
                                                             int in_relu, double* __restrict__ stats) {
    constexpr int TAPS = KS * KS, K = TAPS * CIN, STEPS = (K + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.z % g.n, cot = blockIdx.z / g.n;
    const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TWD;
    const int halo = g.pad, tile_w = TWD + 2 * halo, tile_h = TH + 2 * halo;

    // B operands: lane (channel r, k-half h) holds w[co][k = 2s + h] for every step, k = ci * TAPS + tap (torch layout)
    const int co = cot * 32 + r;
    float bw[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int k = 2 * s + h;
        bw[s] = (k < K && co < g.cout) ? w[(size_t)co * K + k] : 0.f;
    }

    // stage the window: [tile_h][tile_w][CIN], zero outside the image
    const float* src = in + (size_t)n * g.hin * g.win * CIN;
    for (int i = threadIdx.x; i < tile_h * tile_w * CIN; i += 256) {
        const int ci = i % CIN, p = i / CIN;
        const int ty = p / tile_w, tx = p - ty * tile_w;
        const int iy = oy0 - halo + ty, ix = ox0 - halo + tx;
        float v = 0.f;
        if (iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win) {
            v = src[((size_t)iy * g.win + ix) * CIN + ci];
            if (in_relu) v = fmaxf(v, 0.f);
        }
        lds[i] = v;
    }
    __syncthreads();

    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;
    // A operand of step s for this lane: pixel (row 2*wave + m, column r), k = 2s + h -> (ci, ky, kx), all but h known
    // at compile time; both halves are computed and the lane selects (a wave-uniform branch would split the MFMA issue)
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int k0 = 2 * s, k1 = 2 * s + 1;
        const int ci0 = k0 / TAPS, t0 = k0 % TAPS, ci1 = (k1 < K ? k1 : k0) / TAPS, t1 = (k1 < K ? k1 : k0) % TAPS;
        const int off0 = ((t0 / KS) * g.dil * tile_w + (t0 % KS) * g.dil) * CIN + ci0;
        const int off1 = ((t1 / KS) * g.dil * tile_w + (t1 % KS) * g.dil) * CIN + ci1;
        const int off = h ? off1 : off0;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float a = lds[((2 * wave + m) * tile_w + r) * CIN + off];
            if (h && k1 >= K) a = 0.f;
            acc[m] = mfma32(a, bw[s], acc[m]);
        }
    }

    // epilogue: lane = channel r, register v = pixel column acc_row(v, h) of row 2*wave + m
    double sm = 0.0, sq = 0.0;
    const bool cok = co < g.cout;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int oy = oy0 + 2 * wave + m;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int ox = ox0 + acc_row(v, h);
            if (cok && oy < g.hout && ox < g.wout) {
                const float val = acc[m][v];
                out[(((size_t)n * g.hout + oy) * g.wout + ox) * g.cout + co] = val;
                sm += val;
                sq += (double)val * val;
            }
        }
    }
    if (stats != nullptr) {                              // block-level reduction: 2 atomics per channel per block
        __syncthreads();
        double* red = reinterpret_cast<double*>(lds);    // [4 waves][32 channels][2]
        sm += __shfl_xor(sm, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        if (h == 0) { red[(wave * 32 + r) * 2] = sm; red[(wave * 32 + r) * 2 + 1] = sq; }
        __syncthreads();
        if (wave == 0 && h == 0 && cok) {
            for (int wv = 1; wv < 4; ++wv) { sm += red[(wv * 32 + r) * 2]; sq += red[(wv * 32 + r) * 2 + 1]; }
            double* st = stats + ((size_t)n * g.cout + co) * 2;
            atomicAdd(st, sm);
            atomicAdd(st + 1, sq);
        }
    }
}

}  // namespace

bool stem_mfma_ok(const GatherGeom& g) {
    if (g.stride != 1 || g.cin < 1 || g.cin > 4 || g.cout % 32 != 0) return false;
    if (g.kh != g.kw || (g.kh != 7 && g.kh != 3 && g.kh != 5) || g.pad != g.dil * (g.kh / 2)) return false;
    if (g.hout != g.hin || g.wout != g.win || g.wout < 32 || g.hout < 8) return false;
    const size_t bytes = (size_t)(TH + 2 * g.pad) * (TWD + 2 * g.pad) * g.cin * sizeof(float);
    return bytes <= 60 * 1024 && g.kh * g.kw * g.cin <= 160 && (long)g.n * g.hout * g.wout * g.cout < 0x7fffffffL;
}

template <int KS>
static int launch_stem_ks(const GatherGeom& g, const float* in, const float* w, float* out, int in_relu, double* stats,
                          hipStream_t st) {
    size_t bytes = (size_t)(TH + 2 * g.pad) * (TWD + 2 * g.pad) * g.cin * sizeof(float);
    if (bytes < 4 * 32 * 2 * sizeof(double)) bytes = 4 * 32 * 2 * sizeof(double);
    dim3 grid((g.wout + TWD - 1) / TWD, (g.hout + TH - 1) / TH, g.n * (g.cout / 32));
#define SENAS_STEM(CIN_) hipLaunchKernelGGL((conv_stem_mfma_kernel<KS, CIN_>), grid, dim3(256), bytes, st, g, in, w, out, in_relu, stats)
    switch (g.cin) {
        case 1: SENAS_STEM(1); break;
        case 2: SENAS_STEM(2); break;
        case 3: SENAS_STEM(3); break;
        default: SENAS_STEM(4); break;
    }
#undef SENAS_STEM
    return launch_status("conv_stem_mfma");
}

int launch_stem_mfma(const GatherGeom& g, const float* in, const float* w, float* out, int in_relu, double* stats,
                     hipStream_t st) {
    if (g.kh == 7) return launch_stem_ks<7>(g, in, w, out, in_relu, stats, st);
    if (g.kh == 5) return launch_stem_ks<5>(g, in, w, out, in_relu, stats, st);
    return launch_stem_ks<3>(g, in, w, out, in_relu, stats, st);
}

}  // namespace senas

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the stem convolution: dW[b][a][t] = sum_{n,p} x[n, p + t - pad][a] * dy[n, p][b], a = 1..4 input
// channels, t = 49 taps, b = 32 output channels.  The direct-global form gathered one float per lane and tap from L1
// (190 us at 8x1x256x256 for 1.6 GFLOP).  Here: MFMA view M = (channel, tap) pairs in 32-row tiles, N = 32 output
// channels, K = pixels (2 per v_mfma_f32_32x32x2_f32); a block stages the x window of an 8 x 32 pixel tile in LDS (a few
// KB) and the dy tile as it is (NHWC: a pixel's 32 channels are one 128-byte row); wave w owns rows 2w, 2w + 1.  Blocks
// loop over tiles with their accumulators in registers, fold their 4 waves through LDS and leave ONE partial row in the
// torch layout; the sum over the blocks is the batched second stage (kind 1).
namespace senas {
namespace {

template <int MT>       // 32-row tiles of the (channel, tap) axis
__global__ __launch_bounds__(256) void wgrad_stem_kernel(WgradGeom g, const float* __restrict__ X, const float* __restrict__ G,
                                                         float* __restrict__ part, int x_relu, int tiles_x, int tiles_y, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int A = g.A, taps = g.kh * g.kw, mtot = A * taps;
    const int halo = g.pad, WW = TWD + 2 * halo, WH = TH + 2 * halo;
    float* xs = lds;                                  // [WH][WW][A]
    float* gs = lds + WH * WW * A;                    // [TH * 32 pixels][32]
    // this lane's (channel, tap) of every M tile: m = a * taps + t (the torch layout of one output channel's weights)
    int moff[MT];
    bool mok[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 32 + r;
        mok[t] = m < mtot;
        const int mc = mok[t] ? m : 0;
        const int a = mc / taps, tap = mc - a * taps;
        moff[t] = ((tap / g.kw) * WW + (tap % g.kw)) * A + a;
    }
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int oy0 = ty * TH, ox0 = tx * TWD;
        __syncthreads();                              // the previous tile's readers are done
        const float* src = X + (size_t)n * g.hi * g.wi * A;
        for (int i = threadIdx.x; i < WH * WW * A; i += 256) {
            const int a = i % A, p = i / A;
            const int wy = p / WW, wx = p - wy * WW;
            const int iy = oy0 - halo + wy, ix = ox0 - halo + wx;
            float v = 0.f;
            if (iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi) {
                v = src[((size_t)iy * g.wi + ix) * A + a];
                if (x_relu) v = fmaxf(v, 0.f);
            }
            xs[i] = v;
        }
        const float* gsrc = G + (size_t)n * g.hg * g.wg * 32;
        for (int i = threadIdx.x; i < TH * 32 * 8; i += 256) {      // 16-byte pieces of the dy tile
            const int q = i & 7, p = i >> 3;
            const int gy = oy0 + (p >> 5), gx = ox0 + (p & 31);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy < g.hg && gx < g.wg) v = *reinterpret_cast<const float4*>(gsrc + ((size_t)gy * g.wg + gx) * 32 + 4 * q);
            *reinterpret_cast<float4*>(gs + (size_t)p * 32 + 4 * q) = v;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * wave + rr;
            const float* xrow = xs + (size_t)row * WW * A;
            const float* grow = gs + (size_t)row * 32 * 32 + r;
            for (int k = 0; k < 32; k += 2) {            // K-step: pixels k + h of the row
                const float b = grow[(k + h) * 32];
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const float a = mok[t] ? xrow[(k + h) * A + moff[t]] : 0.f;
                    acc[t] = mfma32(a, b, acc[t]);
                }
            }
        }
    }
    // ---- fold the 4 waves through LDS, then this block's partial row in the torch layout: e = (b * A + a) * taps + t = b * mtot + m
    __syncthreads();
    float* red = lds;                                  // [wave][MT][16 regs][64 lanes]
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) red[((wave * MT + t) * 16 + v) * 64 + lane] = acc[t][v];
    __syncthreads();
    const int n_elem = 32 * mtot;
    float* dst = part + (size_t)blockIdx.x * n_elem;
    for (int e = threadIdx.x; e < n_elem; e += 256) {
        const int b = e / mtot, m = e - b * mtot;
        const int t = m >> 5, row = m & 31;                        // accumulator register / lane that holds (row, column b)
        const int hh = (row >> 2) & 1, v = (row & 3) + 4 * (row >> 3), ln = hh * 32 + b;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[((w * MT + t) * 16 + v) * 64 + ln];
        dst[e] = s;
    }
}

}  // namespace

bool stem_wgrad_ok(const WgradGeom& g) {
    if (!(g.A >= 1 && g.A <= 4 && g.B == 32 && g.stride == 1 && g.dil == 1 && g.kh == g.kw && (g.kh == 7 || g.kh == 5 || g.kh == 3))) return false;
    if (g.pad != g.kh / 2 || g.hg != g.hi || g.wg != g.wi) return false;
    return (g.A * g.kh * g.kw + 31) / 32 <= 7 && (long)g.n * g.hi * g.wi * 32 < 0x7fffffffL;
}

static long stem_wgrad_tiles(const WgradGeom& g, int& tx, int& ty) {
    tx = (g.wg + TWD - 1) / TWD;
    ty = (g.hg + TH - 1) / TH;
    return (long)tx * ty * g.n;
}

int stem_wgrad_blocks(const WgradGeom& g) {
    int tx, ty;
    const long nt = stem_wgrad_tiles(g, tx, ty);
    return (int)(nt < 512 ? nt : 512);                 // two resident blocks per CU (the window is a few KB)
}

int64_t stem_wgrad_ws_bytes(const WgradGeom& g) { return (int64_t)stem_wgrad_blocks(g) * 32 * g.A * g.kh * g.kw * sizeof(float); }

int launch_stem_wgrad(const WgradGeom& g, const float* X, const float* G, float* part, int x_relu, int* nblk_out, hipStream_t st) {
    int tx, ty;
    const long nt = stem_wgrad_tiles(g, tx, ty);
    const int nblk = stem_wgrad_blocks(g), mt = (g.A * g.kh * g.kw + 31) / 32;
    const size_t stage = ((size_t)(TH + 2 * g.pad) * (TWD + 2 * g.pad) * g.A + (size_t)TH * 32 * 32) * sizeof(float);
    const size_t fold = (size_t)4 * mt * 16 * 64 * sizeof(float);
    const size_t bytes = stage > fold ? stage : fold;
#define SENAS_WS(MT_)                                                                                                         \
    do {                                                                                                                      \
        if (bytes > 64 * 1024)                                                                                                \
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(&wgrad_stem_kernel<MT_>), 128 * 1024, "wgrad_stem: raising the dynamic LDS limit")) return rc; \
        hipLaunchKernelGGL((wgrad_stem_kernel<MT_>), dim3(nblk), dim3(256), bytes, st, g, X, G, part, x_relu, tx, ty, (int)nt); \
    } while (0)
    switch (mt) {
        case 1: SENAS_WS(1); break;
        case 2: SENAS_WS(2); break;
        case 3: SENAS_WS(3); break;
        case 4: SENAS_WS(4); break;
        case 5: SENAS_WS(5); break;
        case 6: SENAS_WS(6); break;
        default: SENAS_WS(7); break;
    }
#undef SENAS_WS
    *nblk_out = nblk;
    return launch_status("wgrad_stem");
}

}  // namespace senas
