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
