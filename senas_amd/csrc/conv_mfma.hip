// Dense convolution as implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// fp32 in / fp32 accumulate: bit-for-bit an fp32 FMA chain, so parity with the fp32 reference
// holds; rate 64 FLOP/clk/SIMD = 157 TFLOP/s chip-wide.  At 64 cycles per MFMA the operands are
// cheap to feed, so no kernel stages through LDS: A fragments are loads straight from the NHWC
// activation (L1/L2-resident across the taps), B fragments come from a pre-packed weight image in
// which one wave-instruction reads 1 KiB contiguous.  Every kernel is software-pipelined by hand:
// the fragments of step i+1 are in flight while the MFMAs of step i issue.
//
//   forward / data-gradient ("gather"):  M = 32 output pixels, N = 32 output channels, K = taps x c_in
//       A[i][k] = in[pixel i shifted by tap][c],  B[k][j] = w[tap][c][j]
//       D: lane holds output channel (lane & 31) for 16 pixels -> stores are 128-B rows.
//   weight-gradient:                     M = 32 in-channels, N = 32 out-channels, K = pixels
//       A[i][k] = I[pixel k shifted by tap][i],   B[k][j] = G[pixel k][j]; one accumulator per tap,
//       taps spread over the waves of the block, split-K over pixel chunks + fp32 atomics.
#include "common.h"

namespace senas {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row of the 32x32 accumulator held in register v by a lane of half h (cdna_hip_programming.md section 3)
__device__ __forceinline__ int acc_row(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

// ---------------------------------------------------------------------------------------------
// weight image for the gather kernel: dst[nt][tap][cg][h][j(32)][s(4)] = W(a = cg*8 + 4h + s, b = nt*32 + j), zero for b >= B
// src is the torch layout [d0][d1][taps]; swap selects which of d0/d1 is the reduction channel a.
__global__ void pack_weights_mfma_kernel(const float* __restrict__ src, float* __restrict__ dst, int d0, int d1, int taps,
                                         int swap) {
    const int A = swap ? d1 : d0, B = swap ? d0 : d1;
    const int ntiles = (B + 31) / 32;
    const int total = ntiles * taps * A * 32;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int s = i & 3, jj = (i >> 2) & 31, h = (i >> 7) & 1;
    const int cg = (i >> 8) % (A / 8);
    const int t = ((i >> 8) / (A / 8)) % taps, nt = (i >> 8) / ((A / 8) * taps);
    const int a = cg * 8 + 4 * h + s, j = nt * 32 + jj;
    float v = 0.f;
    if (j < B) {
        const int s0 = swap ? j : a, s1 = swap ? a : j;
        v = src[((size_t)s0 * d1 + s1) * taps + t];
    }
    dst[i] = v;
}

// ---------------------------------------------------------------------------------------------
// gather kernel.  Output pixels are enumerated phase-major: for the transposed gather with
// stride 2 a block only holds pixels of one (oy & 1, ox & 1) class, so tap validity is uniform.
// One pipeline step = one tap x 4 channel groups (32 input channels): 4 B + 4*MT A 16-byte loads, 16*MT MFMAs.
template <int MT>
struct GFrag {
    float4 a[4][MT];
    float4 b[4];
};

// KS = 4 / 8: the 4 / 8 waves of a block share ONE 32-pixel sub-tile and split the (tap, chunk) steps between
// them (partial accumulators are summed through LDS) -- for tiny maps, where the serial K loop of a wave is
// the whole launch time: a step is 16 MFMAs of 64 cycles on one SIMD, 0.43 us, and the 8 x 8 maps of the deepest
// down cell fill 32 blocks (KS = 8: 512-thread blocks, for launches of at most 256 of them; 16 waves would
// leave 128 VGPRs each and spill).
// KS = 1: every wave owns MT sub-tiles and runs all steps.
template <bool TG, int MT, int KS>
__global__ __launch_bounds__(KS == 8 ? 512 : 256) void conv_mfma_kernel(GatherGeom g, const float* __restrict__ in,
                                                        const float* __restrict__ wp, float* __restrict__ out,
                                                        int in_relu, const float* __restrict__ mask,
                                                        double* __restrict__ stats, int tiles_per_phase) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int co = blockIdx.y * 32 + r;                               // output channel of this lane
    const bool s2 = TG && g.stride == 2;
    const int phase = blockIdx.x / tiles_per_phase, tile = blockIdx.x % tiles_per_phase;
    const int py = s2 ? (phase >> 1) : 0, px = s2 ? (phase & 1) : 0;
    const int HP = s2 ? g.hout >> 1 : g.hout, WP = s2 ? g.wout >> 1 : g.wout;
    const int QP = HP * WP;
    const long total = (long)g.n * QP;
    const long base = KS > 1 ? (long)tile * 32 : ((long)tile * 4 + wave) * (MT * 32);

    int pn[MT], poy[MT], pox[MT];
    bool live[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const long p = base + m * 32 + r;
        live[m] = p < total;
        const long pc = live[m] ? p : 0;
        pn[m] = (int)(pc / QP);
        const int q = (int)(pc % QP);
        const int qy = q / WP, qx = q % WP;
        poy[m] = s2 ? 2 * qy + py : qy;
        pox[m] = s2 ? 2 * qx + px : qx;
    }
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;

    const int ngroups = g.cin >> 3;
    const int nchunks = (ngroups + 3) >> 2;
    wp += (size_t)blockIdx.y * g.kh * g.kw * ngroups * 256 + lane * 4;

    // ---- software pipeline over (tap, chunk) steps
    auto load_step = [&](int ky, int kx, int chunk, GFrag<MT>& f) {
        const float* wt = wp + (size_t)(ky * g.kw + kx) * ngroups * 256;
        int off[MT];
        bool ok[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            int iy, ix;
            if (!TG) {
                iy = poy[m] * g.stride - g.pad + ky * g.dil;
                ix = pox[m] * g.stride - g.pad + kx * g.dil;
            } else if (s2) {
                iy = (poy[m] + g.pad - ky * g.dil) >> 1;
                ix = (pox[m] + g.pad - kx * g.dil) >> 1;
            } else {
                iy = poy[m] + g.pad - ky * g.dil;
                ix = pox[m] + g.pad - kx * g.dil;
            }
            ok[m] = live[m] && iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win;
            off[m] = ok[m] ? ((pn[m] * g.hin + iy) * g.win + ix) * g.cin + 4 * h : 4 * h;   // clamped: always a valid address
        }
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const int cg = chunk * 4 + c4;
            const bool cok = cg < ngroups;                             // wave-uniform
            const int cgc = cok ? cg : 0;
            f.b[c4] = *reinterpret_cast<const float4*>(wt + cgc * 256);
            if (!cok) f.b[c4] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float4 a = *reinterpret_cast<const float4*>(in + off[m] + cgc * 8);
                if (!ok[m]) a = make_float4(0.f, 0.f, 0.f, 0.f);
                if (in_relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
                f.a[c4][m] = a;
            }
        }
    };
    auto compute_step = [&](const GFrag<MT>& f) {
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = mfma32(f.a[c4][m].x, f.b[c4].x, acc[m]);
                acc[m] = mfma32(f.a[c4][m].y, f.b[c4].y, acc[m]);
                acc[m] = mfma32(f.a[c4][m].z, f.b[c4].z, acc[m]);
                acc[m] = mfma32(f.a[c4][m].w, f.b[c4].w, acc[m]);
            }
    };
    auto tap_valid = [&](int ky, int kx) -> bool {
        if (!s2) return true;
        return !(((py + g.pad - ky * g.dil) | (px + g.pad - kx * g.dil)) & 1);
    };
    // advance (ky, kx, chunk) to the next valid step; returns false at the end
    auto advance1 = [&](int& ky, int& kx, int& chunk) -> bool {
        if (++chunk < nchunks) return true;
        chunk = 0;
        for (;;) {
            if (++kx >= g.kw) { kx = 0; if (++ky >= g.kh) return false; }
            if (tap_valid(ky, kx)) return true;
        }
    };
    int sidx = -1;                                                    // running index of valid steps
    auto advance = [&](int& ky, int& kx, int& chunk) -> bool {
        for (;;) {
            if (!advance1(ky, kx, chunk)) return false;
            ++sidx;
            if (KS == 1 || (sidx & (KS - 1)) == wave) return true;    // this wave's share of the steps
        }
    };
    int ky = 0, kx = -1, chunk = nchunks - 1;                         // "before the first step"
    bool have = advance(ky, kx, chunk);
    GFrag<MT> cur, nxt;
    if (have) load_step(ky, kx, chunk, cur);
    while (have) {
        int nky = ky, nkx = kx, nchunk = chunk;
        const bool more = advance(nky, nkx, nchunk);
        if (more) load_step(nky, nkx, nchunk, nxt);
        compute_step(cur);
        if (more) cur = nxt;
        ky = nky; kx = nkx; chunk = nchunk;
        have = more;
    }

    if (KS > 1) {                                                     // sum the partial tiles through LDS
        __shared__ float red[KS > 1 ? KS - 1 : 1][16][64];
        if (wave > 0) {
#pragma unroll
            for (int v = 0; v < 16; ++v) red[wave - 1][v][lane] = acc[0][v];
        }
        __syncthreads();
        if (wave > 0) return;
        if (KS == 4) {
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[0][v] += red[0][v][lane] + red[1][v][lane] + red[2][v][lane];
        } else {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < KS - 1; ++w) t += red[w][v][lane];
                acc[0][v] += t;
            }
        }
    }
    // ---- epilogue: lane = output channel, registers = pixels
    const bool cok = co < g.cout;
    // per-pixel output offset and image index travel by shuffle from the lane that decoded the pixel
    int n_lo = 0x7fffffff, n_hi = -1;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int my_off = ((pn[m] * g.hout + poy[m]) * g.wout + pox[m]) * g.cout;
        const int my_n = live[m] ? pn[m] : -1;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = acc_row(v, h);
            const int o = __shfl(my_off, row, 64);
            const int nn = __shfl(my_n, row, 64);
            float val = acc[m][v];
            if (nn >= 0 && cok) {
                if (mask != nullptr && !(mask[o + co] > 0.f)) val = 0.f;
                out[o + co] = val;
            }
            acc[m][v] = (nn >= 0) ? val : 0.f;
            if (nn >= 0) { n_lo = min(n_lo, nn); n_hi = max(n_hi, nn); }
        }
    }
    if (stats != nullptr) {
        // images touched by this wave (uniform after the reductions)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { n_lo = min(n_lo, __shfl_xor(n_lo, o, 64)); n_hi = max(n_hi, __shfl_xor(n_hi, o, 64)); }
        for (int nn = n_lo; nn <= n_hi; ++nn) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int my_n = live[m] ? pn[m] : -1;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = acc_row(v, h);
                    if (__shfl(my_n, row, 64) == nn) { const double t = acc[m][v]; s += t; q += t * t; }
                }
            }
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            if (h == 0 && cok) {
                double* st = stats + ((size_t)nn * g.cout + co) * 2;
                atomicAdd(st, s);
                atomicAdd(st + 1, q);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient.  dwp[tap][A][32] += sum_p I[n, p*s - pad + k*d][a] * G[n, p][b]
// unit u = tap * a_tiles + a_tile.  Units are dealt to the waves of a block in groups of UW; when a
// convolution has fewer than 4 groups the spare waves split the block's pixel chunk instead.
struct PixelCursor {          // (n, gy, gx) of a running pixel index, advanced without divisions
    int n, gy, gx;
    __device__ __forceinline__ void init(long p, int per_img, int wg) {
        n = (int)(p / per_img);
        const int q = (int)(p % per_img);
        gy = q / wg;
        gx = q % wg;
    }
    __device__ __forceinline__ void step(int dp, int hg, int wg) {
        gx += dp;
        while (gx >= wg) { gx -= wg; if (++gy >= hg) { gy = 0; ++n; } }
    }
};

template <int UW>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradGeom g, const float* __restrict__ I,
                                                         const float* __restrict__ G, float* __restrict__ dwp,
                                                         int i_relu, int g_relu, int units, int a_tiles, int split) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int ug = blockIdx.y * (4 / split) + wave / split;          // unit group of this wave
    const int part = wave % split;                                   // its share of the pixel chunk
    const int u0 = ug * UW;
    if (u0 >= units) return;                                          // wave-uniform
    const int per_img = g.hg * g.wg;
    const long total = (long)g.n * per_img;
    long c0 = (long)blockIdx.x * g.chunk, c1 = c0 + g.chunk;
    if (c1 > total) c1 = total;
    const long span = ((c1 - c0 + split - 1) / split + 1) & ~1L;      // even
    long p0 = c0 + part * span, p1 = p0 + span;
    if (p1 > c1) p1 = c1;
    if (p0 >= p1) return;

    int uky[UW], ukx[UW], ua[UW];
    bool uok[UW];
#pragma unroll
    for (int t = 0; t < UW; ++t) {
        const int u = min(u0 + t, units - 1);
        const int tap = u / a_tiles;
        uky[t] = (tap / g.kw) * g.dil - g.pad;
        ukx[t] = (tap % g.kw) * g.dil - g.pad;
        ua[t] = (u % a_tiles) * 32 + r;
        uok[t] = (u0 + t < units) && ua[t] < g.A;
    }
    f32x16 acc[UW];
#pragma unroll
    for (int t = 0; t < UW; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

    const bool bok = r < g.B;
    const int rb = bok ? r : 0;
    PixelCursor cur;
    cur.init(p0 + h < p1 ? p0 + h : p0, per_img, g.wg);
    long pp = p0 + h;

    auto load = [&](float (&a)[UW], float& b) {
        const bool valid = pp < p1;
        const long pc = valid ? pp : p0;
        b = G[(size_t)pc * g.B + rb];
        if (!(valid && bok)) b = 0.f;
        if (g_relu) b = fmaxf(b, 0.f);
        const int by = cur.gy * g.stride, bx = cur.gx * g.stride;
        const size_t ibase = (size_t)cur.n * g.hi * g.wi;
#pragma unroll
        for (int t = 0; t < UW; ++t) {
            const int iy = by + uky[t], ix = bx + ukx[t];
            const bool ok = valid && uok[t] && iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi;
            const size_t off = ok ? (ibase + (size_t)(iy * g.wi + ix)) * g.A + ua[t] : 0;
            float v = I[off];
            if (!ok) v = 0.f;
            if (i_relu) v = fmaxf(v, 0.f);
            a[t] = v;
        }
    };
    float a_cur[UW], b_cur, a_nxt[UW], b_nxt;
    load(a_cur, b_cur);
    for (long p = p0; p < p1; p += 2) {
        const bool more = p + 2 < p1;
        if (more) {
            pp += 2;
            cur.step(2, g.hg, g.wg);
            load(a_nxt, b_nxt);
        }
#pragma unroll
        for (int t = 0; t < UW; ++t) acc[t] = mfma32(a_cur[t], b_cur, acc[t]);
        if (more) {
#pragma unroll
            for (int t = 0; t < UW; ++t) a_cur[t] = a_nxt[t];
            b_cur = b_nxt;
        }
    }
#pragma unroll
    for (int t = 0; t < UW; ++t) {
        const int u = u0 + t;
        if (u < units && bok) {
            const int tap = u / a_tiles, abase = (u % a_tiles) * 32;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int a = abase + acc_row(v, h);
                if (a < g.A) atomicAdd(&dwp[((size_t)tap * g.A + a) * 32 + r], acc[t][v]);
            }
        }
    }
}

// Small input-channel count (the 7x7 stem: c_in = 1 or 3): the M axis enumerates (tap, channel)
// pairs, m = tap * A + a, so each lane gathers its own tap.  dwp[m][32]; 4 waves split the pixels.
template <int UW>
__global__ __launch_bounds__(256) void wgrad_smallc_mfma_kernel(WgradGeom g, const float* __restrict__ I,
                                                                const float* __restrict__ G, float* __restrict__ dwp,
                                                                int i_relu, int g_relu, int mtot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int per_img = g.hg * g.wg;
    const long total = (long)g.n * per_img;
    long c0 = (long)blockIdx.x * g.chunk, c1 = c0 + g.chunk;
    if (c1 > total) c1 = total;
    const long span = ((c1 - c0 + 3) / 4 + 1) & ~1L;
    long p0 = c0 + wave * span, p1 = p0 + span;
    if (p1 > c1) p1 = c1;
    if (p0 >= p1) return;
    int mky[UW], mkx[UW], ma[UW];
    bool mok[UW];
#pragma unroll
    for (int t = 0; t < UW; ++t) {
        const int m = t * 32 + r;
        mok[t] = m < mtot;
        const int mc = mok[t] ? m : 0;
        const int tap = mc / g.A;
        ma[t] = mc % g.A;
        mky[t] = (tap / g.kw) * g.dil - g.pad;
        mkx[t] = (tap % g.kw) * g.dil - g.pad;
    }
    f32x16 acc[UW];
#pragma unroll
    for (int t = 0; t < UW; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    const bool bok = r < g.B;
    const int rb = bok ? r : 0;
    PixelCursor cur;
    cur.init(p0 + h < p1 ? p0 + h : p0, per_img, g.wg);
    long pp = p0 + h;
    auto load = [&](float (&a)[UW], float& b) {
        const bool valid = pp < p1;
        const long pc = valid ? pp : p0;
        b = G[(size_t)pc * g.B + rb];
        if (!(valid && bok)) b = 0.f;
        if (g_relu) b = fmaxf(b, 0.f);
        const int by = cur.gy * g.stride, bx = cur.gx * g.stride;
        const size_t ibase = (size_t)cur.n * g.hi * g.wi;
#pragma unroll
        for (int t = 0; t < UW; ++t) {
            const int iy = by + mky[t], ix = bx + mkx[t];
            const bool ok = valid && mok[t] && iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi;
            const size_t off = ok ? (ibase + (size_t)(iy * g.wi + ix)) * g.A + ma[t] : 0;
            float v = I[off];
            if (!ok) v = 0.f;
            if (i_relu) v = fmaxf(v, 0.f);
            a[t] = v;
        }
    };
    float a_cur[UW], b_cur, a_nxt[UW], b_nxt;
    load(a_cur, b_cur);
    for (long p = p0; p < p1; p += 2) {
        const bool more = p + 2 < p1;
        if (more) {
            pp += 2;
            cur.step(2, g.hg, g.wg);
            load(a_nxt, b_nxt);
        }
#pragma unroll
        for (int t = 0; t < UW; ++t) acc[t] = mfma32(a_cur[t], b_cur, acc[t]);
        if (more) {
#pragma unroll
            for (int t = 0; t < UW; ++t) a_cur[t] = a_nxt[t];
            b_cur = b_nxt;
        }
    }
    if (bok) {
#pragma unroll
        for (int t = 0; t < UW; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = t * 32 + acc_row(v, h);
                if (m < mtot) atomicAdd(&dwp[(size_t)m * 32 + r], acc[t][v]);
            }
    }
}

// dwp[tap][A][32] -> torch layout dw[b][a][tap]   (also valid for the small-c image: m = tap*A + a)
__global__ void unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int A, int B, int taps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A * B * taps) return;
    const int t = i % taps, a = (i / taps) % A, b = i / (taps * A);
    dw[i] = dwp[((size_t)t * A + a) * 32 + b];
}

bool mfma_gather_ok(const GatherGeom& g, bool tg) {
    if (g.cin % 8 != 0) return false;
    if (tg && g.stride == 2 && ((g.hout | g.wout) & 1)) return false;
    const long lim = 0x7fffffffL;
    return (long)g.n * g.hin * g.win * g.cin < lim && (long)g.n * g.hout * g.wout * g.cout < lim;
}

template <bool TG>
int launch_mfma_gather(const GatherGeom& g, const float* in, const float* wp, float* out, int in_relu,
                       const float* mask, double* stats, hipStream_t st) {
    const bool s2 = TG && g.stride == 2;
    const long per_phase = (long)g.n * (s2 ? (g.hout / 2) * (g.wout / 2) : g.hout * g.wout);
    const int phases = s2 ? 4 : 1;
    const int ntiles = (g.cout + 31) / 32;
    // 2 sub-tiles per wave (256 pixels per block) keeps >= 2 waves per SIMD on the big maps;
    // small maps take 1 sub-tile so that more CUs get work
    if (per_phase * phases >= 256L * 512) {
        const int tiles = (int)((per_phase + 255) / 256);
        hipLaunchKernelGGL((conv_mfma_kernel<TG, 2, 1>), dim3(tiles * phases, ntiles), dim3(256), 0, st, g, in, wp, out, in_relu, mask, stats, tiles);
    } else if (per_phase * phases <= 32L * 1024) {                    // tiny maps: one sub-tile per block, taps split over its waves
        const int tiles = (int)((per_phase + 31) / 32);
        if ((long)tiles * phases * ntiles <= 256)                     // at most a block per CU: 8 waves each
            hipLaunchKernelGGL((conv_mfma_kernel<TG, 1, 8>), dim3(tiles * phases, ntiles), dim3(512), 0, st, g, in, wp, out, in_relu, mask, stats, tiles);
        else
            hipLaunchKernelGGL((conv_mfma_kernel<TG, 1, 4>), dim3(tiles * phases, ntiles), dim3(256), 0, st, g, in, wp, out, in_relu, mask, stats, tiles);
    } else {
        const int tiles = (int)((per_phase + 127) / 128);
        hipLaunchKernelGGL((conv_mfma_kernel<TG, 1, 1>), dim3(tiles * phases, ntiles), dim3(256), 0, st, g, in, wp, out, in_relu, mask, stats, tiles);
    }
    return launch_status("conv_mfma");
}

template int launch_mfma_gather<false>(const GatherGeom&, const float*, const float*, float*, int, const float*, double*, hipStream_t);
template int launch_mfma_gather<true>(const GatherGeom&, const float*, const float*, float*, int, const float*, double*, hipStream_t);

void launch_pack_mfma(const float* w, float* wp, int d0, int d1, int taps, int swap, hipStream_t st) {
    const int A = swap ? d1 : d0, B = swap ? d0 : d1;
    const int total = ((B + 31) / 32) * taps * A * 32;
    hipLaunchKernelGGL(pack_weights_mfma_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, wp, d0, d1, taps, swap);
}

void launch_unpack_wgrad(const float* ws, float* dw, int A, int B, int taps, hipStream_t st) {
    const int n = A * B * taps;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ws, dw, A, B, taps);
}

bool mfma_wgrad_ok(const WgradGeom& g) {
    if (g.B > 32 || (long)g.n * g.hi * g.wi * g.A >= 0x7fffffffL) return false;
    return g.A % 8 == 0 || g.kh * g.kw * g.A <= 160;
}

int launch_mfma_wgrad(WgradGeom g, const float* I, const float* G, float* dw, float* ws, int i_relu, int g_relu,
                      int ws_is_zero, hipStream_t st) {
    const int taps = g.kh * g.kw;
    const long total = (long)g.n * g.hg * g.wg;
    if (!ws_is_zero) {
        hipError_t e = hipMemsetAsync(ws, 0, (size_t)taps * g.A * 32 * sizeof(float), st);
        if (e != hipSuccess) { set_error("memset wgrad ws", e); return SENAS_ELAUNCH; }
    }
    // pixel chunks: ~512 blocks on the big maps; small maps are bound by the serial K loop of one wave,
    // so they get short chunks (32 pixels = 16 MFMA steps) and more blocks
    long chunk = (total + 511) / 512;
    if (chunk < 32) chunk = 32;
    chunk = (chunk + 7) & ~7L;
    g.chunk = (int)chunk;
    const unsigned gx = (unsigned)((total + chunk - 1) / chunk);
    if (g.A % 8 != 0) {                                   // small c_in: (tap, channel) pairs on the M axis
        const int mtot = taps * g.A;
        hipLaunchKernelGGL((wgrad_smallc_mfma_kernel<5>), dim3(gx), dim3(256), 0, st, g, I, G, ws, i_relu, g_relu, mtot);
    } else {
        constexpr int UW = 7;
        const int a_tiles = (g.A + 31) / 32;
        const int units = taps * a_tiles;
        const int ugroups = (units + UW - 1) / UW;
        const int split = ugroups >= 3 ? 1 : (ugroups == 2 ? 2 : 4);
        const int rows = (ugroups + (4 / split) - 1) / (4 / split);
        hipLaunchKernelGGL((wgrad_mfma_kernel<UW>), dim3(gx, rows), dim3(256), 0, st, g, I, G, ws, i_relu, g_relu, units, a_tiles, split);
    }
    const int n = g.A * g.B * taps;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ws, dw, g.A, g.B, taps);
    return launch_status("wgrad_mfma");
}

}  // namespace senas

// ---------------------------------------------------------------------------------------------
// Batched weight packing: one launch refreshes the fragment images of every convolution of a model
// (weights change once per optimizer step, not once per launch).  items: device array.
namespace senas {

struct PackItem {                // = senas_pack_item
    const float* src;
    float* dst;
    int d0, d1, taps, swap;
    long elems;
};

__global__ void pack_weights_batched_kernel(const PackItem* __restrict__ items) {
    const PackItem it = items[blockIdx.y];
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= it.elems) return;
    const int A = it.swap ? it.d1 : it.d0, B = it.swap ? it.d0 : it.d1;
    const int s = i & 3, jj = (i >> 2) & 31, h = (i >> 7) & 1;
    const int cg = (int)((i >> 8) % (A / 8));
    const int t = (int)(((i >> 8) / (A / 8)) % it.taps), nt = (int)((i >> 8) / ((long)(A / 8) * it.taps));
    const int a = cg * 8 + 4 * h + s, j = nt * 32 + jj;
    float v = 0.f;
    if (j < B) {
        const int s0 = it.swap ? j : a, s1 = it.swap ? a : j;
        v = it.src[((size_t)s0 * it.d1 + s1) * it.taps + t];
    }
    it.dst[i] = v;
}

// rows x row_len floats between two row-strided buffers (the per-edge weights into their slice of a stacked weight
// buffer; the slices of a stacked weight gradient back into the per-edge gradients): one launch for all items
struct CopyItem {
    const float* src;
    float* dst;
    long rows, row_len, src_stride, dst_stride;
    long accumulate;            // != 0: dst += src
};

__global__ void copy_rows_batched_kernel(const CopyItem* __restrict__ items) {
    const CopyItem it = items[blockIdx.y];
    const long total = it.rows * it.row_len;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / it.row_len, c = i - r * it.row_len;
        const float v = it.src[r * it.src_stride + c];
        float* out = it.dst + r * it.dst_stride + c;
        *out = it.accumulate ? *out + v : v;
    }
}

}  // namespace senas

static_assert(sizeof(senas_pack_item) == sizeof(senas::PackItem), "senas_pack_item layout");
static_assert(sizeof(senas_copy_item) == sizeof(senas::CopyItem), "senas_copy_item layout");

extern "C" int senas_copy_rows_batched(const senas_copy_item* items_dev, int n, int64_t max_elems, void* stream) {
    SENAS_REQUIRE(items_dev && n > 0 && max_elems > 0, "copy_rows_batched: bad argument");
    long blocks = (max_elems + 255) / 256;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(senas::copy_rows_batched_kernel, dim3((unsigned)blocks, n), dim3(256), 0, senas::as_stream(stream),
                       reinterpret_cast<const senas::CopyItem*>(items_dev));
    return senas::launch_status("copy_rows_batched");
}

extern "C" int senas_conv2d_pack_layout(const senas_conv_geom* g, int direction, int32_t* d0, int32_t* d1, int32_t* swap,
                                        int64_t* elems) {
    SENAS_REQUIRE(g && d0 && d1 && swap && elems && (direction == 0 || direction == 1), "conv2d_pack_layout: bad argument");
    // torch layouts: Conv2d w[co][ci][taps], ConvTranspose2d w[ci][co][taps]; the reduction channel of the
    // forward pass is ci, of the data gradient co
    *d0 = g->transposed ? g->ci : g->co;
    *d1 = g->transposed ? g->co : g->ci;
    const int reduce_is_d1 = g->transposed ? (direction == 1) : (direction == 0);
    *swap = reduce_is_d1;
    const int A = reduce_is_d1 ? *d1 : *d0, B = reduce_is_d1 ? *d0 : *d1;
    *elems = (g->groups == 1 && A % 8 == 0) ? (int64_t)((B + 31) / 32) * g->kh * g->kw * A * 32 : 0;   // 0: no MFMA image
    return SENAS_OK;
}

extern "C" int senas_pack_batched(const senas_pack_item* items_dev, int n, int64_t max_elems, void* stream) {
    SENAS_REQUIRE(items_dev && n > 0 && max_elems > 0, "pack_batched: bad argument");
    dim3 grid((unsigned)((max_elems + 255) / 256), n);
    hipLaunchKernelGGL(senas::pack_weights_batched_kernel, grid, dim3(256), 0, senas::as_stream(stream),
                       reinterpret_cast<const senas::PackItem*>(items_dev));
    return senas::launch_status("pack_batched");
}
