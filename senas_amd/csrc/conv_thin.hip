// "Thin" convolutions, NHWC fp32, gfx950: one side of the GEMM view is a handful of channels, so the
// launch is bound by HBM (one pass over the fat tensor), not by the matrix cores.  On the path:
//   stem   Conv2d(1 -> c, 7x7)      forward            -> thin-K gather   (K = taps x 1)
//   head   Conv2d(c -> n_class, 3x3) data gradient     -> thin-K transposed gather (K = taps x n_class)
//   head   forward                                     -> thin-N gather   (n_class outputs per pixel)
//   head   weight gradient                             -> thin-N wgrad
// Lane mapping everywhere: the fat tensor's channel axis is split in 16-byte pieces over Q consecutive
// lanes, so a wave touches whole pixels (Q x 16 contiguous bytes each) with every load/store instruction.
// Weights are read in the torch layout and transposed while they are copied into LDS (no repack launch).
#include "common.h"

namespace senas {

namespace {

// torch weight src[d0][d1][taps] viewed as [tap][A][B] (swap == 0: A = d0, B = d1; swap == 1: A = d1, B = d0)
__device__ __forceinline__ float weight_at(const float* __restrict__ src, int d1, int taps, int swap, int t, int a, int b) {
    const int s0 = swap ? b : a, s1 = swap ? a : b;
    return src[((size_t)s0 * d1 + s1) * taps + t];
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// thin-K gather: cin <= 8 (the supernet's 8 -> 8 inner edges included), cout = 4*Q (Q a power of two <= 16).  thread = (pixel lane, 4 output channels).
// grid = (pixel chunks of one image, n); block 256; dynamic LDS = weights [taps*cin][cout] + fp64 [cout][2].
template <bool TG>
__global__ __launch_bounds__(256) void conv_thin_k_kernel(GatherGeom g, const float* __restrict__ in,
                                                          const float* __restrict__ w, int d1, int swap,
                                                          float* __restrict__ out, int in_relu,
                                                          const float* __restrict__ mask, double* __restrict__ stats,
                                                          int passes) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int taps = g.kh * g.kw;
    const int wfloats = taps * g.cin * g.cout;
    for (int i = threadIdx.x; i < wfloats; i += 256) {
        const int b = i % g.cout, a = (i / g.cout) % g.cin, t = i / (g.cout * g.cin);
        lds[i] = weight_at(w, d1, taps, swap, t, a, b);
    }
    double* sred = reinterpret_cast<double*>(lds + ((wfloats + 3) & ~3));
    if (stats != nullptr && threadIdx.x < 2 * g.cout) sred[threadIdx.x] = 0.0;
    __syncthreads();

    const int Q = g.cout >> 2, q = threadIdx.x & (Q - 1), pl = threadIdx.x / Q, ppb = 256 / Q;
    const int n = blockIdx.y, hw = g.hout * g.wout;
    const int p0 = blockIdx.x * ppb * passes;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < passes; ++it) {
        const int pix = p0 + it * ppb + pl;
        if (pix >= hw) break;
        const int oy = pix / g.wout, ox = pix - oy * g.wout;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < g.kh; ++ky) {
            int iy;
            if (!tap_src<TG>(g, oy, ky, g.hin, iy)) continue;
            for (int kx = 0; kx < g.kw; ++kx) {
                int ix;
                if (!tap_src<TG>(g, ox, kx, g.win, ix)) continue;
                const float* ip = in + ((size_t)(n * g.hin + iy) * g.win + ix) * g.cin;
                const float* wt = lds + (ky * g.kw + kx) * g.cin * g.cout + q * 4;
                auto mac = [&](float v, int ci) {
                    if (in_relu) v = fmaxf(v, 0.f);
                    const float4 w4 = *reinterpret_cast<const float4*>(wt + ci * g.cout);
                    acc[0] = fmaf(v, w4.x, acc[0]);
                    acc[1] = fmaf(v, w4.y, acc[1]);
                    acc[2] = fmaf(v, w4.z, acc[2]);
                    acc[3] = fmaf(v, w4.w, acc[3]);
                };
                if ((g.cin & 3) == 0) {                    // 4 or 8 input channels: 16-byte loads
                    for (int c4 = 0; c4 < g.cin; c4 += 4) {
                        const float4 v = *reinterpret_cast<const float4*>(ip + c4);
                        mac(v.x, c4); mac(v.y, c4 + 1); mac(v.z, c4 + 2); mac(v.w, c4 + 3);
                    }
                } else {
                    for (int ci = 0; ci < g.cin; ++ci) mac(ip[ci], ci);
                }
            }
        }
        const size_t o = ((size_t)n * hw + pix) * g.cout + q * 4;
        if (mask != nullptr) {
            const float4 m = *reinterpret_cast<const float4*>(mask + o);
            if (!(m.x > 0.f)) acc[0] = 0.f;
            if (!(m.y > 0.f)) acc[1] = 0.f;
            if (!(m.z > 0.f)) acc[2] = 0.f;
            if (!(m.w > 0.f)) acc[3] = 0.f;
        }
        *reinterpret_cast<float4*>(out + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[j] += (double)acc[j];
            ss[j] += (double)acc[j] * (double)acc[j];
        }
    }
    if (stats != nullptr) {           // lanes that share q (stride Q inside the wave) -> LDS -> one atomic per channel per block
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            for (int o = Q; o < 64; o <<= 1) {
                s[j] += __shfl_xor(s[j], o, 64);
                ss[j] += __shfl_xor(ss[j], o, 64);
            }
        }
        if ((threadIdx.x & 63) < Q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                atomicAdd(&sred[(q * 4 + j) * 2], s[j]);
                atomicAdd(&sred[(q * 4 + j) * 2 + 1], ss[j]);
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * g.cout) atomicAdd(stats + (size_t)n * g.cout * 2 + threadIdx.x, sred[threadIdx.x]);
    }
}

// thin-K PLAIN gather, stride 1, FOUR consecutive output columns x 4 output channels per thread: a weight quad read from LDS
// serves four pixels (16 FMAs per ds_read_b128 instead of 4) and the 16 accumulators give the VALU independent chains --
// the one-pixel form above runs the supernet's 8 -> 8 / 8 -> 16 5x5 inner-edge convolutions at ~9 TF/s.
// flip: taps mirrored (the data gradient of a stride-1 "same" Conv2d is this gather over dy with the kernel turned by
// 180 degrees).  cin in {4, 8}; wout % 4 == 0; no mask.
__global__ __launch_bounds__(256) void conv_thin_k4_kernel(GatherGeom g, const float* __restrict__ in,
                                                           const float* __restrict__ w, int d1, int swap, int flip,
                                                           float* __restrict__ out, int in_relu, double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int taps = g.kh * g.kw;
    const int wfloats = taps * g.cin * g.cout;
    for (int i = threadIdx.x; i < wfloats; i += 256) {
        const int b = i % g.cout, a = (i / g.cout) % g.cin, t = i / (g.cout * g.cin);
        lds[i] = weight_at(w, d1, taps, swap, flip ? taps - 1 - t : t, a, b);
    }
    double* sred = reinterpret_cast<double*>(lds + ((wfloats + 3) & ~3));
    if (stats != nullptr && threadIdx.x < 2 * g.cout) sred[threadIdx.x] = 0.0;
    __syncthreads();

    const int Q = g.cout >> 2, q = threadIdx.x & (Q - 1), pl = threadIdx.x / Q, gpb = 256 / Q;      // pixel GROUPS per block
    const int n = blockIdx.y, wq = g.wout >> 2, groups = g.hout * wq;
    const int grp = blockIdx.x * gpb + pl;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    if (grp < groups) {
        const int oy = grp / wq, ox0 = (grp - oy * wq) * 4;
        float acc[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j][c] = 0.f;
        for (int ky = 0; ky < g.kh; ++ky) {
            const int iy = oy - g.pad + ky * g.dil;
            if (iy < 0 || iy >= g.hin) continue;
            const float* row = in + ((size_t)(n * g.hin + iy) * g.win) * g.cin;
            for (int kx = 0; kx < g.kw; ++kx) {
                const float* wt = lds + (ky * g.kw + kx) * g.cin * g.cout + q * 4;
                const int ixb = ox0 - g.pad + kx * g.dil;
                for (int c4 = 0; c4 < g.cin; c4 += 4) {
                    float4 v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ix = ixb + j;
                        v[j] = (ix >= 0 && ix < g.win) ? *reinterpret_cast<const float4*>(row + (size_t)ix * g.cin + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
                        if (in_relu) { v[j].x = fmaxf(v[j].x, 0.f); v[j].y = fmaxf(v[j].y, 0.f); v[j].z = fmaxf(v[j].z, 0.f); v[j].w = fmaxf(v[j].w, 0.f); }
                    }
                    const float4 w0 = *reinterpret_cast<const float4*>(wt + (c4 + 0) * g.cout), w1 = *reinterpret_cast<const float4*>(wt + (c4 + 1) * g.cout),
                                 w2 = *reinterpret_cast<const float4*>(wt + (c4 + 2) * g.cout), w3 = *reinterpret_cast<const float4*>(wt + (c4 + 3) * g.cout);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[j][0] = fmaf(v[j].x, w0.x, fmaf(v[j].y, w1.x, fmaf(v[j].z, w2.x, fmaf(v[j].w, w3.x, acc[j][0]))));
                        acc[j][1] = fmaf(v[j].x, w0.y, fmaf(v[j].y, w1.y, fmaf(v[j].z, w2.y, fmaf(v[j].w, w3.y, acc[j][1]))));
                        acc[j][2] = fmaf(v[j].x, w0.z, fmaf(v[j].y, w1.z, fmaf(v[j].z, w2.z, fmaf(v[j].w, w3.z, acc[j][2]))));
                        acc[j][3] = fmaf(v[j].x, w0.w, fmaf(v[j].y, w1.w, fmaf(v[j].z, w2.w, fmaf(v[j].w, w3.w, acc[j][3]))));
                    }
                }
            }
        }
        const size_t o = (((size_t)n * g.hout + oy) * g.wout + ox0) * g.cout + q * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<float4*>(out + o + (size_t)j * g.cout) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s[c] += (double)acc[j][c];
                ss[c] += (double)acc[j][c] * (double)acc[j][c];
            }
        }
    }
    if (stats != nullptr) {           // lanes that share q (stride Q inside the wave) -> LDS -> one atomic per channel per block
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            for (int o = Q; o < 64; o <<= 1) {
                s[j] += __shfl_xor(s[j], o, 64);
                ss[j] += __shfl_xor(ss[j], o, 64);
            }
        }
        if ((threadIdx.x & 63) < Q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                atomicAdd(&sred[(q * 4 + j) * 2], s[j]);
                atomicAdd(&sred[(q * 4 + j) * 2 + 1], ss[j]);
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * g.cout) atomicAdd(stats + (size_t)n * g.cout * 2 + threadIdx.x, sred[threadIdx.x]);
    }
}

// thin-K gather, 3x3, stride 1, dilation 1, padding 1 (the segmentation head's data gradient: n_class -> 32 channels at full
// resolution), CIN in {2, 4}: the generic kernel above walks its taps behind bounds tests, and hipcc puts a full
// `s_waitcnt vmcnt(0)` after every load that sits behind control flow -- 18 dependent L2 round trips per thread (92 us where the
// output alone is 67 MB = 15 us of HBM time).  Here the nine taps are nine unconditional loads of clamped addresses, issued
// back to back; borders are zeroed by selects.  Same thread mapping, weights, mask and statistics as above.
template <int CIN, bool TG>
__global__ __launch_bounds__(256) void conv_thin_k3_kernel(GatherGeom g, const float* __restrict__ in, const float* __restrict__ w, int d1,
                                                           int swap, float* __restrict__ out, int in_relu, const float* __restrict__ mask,
                                                           double* __restrict__ stats, int passes) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wfloats = 9 * CIN * g.cout;
    for (int i = threadIdx.x; i < wfloats; i += 256) {
        const int b = i % g.cout, a = (i / g.cout) % CIN, t = i / (g.cout * CIN);
        lds[i] = weight_at(w, d1, 9, swap, t, a, b);
    }
    double* sred = reinterpret_cast<double*>(lds + ((wfloats + 3) & ~3));
    if (stats != nullptr && threadIdx.x < 2 * g.cout) sred[threadIdx.x] = 0.0;
    __syncthreads();
    const int Q = g.cout >> 2, q = threadIdx.x & (Q - 1), pl = threadIdx.x / Q, ppb = 256 / Q;
    const int n = blockIdx.y, hw = g.hout * g.wout;
    int bx = blockIdx.x;                                   // XCD-aware order (see conv_thin_n3_kernel)
    if ((gridDim.x & 7u) == 0) bx = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int p0 = bx * ppb * passes;
    const float* __restrict__ img = in + (size_t)n * g.hin * g.win * CIN;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < passes; ++it) {
        const int pix = p0 + it * ppb + pl;
        const bool live = pix < hw;
        const int pc = live ? pix : hw - 1;
        const int oy = pc / g.wout, ox = pc - oy * g.wout;
        float v[9][CIN];
        bool ok[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - 3 * ky;
            const int iy = TG ? oy + 1 - ky : oy - 1 + ky, ix = TG ? ox + 1 - kx : ox - 1 + kx;
            ok[t] = iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win;
            const int cy = min(max(iy, 0), g.hin - 1), cx = min(max(ix, 0), g.win - 1);
            const float* ip = img + ((size_t)cy * g.win + cx) * CIN;
            if (CIN == 2) {
                const float2 u = *reinterpret_cast<const float2*>(ip);
                v[t][0] = u.x; v[t][1] = u.y;
            } else {
                const float4 u = *reinterpret_cast<const float4*>(ip);
                v[t][0] = u.x; v[t][1] = u.y; v[t][CIN - 2] = u.z; v[t][CIN - 1] = u.w;
            }
        }
        const size_t o = ((size_t)n * hw + pc) * g.cout + q * 4;
        float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
        if (mask != nullptr) m = *reinterpret_cast<const float4*>(mask + o);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                float x = ok[t] ? v[t][ci] : 0.f;
                if (in_relu) x = fmaxf(x, 0.f);
                const float4 w4 = *reinterpret_cast<const float4*>(lds + (t * CIN + ci) * g.cout + q * 4);
                acc[0] = fmaf(x, w4.x, acc[0]);
                acc[1] = fmaf(x, w4.y, acc[1]);
                acc[2] = fmaf(x, w4.z, acc[2]);
                acc[3] = fmaf(x, w4.w, acc[3]);
            }
        }
        if (!(m.x > 0.f)) acc[0] = 0.f;
        if (!(m.y > 0.f)) acc[1] = 0.f;
        if (!(m.z > 0.f)) acc[2] = 0.f;
        if (!(m.w > 0.f)) acc[3] = 0.f;
        if (live) {
            *reinterpret_cast<float4*>(out + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[j] += (double)acc[j];
                ss[j] += (double)acc[j] * (double)acc[j];
            }
        }
    }
    if (stats != nullptr) {           // as conv_thin_k_kernel: lanes that share q -> LDS -> one atomic per channel per block
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            for (int o = Q; o < 64; o <<= 1) {
                s[j] += __shfl_xor(s[j], o, 64);
                ss[j] += __shfl_xor(ss[j], o, 64);
            }
        }
        if ((threadIdx.x & 63) < Q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                atomicAdd(&sred[(q * 4 + j) * 2], s[j]);
                atomicAdd(&sred[(q * 4 + j) * 2 + 1], ss[j]);
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * g.cout) atomicAdd(stats + (size_t)n * g.cout * 2 + threadIdx.x, sred[threadIdx.x]);
    }
}

bool thin_k3_ok(const GatherGeom& g) {
    return g.kh == 3 && g.kw == 3 && g.stride == 1 && g.dil == 1 && g.pad == 1 && g.hout == g.hin && g.wout == g.win && (g.cin == 2 || g.cin == 4);
}

bool thin_k4_ok(const GatherGeom& g) {
    const int q = g.cout >> 2;
    // (a quarter of the one-pixel form's threads: only where that still fills the chip -- measured: 128x128x4 images up)
    if ((long)g.n * g.hout * g.wout < 4L * 128 * 128) return false;
    return g.stride == 1 && (g.cin == 4 || g.cin == 8) && g.cout % 4 == 0 && q >= 1 && q <= 4 && (q & (q - 1)) == 0 && g.wout % 4 == 0 &&
           g.hout == g.hin && g.wout == g.win && g.kh == g.kw && g.pad == g.dil * (g.kh / 2) && g.kh * g.kw * g.cin * g.cout <= 8192 &&
           (long)g.n * g.hout * g.wout * g.cout < 0x7fffffffL && g.n <= 65535;
}

int launch_thin_k4(const GatherGeom& g, const float* in, const float* w, int d1, int swap, int flip, float* out, int in_relu,
                   double* stats, hipStream_t st) {
    const int gpb = 256 / (g.cout >> 2), groups = g.hout * (g.wout >> 2);
    dim3 grid((groups + gpb - 1) / gpb, g.n);
    const size_t bytes = (size_t)((g.kh * g.kw * g.cin * g.cout + 3) & ~3) * sizeof(float) + (size_t)2 * g.cout * sizeof(double);
    hipLaunchKernelGGL(conv_thin_k4_kernel, grid, dim3(256), bytes, st, g, in, w, d1, swap, flip, out, in_relu, stats);
    return launch_status("conv_thin_k4");
}

bool thin_k_ok(const GatherGeom& g) {
    const int q = g.cout >> 2;
    return g.cin <= 8 && g.cout % 4 == 0 && q >= 1 && q <= 16 && (q & (q - 1)) == 0 && g.kh * g.kw * g.cin * g.cout <= 8192 &&
           (long)g.n * g.hout * g.wout * g.cout < 0x7fffffffL && g.n <= 65535;
}

template <bool TG>
int launch_thin_k(const GatherGeom& g, const float* in, const float* w, int d1, int swap, float* out, int in_relu,
                  const float* mask, double* stats, hipStream_t st) {
    const int ppb = 256 / (g.cout >> 2), hw = g.hout * g.wout;
    int passes = 8;
    while (passes > 1 && (long)g.n * ((hw + ppb * passes - 1) / (ppb * passes)) < 1024) passes >>= 1;
    dim3 grid((hw + ppb * passes - 1) / (ppb * passes), g.n);
    const size_t bytes = (size_t)((g.kh * g.kw * g.cin * g.cout + 3) & ~3) * sizeof(float) + (size_t)2 * g.cout * sizeof(double);
    if (thin_k3_ok(g)) {
        if (g.cin == 2) hipLaunchKernelGGL((conv_thin_k3_kernel<2, TG>), grid, dim3(256), bytes, st, g, in, w, d1, swap, out, in_relu, mask, stats, passes);
        else hipLaunchKernelGGL((conv_thin_k3_kernel<4, TG>), grid, dim3(256), bytes, st, g, in, w, d1, swap, out, in_relu, mask, stats, passes);
        return launch_status("conv_thin_k3");
    }
    hipLaunchKernelGGL((conv_thin_k_kernel<TG>), grid, dim3(256), bytes, st, g, in, w, d1, swap, out, in_relu, mask, stats, passes);
    return launch_status("conv_thin_k");
}
template int launch_thin_k<false>(const GatherGeom&, const float*, const float*, int, int, float*, int, const float*, double*, hipStream_t);
template int launch_thin_k<true>(const GatherGeom&, const float*, const float*, int, int, float*, int, const float*, double*, hipStream_t);

// ---------------------------------------------------------------------------------------------
// thin-N gather: cout <= CO (2, 4 or 8), cin = 4*Q (Q a power of two <= 16); TG: transposed gather (ConvTranspose2d
// forward).  thread = (pixel lane, 4 input channels); the Q partial sums of a pixel are folded by DPP group sums.
template <int CO, bool TG>
__global__ __launch_bounds__(256) void conv_thin_n_kernel(GatherGeom g, const float* __restrict__ in,
                                                          const float* __restrict__ w, int d1, int swap,
                                                          float* __restrict__ out, int in_relu, double* __restrict__ stats,
                                                          int passes) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [taps*cin][CO] (zero-padded columns), then fp64 [CO][2]
    const int taps = g.kh * g.kw;
    const int rows = taps * g.cin;
    for (int i = threadIdx.x; i < rows * CO; i += 256) {
        const int b = i % CO, a = (i / CO) % g.cin, t = i / (CO * g.cin);
        lds[i] = b < g.cout ? weight_at(w, d1, taps, swap, t, a, b) : 0.f;
    }
    double* sred = reinterpret_cast<double*>(lds + ((rows * CO + 3) & ~3));
    if (threadIdx.x < 2 * CO) sred[threadIdx.x] = 0.0;
    __syncthreads();

    const int Q = g.cin >> 2, q = threadIdx.x & (Q - 1), pl = threadIdx.x / Q, ppb = 256 / Q;
    const int n = blockIdx.y, hw = g.hout * g.wout;
    const int p0 = blockIdx.x * ppb * passes;
    double s[CO], ss[CO];
#pragma unroll
    for (int j = 0; j < CO; ++j) s[j] = ss[j] = 0.0;
    for (int it = 0; it < passes; ++it) {
        const int pix = p0 + it * ppb + pl;
        const bool live = pix < hw;                       // dead lanes keep shuffling with the live ones
        const int pc = live ? pix : 0;
        const int oy = pc / g.wout, ox = pc - oy * g.wout;
        float acc[CO];
#pragma unroll
        for (int j = 0; j < CO; ++j) acc[j] = 0.f;
        for (int ky = 0; ky < g.kh; ++ky) {
            int iy;
            if (!tap_src<TG>(g, oy, ky, g.hin, iy)) continue;
            for (int kx = 0; kx < g.kw; ++kx) {
                int ix;
                if (!tap_src<TG>(g, ox, kx, g.win, ix)) continue;
                float4 v = *reinterpret_cast<const float4*>(in + ((size_t)(n * g.hin + iy) * g.win + ix) * g.cin + q * 4);
                if (in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                const float* wt = lds + ((ky * g.kw + kx) * g.cin + q * 4) * CO;
#pragma unroll
                for (int j = 0; j < CO; ++j) {
                    acc[j] = fmaf(v.x, wt[j], acc[j]);
                    acc[j] = fmaf(v.y, wt[CO + j], acc[j]);
                    acc[j] = fmaf(v.z, wt[2 * CO + j], acc[j]);
                    acc[j] = fmaf(v.w, wt[3 * CO + j], acc[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CO; ++j) acc[j] = group_sum(acc[j], Q);          // Q <= 16: inside a 16-lane row, DPP
        if (live && q == 0) {
            float* op = out + ((size_t)n * hw + pix) * g.cout;
#pragma unroll
            for (int j = 0; j < CO; ++j) {
                if (j < g.cout) {
                    op[j] = acc[j];
                    s[j] += (double)acc[j];
                    ss[j] += (double)acc[j] * (double)acc[j];
                }
            }
        }
    }
    if (stats != nullptr) {
        // only the q == 0 lanes hold values and they are congruent modulo Q: a strided row sum puts each 16-lane row's
        // total in its first lane (DPP, no shuffles); the 16 row totals of the block meet in LDS
#pragma unroll
        for (int j = 0; j < CO; ++j) {
            s[j] = row_strided_sum(s[j], Q);
            ss[j] = row_strided_sum(ss[j], Q);
        }
        if ((threadIdx.x & 15) == 0) {
#pragma unroll
            for (int j = 0; j < CO; ++j) {
                atomicAdd(&sred[j * 2], s[j]);
                atomicAdd(&sred[j * 2 + 1], ss[j]);
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * g.cout) atomicAdd(stats + (size_t)n * g.cout * 2 + threadIdx.x, sred[threadIdx.x]);
    }
}

// thin-N gather, 3x3, stride 1, dilation 1, padding 1 (the segmentation head: 32 -> n_class at full resolution): as
// conv_thin_k3_kernel, the nine taps are nine unconditional loads of clamped addresses issued back to back (the generic
// kernel's bounds tests serialise its loads), borders zeroed by selects.
template <int CO, bool TG>
__global__ __launch_bounds__(256) void conv_thin_n3_kernel(GatherGeom g, const float* __restrict__ in, const float* __restrict__ w, int d1,
                                                           int swap, float* __restrict__ out, int in_relu, double* __restrict__ stats,
                                                           int passes) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [9*cin][CO] (zero-padded columns), then fp64 [CO][2]
    const int rows = 9 * g.cin;
    for (int i = threadIdx.x; i < rows * CO; i += 256) {
        const int b = i % CO, a = (i / CO) % g.cin, t = i / (CO * g.cin);
        lds[i] = b < g.cout ? weight_at(w, d1, 9, swap, t, a, b) : 0.f;
    }
    double* sred = reinterpret_cast<double*>(lds + ((rows * CO + 3) & ~3));
    if (threadIdx.x < 2 * CO) sred[threadIdx.x] = 0.0;
    __syncthreads();
    const int Q = g.cin >> 2, q = threadIdx.x & (Q - 1), pl = threadIdx.x / Q, ppb = 256 / Q;
    const int n = blockIdx.y, hw = g.hout * g.wout;
    // XCD-aware order: workgroups b and b + 8 share an XCD (round-robin placement), and a block's three input rows are its
    // neighbours' too -- an XCD takes a CONTIGUOUS eighth of the image so that those re-reads meet in one L2 (else every row
    // is fetched by three XCDs: 47 us for one pass over 67 MB)
    int bx = blockIdx.x;
    if ((gridDim.x & 7u) == 0) bx = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int p0 = bx * ppb * passes;
    const float* __restrict__ img = in + (size_t)n * g.hin * g.win * g.cin + q * 4;
    double s[CO], ss[CO];
#pragma unroll
    for (int j = 0; j < CO; ++j) s[j] = ss[j] = 0.0;
    for (int it = 0; it < passes; ++it) {
        const int pix = p0 + it * ppb + pl;
        const bool live = pix < hw;                       // dead lanes keep shuffling with the live ones
        const int pc = live ? pix : 0;
        const int oy = pc / g.wout, ox = pc - oy * g.wout;
        float4 v[9];
        bool ok[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - 3 * ky;
            const int iy = TG ? oy + 1 - ky : oy - 1 + ky, ix = TG ? ox + 1 - kx : ox - 1 + kx;
            ok[t] = iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win;
            const int cy = min(max(iy, 0), g.hin - 1), cx = min(max(ix, 0), g.win - 1);
            v[t] = *reinterpret_cast<const float4*>(img + ((size_t)cy * g.win + cx) * g.cin);
        }
        float acc[CO];
#pragma unroll
        for (int j = 0; j < CO; ++j) acc[j] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float4 x = ok[t] ? v[t] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (in_relu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
            const float* wt = lds + (t * g.cin + q * 4) * CO;
#pragma unroll
            for (int j = 0; j < CO; ++j) {
                acc[j] = fmaf(x.x, wt[j], acc[j]);
                acc[j] = fmaf(x.y, wt[CO + j], acc[j]);
                acc[j] = fmaf(x.z, wt[2 * CO + j], acc[j]);
                acc[j] = fmaf(x.w, wt[3 * CO + j], acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < CO; ++j) acc[j] = group_sum(acc[j], Q);          // Q <= 16: inside a 16-lane row, DPP
        if (live && q == 0) {
            float* op = out + ((size_t)n * hw + pix) * g.cout;
#pragma unroll
            for (int j = 0; j < CO; ++j) {
                if (j < g.cout) {
                    op[j] = acc[j];
                    s[j] += (double)acc[j];
                    ss[j] += (double)acc[j] * (double)acc[j];
                }
            }
        }
    }
    if (stats != nullptr) {
#pragma unroll
        for (int j = 0; j < CO; ++j) {
            s[j] = row_strided_sum(s[j], Q);
            ss[j] = row_strided_sum(ss[j], Q);
        }
        if ((threadIdx.x & 15) == 0) {
#pragma unroll
            for (int j = 0; j < CO; ++j) {
                atomicAdd(&sred[j * 2], s[j]);
                atomicAdd(&sred[j * 2 + 1], ss[j]);
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * g.cout) atomicAdd(stats + (size_t)n * g.cout * 2 + threadIdx.x, sred[threadIdx.x]);
    }
}

bool thin_n3_ok(const GatherGeom& g) {
    return g.kh == 3 && g.kw == 3 && g.stride == 1 && g.dil == 1 && g.pad == 1 && g.hout == g.hin && g.wout == g.win && g.cout <= 4;
}

bool thin_n_ok(const GatherGeom& g) {
    const int q = g.cin >> 2;
    return g.cout <= 8 && g.cin % 4 == 0 && q >= 1 && q <= 16 && (q & (q - 1)) == 0 && g.kh * g.kw * g.cin <= 2048 &&
           (long)g.n * g.hin * g.win * g.cin < 0x7fffffffL && g.n <= 65535;
}

template <bool TG>
int launch_thin_n(const GatherGeom& g, const float* in, const float* w, int d1, int swap, float* out, int in_relu,
                  double* stats, hipStream_t st) {
    const int ppb = 256 / (g.cin >> 2), hw = g.hout * g.wout;
    int passes = 8;
    while (passes > 1 && (long)g.n * ((hw + ppb * passes - 1) / (ppb * passes)) < 1024) passes >>= 1;
    dim3 grid((hw + ppb * passes - 1) / (ppb * passes), g.n);
    const int co = g.cout <= 2 ? 2 : (g.cout <= 4 ? 4 : 8);
    const size_t bytes = (size_t)((g.kh * g.kw * g.cin * co + 3) & ~3) * sizeof(float) + (size_t)2 * co * sizeof(double);
    if (thin_n3_ok(g)) {
        if (co == 2) hipLaunchKernelGGL((conv_thin_n3_kernel<2, TG>), grid, dim3(256), bytes, st, g, in, w, d1, swap, out, in_relu, stats, passes);
        else hipLaunchKernelGGL((conv_thin_n3_kernel<4, TG>), grid, dim3(256), bytes, st, g, in, w, d1, swap, out, in_relu, stats, passes);
        return launch_status("conv_thin_n3");
    }
    if (co == 2) hipLaunchKernelGGL((conv_thin_n_kernel<2, TG>), grid, dim3(256), bytes, st, g, in, w, d1, swap, out, in_relu, stats, passes);
    else if (co == 4) hipLaunchKernelGGL((conv_thin_n_kernel<4, TG>), grid, dim3(256), bytes, st, g, in, w, d1, swap, out, in_relu, stats, passes);
    else hipLaunchKernelGGL((conv_thin_n_kernel<8, TG>), grid, dim3(256), bytes, st, g, in, w, d1, swap, out, in_relu, stats, passes);
    return launch_status("conv_thin_n");
}
template int launch_thin_n<false>(const GatherGeom&, const float*, const float*, int, int, float*, int, double*, hipStream_t);
template int launch_thin_n<true>(const GatherGeom&, const float*, const float*, int, int, float*, int, double*, hipStream_t);

// ---------------------------------------------------------------------------------------------
// thin-N weight gradient: B <= BB (2 or 4) gradient channels, A = 4*Q fat channels, KS x KS taps.
//   part[block][(b*A + a)*taps + t] = sum over the block's pixels of I[p*s - pad + k*d][a] * G[p][b]
// thread = (pixel lane, 4 channels of I) with taps*4*BB partial sums in registers; folded over the pixel
// lanes by shuffles and over the 4 waves through LDS.  A second launch (dwconv_wgrad_sum_kernel) adds the
// blocks' partials in a fixed order: no atomics, bitwise reproducible.
template <int KS, int BB>
__global__ __launch_bounds__(256) void wgrad_thin_n_kernel(WgradGeom g, const float* __restrict__ I,
                                                           const float* __restrict__ G, float* __restrict__ part,
                                                           int i_relu, int g_relu) {
    constexpr int TAPS = KS * KS;
    extern __shared__ __attribute__((aligned(16))) float red[];     // [4 waves x row slots][Q][TAPS*4*BB]
    const int Q = g.A >> 2, q = threadIdx.x & (Q - 1), pl = threadIdx.x / Q, lanes = 256 / Q;
    const int per_img = g.hg * g.wg;
    const long total = (long)g.n * per_img;
    // XCD-aware order: an XCD (workgroups b, b + 8, ...) takes a contiguous eighth of the pixels, so the taps' re-reads of
    // neighbouring rows meet in one L2; the partial row stays indexed by blockIdx.x (any order: the sum launch adds them all)
    unsigned bx = blockIdx.x;
    if ((gridDim.x & 7u) == 0) bx = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    long p0 = (long)bx * g.chunk, p1 = p0 + g.chunk;
    if (p1 > total) p1 = total;
    float acc[TAPS][4][BB];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < BB; ++j) acc[t][i][j] = 0.f;
    for (long p = p0 + pl; p < p1; p += lanes) {
        const int n = (int)(p / per_img), r = (int)(p - (long)n * per_img);
        const int gy = r / g.wg, gx = r - gy * g.wg;
        float gv[BB];
#pragma unroll
        for (int j = 0; j < BB; ++j) {
            gv[j] = j < g.B ? G[(size_t)p * g.B + j] : 0.f;
            if (g_relu) gv[j] = fmaxf(gv[j], 0.f);
        }
        const float* In = I + (size_t)n * g.hi * g.wi * g.A + q * 4;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int iy = gy * g.stride - g.pad + ky * g.dil;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int ix = gx * g.stride - g.pad + kx * g.dil;
                const bool ok = iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi;
                float iv[4];
                ldv<4>(In + (ok ? (size_t)(iy * g.wi + ix) * g.A : 0), iv);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = ok ? iv[i] : 0.f;
                    if (i_relu) v = fmaxf(v, 0.f);
#pragma unroll
                    for (int j = 0; j < BB; ++j) acc[ky * KS + kx][i][j] = fmaf(v, gv[j], acc[ky * KS + kx][i][j]);
                }
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slots = wave_slots(Q), nparts = 4 * slots;
    const bool holder = lane_holds_partial(lane, Q);
    float* dst = red + (size_t)((wave * slots + lane_slot(lane, Q)) * Q + q) * TAPS * 4 * BB;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < BB; ++j) {
                const float v = row_strided_sum(acc[t][i][j], Q);
                if (holder) dst[t * 4 * BB + i * BB + j] = v;
            }
    __syncthreads();
    const int n_elem = g.B * g.A * TAPS;
    for (int e = threadIdx.x; e < n_elem; e += 256) {           // e = (b*A + a)*TAPS + t
        const int t = e % TAPS, a = (e / TAPS) % g.A, b = e / (TAPS * g.A);
        const int qq = a >> 2, i = a & 3;
        float v = 0.f;
        for (int wv = 0; wv < nparts; ++wv) v += red[((size_t)(wv * Q + qq) * TAPS + t) * 4 * BB + i * BB + b];
        part[(size_t)blockIdx.x * n_elem + e] = v;
    }
}

bool thin_n_wgrad_ok(const WgradGeom& g) {
    const int q = g.A >> 2;
    return g.B <= 4 && g.A % 4 == 0 && q >= 1 && q <= 16 && (q & (q - 1)) == 0 && g.kh == g.kw && (g.kh == 1 || g.kh == 3);
}

// (at most 2048 partial rows: 512 left two waves per SIMD walking 32 dependent iterations each on the head's 256 x 256 maps -- 58 us
// for one pass over 67 MB)
int64_t thin_n_wgrad_ws_bytes(const WgradGeom& g) { return (int64_t)2048 * g.B * g.A * g.kh * g.kw * sizeof(float); }

// part: >= thin_n_wgrad_ws_bytes(g); returns the number of partial rows (blocks) written
int launch_thin_n_wgrad(WgradGeom g, const float* I, const float* G, float* part, int i_relu, int g_relu, int* nblk_out,
                        hipStream_t st) {
    const long total = (long)g.n * g.hg * g.wg;
    long nblk = (total + 255) / 256;
    if (nblk > 2048) nblk = 2048;
    g.chunk = (int)((total + nblk - 1) / nblk);
    nblk = (total + g.chunk - 1) / g.chunk;
    const int bb = g.B <= 2 ? 2 : 4;
    const int q4 = g.A >> 2;
    const size_t bytes = (size_t)4 * (q4 <= 16 ? 4 : 64 / q4) * q4 * g.kh * g.kw * 4 * bb * sizeof(float);
#define SENAS_TW(KS, BB) hipLaunchKernelGGL((wgrad_thin_n_kernel<KS, BB>), dim3((unsigned)nblk), dim3(256), bytes, st, g, I, G, part, i_relu, g_relu)
    if (g.kh == 3) { if (bb == 2) SENAS_TW(3, 2); else SENAS_TW(3, 4); }
    else { if (bb == 2) SENAS_TW(1, 2); else SENAS_TW(1, 4); }
#undef SENAS_TW
    *nblk_out = (int)nblk;
    return launch_status("wgrad_thin_n");
}

// ---------------------------------------------------------------------------------------------
// weight gradient with BOTH sides narrow: A = 4 or 8 fine-grid channels, B <= BB gradient channels (BB = 8: the
// supernet's 8 -> 8 inner edges; BB = 16: two of them stacked; the 32-wide MFMA tile would be 1/16 - 1/8 full).
// thread = (role, pixel lane), role = (tap, 4 channels of A): 4 x BB partial sums in registers; the pixel lanes of a
// role meet in LDS, the blocks' partial rows in the sum launch.
//   part[block][(b*A + a)*taps + t]
template <int BB>
__global__ __launch_bounds__(256) void wgrad_c8_kernel(WgradGeom g, const float* __restrict__ I, const float* __restrict__ G,
                                                       float* __restrict__ part, int i_relu, int g_relu) {
    __shared__ float red[256 * 4 * BB];
    const int taps = g.kh * g.kw, a4s = g.A >> 2;
    const int R = taps * a4s, L = 256 / R;
    const int role = threadIdx.x % R, pl = threadIdx.x / R;
    const int tap = role / a4s, a4 = role - tap * a4s;
    const int ky = tap / g.kw, kx = tap - ky * g.kw;
    const int per_img = g.hg * g.wg;
    const long total = (long)g.n * per_img;
    long p0 = (long)blockIdx.x * g.chunk, p1 = p0 + g.chunk;
    if (p1 > total) p1 = total;
    float acc[4][BB];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < BB; ++j) acc[i][j] = 0.f;
    if (pl < L) {
        for (long p = p0 + pl; p < p1; p += L) {
            const int n = (int)(p / per_img), r = (int)(p - (long)n * per_img);
            const int gy = r / g.wg, gx = r - gy * g.wg;
            const int iy = gy * g.stride - g.pad + ky * g.dil, ix = gx * g.stride - g.pad + kx * g.dil;
            if (iy < 0 || iy >= g.hi || ix < 0 || ix >= g.wi) continue;
            float gv[BB];
            if (g.B == BB) {
#pragma unroll
                for (int j = 0; j < BB; j += 4) ldv<4>(G + (size_t)p * BB + j, reinterpret_cast<float(&)[4]>(gv[j]));
            } else {
#pragma unroll
                for (int j = 0; j < BB; ++j) gv[j] = j < g.B ? G[(size_t)p * g.B + j] : 0.f;
            }
            float iv[4];
            ldv<4>(I + ((size_t)(n * g.hi + iy) * g.wi + ix) * g.A + a4 * 4, iv);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = i_relu ? fmaxf(iv[i], 0.f) : iv[i];
#pragma unroll
                for (int j = 0; j < BB; ++j) acc[i][j] = fmaf(v, g_relu ? fmaxf(gv[j], 0.f) : gv[j], acc[i][j]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < BB; ++j) red[threadIdx.x * (4 * BB) + i * BB + j] = acc[i][j];
    __syncthreads();
    const int n_elem = g.B * g.A * taps;
    for (int e = threadIdx.x; e < n_elem; e += 256) {           // e = (b*A + a)*taps + t
        const int t = e % taps, a = (e / taps) % g.A, b = e / (taps * g.A);
        const int ro = t * a4s + (a >> 2), idx = (a & 3) * BB + b;
        float v = 0.f;
        for (int k = 0; k < L; ++k) v += red[(k * R + ro) * (4 * BB) + idx];
        part[(size_t)blockIdx.x * n_elem + e] = v;
    }
}

bool wgrad_c8_ok(const WgradGeom& g) {
    return (g.A == 4 || g.A == 8) && g.B >= 1 && g.B <= 16 && g.kh * g.kw * (g.A >> 2) <= 128;
}

static long wgrad_c8_blocks(const WgradGeom& g, int& chunk) {
    const long total = (long)g.n * g.hg * g.wg;
    const int L = 256 / (g.kh * g.kw * (g.A >> 2));
    long nblk = (total + 16L * L - 1) / (16L * L);              // ~16 pixels per thread
    if (nblk > 1024) nblk = 1024;
    if (nblk < 1) nblk = 1;
    chunk = (int)((total + nblk - 1) / nblk);
    return (total + chunk - 1) / chunk;
}

int64_t wgrad_c8_ws_bytes(const WgradGeom& g) {
    int chunk;
    return (int64_t)wgrad_c8_blocks(g, chunk) * g.A * g.B * g.kh * g.kw * sizeof(float);
}

int launch_wgrad_c8(WgradGeom g, const float* I, const float* G, float* part, int i_relu, int g_relu, int* nblk_out, hipStream_t st) {
    int chunk;
    const long nblk = wgrad_c8_blocks(g, chunk);
    g.chunk = chunk;
    if (g.B <= 8) hipLaunchKernelGGL(wgrad_c8_kernel<8>, dim3((unsigned)nblk), dim3(256), 0, st, g, I, G, part, i_relu, g_relu);
    else hipLaunchKernelGGL(wgrad_c8_kernel<16>, dim3((unsigned)nblk), dim3(256), 0, st, g, I, G, part, i_relu, g_relu);
    *nblk_out = (int)nblk;
    return launch_status("wgrad_c8");
}

}  // namespace senas
