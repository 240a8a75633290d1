// Elementwise + per-channel-reduction kernels: ReLU, batch-norm statistics / finalize, and the
// fused "normalise, gate, mix, add, activate" node kernel with its backward.  NHWC fp32, gfx950.
// All of these are HBM-bound: 16-byte loads along the channel axis, every tensor read once.
#include "common.h"

namespace senas {

struct PtrTable {
    const float* p[SENAS_MAX_TERMS];
};
struct MutPtrTable {
    float* p[SENAS_MAX_TERMS];
};

// ------------------------------------------------------------------------------------- ReLU
__global__ void relu_fwd_kernel(long n4, long n, const float* __restrict__ x, float* __restrict__ y) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        reinterpret_cast<float4*>(y)[i] = v;
    }
    if (blockIdx.x == 0) for (long j = n4 * 4 + threadIdx.x; j < n; j += blockDim.x) y[j] = fmaxf(x[j], 0.f);
}

__global__ void relu_bwd_kernel(long n4, long n, const float* __restrict__ dy, const float* __restrict__ y,
                                float* __restrict__ dx) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 g = reinterpret_cast<const float4*>(dy)[i];
        float4 v = reinterpret_cast<const float4*>(y)[i];
        g.x = v.x > 0.f ? g.x : 0.f; g.y = v.y > 0.f ? g.y : 0.f; g.z = v.z > 0.f ? g.z : 0.f; g.w = v.w > 0.f ? g.w : 0.f;
        reinterpret_cast<float4*>(dx)[i] = g;
    }
    if (blockIdx.x == 0) for (long j = n4 * 4 + threadIdx.x; j < n; j += blockDim.x) dx[j] = y[j] > 0.f ? dy[j] : 0.f;
}

// ------------------------------------------------------------------------------------- channel statistics
// x: [n][hw][c].  block = rows x c lanes (rows = 256 / c), grid = (chunks, n).
// per-image per-channel sum / sumsq in fp64, one atomic pair per block and channel.
__global__ __launch_bounds__(256) void chan_stats_kernel(long hw, int c, long chunk, const float* __restrict__ x,
                                                         double* __restrict__ stats) {
    __shared__ double red[2][256];
    const int rows = 256 / c;
    const int ch = threadIdx.x % c, row = threadIdx.x / c;
    const int n = blockIdx.y;
    long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk;
    if (p1 > hw) p1 = hw;
    double s = 0.0, q = 0.0;
    if (row < rows) {
        const float* xp = x + (size_t)n * hw * c + ch;
        for (long p = p0 + row; p < p1; p += rows) {
            const float v = xp[p * c];
            s += v;
            q += (double)v * v;
        }
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (row == 0) {
        for (int r = 1; r < rows; ++r) { s += red[0][r * c + ch]; q += red[1][r * c + ch]; }
        double* st = stats + ((size_t)n * c + ch) * 2;
        atomicAdd(st, s);
        atomicAdd(st + 1, q);
    }
}

// more than 256 channels is not on the path; channel counts are 1..256
__global__ void bn_finalize_kernel(int n, long hw, int c, const double* __restrict__ stats,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rmean, float* __restrict__ rvar, int64_t* __restrict__ nbt,
                                   float momentum, float eps, int training, float* __restrict__ mean_o,
                                   float* __restrict__ invstd_o, float* __restrict__ scale_o, float* __restrict__ shift_o) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch == 0 && training && nbt != nullptr) *nbt += 1;
    if (ch >= c) return;
    float mean, invstd;
    if (training) {
        double s = 0.0, q = 0.0;
        for (int i = 0; i < n; ++i) { s += stats[((size_t)i * c + ch) * 2]; q += stats[((size_t)i * c + ch) * 2 + 1]; }
        const double m = (double)n * (double)hw;
        const double mu = s / m;
        double var = q / m - mu * mu;
        if (var < 0.0) var = 0.0;
        mean = (float)mu;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (rmean != nullptr) {
            const double unbiased = m > 1.0 ? var * m / (m - 1.0) : var;
            rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * mean;
            rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * (float)unbiased;
        }
    } else {
        mean = rmean[ch];
        invstd = 1.f / sqrtf(rvar[ch] + eps);
    }
    const float sc = gamma[ch] * invstd;
    mean_o[ch] = mean;
    invstd_o[ch] = invstd;
    scale_o[ch] = sc;
    shift_o[ch] = beta[ch] - mean * sc;
}

// ------------------------------------------------------------------------------------- combine
// y = act(sum_t coef[t][n][c] * z_t + bias[n][c] + residual).  One thread = V channels of one pixel.
template <int V>
__global__ __launch_bounds__(256) void combine_fwd_kernel(long hw, int c, int nterms, PtrTable z,
                                                          const float* __restrict__ coef, const float* __restrict__ bias,
                                                          const float* __restrict__ residual, int relu,
                                                          float* __restrict__ y, int nimg) {
    const int cv = c / V;
    const long per_img = hw * cv;
    const int n = blockIdx.y;
    const size_t img_off = (size_t)n * hw * c;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_img; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % cv) * V;
        const size_t off = img_off + (size_t)(i / cv) * c + ch;
        float acc[V], tmp[V], kf[V];
        ldv<V>(bias + n * c + ch, acc);
        if (residual != nullptr) {
            ldv<V>(residual + off, tmp);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += tmp[j];
        }
        for (int t = 0; t < nterms; ++t) {
            ldv<V>(z.p[t] + off, tmp);
            ldv<V>(coef + ((size_t)t * nimg + n) * c + ch, kf);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] = fmaf(kf[j], tmp[j], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = relu ? fmaxf(acc[j], 0.f) : acc[j];
        stv<V>(y + off, acc);
    }
}

// backward reductions: p1[n][c] += sum_p ds ; p2[t][n][c] += sum_p ds * z_t   (t in [t0, t0+TT))
template <int TT>
__global__ __launch_bounds__(256) void combine_bwd_reduce_kernel(long hw, int c, long chunk, int t0, int tt, int nimg,
                                                                 PtrTable z, const float* __restrict__ dy,
                                                                 const float* __restrict__ y, int relu, int do_p1,
                                                                 double* __restrict__ p1, double* __restrict__ p2) {
    __shared__ double red[256];
    const int rows = 256 / c;
    const int ch = threadIdx.x % c, row = threadIdx.x / c;
    const int n = blockIdx.y;
    long q0 = (long)blockIdx.x * chunk, q1 = q0 + chunk;
    if (q1 > hw) q1 = hw;
    double a1 = 0.0, a2[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) a2[t] = 0.0;
    if (row < rows) {
        const size_t base = (size_t)n * hw * c + ch;
        for (long p = q0 + row; p < q1; p += rows) {
            const size_t o = base + (size_t)p * c;
            float ds = dy[o];
            if (relu && !(y[o] > 0.f)) ds = 0.f;
            a1 += ds;
#pragma unroll
            for (int t = 0; t < TT; ++t) if (t < tt) a2[t] += (double)ds * (double)z.p[t0 + t][o];
        }
    }
    // cross-row reduction through LDS, one quantity at a time
    auto reduce_to = [&](double v, double* dst) {
        __syncthreads();
        red[threadIdx.x] = v;
        __syncthreads();
        if (row == 0) {
            for (int r = 1; r < rows; ++r) v += red[r * c + ch];
            atomicAdd(dst, v);
        }
    };
    if (do_p1) reduce_to(a1, p1 + (size_t)n * c + ch);
#pragma unroll
    for (int t = 0; t < TT; ++t) if (t < tt) reduce_to(a2[t], p2 + ((size_t)(t0 + t) * nimg + n) * c + ch);
}

// backward apply: dz_t = a*ds + b*z_t + k ; optional ds_out
template <int V>
__global__ __launch_bounds__(256) void combine_bwd_apply_kernel(long hw, int c, int nterms, int nimg, PtrTable z,
                                                                const float* __restrict__ dy, const float* __restrict__ y,
                                                                int relu, const float* __restrict__ A,
                                                                const float* __restrict__ B, const float* __restrict__ K,
                                                                MutPtrTable dz, float* __restrict__ ds_out) {
    const int cv = c / V;
    const long per_img = hw * cv;
    const int n = blockIdx.y;
    const size_t img_off = (size_t)n * hw * c;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_img; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % cv) * V;
        const size_t off = img_off + (size_t)(i / cv) * c + ch;
        float ds[V], yv[V], zv[V], av[V], bv[V], kv[V];
        ldv<V>(dy + off, ds);
        if (relu) {
            ldv<V>(y + off, yv);
#pragma unroll
            for (int j = 0; j < V; ++j) if (!(yv[j] > 0.f)) ds[j] = 0.f;
        }
        if (ds_out != nullptr) stv<V>(ds_out + off, ds);
        for (int t = 0; t < nterms; ++t) {
            float* out = dz.p[t];
            if (out == nullptr) continue;
            const size_t ko = ((size_t)t * nimg + n) * c + ch;
            ldv<V>(z.p[t] + off, zv);
            ldv<V>(A + ko, av);
            ldv<V>(B + ko, bv);
            ldv<V>(K + ko, kv);
#pragma unroll
            for (int j = 0; j < V; ++j) zv[j] = fmaf(av[j], ds[j], fmaf(bv[j], zv[j], kv[j]));
            stv<V>(out + off, zv);
        }
    }
}

static inline unsigned stream_grid(long work) {
    long b = (work + 255) / 256;
    if (b > 2048) b = 2048;   // 256 CUs x 8 blocks, grid-stride the rest
    if (b < 1) b = 1;
    return (unsigned)b;
}

static long reduce_chunk(long hw, int n) {
    // aim for >= ~1024 blocks in total, but at least 256 pixels per block
    long blocks_per_img = 1024 / (n > 0 ? n : 1);
    if (blocks_per_img < 1) blocks_per_img = 1;
    long chunk = (hw + blocks_per_img - 1) / blocks_per_img;
    if (chunk < 256) chunk = 256;
    return chunk;
}

// ---------------------------------------------------------------------------------------------
// out = sum of n tensors (n <= SENAS_MAX_TERMS): the gradient of a tensor with n consumers in ONE pass, instead of
// autograd's n-1 binary accumulations (each a read-read-write pass).
struct SumTable {
    const float* p[SENAS_MAX_TERMS];
};
__global__ __launch_bounds__(256) void sum_n_kernel(SumTable t, int n, long n4, long numel, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 a = reinterpret_cast<const float4*>(t.p[0])[i];
        for (int k = 1; k < n; ++k) {
            const float4 b = reinterpret_cast<const float4*>(t.p[k])[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        reinterpret_cast<float4*>(out)[i] = a;
    }
    if (blockIdx.x == 0 && threadIdx.x < (numel & 3)) {          // tail (numel % 4 floats)
        const long i = (numel & ~3L) + threadIdx.x;
        float a = t.p[0][i];
        for (int k = 1; k < n; ++k) a += t.p[k][i];
        out[i] = a;
    }
}

// the same with sources that are channel slices of wider NHWC tensors (pixel stride s_k floats >= c): the gradient of a
// tensor one of whose consumers is a torch.cat along channels arrives as such a slice
struct SumTableS {
    const float* p[SENAS_MAX_TERMS];
    int s4[SENAS_MAX_TERMS];            // pixel stride in 16-byte pieces
};
__global__ __launch_bounds__(256) void sum_n_strided_kernel(SumTableS t, int n, long total4, int cq, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long pix = i / cq;
        const int q = (int)(i - pix * cq);
        float4 a = reinterpret_cast<const float4*>(t.p[0])[pix * t.s4[0] + q];
        for (int k = 1; k < n; ++k) {
            const float4 b = reinterpret_cast<const float4*>(t.p[k])[pix * t.s4[k] + q];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        reinterpret_cast<float4*>(out)[i] = a;
    }
}

// ---- gamma-gated blend of two skip candidates (search/senas_search.py:98-102): y = g[0] * x1 + g[1] * x2, g on the device
__global__ __launch_bounds__(256) void blend2_fwd_kernel(const float4* __restrict__ x1, const float4* __restrict__ x2,
                                                         const float* __restrict__ g, float4* __restrict__ y, long n4) {
    const float a = g[0], b = g[1];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 u = x1[i], v = x2[i];
        y[i] = make_float4(fmaf(a, u.x, b * v.x), fmaf(a, u.y, b * v.y), fmaf(a, u.z, b * v.z), fmaf(a, u.w, b * v.w));
    }
}

// dx1 = g[0] * dy, dx2 = g[1] * dy (either may be NULL), dg[0] += sum dy * x1, dg[1] += sum dy * x2 (fp64, caller zeroes)
__global__ __launch_bounds__(256) void blend2_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ x1,
                                                         const float4* __restrict__ x2, const float* __restrict__ g,
                                                         float4* __restrict__ dx1, float4* __restrict__ dx2,
                                                         double* __restrict__ dg, long n4) {
    __shared__ double red[2][4];
    const float a = g[0], b = g[1];
    double s1 = 0.0, s2 = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 d = dy[i], u = x1[i], v = x2[i];
        if (dx1 != nullptr) dx1[i] = make_float4(a * d.x, a * d.y, a * d.z, a * d.w);
        if (dx2 != nullptr) dx2[i] = make_float4(b * d.x, b * d.y, b * d.z, b * d.w);
        s1 += (double)d.x * u.x + (double)d.y * u.y + (double)d.z * u.z + (double)d.w * u.w;
        s2 += (double)d.x * v.x + (double)d.y * v.y + (double)d.z * v.z + (double)d.w * v.w;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(dg, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(dg + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// ---- in0 of up cell (i, j) of the supernet in ONE pass (search/senas_search.py:96-103: torch.cat of the column's down-path
// output and the gamma-gated blends of neighbouring outputs below it): m tensors x_0 .. x_{m-1} of [npix][c] in, one
// [npix][m * c] tensor out -- slice 0 = x_0, slice k = g_k[0] * x_{k-1} + g_k[1] * x_k (g_k = row idx[k] of the softmax(gamma)
// table).  Every x_k is read once and every output byte written once (m - 1 blends + a concatenation read x 3m - 2 times and
// write 2m - 1 slices).
struct SkipTab {
    const float* x[SENAS_SKIP_MAX];
    float* dx[SENAS_SKIP_MAX];
    int idx[SENAS_SKIP_MAX];
};

template <int M>
__global__ __launch_bounds__(256) void skipcat_fwd_kernel(SkipTab t, const float* __restrict__ table, float4* __restrict__ y,
                                                          long total4, int cq, int m_rt) {
    const int m = M > 0 ? M : m_rt;
    float ga[SENAS_SKIP_MAX], gb[SENAS_SKIP_MAX];
#pragma unroll
    for (int k = 0; k < SENAS_SKIP_MAX; ++k) {
        ga[k] = gb[k] = 0.f;
        if (k >= 1 && k < m) { ga[k] = table[2 * t.idx[k]]; gb[k] = table[2 * t.idx[k] + 1]; }
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long pix = i / cq;
        const int q = (int)(i - pix * cq);
        float4* o = y + pix * (long)(m * cq) + q;
        float4 v[SENAS_SKIP_MAX];
#pragma unroll
        for (int k = 0; k < (M > 0 ? M : SENAS_SKIP_MAX); ++k)
            if (k < m) v[k] = reinterpret_cast<const float4*>(t.x[k])[i];
        o[0] = v[0];
#pragma unroll
        for (int k = 1; k < (M > 0 ? M : SENAS_SKIP_MAX); ++k)
            if (k < m) {
                const float a = ga[k], b = gb[k];
                const float4 u = v[k - 1], w = v[k];
                o[(long)k * cq] = make_float4(fmaf(a, u.x, b * w.x), fmaf(a, u.y, b * w.y), fmaf(a, u.z, b * w.z), fmaf(a, u.w, b * w.w));
            }
    }
}

// its backward pass in one more: d x_k = [k == 0] dy_0 + [k >= 1] g_k[1] dy_k + [k + 1 < m] g_{k+1}[0] dy_{k+1} (NULL: not
// wanted), acc[idx[k]] += (sum dy_k x_{k-1}, sum dy_k x_k) in fp64 (the caller zeroes the table once per pass)
template <int M>
__global__ __launch_bounds__(256) void skipcat_bwd_kernel(SkipTab t, const float* __restrict__ table, const float4* __restrict__ dy,
                                                          double* __restrict__ acc, long total4, int cq, int m_rt) {
    constexpr int MM = M > 0 ? M : SENAS_SKIP_MAX;
    const int m = M > 0 ? M : m_rt;
    __shared__ double red[2 * SENAS_SKIP_MAX][4];
    float ga[MM + 1], gb[MM + 1];
    double s[2 * MM];
#pragma unroll
    for (int k = 0; k < MM; ++k) {
        s[2 * k] = s[2 * k + 1] = 0.0;
        ga[k] = gb[k] = 0.f;
        if (k >= 1 && k < m) { ga[k] = table[2 * t.idx[k]]; gb[k] = table[2 * t.idx[k] + 1]; }
    }
    ga[MM] = gb[MM] = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long pix = i / cq;
        const int q = (int)(i - pix * cq);
        const float4* d0 = dy + pix * (long)(m * cq) + q;
        float4 d[MM + 1], x[MM];
#pragma unroll
        for (int k = 0; k < MM; ++k)
            if (k < m) { d[k] = d0[(long)k * cq]; x[k] = reinterpret_cast<const float4*>(t.x[k])[i]; }
            else d[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        d[MM] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < MM; ++k)
            if (k < m) {
                const float b = k == 0 ? 1.f : gb[k];
                const float a = (k + 1 < m) ? ga[k + 1] : 0.f;
                const float4 u = d[k], w = d[k + 1];
                if (t.dx[k] != nullptr)
                    reinterpret_cast<float4*>(t.dx[k])[i] = make_float4(fmaf(a, w.x, b * u.x), fmaf(a, w.y, b * u.y), fmaf(a, w.z, b * u.z),
                                                                        fmaf(a, w.w, b * u.w));
                if (k >= 1) {
                    const float4 p = x[k - 1], c = x[k];
                    s[2 * k] += (double)u.x * p.x + (double)u.y * p.y + (double)u.z * p.z + (double)u.w * p.w;
                    s[2 * k + 1] += (double)u.x * c.x + (double)u.y * c.y + (double)u.z * c.z + (double)u.w * c.w;
                }
            }
    }
#pragma unroll
    for (int k = 1; k < MM; ++k)
        if (k < m) {
            const double s0 = wave_sum(s[2 * k]), s1 = wave_sum(s[2 * k + 1]);
            if ((threadIdx.x & 63) == 0) { red[2 * k][threadIdx.x >> 6] = s0; red[2 * k + 1][threadIdx.x >> 6] = s1; }
        }
    __syncthreads();
    if (threadIdx.x >= 2 && threadIdx.x < 2 * m) {
        const int k = threadIdx.x >> 1, h = threadIdx.x & 1;
        atomicAdd(acc + 2 * t.idx[k] + h, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
    }
}

template <int M>
static void skipcat_fwd_launch(const SkipTab& t, const float* table, float* y, long total4, int cq, int m, hipStream_t s) {
    hipLaunchKernelGGL(skipcat_fwd_kernel<M>, dim3(stream_grid(total4)), dim3(256), 0, s, t, table, reinterpret_cast<float4*>(y), total4, cq, m);
}
template <int M>
static void skipcat_bwd_launch(const SkipTab& t, const float* table, const float* dy, double* acc, long total4, int cq, int m, hipStream_t s) {
    long blocks = (total4 + 255) / 256;
    if (blocks > 256) blocks = 256;                                       // (every block ends in 2 (m - 1) fp64 atomics on the same few addresses)
    hipLaunchKernelGGL(skipcat_bwd_kernel<M>, dim3((unsigned)blocks), dim3(256), 0, s, t, table, reinterpret_cast<const float4*>(dy), acc, total4, cq, m);
}

}  // namespace senas

using namespace senas;

static int skipcat_table(SkipTab& t, int64_t npix, int c, int m, const float* const* xs, const int32_t* idx, int rows) {
    SENAS_REQUIRE(npix > 0 && c >= 4 && c % 4 == 0 && m >= 2 && m <= SENAS_SKIP_MAX && xs && idx && rows > 0, "skipcat: bad argument (2 <= m <= SENAS_SKIP_MAX, c % 4 == 0)");
    for (int k = 0; k < m; ++k) {
        SENAS_REQUIRE(xs[k] && (reinterpret_cast<uintptr_t>(xs[k]) & 15) == 0, "skipcat: null or misaligned source");
        SENAS_REQUIRE(k == 0 || (idx[k] >= 0 && idx[k] < rows), "skipcat: gamma row out of range");
        t.x[k] = xs[k];
        t.idx[k] = k == 0 ? 0 : idx[k];
    }
    return SENAS_OK;
}

extern "C" int senas_skipcat_fwd(int64_t npix, int c, int m, const float* const* xs, const float* table, int rows, const int32_t* idx,
                                 float* y, void* stream) {
    SkipTab t{};
    const int rc = skipcat_table(t, npix, c, m, xs, idx, rows);
    if (rc != SENAS_OK) return rc;
    SENAS_REQUIRE(table && y && (reinterpret_cast<uintptr_t>(y) & 15) == 0, "skipcat_fwd: null or misaligned destination");
    const long total4 = (long)npix * (c / 4);
    hipStream_t s = as_stream(stream);
    switch (m) {
        case 2: skipcat_fwd_launch<2>(t, table, y, total4, c / 4, m, s); break;
        case 3: skipcat_fwd_launch<3>(t, table, y, total4, c / 4, m, s); break;
        case 4: skipcat_fwd_launch<4>(t, table, y, total4, c / 4, m, s); break;
        default: skipcat_fwd_launch<0>(t, table, y, total4, c / 4, m, s); break;
    }
    return launch_status("skipcat_fwd");
}

extern "C" int senas_skipcat_bwd(int64_t npix, int c, int m, const float* dy, const float* const* xs, const float* table, int rows,
                                 const int32_t* idx, float* const* dxs, double* acc, void* stream) {
    SkipTab t{};
    const int rc = skipcat_table(t, npix, c, m, xs, idx, rows);
    if (rc != SENAS_OK) return rc;
    SENAS_REQUIRE(table && dy && dxs && acc && (reinterpret_cast<uintptr_t>(dy) & 15) == 0, "skipcat_bwd: null or misaligned argument");
    for (int k = 0; k < m; ++k) {
        SENAS_REQUIRE((reinterpret_cast<uintptr_t>(dxs[k]) & 15) == 0, "skipcat_bwd: misaligned gradient");
        t.dx[k] = dxs[k];
    }
    const long total4 = (long)npix * (c / 4);
    hipStream_t s = as_stream(stream);
    switch (m) {
        case 2: skipcat_bwd_launch<2>(t, table, dy, acc, total4, c / 4, m, s); break;
        case 3: skipcat_bwd_launch<3>(t, table, dy, acc, total4, c / 4, m, s); break;
        case 4: skipcat_bwd_launch<4>(t, table, dy, acc, total4, c / 4, m, s); break;
        default: skipcat_bwd_launch<0>(t, table, dy, acc, total4, c / 4, m, s); break;
    }
    return launch_status("skipcat_bwd");
}

extern "C" int senas_blend2_fwd(int64_t numel, const float* x1, const float* x2, const float* g, float* y, void* stream) {
    SENAS_REQUIRE(x1 && x2 && g && y && numel > 0 && numel % 4 == 0, "blend2_fwd: bad argument (numel % 4 == 0)");
    SENAS_REQUIRE(((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)y) % 16 == 0, "blend2_fwd: tensors must be 16-byte aligned");
    const long n4 = numel / 4;
    hipLaunchKernelGGL(blend2_fwd_kernel, dim3(stream_grid(n4)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4*>(x1),
                       reinterpret_cast<const float4*>(x2), g, reinterpret_cast<float4*>(y), n4);
    return launch_status("blend2_fwd");
}

extern "C" int senas_blend2_bwd(int64_t numel, const float* dy, const float* x1, const float* x2, const float* g, float* dx1,
                                float* dx2, double* dg, void* stream) {
    SENAS_REQUIRE(dy && x1 && x2 && g && dg && numel > 0 && numel % 4 == 0, "blend2_bwd: bad argument (numel % 4 == 0)");
    SENAS_REQUIRE(((uintptr_t)dy | (uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)dx1 | (uintptr_t)dx2) % 16 == 0,
                  "blend2_bwd: tensors must be 16-byte aligned");
    const long n4 = numel / 4;
    long blocks = (n4 + 255) / 256;
    if (blocks > 128) blocks = 128;                                    // every block ends in two fp64 atomics on the SAME pair: 512 blocks spent 15 of their 19 us queueing there
    hipLaunchKernelGGL(blend2_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4*>(dy),
                       reinterpret_cast<const float4*>(x1), reinterpret_cast<const float4*>(x2), g, reinterpret_cast<float4*>(dx1),
                       reinterpret_cast<float4*>(dx2), dg, n4);
    return launch_status("blend2_bwd");
}


extern "C" int senas_relu_fwd(int64_t numel, const float* x, float* y, void* stream) {
    SENAS_REQUIRE(x && y && numel >= 0, "relu_fwd: bad argument");
    if (numel == 0) return SENAS_OK;
    const long n4 = ((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0) ? numel / 4 : 0;
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(stream_grid(n4 > 0 ? n4 : 256)), dim3(256), 0, as_stream(stream), n4, (long)numel, x, y);
    return launch_status("relu_fwd");
}

extern "C" int senas_relu_bwd(int64_t numel, const float* dy, const float* y, float* dx, void* stream) {
    SENAS_REQUIRE(dy && y && dx && numel >= 0, "relu_bwd: bad argument");
    if (numel == 0) return SENAS_OK;
    const bool al = (uintptr_t)dy % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)dx % 16 == 0;
    const long n4 = al ? numel / 4 : 0;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(stream_grid(n4 > 0 ? n4 : 256)), dim3(256), 0, as_stream(stream), n4, (long)numel, dy, y, dx);
    return launch_status("relu_bwd");
}

extern "C" int senas_chan_stats(int n, int64_t hw, int c, const float* x, double* stats, void* stream) {
    SENAS_REQUIRE(x && stats && n > 0 && hw > 0 && c > 0 && c <= 256, "chan_stats: bad argument");
    const long chunk = reduce_chunk(hw, n);
    dim3 grid((unsigned)((hw + chunk - 1) / chunk), n);
    hipLaunchKernelGGL(chan_stats_kernel, grid, dim3(256), 0, as_stream(stream), (long)hw, c, chunk, x, stats);
    return launch_status("chan_stats");
}

extern "C" int senas_bn_finalize(int n, int64_t hw, int c, const double* stats, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                 float eps, int training, float* mean, float* invstd, float* scale, float* shift,
                                 void* stream) {
    SENAS_REQUIRE(gamma && beta && mean && invstd && scale && shift && c > 0, "bn_finalize: null pointer");
    SENAS_REQUIRE(training ? stats != nullptr : (running_mean && running_var), "bn_finalize: missing statistics");
    SENAS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running buffers must come in pairs");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 63) / 64), dim3(64), 0, as_stream(stream), n, (long)hw, c, stats, gamma,
                       beta, running_mean, running_var, num_batches_tracked, momentum, eps, training, mean, invstd, scale, shift);
    return launch_status("bn_finalize");
}

extern "C" int senas_combine_fwd(int n, int64_t hw, int c, int nterms, const float* const* z, const float* coef,
                                 const float* bias, const float* residual, int relu, float* y, void* stream) {
    SENAS_REQUIRE(n > 0 && hw > 0 && c > 0 && nterms >= 0 && nterms <= SENAS_MAX_TERMS, "combine_fwd: bad sizes");
    SENAS_REQUIRE(bias && y && (nterms == 0 || (z && coef)), "combine_fwd: null pointer");
    PtrTable tab{};
    for (int t = 0; t < nterms; ++t) { SENAS_REQUIRE(z[t], "combine_fwd: null term"); tab.p[t] = z[t]; }
    const int V = (c % 4 == 0) ? 4 : 1;
    dim3 grid(stream_grid(hw * (c / V)), n);
    if (grid.x * (unsigned)n > 4096) grid.x = (4096 + n - 1) / n;
    if (V == 4) hipLaunchKernelGGL((combine_fwd_kernel<4>), grid, dim3(256), 0, as_stream(stream), (long)hw, c, nterms, tab, coef, bias, residual, relu, y, n);
    else hipLaunchKernelGGL((combine_fwd_kernel<1>), grid, dim3(256), 0, as_stream(stream), (long)hw, c, nterms, tab, coef, bias, residual, relu, y, n);
    return launch_status("combine_fwd");
}

extern "C" int senas_combine_bwd_reduce(int n, int64_t hw, int c, int nterms, const float* const* z, const float* dy,
                                        const float* y, int relu, double* p1, double* p2, void* stream) {
    SENAS_REQUIRE(n > 0 && hw > 0 && c > 0 && c <= 256 && nterms >= 0 && nterms <= SENAS_MAX_TERMS, "combine_bwd_reduce: bad sizes");
    SENAS_REQUIRE(dy && p1 && (!relu || y) && (nterms == 0 || (z && p2)), "combine_bwd_reduce: null pointer");
    PtrTable tab{};
    for (int t = 0; t < nterms; ++t) { SENAS_REQUIRE(z[t], "combine_bwd_reduce: null term"); tab.p[t] = z[t]; }
    const long chunk = reduce_chunk(hw, n);
    dim3 grid((unsigned)((hw + chunk - 1) / chunk), n);
    hipStream_t st = as_stream(stream);
    int t0 = 0, first = 1;
    do {
        const int left = nterms - t0;
        const int tt = left >= 4 ? 4 : left;
        if (tt > 2) hipLaunchKernelGGL((combine_bwd_reduce_kernel<4>), grid, dim3(256), 0, st, (long)hw, c, chunk, t0, tt, n, tab, dy, y, relu, first, p1, p2);
        else if (tt == 2) hipLaunchKernelGGL((combine_bwd_reduce_kernel<2>), grid, dim3(256), 0, st, (long)hw, c, chunk, t0, tt, n, tab, dy, y, relu, first, p1, p2);
        else hipLaunchKernelGGL((combine_bwd_reduce_kernel<1>), grid, dim3(256), 0, st, (long)hw, c, chunk, t0, tt, n, tab, dy, y, relu, first, p1, p2);
        t0 += tt > 0 ? tt : 1;
        first = 0;
    } while (t0 < nterms);
    return launch_status("combine_bwd_reduce");
}

extern "C" int senas_combine_bwd_apply(int n, int64_t hw, int c, int nterms, const float* const* z, const float* dy,
                                       const float* y, int relu, const float* a, const float* b, const float* k,
                                       float* const* dz, float* ds_out, void* stream) {
    SENAS_REQUIRE(n > 0 && hw > 0 && c > 0 && nterms >= 0 && nterms <= SENAS_MAX_TERMS, "combine_bwd_apply: bad sizes");
    SENAS_REQUIRE(dy && (!relu || y) && (nterms == 0 || (z && dz && a && b && k)), "combine_bwd_apply: null pointer");
    PtrTable tab{};
    MutPtrTable out{};
    for (int t = 0; t < nterms; ++t) { tab.p[t] = z[t]; out.p[t] = dz[t]; SENAS_REQUIRE(z[t] || !dz[t], "combine_bwd_apply: null term"); }
    const int V = (c % 4 == 0) ? 4 : 1;
    dim3 grid(stream_grid(hw * (c / V)), n);
    if (grid.x * (unsigned)n > 4096) grid.x = (4096 + n - 1) / n;
    if (V == 4) hipLaunchKernelGGL((combine_bwd_apply_kernel<4>), grid, dim3(256), 0, as_stream(stream), (long)hw, c, nterms, n, tab, dy, y, relu, a, b, k, out, ds_out);
    else hipLaunchKernelGGL((combine_bwd_apply_kernel<1>), grid, dim3(256), 0, as_stream(stream), (long)hw, c, nterms, n, tab, dy, y, relu, a, b, k, out, ds_out);
    return launch_status("combine_bwd_apply");
}

extern "C" int senas_sum_n_strided(int n, int64_t npix, int c, const float* const* srcs, const int32_t* src_pixel_stride, float* out,
                                   void* stream) {
    SENAS_REQUIRE(n >= 1 && n <= SENAS_MAX_TERMS && npix > 0 && c >= 4 && c % 4 == 0 && srcs && src_pixel_stride && out, "sum_n_strided: bad argument");
    senas::SumTableS t{};
    for (int k = 0; k < n; ++k) {
        SENAS_REQUIRE(srcs[k] && (reinterpret_cast<uintptr_t>(srcs[k]) & 15) == 0 && src_pixel_stride[k] >= c && src_pixel_stride[k] % 4 == 0,
                      "sum_n_strided: null, misaligned or too narrow source");
        t.p[k] = srcs[k];
        t.s4[k] = src_pixel_stride[k] / 4;
    }
    SENAS_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "sum_n_strided: misaligned destination");
    const long total4 = (long)npix * (c / 4);
    long blocks = (total4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(senas::sum_n_strided_kernel, dim3((unsigned)blocks), dim3(256), 0, senas::as_stream(stream), t, n, total4, c / 4, out);
    return senas::launch_status("sum_n_strided");
}

extern "C" int senas_sum_n(int n, int64_t numel, const float* const* srcs, float* out, void* stream) {
    SENAS_REQUIRE(n >= 1 && n <= SENAS_MAX_TERMS && numel > 0 && srcs && out, "sum_n: bad argument");
    senas::SumTable t{};
    for (int k = 0; k < n; ++k) {
        SENAS_REQUIRE(srcs[k] && (reinterpret_cast<uintptr_t>(srcs[k]) & 15) == 0, "sum_n: null or misaligned source");
        t.p[k] = srcs[k];
    }
    SENAS_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "sum_n: misaligned destination");
    const long n4 = numel >> 2;
    long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(senas::sum_n_kernel, dim3((unsigned)blocks), dim3(256), 0, senas::as_stream(stream), t, n, n4, (long)numel, out);
    return senas::launch_status("sum_n");
}
