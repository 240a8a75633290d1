// Cell-node kernels: everything between the raw convolution outputs and the node tensor.
//
//   forward : [prepare]  per term: batch statistics -> mean / invstd / scale / shift, running-stat
//                        update, SE gate (squeeze from the statistics, two tiny mat-vecs, sigmoid),
//                        mixing weight folded in  ->  coef[t][n][c], shiftc[t][n][c]
//             [combine]  y = act(sum_t coef * z_t + sum_t shiftc + residual)      (one pass over HBM)
//   backward: [reduce]   p1 = sum ds, p2_t = sum ds * z_t per (n, c)             (one pass)
//             [prepare]  d gamma, d beta, d mix, d SE weights, and the per-(n,c) coefficients of
//                        dz_t = A * ds + B * z_t + K
//             [apply]    writes every dz_t (and ds for a residual input)         (one pass)
//
// The prepare kernels are tiny (one block per term) and replace ~25 / ~40 elementwise launches.
#include "common.h"

#include <stdlib.h>

namespace senas {

constexpr int kMaxMid = 16;      // SE hidden width: c/16, c <= 256

struct NodeDesc {                // device-visible copy of senas_node_desc
    int nterms, n, c, training, relu;
    long hw;
    float eps, momentum;
    const double* stats[SENAS_MAX_TERMS];
    const float* gamma[SENAS_MAX_TERMS];
    const float* beta[SENAS_MAX_TERMS];
    float* rmean[SENAS_MAX_TERMS];
    float* rvar[SENAS_MAX_TERMS];
    int64_t* nbt[SENAS_MAX_TERMS];
    const float* w1[SENAS_MAX_TERMS];
    const float* w2[SENAS_MAX_TERMS];
    int mid[SENAS_MAX_TERMS];
    int sstride[SENAS_MAX_TERMS];   // doubles between the statistics of consecutive images (2c when dense; wider: the term
                                    // is a channel slice of a stacked convolution output and so are its statistics)
    const float* mix;
};

struct ZTable {
    const float* p[SENAS_MAX_TERMS];
    int s[SENAS_MAX_TERMS];         // pixel stride of z_t in ELEMENTS (c when dense; wider: a channel slice of a stacked tensor)
    int bf[SENAS_MAX_TERMS];        // 1: z_t is a bf16 tensor (a bf16-stored convolution output, "bf16s"); few-term vector kernels only
};
struct DzTable {
    float* p[SENAS_MAX_TERMS];
    int s[SENAS_MAX_TERMS];       // pixel stride of dz_t in elements (c when dense; wider: a channel slice of a stacked tensor)
    int bf[SENAS_MAX_TERMS];      // 1: dz_t is written as bf16 (the gradient of a bf16-stored tensor)
};

// four channels of a term that is fp32 or bf16 in memory (element offset `off`)
__device__ __forceinline__ void ldz4(const float* base, size_t off, int bf, float (&v)[4]) {
    if (bf) {
        const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off);
        v[0] = __builtin_bit_cast(float, r.x << 16); v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, r.y << 16); v[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
    } else {
        ldv<4>(base + off, v);
    }
}
__device__ __forceinline__ void stz4(float* base, size_t off, int bf, const float (&v)[4]) {
    if (bf) {
        typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
        typedef float f2_t __attribute__((ext_vector_type(2)));
        const f2_t a = {v[0], v[1]}, b = {v[2], v[3]};
        const uint2 r = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(a, bf2_t)),
                                   __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf2_t)));      // round to nearest even
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + off) = r;
    } else {
        stv<4>(base + off, v);
    }
}
struct SeGradTable {            // per-term gradient destinations (each its own tensor on the host side)
    float* w1[SENAS_MAX_TERMS];
    float* w2[SENAS_MAX_TERMS];
    float* dgamma[SENAS_MAX_TERMS];
    float* dbeta[SENAS_MAX_TERMS];
    int dmix_accumulate;        // d mix is added to what the buffer holds (shared mixing weights of the cells of one kind)
};

// (1 - momentum) * old + momentum * now, the same bits from every kernel below: in fp32 the compiler is free to contract the two
// products and the sum into one FMA or another, differently from kernel to kernel; in fp64 both products of two floats are exact,
// so the sum is rounded once whichever way it is contracted, and once more to fp32
__device__ __forceinline__ float blend_stat(float momentum, float old, float now) {
    return (float)((double)(1.f - momentum) * (double)old + (double)momentum * (double)now);
}

// ------------------------------------------------------------------------------------------ forward prepare
// grid = nterms blocks of 256 threads; c <= 256.  thread = (image row, channel): 256/c images are handled
// side by side, every global operand is requested before the first dependent instruction, and everything
// after that runs out of LDS -- the kernel is one memory round trip long.
// dynamic LDS: part[R][c][2] doubles | zsum[n][c] doubles | m[n][c] | a1[n][kMaxMid] | w1[mid][c] | w2[c][mid] floats
__global__ __launch_bounds__(256) void node_prepare_fwd_kernel(NodeDesc d, float* __restrict__ coefs, float* __restrict__ gate,
                                                               float* __restrict__ coef, float* __restrict__ shiftc,
                                                               float* __restrict__ se_m, float* __restrict__ se_a1) {
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int t = blockIdx.x, n = d.n, c = d.c;
    const int R = 256 / c, ch = threadIdx.x % c, row = threadIdx.x / c;
    const bool act = row < R;
    const double* st = d.stats[t];
    const bool se = d.w1[t] != nullptr;
    const int mid = se ? d.mid[t] : 0;
    double* part = ldsd;                                            // [R][c][2]
    double* zsum_s = part + (size_t)R * c * 2;                      // [n][c]: per-image channel sums (SE squeeze)
    float* m_s = reinterpret_cast<float*>(zsum_s + (size_t)n * c);  // [n][c]
    float* a_s = m_s + (size_t)n * c;                               // [n][kMaxMid]
    float* w1_s = a_s + (size_t)n * kMaxMid;                        // [mid][c]
    float* w2_s = w1_s + (size_t)kMaxMid * c;                       // [c][mid]

    // ---- all global reads up front
    double s = 0.0, q = 0.0;
    if (act && st != nullptr && (d.training || se))
        for (int i = row; i < n; i += R) {
            const double v0 = st[(size_t)i * d.sstride[t] + ch * 2], v1 = st[(size_t)i * d.sstride[t] + ch * 2 + 1];
            s += v0; q += v1;
            if (se) zsum_s[i * c + ch] = v0;
        }
    float gam = 0.f, bet = 0.f, rm = 0.f, rv = 0.f;
    if (threadIdx.x < c) {
        gam = d.gamma[t][ch]; bet = d.beta[t][ch];
        if (d.rmean[t] != nullptr) { rm = d.rmean[t][ch]; rv = d.rvar[t][ch]; }
    }
    const float w = d.mix != nullptr ? d.mix[t] : 1.f;
    if (se) {
        for (int i = threadIdx.x; i < mid * c; i += 256) { w1_s[i] = d.w1[t][i]; w2_s[i] = d.w2[t][i]; }
    }
    if (act) { part[((size_t)row * c + ch) * 2] = s; part[((size_t)row * c + ch) * 2 + 1] = q; }
    __syncthreads();
    // ---- batch statistics -> mean / invstd / scale / shift (threads 0..c-1 own a channel; broadcast through LDS)
    float* bc = reinterpret_cast<float*>(part);                     // reused after the sums are read: [2][c] scale, shift
    float mean = 0.f, invstd = 0.f, scale = 0.f, shift = 0.f;
    if (threadIdx.x < c) {
        if (d.training) {
            double ss = 0.0, qq = 0.0;
            for (int rr = 0; rr < R; ++rr) { ss += part[((size_t)rr * c + ch) * 2]; qq += part[((size_t)rr * c + ch) * 2 + 1]; }
            const double mm = (double)n * (double)d.hw, mu = ss / mm;
            double var = qq / mm - mu * mu;
            if (var < 0.0) var = 0.0;
            mean = (float)mu;
            invstd = (float)(1.0 / sqrt(var + (double)d.eps));
            if (d.rmean[t] != nullptr) {
                const double unbiased = mm > 1.0 ? var * mm / (mm - 1.0) : var;
                d.rmean[t][ch] = blend_stat(d.momentum, rm, mean);
                d.rvar[t][ch] = blend_stat(d.momentum, rv, (float)unbiased);
            }
        } else {
            mean = rm;
            invstd = 1.f / sqrtf(rv + d.eps);
        }
        scale = gam * invstd;
        shift = fmaf(-mean, scale, bet);              // (explicit: the same bits from every kernel, whatever the compiler would contract)
        float* co = coefs + (size_t)t * 4 * c;
        co[ch] = mean; co[c + ch] = invstd; co[2 * c + ch] = scale; co[3 * c + ch] = shift;
    }
    if (threadIdx.x == 0 && d.training && d.nbt[t] != nullptr) *d.nbt[t] += 1;
    __syncthreads();                                                // everyone is done reading part[]
    if (threadIdx.x < c) { bc[ch] = scale; bc[c + ch] = shift; }
    __syncthreads();
    scale = act ? bc[ch] : 0.f;
    shift = act ? bc[c + ch] : 0.f;
    const size_t tb = (size_t)t * n * c;
    if (!se) {
        if (act)
            for (int i = row; i < n; i += R) {
                gate[tb + (size_t)i * c + ch] = 1.f;
                coef[tb + (size_t)i * c + ch] = w * scale;
                shiftc[tb + (size_t)i * c + ch] = w * shift;
            }
        return;
    }
    // ---- squeeze-and-excitation: m = mean_hw(BN(z)) = scale * mean_hw(z) + shift
    if (act)
        for (int i = row; i < n; i += R) {
            const double zbar = st != nullptr ? zsum_s[i * c + ch] / (double)d.hw : 0.0;
            const float mv = (float)((double)scale * zbar + (double)shift);
            m_s[i * c + ch] = mv;
            se_m[tb + (size_t)i * c + ch] = mv;
        }
    __syncthreads();
    for (int idx = threadIdx.x; idx < n * mid; idx += 256) {
        const int i = idx / mid, j = idx % mid;
        float a = 0.f;
        for (int k = 0; k < c; ++k) a = fmaf(m_s[i * c + k], w1_s[j * c + k], a);
        a_s[i * kMaxMid + j] = a;
        se_a1[((size_t)t * n + i) * kMaxMid + j] = a;
    }
    __syncthreads();
    if (act)
        for (int i = row; i < n; i += R) {
            float a = 0.f;
            for (int j = 0; j < mid; ++j) a = fmaf(fmaxf(a_s[i * kMaxMid + j], 0.f), w2_s[ch * mid + j], a);
            const float g = 1.f / (1.f + expf(-a));
            gate[tb + (size_t)i * c + ch] = g;
            coef[tb + (size_t)i * c + ch] = w * g * scale;
            shiftc[tb + (size_t)i * c + ch] = w * g * shift;
        }
}

static size_t prepare_fwd_lds(const NodeDesc& d) {
    const int R = 256 / d.c;
    return ((size_t)R * d.c * 2 + (size_t)d.n * d.c) * sizeof(double) +
           ((size_t)d.n * d.c + (size_t)d.n * kMaxMid + (size_t)2 * kMaxMid * d.c) * sizeof(float);
}

// one pass over the raw terms of image n: y = act(bias + residual + sum_t cf[t] * z_t)   (cf, bias: per channel, in LDS)
template <int V>
__device__ __forceinline__ void combine_stream(long hw, int c, int nterms, int n, const ZTable& z, const float* lds,
                                               const float* bias, const float* __restrict__ residual, int relu,
                                               float* __restrict__ y, uint8_t* __restrict__ mask8, double* __restrict__ out_stats,
                                               float* __restrict__ y2, int y2s, int y2pad = 0) {
    // y2pad: zero channels written behind the slice (the padding that brings a 24-channel concatenation to a full 32-channel
    // tile, search cell post-process: the thread that owns a pixel's last channel group writes them)
    // y2 (optional): the same values again as a channel slice of a wider NHWC tensor (pixel stride y2s floats) -- the node's
    // place in the concatenation the cell's post-process convolution reads (models/senas_model.py:64), so that no
    // torch.cat launch copies it there; y itself may then be NULL (a node nothing else reads)
    // out_stats (V == 4 only): per-image channel sums of y itself, for the BatchNorm2d of an 'identity' candidate that
    // reads this node (search cell inner edges, utils/operations.py:167-183 without a 1x1 adapter) -- saves its own pass
    Stats4 ost;
    stats_init4(ost);
    const int cv = c / V;
    int ch_thr = (int)(threadIdx.x % cv) * V;            // (every element of this thread has that channel: 256 % cv == 0)
    const long per_img = hw * cv;
    const size_t img_off = (size_t)n * hw * c;
    // (latency-bound regime: every thread has at most two elements of its image)
    const float* safe = nullptr;
    for (int t = 0; t < nterms && safe == nullptr; ++t) safe = z.p[t];
    const bool batched = nterms > 4 && safe != nullptr && per_img <= 2L * gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_img; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % cv) * V;
        const size_t pixel = (size_t)n * hw + (size_t)(i / cv);
        const size_t off = img_off + (size_t)(i / cv) * c + ch;
        float acc[V], tmp[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = bias[ch + j];
        if (residual != nullptr) {
            ldv<V>(residual + off, tmp);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += tmp[j];
        }
        if (batched) {
            // small maps: a thread has one or two elements, and the walk over 12 - 24 terms is a chain of that many dependent
            // L2 round trips (13.7 us for a node on a 2 x 2 map).  Eight requests at a time, no control flow between them: an
            // absent term ('none') loads from the first present one and is multiplied by zero.
            constexpr int NB = 8;
            for (int t0 = 0; t0 < nterms; t0 += NB) {
                float tv[NB][V], cf[NB][V];
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int t = t0 + k < nterms ? t0 + k : nterms - 1;
                    const bool ok = t0 + k < nterms && z.p[t] != nullptr;
                    const float* src = ok ? z.p[t] + pixel * z.s[t] + ch : safe;
                    ldv<V>(src, tv[k]);
#pragma unroll
                    for (int j = 0; j < V; ++j) cf[k][j] = ok ? lds[t * c + ch + j] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < NB; ++k)
#pragma unroll
                    for (int j = 0; j < V; ++j) acc[j] = fmaf(cf[k][j], cf[k][j] != 0.f ? tv[k][j] : 0.f, acc[j]);
            }
        } else {
            for (int t = 0; t < nterms; ++t) {
                if (z.p[t] == nullptr) continue;
                if constexpr (V == 4) ldz4(z.p[t], pixel * z.s[t] + ch, z.bf[t], tmp);
                else ldv<V>(z.p[t] + pixel * z.s[t] + ch, tmp);
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] = fmaf(lds[t * c + ch + j], tmp[j], acc[j]);
            }
        }
        if (V == 4 && mask8 != nullptr) {           // one byte per 16-byte piece: bit j = (y_j > 0), the backward pass's ReLU mask
            const unsigned bits = (acc[0] > 0.f ? 1u : 0u) | (acc[1 % V] > 0.f ? 2u : 0u) | (acc[2 % V] > 0.f ? 4u : 0u) |
                                  (acc[3 % V] > 0.f ? 8u : 0u);
            mask8[off >> 2] = (uint8_t)bits;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = relu ? fmaxf(acc[j], 0.f) : acc[j];
        if (y != nullptr) stv<V>(y + off, acc);
        if (y2 != nullptr) {
            stv<V>(y2 + pixel * y2s + ch, acc);
            if (y2pad > 0 && ch + V == c) {
                float zero[V];
#pragma unroll
                for (int j = 0; j < V; ++j) zero[j] = 0.f;
                for (int q = 0; q < y2pad; q += V) stv<V>(y2 + pixel * y2s + c + q, zero);
            }
        }
        if constexpr (V == 4) {
            ch_thr = ch;
            stats_accumulate4(ost, out_stats, true, n, c, ch, acc, true);
        }
    }
    if constexpr (V == 4) stats_flush4(ost, out_stats, true, n, c, ch_thr);
}

// ------------------------------------------------------------------------------------------ forward combine
// grid = (tiles, n); coefficients of image n are staged in LDS once per block.
template <int V>
__global__ __launch_bounds__(256) void node_combine_fwd_kernel(long hw, int c, int nterms, int nimg, ZTable z,
                                                               const float* __restrict__ coef, const float* __restrict__ shiftc,
                                                               const float* __restrict__ residual, int relu,
                                                               float* __restrict__ y, uint8_t* __restrict__ mask8,
                                                               double* __restrict__ out_stats, float* __restrict__ y2, int y2s, int y2pad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];        // coef[nterms][c], bias[c], shift[nterms][c]
    const int n = blockIdx.y;
    float* bias = lds + nterms * c;
    float* sh = bias + c;
    // (every global operand requested at once: summing the shifts term by term from global memory was a chain of nterms
    // dependent round trips -- 12 of the 13.7 us of a 24-term node on a small map)
    for (int i = threadIdx.x; i < nterms * c; i += 256) {
        const int t = i / c, ch = i % c;
        lds[i] = coef[((size_t)t * nimg + n) * c + ch];
        sh[i] = shiftc[((size_t)t * nimg + n) * c + ch];
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        float b = 0.f;
        for (int t = 0; t < nterms; ++t) b += sh[t * c + ch];
        bias[ch] = b;
    }
    __syncthreads();
    combine_stream<V>(hw, c, nterms, n, z, lds, bias, residual, relu, y, mask8, out_stats, y2, y2s, y2pad);
}

// ------------------------------------------------------------------------------------------ forward, fused
// Few-term nodes (the derived network's cells: <= kFuseTerms terms): every block derives the per-channel
// coefficients of ITS image in a short prologue (the batch statistics of all images are 4 KiB per term, L2-resident)
// and then streams -- one launch per node instead of two; the block (0, image) also leaves the quantities the
// backward pass needs (coefs, gate, se_m, se_a1), block (0, 0) updates the running statistics.
// dynamic LDS: doubles part[R][c][2] | zown[c]; floats cf[T][c] | bias[c] | sc[2][c] | m[c] | a1[kMaxMid]
constexpr int kFuseTerms = 4;

template <int V>
__global__ __launch_bounds__(256) void node_fused_fwd_kernel(NodeDesc d, ZTable z, const float* __restrict__ residual,
                                                             float* __restrict__ y, uint8_t* __restrict__ mask8,
                                                             float* __restrict__ coefs, float* __restrict__ gate,
                                                             float* __restrict__ se_m, float* __restrict__ se_a1,
                                                             double* __restrict__ out_stats, float* __restrict__ y2, int y2s, int y2pad) {
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int n = blockIdx.y, nimg = d.n, c = d.c, T = d.nterms;
    const bool first = blockIdx.x == 0, writer0 = first && n == 0;
    const int R = 256 / c, ch = threadIdx.x % c, row = threadIdx.x / c;
    const bool act = row < R, owner = threadIdx.x < c;              // owner: row 0, one thread per channel
    double* part = ldsd;                                            // [R][c][2]
    double* zown = part + (size_t)R * c * 2;                        // [c]: this image's channel sums
    float* cf = reinterpret_cast<float*>(zown + c);                 // [T][c]
    float* bias = cf + (size_t)T * c;                               // [c]
    float* sc = bias + c;                                           // [2][c]: scale, shift
    float* m_s = sc + 2 * c;                                        // [c]
    float* a_s = m_s + c;                                           // [kMaxMid]
    float bsum = 0.f;                                               // owner threads: running bias of their channel
    for (int t = 0; t < T; ++t) {
        const double* st = d.stats[t];
        const bool se = d.w1[t] != nullptr;
        const int mid = se ? d.mid[t] : 0;
        // ---- global reads first
        double s = 0.0, q = 0.0;
        if (act && st != nullptr && (d.training || se))
            for (int i = row; i < nimg; i += R) {
                const double v0 = st[(size_t)i * d.sstride[t] + ch * 2], v1 = st[(size_t)i * d.sstride[t] + ch * 2 + 1];
                s += v0; q += v1;
                if (i == n) zown[ch] = v0;
            }
        float gam = 0.f, bet = 0.f, rm = 0.f, rv = 0.f;
        if (owner) {
            gam = d.gamma[t][ch]; bet = d.beta[t][ch];
            if (d.rmean[t] != nullptr) { rm = d.rmean[t][ch]; rv = d.rvar[t][ch]; }
        }
        const float w = d.mix != nullptr ? d.mix[t] : 1.f;
        if (act) { part[((size_t)row * c + ch) * 2] = s; part[((size_t)row * c + ch) * 2 + 1] = q; }
        __syncthreads();
        float scale = 0.f, shift = 0.f;
        if (owner) {
            float mean, invstd;
            if (d.training) {
                double ss = 0.0, qq = 0.0;
                for (int rr = 0; rr < R; ++rr) { ss += part[((size_t)rr * c + ch) * 2]; qq += part[((size_t)rr * c + ch) * 2 + 1]; }
                const double mm = (double)nimg * (double)d.hw, mu = ss / mm;
                double var = qq / mm - mu * mu;
                if (var < 0.0) var = 0.0;
                mean = (float)mu;
                invstd = (float)(1.0 / sqrt(var + (double)d.eps));
                if (writer0 && d.rmean[t] != nullptr) {
                    const double unbiased = mm > 1.0 ? var * mm / (mm - 1.0) : var;
                    d.rmean[t][ch] = blend_stat(d.momentum, rm, mean);
                    d.rvar[t][ch] = blend_stat(d.momentum, rv, (float)unbiased);
                }
            } else {
                mean = rm;
                invstd = 1.f / sqrtf(rv + d.eps);
            }
            scale = gam * invstd;
            shift = fmaf(-mean, scale, bet);              // (explicit: the same bits from every kernel, whatever the compiler would contract)
            if (writer0) {
                float* co = coefs + (size_t)t * 4 * c;
                co[ch] = mean; co[c + ch] = invstd; co[2 * c + ch] = scale; co[3 * c + ch] = shift;
            }
            if (se) m_s[ch] = (float)((double)scale * (st != nullptr ? zown[ch] / (double)d.hw : 0.0) + (double)shift);
        }
        if (writer0 && threadIdx.x == 0 && d.training && d.nbt[t] != nullptr) *d.nbt[t] += 1;
        const size_t tb = ((size_t)t * nimg + n) * c;
        float g = 1.f;
        if (se) {                                                   // block-uniform
            __syncthreads();
            if (first && owner) se_m[tb + ch] = m_s[ch];
            if ((int)threadIdx.x < mid) {
                float a = 0.f;
                for (int k = 0; k < c; ++k) a = fmaf(m_s[k], d.w1[t][threadIdx.x * c + k], a);
                a_s[threadIdx.x] = a;
                if (first) se_a1[((size_t)t * nimg + n) * kMaxMid + threadIdx.x] = a;
            }
            __syncthreads();
            if (owner) {
                float a = 0.f;
                for (int j = 0; j < mid; ++j) a = fmaf(fmaxf(a_s[j], 0.f), d.w2[t][ch * mid + j], a);
                g = 1.f / (1.f + expf(-a));
            }
        }
        if (owner) {
            cf[t * c + ch] = w * g * scale;
            bsum += w * g * shift;
            if (first) gate[tb + ch] = g;
        }
        __syncthreads();                                            // part / zown / m_s / a_s are reused by the next term
    }
    if (owner) bias[ch] = bsum;
    __syncthreads();
    combine_stream<V>(d.hw, c, T, n, z, cf, bias, residual, d.relu, y, mask8, out_stats, y2, y2s, y2pad);
}

static size_t fused_fwd_lds(const NodeDesc& d) {
    const int R = 256 / d.c;
    return ((size_t)R * d.c * 2 + d.c) * sizeof(double) + ((size_t)d.nterms * d.c + 4 * d.c + kMaxMid) * sizeof(float);
}

// ------------------------------------------------------------------------------------------ forward, wide
// Many-term nodes on SMALL maps (the search cell's 12 - 24 addends on maps up to ~64 x 64: a handful of blocks): the
// preparation runs as a prologue of every block with one thread per (term, channel) -- one launch instead of two on the
// launch-bound chain of small cells that is the forward critical path of a search pass (profiles/r5_search_cells.txt).
// Every expression and every order of summation is node_prepare_fwd_kernel's / node_combine_fwd_kernel's, so the result is
// bit-identical to the two-launch form (tests/test_gpu_parity.py::test_wide_node_is_the_two_launch_node_bit_for_bit).
// dynamic LDS floats: sc[TC] | sf[TC] | cf[TC] | sh[TC] | m[TC] | bias[c] | a1[T][kMaxMid]   (TC = T * c <= kWideTC)
constexpr int kWideTC = 1024;

template <int V>
__global__ __launch_bounds__(256) void node_wide_fwd_kernel(NodeDesc d, ZTable z, const float* __restrict__ residual,
                                                            float* __restrict__ y, uint8_t* __restrict__ mask8,
                                                            float* __restrict__ coefs, float* __restrict__ gate,
                                                            float* __restrict__ se_m, float* __restrict__ se_a1,
                                                            double* __restrict__ out_stats, float* __restrict__ y2, int y2s, int y2pad) {
    extern __shared__ __attribute__((aligned(16))) float ldsw[];
    SENAS_PHASE(8);
    const int n = blockIdx.y, nimg = d.n, c = d.c, T = d.nterms, TC = T * c;
    const bool first = blockIdx.x == 0, writer0 = first && n == 0;
    const int R = 256 / c;                  // the prepare kernel's image rows: the batch sums are associated as it associates them
    float* sc = ldsw;
    float* sf = sc + TC;
    float* cf = sf + TC;
    float* sh = cf + TC;
    float* m_s = sh + TC;
    float* bias = m_s + TC;
    float* a_s = bias + c;
    // ---- this thread's element (the host launches one thread per element of the image: grid.x * 256 >= hw * c / V) and its
    // operands, requested before the preparation's own chain of loads: the two overlap instead of following each other
    const int cv = c / V;
    const long per_img = d.hw * cv, ei = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = ei < per_img;
    const int ech = live ? (int)(ei % cv) * V : 0;
    const size_t pixel = (size_t)n * d.hw + (size_t)(live ? ei / cv : 0);
    const size_t off = (size_t)n * d.hw * c + (size_t)(live ? ei / cv : 0) * c + ech;
    float tv[SENAS_MAX_TERMS][V], rsd[V];
#pragma unroll
    for (int t = 0; t < SENAS_MAX_TERMS; ++t) {
        if (live && t < T && z.p[t] != nullptr) ldv<V>(z.p[t] + pixel * z.s[t] + ech, tv[t]);
        else {
#pragma unroll
            for (int j = 0; j < V; ++j) tv[t][j] = 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) rsd[j] = 0.f;
    if (live && residual != nullptr) ldv<V>(residual + off, rsd);
    SENAS_PHASE(9);
    // ---- the preparation: one thread per (term, channel)
    for (int i = threadIdx.x; i < TC; i += 256) {
        const int t = i / c, ch = i - t * c;
        const double* st = d.stats[t];
        const bool se = d.w1[t] != nullptr;
        double ss = 0.0, qq = 0.0, zown = 0.0;
        if (st != nullptr && (d.training || se))
            for (int rr = 0; rr < R; ++rr) {
                double s = 0.0, q = 0.0;
                for (int img = rr; img < nimg; img += R) {
                    const double v0 = st[(size_t)img * d.sstride[t] + ch * 2], v1 = st[(size_t)img * d.sstride[t] + ch * 2 + 1];
                    s += v0; q += v1;
                    if (img == n) zown = v0;
                }
                ss += s; qq += q;
            }
        const float gam = d.gamma[t][ch], bet = d.beta[t][ch];
        float rm = 0.f, rv = 0.f;
        if (d.rmean[t] != nullptr) { rm = d.rmean[t][ch]; rv = d.rvar[t][ch]; }
        float mean, invstd;
        if (d.training) {
            const double mm = (double)nimg * (double)d.hw, mu = ss / mm;
            double var = qq / mm - mu * mu;
            if (var < 0.0) var = 0.0;
            mean = (float)mu;
            invstd = (float)(1.0 / sqrt(var + (double)d.eps));
            if (writer0 && d.rmean[t] != nullptr) {
                const double unbiased = mm > 1.0 ? var * mm / (mm - 1.0) : var;
                d.rmean[t][ch] = blend_stat(d.momentum, rm, mean);
                d.rvar[t][ch] = blend_stat(d.momentum, rv, (float)unbiased);
            }
        } else {
            mean = rm;
            invstd = 1.f / sqrtf(rv + d.eps);
        }
        const float scale = gam * invstd;
        const float shift = fmaf(-mean, scale, bet);              // (explicit: the same bits from every kernel, whatever the compiler would contract)
        if (writer0) {
            float* co = coefs + (size_t)t * 4 * c;
            co[ch] = mean; co[c + ch] = invstd; co[2 * c + ch] = scale; co[3 * c + ch] = shift;
        }
        sc[i] = scale;
        sf[i] = shift;
        if (se) {
            const double zbar = st != nullptr ? zown / (double)d.hw : 0.0;
            const float mv = (float)((double)scale * zbar + (double)shift);
            m_s[i] = mv;
            if (first) se_m[((size_t)t * nimg + n) * c + ch] = mv;
        }
    }
    SENAS_PHASE(10);
    if (writer0 && d.training)
        for (int t = threadIdx.x; t < T; t += 256) if (d.nbt[t] != nullptr) *d.nbt[t] += 1;
    __syncthreads();
    SENAS_PHASE(11);
    for (int idx = threadIdx.x; idx < T * kMaxMid; idx += 256) {
        const int t = idx / kMaxMid, j = idx - t * kMaxMid;
        if (d.w1[t] == nullptr || j >= d.mid[t]) continue;
        float a = 0.f;
        for (int k = 0; k < c; ++k) a = fmaf(m_s[t * c + k], d.w1[t][j * c + k], a);
        a_s[idx] = a;
        if (first) se_a1[((size_t)t * nimg + n) * kMaxMid + j] = a;
    }
    __syncthreads();
    SENAS_PHASE(12);
    for (int i = threadIdx.x; i < TC; i += 256) {
        const int t = i / c, ch = i - t * c;
        const float w = d.mix != nullptr ? d.mix[t] : 1.f;
        const size_t o = ((size_t)t * nimg + n) * c + ch;
        if (d.w1[t] == nullptr) {
            if (first) gate[o] = 1.f;
            cf[i] = w * sc[i];
            sh[i] = w * sf[i];
        } else {
            const int mid = d.mid[t];
            float a = 0.f;
            for (int j = 0; j < mid; ++j) a = fmaf(fmaxf(a_s[t * kMaxMid + j], 0.f), d.w2[t][ch * mid + j], a);
            const float g = 1.f / (1.f + expf(-a));
            if (first) gate[o] = g;
            cf[i] = w * g * sc[i];
            sh[i] = w * g * sf[i];
        }
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        float b = 0.f;
        for (int t = 0; t < T; ++t) b += sh[t * c + ch];
        bias[ch] = b;
    }
    __syncthreads();
    SENAS_PHASE(13);
    // ---- the stream: y = act(bias + residual + sum_t cf[t] * z_t), exactly combine_stream's arithmetic for one element
    Stats4 ost;
    stats_init4(ost);
    if (live) {
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = bias[ech + j];
        if (residual != nullptr) {
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += rsd[j];
        }
#pragma unroll
        for (int t = 0; t < SENAS_MAX_TERMS; ++t)
            if (t < T && z.p[t] != nullptr) {
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] = fmaf(cf[t * c + ech + j], tv[t][j], acc[j]);
            }
        if (V == 4 && mask8 != nullptr) {
            const unsigned bits = (acc[0] > 0.f ? 1u : 0u) | (acc[1 % V] > 0.f ? 2u : 0u) | (acc[2 % V] > 0.f ? 4u : 0u) |
                                  (acc[3 % V] > 0.f ? 8u : 0u);
            mask8[off >> 2] = (uint8_t)bits;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = d.relu ? fmaxf(acc[j], 0.f) : acc[j];
        if (y != nullptr) stv<V>(y + off, acc);
        if (y2 != nullptr) {
            stv<V>(y2 + pixel * y2s + ech, acc);
            if (y2pad > 0 && ech + V == c) {
                float zero[V];
#pragma unroll
                for (int j = 0; j < V; ++j) zero[j] = 0.f;
                for (int q = 0; q < y2pad; q += V) stv<V>(y2 + pixel * y2s + c + q, zero);
            }
        }
        if constexpr (V == 4) stats_accumulate4(ost, out_stats, true, n, c, ech, acc, true);
    }
    SENAS_PHASE(14);
    if constexpr (V == 4) stats_flush4(ost, out_stats, true, n, c, (int)(threadIdx.x % cv) * V);
    SENAS_PHASE(15);
}

static size_t wide_fwd_lds(const NodeDesc& d) {
    return ((size_t)5 * d.nterms * d.c + d.c + (size_t)d.nterms * kMaxMid) * sizeof(float);
}

// ------------------------------------------------------------------------------------------ backward reduce
// p1[n][c] += sum_p ds ; p2[t][n][c] += sum_p ds * z_t for t in [t0, t0 + TT).  block = rows x c lanes.
template <int TT>
__global__ __launch_bounds__(256) void node_reduce_kernel(long hw, int c, long chunk, int t0, int tt, int nimg, ZTable z,
                                                          const float* __restrict__ dy, int dys, const float* __restrict__ y,
                                                          const uint8_t* __restrict__ mask8, int relu, int do_p1,
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
            float ds = dy[((size_t)n * hw + p) * dys + ch];
            if (relu) {                                  // byte map (c % 4 == 0: pieces never straddle pixels) or y itself
                const bool pos = mask8 != nullptr ? ((mask8[o >> 2] >> (o & 3)) & 1u) != 0 : y[o] > 0.f;
                if (!pos) ds = 0.f;
            }
            a1 += ds;
#pragma unroll
            for (int t = 0; t < TT; ++t)
                if (t < tt && z.p[t0 + t] != nullptr) a2[t] += (double)ds * (double)z.p[t0 + t][((size_t)n * hw + p) * z.s[t0 + t] + ch];
        }
    }
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
    for (int t = 0; t < TT; ++t)
        if (t < tt && z.p[t0 + t] != nullptr) reduce_to(a2[t], p2 + ((size_t)(t0 + t) * nimg + n) * c + ch);
}

// Vector form for c = 4*Q, Q a power of two (every width on the path): thread = (pixel lane, 4 channels),
// 16-byte loads with U pixels in flight per thread, the ReLU mask from the forward pass's byte map (or y),
// fp64 partials folded over the pixel lanes by shuffles and over the 4 waves through LDS; one fp64 atomic
// per (n, c) and term per block.  grid = (chunks of `chunk` pixels, n).
template <int TT, int U>
__global__ __launch_bounds__(256) void node_reduce_vec_kernel(long hw, int c, long chunk, int t0, int tt_all, int nimg, ZTable z,
                                                              const float* __restrict__ dy, int dys, const float* __restrict__ y,
                                                              const uint8_t* __restrict__ mask8, int relu, int do_p1,
                                                              double* __restrict__ p1, double* __restrict__ p2) {
    extern __shared__ __attribute__((aligned(16))) double redv[];      // [4 waves x row slots][1 + TT][c]
    const int dq = dys >> 2;                                            // dy's pixel stride in 16-byte pieces (Q when dense)
    const int Q = c >> 2, q = threadIdx.x & (Q - 1), pl = threadIdx.x / Q, lanes = 256 / Q;
    const int n = blockIdx.y;
    long q0 = (long)blockIdx.x * chunk, q1 = q0 + chunk;
    if (q1 > hw) q1 = hw;
    // `tt` terms starting at t0, TT at a time.  Nodes with more than TT terms (the search cell: 12 - 24): blockIdx.z takes
    // one group of TT each -- side by side instead of one after the other (every group ends in a fold through LDS and a
    // round of atomics: ~5 us of latency each on a small map); the re-read of dy / the mask comes from L2
    if (gridDim.z > 1) {
        const int first = t0 + (int)blockIdx.z * TT, left = t0 + tt_all - first;
        t0 = first;
        tt_all = left < TT ? left : TT;
        do_p1 = do_p1 && blockIdx.z == 0;
    }
    const int tend = t0 + tt_all;
    for (; t0 < tend; t0 += TT, do_p1 = 0) {
    const int tt = tend - t0 < TT ? tend - t0 : TT;
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[TT][4];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) a2[t][j] = 0.0;
    const size_t img = (size_t)n * hw;
    for (long p = q0 + pl; p < q1; p += (long)lanes * U) {
        float4 dv[U], zv[TT][U];
        unsigned mk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long pp = p + (long)u * lanes;
            const bool ok = pp < q1;
            const size_t o4 = (img + (ok ? pp : q0)) * Q + q;            // index in 16-byte pieces
            dv[u] = reinterpret_cast<const float4*>(dy)[(img + (ok ? pp : q0)) * dq + q];
            if (!ok) dv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            mk[u] = 15u;
            if (relu) {
                if (mask8 != nullptr) mk[u] = mask8[o4];
                else {
                    const float4 yv = reinterpret_cast<const float4*>(y)[o4];
                    mk[u] = (yv.x > 0.f ? 1u : 0u) | (yv.y > 0.f ? 2u : 0u) | (yv.z > 0.f ? 4u : 0u) | (yv.w > 0.f ? 8u : 0u);
                }
            }
#pragma unroll
            for (int t = 0; t < TT; ++t)
                if (t < tt && z.p[t0 + t] != nullptr) {
                    float zt4[4];
                    ldz4(z.p[t0 + t], ((img + (ok ? pp : q0)) * (size_t)(z.s[t0 + t] >> 2) + q) * 4, z.bf[t0 + t], zt4);
                    zv[t][u] = make_float4(zt4[0], zt4[1], zt4[2], zt4[3]);
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float ds[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!((mk[u] >> j) & 1u)) ds[j] = 0.f;
                a1[j] += (double)ds[j];
            }
#pragma unroll
            for (int t = 0; t < TT; ++t)
                if (t < tt && z.p[t0 + t] != nullptr) {
                    const float zz[4] = {zv[t][u].x, zv[t][u].y, zv[t][u].z, zv[t][u].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) a2[t][j] += (double)ds[j] * (double)zz[j];
                }
        }
    }
    // fold: pixel lanes inside a 16-lane row with DPP, then the rows and the 4 waves through LDS
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a1[j] = row_strided_sum(a1[j], Q);
#pragma unroll
        for (int t = 0; t < TT; ++t) a2[t][j] = row_strided_sum(a2[t][j], Q);
    }
    const int slots = wave_slots(Q), nparts = 4 * slots;
    if (lane_holds_partial(lane, Q)) {
        double* dst = redv + (size_t)(wave * slots + lane_slot(lane, Q)) * (1 + TT) * c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dst[q * 4 + j] = a1[j];
#pragma unroll
            for (int t = 0; t < TT; ++t) dst[(size_t)(1 + t) * c + q * 4 + j] = a2[t][j];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (1 + TT) * c; i += 256) {
        const int k = i / c, ch = i - k * c;
        if (k == 0 && !do_p1) continue;
        if (k > 0 && !(k - 1 < tt && z.p[t0 + k - 1] != nullptr)) continue;
        double v = 0.0;
        for (int wv = 0; wv < nparts; ++wv) v += redv[((size_t)wv * (1 + TT) + k) * c + ch];
        if (k == 0) atomicAdd(p1 + (size_t)n * c + ch, v);
        else atomicAdd(p2 + ((size_t)(t0 + k - 1) * nimg + n) * c + ch, v);
    }
    __syncthreads();                                      // the fold scratch is reused by the next group
    }
}

// ------------------------------------------------------------------------------------------ backward prepare
__device__ __forceinline__ double block_sum(double v, double* red) {
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    return red[0];
}

__global__ __launch_bounds__(256) void node_prepare_bwd_generic_kernel(NodeDesc d, const double* __restrict__ p1,
                                                               const double* __restrict__ p2, const float* __restrict__ coefs,
                                                               const float* __restrict__ gate, const float* __restrict__ se_m,
                                                               const float* __restrict__ se_a1, float* __restrict__ dmix,
                                                               float* __restrict__ A, float* __restrict__ B, float* __restrict__ K,
                                                               SeGradTable seg) {
    extern __shared__ __attribute__((aligned(16))) double ldsd[];     // da2[n][c], da1[n][kMaxMid]
    __shared__ double red[256];
    const int t = blockIdx.x, ch = threadIdx.x, n = d.n, c = d.c;
    const bool act = ch < c;
    const size_t tb = (size_t)t * n * c;
    const double hw = (double)d.hw, M = (double)n * hw;
    const float* co = coefs + (size_t)t * 4 * c;
    const double mean = act ? co[ch] : 0.0, invstd = act ? co[c + ch] : 0.0, scale = act ? co[2 * c + ch] : 0.0,
                 shift = act ? co[3 * c + ch] : 0.0;
    const double w = d.mix != nullptr ? (double)d.mix[t] : 1.0;
    const double* st = d.stats[t];
    const bool se = d.w1[t] != nullptr;
    const int mid = se ? d.mid[t] : 0;
    double* da2 = ldsd;
    double* da1 = ldsd + (size_t)n * c;

    // d loss / d(mix_t * gate) per (n, c): full-tensor dot product of ds with BN_t(z_t)
    double dmix_part = 0.0;
    if (act)
        for (int i = 0; i < n; ++i) {
            const double dot = scale * p2[tb + (size_t)i * c + ch] + shift * p1[(size_t)i * c + ch];
            const double g = gate[tb + (size_t)i * c + ch];
            dmix_part += g * dot;
            if (se) da2[i * c + ch] = w * dot * g * (1.0 - g);
        }
    const double dmix_tot = block_sum(dmix_part, red);
    if (threadIdx.x == 0 && dmix != nullptr) dmix[t] = (seg.dmix_accumulate ? dmix[t] : 0.f) + (float)dmix_tot;

    if (se) {
        __syncthreads();
        // dW2[c][j] = sum_n da2[n][c] * relu(a1[n][j])
        if (act)
            for (int j = 0; j < mid; ++j) {
                double s = 0.0;
                for (int i = 0; i < n; ++i) s += da2[i * c + ch] * fmaxf(se_a1[((size_t)t * n + i) * kMaxMid + j], 0.f);
                seg.w2[t][ch * mid + j] = (float)s;
            }
        // da1[n][j] = (a1 > 0) * sum_c da2[n][c] * W2[c][j]
        for (int idx = threadIdx.x; idx < n * mid; idx += blockDim.x) {
            const int i = idx / mid, j = idx % mid;
            double s = 0.0;
            for (int k = 0; k < c; ++k) s += da2[i * c + k] * d.w2[t][k * mid + j];
            da1[i * kMaxMid + j] = se_a1[((size_t)t * n + i) * kMaxMid + j] > 0.f ? s : 0.0;
        }
        __syncthreads();
        // dW1[j][c] = sum_n da1[n][j] * m[n][c]
        if (act)
            for (int j = 0; j < mid; ++j) {
                double s = 0.0;
                for (int i = 0; i < n; ++i) s += da1[i * kMaxMid + j] * se_m[tb + (size_t)i * c + ch];
                seg.w1[t][j * c + ch] = (float)s;
            }
    }
    if (!act) return;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < n; ++i) {
        const double u1 = w * gate[tb + (size_t)i * c + ch];
        double e = 0.0;
        if (se) {
            for (int j = 0; j < mid; ++j) e += da1[i * kMaxMid + j] * d.w1[t][j * c + ch];
            e /= hw;
        }
        const double Z = st != nullptr ? st[(size_t)i * d.sstride[t] + ch * 2] : 0.0;
        s1 += u1 * p1[(size_t)i * c + ch] + hw * e;
        s2 += u1 * p2[tb + (size_t)i * c + ch] + e * Z;
    }
    const double kk = s2 - mean * s1;
    seg.dbeta[t][ch] = (float)s1;
    seg.dgamma[t][ch] = (float)(invstd * kk);
    const double bcoef = d.training ? -scale * invstd * invstd * kk / M : 0.0;
    const double kconst = d.training ? (-scale * s1 / M + scale * invstd * invstd * mean * kk / M) : 0.0;
    for (int i = 0; i < n; ++i) {
        const double u1 = w * gate[tb + (size_t)i * c + ch];
        double e = 0.0;
        if (se) {
            for (int j = 0; j < mid; ++j) e += da1[i * kMaxMid + j] * d.w1[t][j * c + ch];
            e /= hw;
        }
        A[tb + (size_t)i * c + ch] = (float)(scale * u1);
        B[tb + (size_t)i * c + ch] = (float)bcoef;
        K[tb + (size_t)i * c + ch] = (float)(scale * e + kconst);
    }
}

// Fast form of the above for n <= 8 * (256 / c) (every batch on the path): thread = (image row, channel) with up
// to kImgs images per thread in registers; every global operand is requested up front, the SE weights and
// activations live in LDS, reductions over the images go through LDS -- one memory round trip instead of ~10.
constexpr int kImgs = 8;

// One term.  write_grads: this block owns the parameter gradients (d gamma, d beta, d mix, d SE weights).
// A/B/K non-NULL: coefficients of every image go to global memory (stand-alone kernel); keep >= 0: those of image
// `keep` go to A_s/B_s/K_s[c] (fused apply kernel).  Every thread of the block must call it; ends with a barrier.
__device__ __forceinline__ void prepare_bwd_term(const NodeDesc& d, int t, double* ldsd, const double* __restrict__ p1,
                                                 const double* __restrict__ p2, const float* __restrict__ coefs,
                                                 const float* __restrict__ gate, const float* __restrict__ se_m,
                                                 const float* __restrict__ se_a1, float* __restrict__ dmix,
                                                 const SeGradTable& seg, bool write_grads, float* __restrict__ A,
                                                 float* __restrict__ B, float* __restrict__ K, int keep, float* A_s,
                                                 float* B_s, float* K_s) {
    const int n = d.n, c = d.c;
    const int R = 256 / c, ch = threadIdx.x % c, row = threadIdx.x / c;
    const bool act = row < R;
    const size_t tb = (size_t)t * n * c;
    const double hw = (double)d.hw, M = (double)n * hw;
    const bool se = d.w1[t] != nullptr;
    const int mid = se ? d.mid[t] : 0;
    // LDS: doubles da2[n][c] | da1[n][kMaxMid] | part[R][c][2] | red[4]; floats a1[n][kMaxMid] | m[n][c] | w1[mid][c] | w2[c][mid]
    double* da2 = ldsd;
    double* da1 = da2 + (size_t)n * c;
    double* part = da1 + (size_t)n * kMaxMid;
    double* red = part + (size_t)R * c * 2;
    float* a1_s = reinterpret_cast<float*>(red + 4);
    float* m_s = a1_s + (size_t)n * kMaxMid;
    float* w1_s = m_s + (size_t)n * c;
    float* w2_s = w1_s + (size_t)kMaxMid * c;

    SENAS_PHASE(0);
    // ---- all global reads up front
    const float* co = coefs + (size_t)t * 4 * c;
    const double mean = act ? co[ch] : 0.0, invstd = act ? co[c + ch] : 0.0, scale = act ? co[2 * c + ch] : 0.0,
                 shift = act ? co[3 * c + ch] : 0.0;
    const double w = d.mix != nullptr ? (double)d.mix[t] : 1.0;
    const double* st = d.stats[t];
    double P1[kImgs], P2[kImgs], Gt[kImgs], Zs[kImgs];
#pragma unroll
    for (int k = 0; k < kImgs; ++k) {
        const int i = row + k * R;
        const bool ok = act && i < n;
        const size_t o = (size_t)(ok ? i : 0) * c + ch;
        P1[k] = ok ? p1[o] : 0.0;
        P2[k] = ok ? p2[tb + o] : 0.0;
        Gt[k] = ok ? (double)gate[tb + o] : 0.0;
        Zs[k] = ok && st != nullptr ? st[(size_t)(ok ? i : 0) * d.sstride[t] + ch * 2] : 0.0;
    }
    if (se) {
        for (int i = threadIdx.x; i < mid * c; i += 256) { w1_s[i] = d.w1[t][i]; w2_s[i] = d.w2[t][i]; }
        for (int i = threadIdx.x; i < n * kMaxMid; i += 256) a1_s[i] = se_a1[(size_t)t * n * kMaxMid + i];
        for (int i = threadIdx.x; i < n * c; i += 256) m_s[i] = se_m[tb + i];
    }
    // ---- d loss / d(mix_t * gate) per (n, c): full-tensor dot product of ds with BN_t(z_t)
    double dmix_part = 0.0;
#pragma unroll
    for (int k = 0; k < kImgs; ++k) {
        const int i = row + k * R;
        if (act && i < n) {
            const double dot = scale * P2[k] + shift * P1[k];
            dmix_part += Gt[k] * dot;
            if (se) da2[i * c + ch] = w * dot * Gt[k] * (1.0 - Gt[k]);
        }
    }
    SENAS_PHASE(1);
    dmix_part = wave_sum(dmix_part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dmix_part;
    __syncthreads();                                             // red[], da2[], SE operands visible
    SENAS_PHASE(2);
    if (write_grads && threadIdx.x == 0 && dmix != nullptr)
        dmix[t] = (seg.dmix_accumulate ? dmix[t] : 0.f) + (float)(red[0] + red[1] + red[2] + red[3]);

    double e[kImgs];
#pragma unroll
    for (int k = 0; k < kImgs; ++k) e[k] = 0.0;
    if (se) {
        // dW2[c][j] = sum_n da2[n][c] * relu(a1[n][j])
        for (int idx = threadIdx.x; idx < c * mid; idx += 256) {
            const int cc = idx / mid, j = idx - cc * mid;
            double sacc = 0.0;
            for (int i = 0; i < n; ++i) sacc += da2[i * c + cc] * fmaxf(a1_s[i * kMaxMid + j], 0.f);
            if (write_grads) seg.w2[t][idx] = (float)sacc;
        }
        // da1[n][j] = (a1 > 0) * sum_c da2[n][c] * W2[c][j]
        for (int idx = threadIdx.x; idx < n * mid; idx += 256) {
            const int i = idx / mid, j = idx - i * mid;
            double sacc = 0.0;
            for (int k = 0; k < c; ++k) sacc += da2[i * c + k] * w2_s[k * mid + j];
            da1[i * kMaxMid + j] = a1_s[i * kMaxMid + j] > 0.f ? sacc : 0.0;
        }
        __syncthreads();
        // dW1[j][c] = sum_n da1[n][j] * m[n][c]
        for (int idx = threadIdx.x; idx < mid * c; idx += 256) {
            const int j = idx / c, cc = idx - j * c;
            double sacc = 0.0;
            for (int i = 0; i < n; ++i) sacc += da1[i * kMaxMid + j] * m_s[i * c + cc];
            if (write_grads) seg.w1[t][idx] = (float)sacc;
        }
#pragma unroll
        for (int k = 0; k < kImgs; ++k) {
            const int i = row + k * R;
            if (act && i < n) {
                double ev = 0.0;
                for (int j = 0; j < mid; ++j) ev += da1[i * kMaxMid + j] * w1_s[j * c + ch];
                e[k] = ev / hw;
            }
        }
    }
    // ---- sums over the images
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < kImgs; ++k) {
        const int i = row + k * R;
        if (act && i < n) {
            const double u1 = w * Gt[k];
            s1 += u1 * P1[k] + hw * e[k];
            s2 += u1 * P2[k] + e[k] * Zs[k];
        }
    }
    SENAS_PHASE(3);
    if (act) { part[((size_t)row * c + ch) * 2] = s1; part[((size_t)row * c + ch) * 2 + 1] = s2; }
    __syncthreads();
    SENAS_PHASE(4);
    if (act) {
        s1 = 0.0; s2 = 0.0;
        for (int rr = 0; rr < R; ++rr) { s1 += part[((size_t)rr * c + ch) * 2]; s2 += part[((size_t)rr * c + ch) * 2 + 1]; }
        const double kk = s2 - mean * s1;
        if (write_grads && row == 0) {
            seg.dbeta[t][ch] = (float)s1;
            seg.dgamma[t][ch] = (float)(invstd * kk);
        }
        const double bcoef = d.training ? -scale * invstd * invstd * kk / M : 0.0;
        const double kconst = d.training ? (-scale * s1 / M + scale * invstd * invstd * mean * kk / M) : 0.0;
#pragma unroll
        for (int k = 0; k < kImgs; ++k) {
            const int i = row + k * R;
            if (i < n) {
                const float av = (float)(scale * w * Gt[k]), bv = (float)bcoef, kv = (float)(scale * e[k] + kconst);
                if (A != nullptr) {
                    const size_t o = tb + (size_t)i * c + ch;
                    A[o] = av; B[o] = bv; K[o] = kv;
                }
                if (i == keep) { A_s[ch] = av; B_s[ch] = bv; K_s[ch] = kv; }
            }
        }
    }
    SENAS_PHASE(5);
    __syncthreads();                                             // the LDS scratch is reused by the caller's next term
    SENAS_PHASE(6);
}

__global__ __launch_bounds__(256) void node_prepare_bwd_kernel(NodeDesc d, const double* __restrict__ p1,
                                                               const double* __restrict__ p2, const float* __restrict__ coefs,
                                                               const float* __restrict__ gate, const float* __restrict__ se_m,
                                                               const float* __restrict__ se_a1, float* __restrict__ dmix,
                                                               float* __restrict__ A, float* __restrict__ B, float* __restrict__ K,
                                                               SeGradTable seg) {
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    prepare_bwd_term(d, blockIdx.x, ldsd, p1, p2, coefs, gate, se_m, se_a1, dmix, seg, true, A, B, K, -1, nullptr, nullptr, nullptr);
}

static size_t prepare_bwd_lds(const NodeDesc& d) {
    const int R = 256 / d.c;
    return ((size_t)d.n * d.c + (size_t)d.n * kMaxMid + (size_t)R * d.c * 2 + 4) * sizeof(double) +
           ((size_t)d.n * kMaxMid + (size_t)d.n * d.c + (size_t)2 * kMaxMid * d.c) * sizeof(float);
}

// ------------------------------------------------------------------------------------------ backward apply
// dz_t = A * ds + B * z_t + K for every term of image n (ds = dy under the ReLU mask); coefficient (t, ch) sits at
// A[t * kt + kbase + ch] -- global [t][n][c] arrays (kt = nimg*c, kbase = n*c) or per-block LDS copies (kt = c, kbase = 0)
template <int V>
__device__ __forceinline__ void apply_stream(long hw, int c, int nterms, int n, const ZTable& z, const float* __restrict__ dy,
                                             int dys, const float* __restrict__ y, const uint8_t* __restrict__ mask8, int relu,
                                             const float* A, const float* B, const float* K, int kt, int kbase, const DzTable& dz,
                                             float* __restrict__ ds_out, long first = 0) {
    // first: elements of the image that somebody else has done (the fused kernel's prefetched first element per thread)
    const int cv = c / V;
    const long per_img = hw * cv;
    const size_t img_off = (size_t)n * hw * c;
    const float* safe = nullptr;
    for (int t = 0; t < nterms && safe == nullptr; ++t) safe = z.p[t];
    const bool batched = nterms > 4 && safe != nullptr && per_img <= 2L * gridDim.x * 256;      // (latency-bound regime)
    for (long i = first + (long)blockIdx.x * 256 + threadIdx.x; i < per_img; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % cv) * V;
        const size_t off = img_off + (size_t)(i / cv) * c + ch;
        float ds[V], yv[V], zv[V], av[V], bv[V], kv[V];
        ldv<V>(dy + ((size_t)n * hw + (size_t)(i / cv)) * dys + ch, ds);
        if (relu) {
            if (V == 4 && mask8 != nullptr) {
                const unsigned mk = mask8[off >> 2];
#pragma unroll
                for (int j = 0; j < V; ++j) if (!((mk >> j) & 1u)) ds[j] = 0.f;
            } else {
                ldv<V>(y + off, yv);
#pragma unroll
                for (int j = 0; j < V; ++j) if (!(yv[j] > 0.f)) ds[j] = 0.f;
            }
        }
        if (ds_out != nullptr) stv<V>(ds_out + off, ds);
        if (batched) {
            // small maps: one or two elements per thread, and a term-by-term walk is a chain of nterms dependent L2 round trips
            // (9 - 15 us for a node of a search cell).  Eight terms' operands requested together, then their stores.
            constexpr int NB = 8;
            const size_t pix = (size_t)n * hw + (size_t)(i / cv);
            for (int t0 = 0; t0 < nterms; t0 += NB) {
                float zb[NB][V], ab[NB][V], bb[NB][V], kb[NB][V];
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int t = t0 + k < nterms ? t0 + k : nterms - 1;
                    const bool ok = t0 + k < nterms && dz.p[t] != nullptr && z.p[t] != nullptr;
                    const int ko = t * kt + kbase + ch;
                    ldv<V>(ok ? z.p[t] + pix * z.s[t] + ch : safe, zb[k]);
                    ldv<V>(A + ko, ab[k]);
                    ldv<V>(B + ko, bb[k]);
                    ldv<V>(K + ko, kb[k]);
                }
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int t = t0 + k;
                    if (t < nterms && dz.p[t] != nullptr && z.p[t] != nullptr) {
#pragma unroll
                        for (int j = 0; j < V; ++j) zb[k][j] = fmaf(ab[k][j], ds[j], fmaf(bb[k][j], zb[k][j], kb[k][j]));
                        stv<V>(dz.p[t] + pix * dz.s[t] + ch, zb[k]);
                    }
                }
            }
            continue;
        }
        for (int t = 0; t < nterms; ++t) {
            float* out = dz.p[t];
            if (out == nullptr) continue;
            const int ko = t * kt + kbase + ch;
            if constexpr (V == 4) ldz4(z.p[t], ((size_t)n * hw + (size_t)(i / cv)) * z.s[t] + ch, z.bf[t], zv);
            else ldv<V>(z.p[t] + ((size_t)n * hw + (size_t)(i / cv)) * z.s[t] + ch, zv);
            ldv<V>(A + ko, av);
            ldv<V>(B + ko, bv);
            ldv<V>(K + ko, kv);
#pragma unroll
            for (int j = 0; j < V; ++j) zv[j] = fmaf(av[j], ds[j], fmaf(bv[j], zv[j], kv[j]));
            if constexpr (V == 4) stz4(out, ((size_t)n * hw + (size_t)(i / cv)) * dz.s[t] + ch, dz.bf[t], zv);
            else stv<V>(out + ((size_t)n * hw + (size_t)(i / cv)) * dz.s[t] + ch, zv);
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void node_apply_kernel(long hw, int c, int nterms, int nimg, ZTable z,
                                                         const float* __restrict__ dy, int dys, const float* __restrict__ y,
                                                         const uint8_t* __restrict__ mask8, int relu,
                                                         const float* __restrict__ A, const float* __restrict__ B,
                                                         const float* __restrict__ K, DzTable dz, float* __restrict__ ds_out) {
    apply_stream<V>(hw, c, nterms, blockIdx.y, z, dy, dys, y, mask8, relu, A, B, K, nimg * c, blockIdx.y * c, dz, ds_out);
}

// ------------------------------------------------------------------------------------------ backward apply, fused
// Few-term nodes (<= kFuseTerms: the derived network's cells, the single-term batch-norms around every cell): the backward
// PREPARATION runs as a prologue of every apply block (prepare_bwd_term with keep = the block's image: the coefficients of that
// image land in LDS), so a node's backward pass is two launches (reduce, apply) instead of three -- one launch less on every
// node of the chain of SMALL cells that is the backward critical path (the launcher takes this form up to 32 blocks).  Round 2
// measured the form at every size and dropped it (20.2 ms against 19.7); here the thread's first element -- dy, its ReLU mask,
// every term -- is requested BEFORE the prologue, whose own loads (a few L2-resident rows) and barriers then run under the
// stream's latency -- which helps the small grids and does not rescue the large ones (thousands of blocks each repeating it).
// Block (0, 0) owns the parameter gradients.  The arithmetic is prepare_bwd_term's and apply_stream's: results are bit-identical
// to the three-launch form (tests/test_gpu_parity.py::test_fused_apply_is_the_three_launch_backward_bit_for_bit).
// dynamic LDS: prepare_bwd_term's scratch | A_s[T][c] | B_s[T][c] | K_s[T][c] floats
template <int V>
__global__ __launch_bounds__(256) void node_apply_fused_kernel(NodeDesc d, ZTable z, const float* __restrict__ dy, int dys,
                                                               const float* __restrict__ y, const uint8_t* __restrict__ mask8,
                                                               const double* __restrict__ p1, const double* __restrict__ p2,
                                                               const float* __restrict__ coefs, const float* __restrict__ gate,
                                                               const float* __restrict__ se_m, const float* __restrict__ se_a1,
                                                               float* __restrict__ dmix, SeGradTable seg, DzTable dz,
                                                               float* __restrict__ ds_out, size_t scratch_bytes) {
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int n = blockIdx.y, c = d.c, T = d.nterms;
    float* A_s = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(ldsd) + scratch_bytes);
    float* B_s = A_s + T * c;
    float* K_s = B_s + T * c;
    // ---- this thread's first element: requested before the preparation
    const int cv = c / V;
    const long per_img = d.hw * cv, e0 = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = e0 < per_img;
    const int ech = live ? (int)(e0 % cv) * V : 0;
    const size_t pix = (size_t)n * d.hw + (size_t)(live ? e0 / cv : 0);
    const size_t off = (size_t)n * d.hw * c + (size_t)(live ? e0 / cv : 0) * c + ech;
    float ds[V], zt[kFuseTerms][V];
#pragma unroll
    for (int j = 0; j < V; ++j) ds[j] = 0.f;
    unsigned mk = 15u;
    float yv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) yv[j] = 1.f;
    if (live) {
        ldv<V>(dy + pix * dys + ech, ds);
        if (d.relu) {
            if (V == 4 && mask8 != nullptr) mk = mask8[off >> 2];
            else ldv<V>(y + off, yv);
        }
    }
#pragma unroll
    for (int t = 0; t < kFuseTerms; ++t) {
        const bool ok = live && t < T && dz.p[t] != nullptr && z.p[t] != nullptr;
        if (ok) {
            if constexpr (V == 4) ldz4(z.p[t], pix * z.s[t] + ech, z.bf[t], zt[t]);
            else ldv<V>(z.p[t] + pix * z.s[t] + ech, zt[t]);
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) zt[t][j] = 0.f;
        }
    }
    // ---- the preparation (every thread of the block takes part; block (0, 0) writes the parameter gradients)
    const bool owner = blockIdx.x == 0 && n == 0;
    for (int t = 0; t < T; ++t)
        prepare_bwd_term(d, t, ldsd, p1, p2, coefs, gate, se_m, se_a1, dmix, seg, owner, nullptr, nullptr, nullptr, n, A_s + t * c, B_s + t * c,
                         K_s + t * c);
    __syncthreads();
    // ---- the first element from the registers ...
    if (live) {
        if (d.relu) {
            if (V == 4 && mask8 != nullptr) {
#pragma unroll
                for (int j = 0; j < V; ++j) if (!((mk >> j) & 1u)) ds[j] = 0.f;
            } else {
#pragma unroll
                for (int j = 0; j < V; ++j) if (!(yv[j] > 0.f)) ds[j] = 0.f;
            }
        }
        if (ds_out != nullptr) stv<V>(ds_out + off, ds);
#pragma unroll
        for (int t = 0; t < kFuseTerms; ++t) {
            if (t < T && dz.p[t] != nullptr && z.p[t] != nullptr) {
                float r[V];
#pragma unroll
                for (int j = 0; j < V; ++j) r[j] = fmaf(A_s[t * c + ech + j], ds[j], fmaf(B_s[t * c + ech + j], zt[t][j], K_s[t * c + ech + j]));
                if constexpr (V == 4) stz4(dz.p[t], pix * dz.s[t] + ech, dz.bf[t], r);
                else stv<V>(dz.p[t] + pix * dz.s[t] + ech, r);
            }
        }
    }
    // ---- ... the rest of this thread's elements the usual way (coefficients from LDS)
    apply_stream<V>(d.hw, c, T, n, z, dy, dys, y, mask8, d.relu, A_s, B_s, K_s, c, 0, dz, ds_out, (long)gridDim.x * 256);
}

static bool fill_desc(const senas_node_desc* s, NodeDesc& d) {
    if (!s || s->nterms < 1 || s->nterms > SENAS_MAX_TERMS || s->n < 1 || s->c < 1 || s->c > 256 || s->hw < 1) return false;
    d.nterms = s->nterms; d.n = s->n; d.c = s->c; d.training = s->training; d.relu = s->relu; d.hw = (long)s->hw;
    d.eps = s->eps; d.momentum = s->momentum; d.mix = s->mix;
    for (int t = 0; t < SENAS_MAX_TERMS; ++t) {
        const bool in = t < s->nterms;
        d.stats[t] = in ? s->stats[t] : nullptr;
        d.gamma[t] = in ? s->gamma[t] : nullptr;
        d.beta[t] = in ? s->beta[t] : nullptr;
        d.rmean[t] = in ? s->running_mean[t] : nullptr;
        d.rvar[t] = in ? s->running_var[t] : nullptr;
        d.nbt[t] = in ? s->num_batches_tracked[t] : nullptr;
        d.w1[t] = in ? s->se_w1[t] : nullptr;
        d.w2[t] = in ? s->se_w2[t] : nullptr;
        d.mid[t] = in ? s->se_mid[t] : 0;
        d.sstride[t] = (in && s->stats_image_stride[t] > 0) ? s->stats_image_stride[t] : 2 * s->c;
        if (in) {
            if (!d.gamma[t] || !d.beta[t]) return false;
            if ((d.rmean[t] == nullptr) != (d.rvar[t] == nullptr)) return false;
            if (!s->training && !d.rmean[t]) return false;
            if (d.w1[t] && (!d.w2[t] || d.mid[t] < 1 || d.mid[t] > kMaxMid)) return false;
        }
    }
    return true;
}

static long node_chunk(long hw, int n) {
    long per_img = 1024 / (n > 0 ? n : 1);
    if (per_img < 1) per_img = 1;
    long chunk = (hw + per_img - 1) / per_img;
    return chunk < 256 ? 256 : chunk;
}

static unsigned node_grid(long work, int n) {
    long b = (work + 255) / 256;
    const long cap = 4096 / (n > 0 ? n : 1);
    if (b > cap) b = cap;
    return (unsigned)(b < 1 ? 1 : b);
}

SENAS_PHASE_READER(node)

}  // namespace senas

using namespace senas;

// z pixel strides: dense (c) unless given; a strided term must keep 16-byte alignment for the vector kernels
static bool fill_ztable(const NodeDesc& d, const float* const* z, const int32_t* z_pixel_stride, ZTable& zt) {
    for (int t = 0; t < SENAS_MAX_TERMS; ++t) { zt.p[t] = nullptr; zt.s[t] = d.c; zt.bf[t] = 0; }
    for (int t = 0; t < d.nterms; ++t) {
        zt.p[t] = z[t];
        const int st = z_pixel_stride != nullptr ? z_pixel_stride[t] : 0;
        zt.bf[t] = st < 0;                                   // a NEGATIVE pixel stride: the term is a bf16 tensor of that stride (in elements)
        zt.s[t] = st != 0 ? (st < 0 ? -st : st) : d.c;
        if (zt.s[t] < d.c) return false;
        if (zt.bf[t] && (d.c % 4 != 0 || zt.s[t] % 4 != 0 || (reinterpret_cast<uintptr_t>(z[t]) & 7) != 0 || d.nterms > kFuseTerms)) return false;
        if (!zt.bf[t] && zt.s[t] != d.c && d.c % 4 == 0 && (zt.s[t] % 4 != 0 || (reinterpret_cast<uintptr_t>(z[t]) & 15) != 0)) return false;
        if (d.sstride[t] < 2 * d.c) return false;
    }
    return true;
}

extern "C" int senas_node_fwd(const senas_node_desc* desc, const float* const* z, const int32_t* z_pixel_stride, const float* residual, float* y,
                              float* coefs, float* gate, float* coef, float* shiftc, float* se_m, float* se_a1,
                              uint8_t* mask8, double* out_stats, float* y2, int64_t y2_pixel_stride, int y2_zero_pad, void* stream) {
    NodeDesc d;
    SENAS_REQUIRE(fill_desc(desc, d), "node_fwd: bad descriptor");
    SENAS_REQUIRE(z && (y || y2) && coefs && gate && coef && shiftc, "node_fwd: null pointer");
    SENAS_REQUIRE(y2 == nullptr || (y2_pixel_stride >= desc->c && y2_pixel_stride < (1 << 30) &&
                                    (desc->c % 4 != 0 || (y2_pixel_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(y2) & 15) == 0))),
                  "node_fwd: the second destination must keep 16-byte alignment and a pixel stride >= c");
    const int y2s = (int)y2_pixel_stride;
    const int y2pad = y2 != nullptr ? y2_zero_pad : 0;
    SENAS_REQUIRE(y2pad >= 0 && (y2pad == 0 || (desc->c % 4 == 0 && y2pad % 4 == 0 && desc->c + y2pad <= y2s)),
                  "node_fwd: the zero padding behind the second destination must be whole 16-byte pieces inside the pixel");
    ZTable zt{};
    SENAS_REQUIRE(fill_ztable(d, z, z_pixel_stride, zt), "node_fwd: a strided term must keep 16-byte alignment and stride >= c");
    bool any_se = false;
    for (int t = 0; t < d.nterms; ++t) {
        if (d.w1[t]) { any_se = true; SENAS_REQUIRE(d.stats[t], "node_fwd: SE term without statistics"); }
        SENAS_REQUIRE(!(d.training && z[t] && !d.stats[t]), "node_fwd: training-mode term without statistics");
    }
    SENAS_REQUIRE(!any_se || (se_m && se_a1), "node_fwd: SE scratch missing");
    {
        const int cq = d.c / 4;
        SENAS_REQUIRE(out_stats == nullptr || (d.c % 4 == 0 && (cq & (cq - 1)) == 0 && cq <= 64), "node_fwd: output statistics need c = 4 * 2^k <= 256");
    }
    hipStream_t st = as_stream(stream);
    if (d.nterms <= kFuseTerms && fused_fwd_lds(d) <= 48 * 1024) {          // one launch: prologue + stream
        const int V = (d.c % 4 == 0) ? 4 : 1;
        // every block repeats the preparation (statistics -> coefficients, ~3 us): at most 1024 blocks in all, a few
        // elements per thread, instead of 4096 one-element blocks
        unsigned gx = node_grid(d.hw * (d.c / V), d.n);
        const unsigned fat = (unsigned)(1024 / (d.n > 0 ? d.n : 1));
        if (gx > fat && fat >= 1) gx = fat;
        dim3 grid(gx, d.n);
        if (V == 4) hipLaunchKernelGGL((node_fused_fwd_kernel<4>), grid, dim3(256), fused_fwd_lds(d), st, d, zt, residual, y, d.relu ? mask8 : nullptr, coefs, gate, se_m, se_a1, out_stats, y2, y2s, y2pad);
        else hipLaunchKernelGGL((node_fused_fwd_kernel<1>), grid, dim3(256), fused_fwd_lds(d), st, d, zt, residual, y, (uint8_t*)nullptr, coefs, gate, se_m, se_a1, (double*)nullptr, y2, y2s, y2pad);
        return launch_status("node_fwd (fused)");
    }
    {
        // many terms on a small map: prologue + stream in one launch (node_wide_fwd_kernel) while the whole grid is a few
        // dozen blocks -- beyond that every block's prologue costs more than the launch it saves
        const int V = (d.c % 4 == 0) ? 4 : 1;
        const unsigned gx = node_grid(d.hw * (d.c / V), d.n);
        const char* sw = getenv("SENAS_NODE_WIDE");                        // ("0": the two-launch form, for the bit-identity test)
        const bool wide_on = !(sw && sw[0] == '0');
        if (wide_on && d.nterms * d.c <= kWideTC && (long)gx * d.n <= 128 && (long)gx * 256 >= d.hw * (d.c / V) && wide_fwd_lds(d) <= 48 * 1024) {
            dim3 grid(gx, d.n);
            if (V == 4) hipLaunchKernelGGL((node_wide_fwd_kernel<4>), grid, dim3(256), wide_fwd_lds(d), st, d, zt, residual, y, d.relu ? mask8 : nullptr, coefs, gate, se_m, se_a1, out_stats, y2, y2s, y2pad);
            else hipLaunchKernelGGL((node_wide_fwd_kernel<1>), grid, dim3(256), wide_fwd_lds(d), st, d, zt, residual, y, (uint8_t*)nullptr, coefs, gate, se_m, se_a1, (double*)nullptr, y2, y2s, y2pad);
            return launch_status("node_fwd (wide)");
        }
    }
    const size_t lds1 = prepare_fwd_lds(d);
    SENAS_REQUIRE(lds1 <= 64 * 1024, "node_fwd: batch x channels too large for the prepare kernel");
    hipLaunchKernelGGL(node_prepare_fwd_kernel, dim3(d.nterms), dim3(256), lds1, st, d, coefs, gate, coef, shiftc, se_m, se_a1);
    const int V = (d.c % 4 == 0) ? 4 : 1;
    dim3 grid(node_grid(d.hw * (d.c / V), d.n), d.n);
    const size_t lds2 = ((size_t)2 * d.nterms * d.c + d.c) * sizeof(float);
    SENAS_REQUIRE(lds2 <= 64 * 1024, "node_fwd: terms x channels too large for the combine kernel's coefficient stage (split the node: 2*T*c + c <= 16384)");
    if (V == 4) hipLaunchKernelGGL((node_combine_fwd_kernel<4>), grid, dim3(256), lds2, st, d.hw, d.c, d.nterms, d.n, zt, coef, shiftc, residual, d.relu, y, d.relu ? mask8 : nullptr, out_stats, y2, y2s, y2pad);
    else hipLaunchKernelGGL((node_combine_fwd_kernel<1>), grid, dim3(256), lds2, st, d.hw, d.c, d.nterms, d.n, zt, coef, shiftc, residual, d.relu, y, (uint8_t*)nullptr, (double*)nullptr, y2, y2s, y2pad);
    return launch_status("node_fwd");
}

extern "C" int senas_node_bwd(const senas_node_desc* desc, const float* const* z, const int32_t* z_pixel_stride, const float* dy, int64_t dy_pixel_stride,
                              const float* y, const uint8_t* mask8, const float* coefs, const float* gate, const float* se_m, const float* se_a1,
                              double* p1, double* p2, float* const* dgamma, float* const* dbeta, float* dmix, int dmix_accumulate,
                              float* const* dse_w1, float* const* dse_w2, float* abk, float* const* dz, const int32_t* dz_pixel_stride,
                              float* ds_out, void* stream) {
    NodeDesc d;
    SENAS_REQUIRE(fill_desc(desc, d), "node_bwd: bad descriptor");
    SENAS_REQUIRE(z && dy && coefs && gate && p1 && p2 && dgamma && dbeta && abk && dz && (!d.relu || y || mask8), "node_bwd: null pointer");
    if (d.c % 4 != 0) { SENAS_REQUIRE(!d.relu || y, "node_bwd: the byte mask needs c % 4 == 0"); mask8 = nullptr; }
    if (dy_pixel_stride <= 0) dy_pixel_stride = d.c;
    SENAS_REQUIRE(dy_pixel_stride >= d.c && dy_pixel_stride < (1 << 30), "node_bwd: dy pixel stride smaller than c");
    SENAS_REQUIRE(d.c % 4 != 0 || dy_pixel_stride == d.c || (dy_pixel_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0),
                  "node_bwd: a strided dy must keep 16-byte alignment");
    const int dys = (int)dy_pixel_stride;
    ZTable zt{};
    SENAS_REQUIRE(fill_ztable(d, z, z_pixel_stride, zt), "node_bwd: a strided term must keep 16-byte alignment and stride >= c");
    DzTable dzt{};
    SeGradTable seg{};
    seg.dmix_accumulate = dmix_accumulate;
    bool any_se = false, any_dz = false;
    for (int t = 0; t < d.nterms; ++t) {
        dzt.p[t] = dz[t];
        const int dst = dz_pixel_stride != nullptr ? dz_pixel_stride[t] : 0;
        dzt.bf[t] = dst < 0;                                 // a NEGATIVE stride: dz_t is written as bf16
        dzt.s[t] = dst != 0 ? (dst < 0 ? -dst : dst) : d.c;
        SENAS_REQUIRE(!dzt.bf[t] || (d.c % 4 == 0 && dzt.s[t] % 4 == 0 && (reinterpret_cast<uintptr_t>(dz[t]) & 7) == 0 && d.nterms <= kFuseTerms),
                      "node_bwd: a bf16 dz needs c % 4 == 0, 8-byte alignment and at most 4 terms");
        SENAS_REQUIRE(dzt.bf[t] || (dzt.s[t] >= d.c && (d.c % 4 != 0 || dzt.s[t] == d.c ||
                                          (dzt.s[t] % 4 == 0 && (reinterpret_cast<uintptr_t>(dz[t]) & 15) == 0))),
                      "node_bwd: a strided dz must keep 16-byte alignment");
        SENAS_REQUIRE(dgamma[t] && dbeta[t], "node_bwd: null batch-norm gradient destination");
        seg.dgamma[t] = dgamma[t];
        seg.dbeta[t] = dbeta[t];
        SENAS_REQUIRE(z[t] || !dz[t], "node_bwd: dz without z");
        any_dz = any_dz || dz[t] != nullptr;
        if (d.w1[t]) {
            any_se = true;
            SENAS_REQUIRE(dse_w1 && dse_w2 && dse_w1[t] && dse_w2[t] && se_m && se_a1, "node_bwd: SE gradient buffers missing");
            seg.w1[t] = dse_w1[t];
            seg.w2[t] = dse_w2[t];
        }
    }
    hipStream_t st = as_stream(stream);
    const int Q = d.c >> 2;
    const bool vec = d.c % 4 == 0 && Q >= 1 && Q <= 64 && (Q & (Q - 1)) == 0;
    int t0 = 0, first = 1;
    do {
        const int left = d.nterms - t0;
        const int tt = (left >= 8 && !vec) ? 8 : left;          // the vectorised kernel loops over its 8-term groups itself
        if (vec) {
            // pixels per block: a few U-deep iterations of the 256/Q pixel lanes; <= 64 blocks per image so that
            // at most 64 blocks contend for one (n, c) accumulator
            const int U = tt <= 2 ? 4 : (tt <= 4 ? 2 : 1);
            const long per_iter = (long)(256 / Q) * U;
            // blocks per image: every block ends in one fp64 atomic per (term, channel) on the image's accumulators; 256
            // blocks per image measured 93 us at 256 x 256 where 64 take 36 (contention), 128 are the best at 128 x 128
            const int iters = d.hw <= 16384 ? 1 : 8;
            const long cap = d.hw <= 4096 ? 64 : (d.hw <= 16384 ? 128 : 64);
            long chunk = per_iter * iters;
            if ((d.hw + chunk - 1) / chunk > cap) chunk = ((d.hw + cap - 1) / cap + per_iter - 1) / per_iter * per_iter;
            dim3 rgrid((unsigned)((d.hw + chunk - 1) / chunk), d.n, tt > 8 ? (unsigned)((tt + 7) / 8) : 1u);
#define SENAS_RV(TT, UU) hipLaunchKernelGGL((node_reduce_vec_kernel<TT, UU>), rgrid, dim3(256), (size_t)4 * (Q <= 16 ? 4 : 64 / Q) * (1 + TT) * d.c * sizeof(double), st, \
                                             d.hw, d.c, chunk, t0, tt, d.n, zt, dy, dys, y, mask8, d.relu, first, p1, p2)
            if (tt > 4) SENAS_RV(8, 1);
            else if (tt > 2) SENAS_RV(4, 2);
            else if (tt == 2) SENAS_RV(2, 4);
            else SENAS_RV(1, 4);
#undef SENAS_RV
        } else {
            const long chunk = node_chunk(d.hw, d.n);
            dim3 rgrid((unsigned)((d.hw + chunk - 1) / chunk), d.n);
            if (tt > 4) hipLaunchKernelGGL((node_reduce_kernel<8>), rgrid, dim3(256), 0, st, d.hw, d.c, chunk, t0, tt, d.n, zt, dy, dys, y, mask8, d.relu, first, p1, p2);
            else if (tt > 2) hipLaunchKernelGGL((node_reduce_kernel<4>), rgrid, dim3(256), 0, st, d.hw, d.c, chunk, t0, tt, d.n, zt, dy, dys, y, mask8, d.relu, first, p1, p2);
            else if (tt == 2) hipLaunchKernelGGL((node_reduce_kernel<2>), rgrid, dim3(256), 0, st, d.hw, d.c, chunk, t0, tt, d.n, zt, dy, dys, y, mask8, d.relu, first, p1, p2);
            else hipLaunchKernelGGL((node_reduce_kernel<1>), rgrid, dim3(256), 0, st, d.hw, d.c, chunk, t0, tt, d.n, zt, dy, dys, y, mask8, d.relu, first, p1, p2);
        }
        t0 += tt;
        first = 0;
    } while (t0 < d.nterms);
    const size_t tnc = (size_t)d.nterms * d.n * d.c;
    const size_t lds_fast = prepare_bwd_lds(d);
    const bool fast_ok = d.n <= kImgs * (256 / d.c) && lds_fast <= 64 * 1024;
    {
        // few terms: the preparation as a prologue of the apply launch (node_apply_fused_kernel), the operands requested before it
        const char* sw = getenv("SENAS_NODE_FUSED_APPLY");                 // ("0": the three-launch form, for the bit-identity test)
        const size_t scratch = (lds_fast + 15) & ~(size_t)15;
        const size_t lds_fused = scratch + (size_t)3 * d.nterms * d.c * sizeof(float);
        const int V = (d.c % 4 == 0) ? 4 : 1;
        dim3 grid(node_grid(d.hw * (d.c / V), d.n), d.n);
        // ... on SMALL grids only: with thousands of blocks every one of them repeats the prologue and the launch it saves is
        // nothing beside that (measured, round 5: derived train step 15.7 -> 18.9 ms with the fused form at every size -- round 2's
        // finding stands for the large maps, prefetch or not); up to 32 blocks the chain of small cells saves a launch per node
        // (thresholds 0 / 32 / 128 / 512 are within the run-to-run spread of the search step: profiles/r5_planar_wide_ab.txt)
        long fuse_blocks = 32;
        if (const char* fb = getenv("SENAS_NODE_FUSED_BLOCKS")) fuse_blocks = atol(fb);
        if (!(sw && sw[0] == '0') && d.nterms <= kFuseTerms && fast_ok && lds_fused <= 48 * 1024 && (any_dz || ds_out) &&
            (long)grid.x * d.n <= fuse_blocks) {
            if (V == 4) hipLaunchKernelGGL((node_apply_fused_kernel<4>), grid, dim3(256), lds_fused, st, d, zt, dy, dys, y, mask8, p1, p2, coefs, gate, se_m, se_a1, dmix, seg, dzt, ds_out, scratch);
            else hipLaunchKernelGGL((node_apply_fused_kernel<1>), grid, dim3(256), lds_fused, st, d, zt, dy, dys, y, mask8, p1, p2, coefs, gate, se_m, se_a1, dmix, seg, dzt, ds_out, scratch);
            return launch_status("node_bwd (fused apply)");
        }
    }
    // (running this preparation as a prologue of every apply block was measured: the block cannot stream before its
    // prologue is done, so the launch it saves buys nothing -- 20.2 ms vs 19.7 ms per step; kept as its own launch)
    if (fast_ok) {
        hipLaunchKernelGGL(node_prepare_bwd_kernel, dim3(d.nterms), dim3(256), lds_fast, st, d, p1, p2, coefs, gate, se_m, se_a1,
                           dmix, abk, abk + tnc, abk + 2 * tnc, seg);
    } else {
        const size_t lds = any_se ? ((size_t)d.n * d.c + (size_t)d.n * kMaxMid) * sizeof(double) : 0;
        hipLaunchKernelGGL(node_prepare_bwd_generic_kernel, dim3(d.nterms), dim3(256), lds, st, d, p1, p2, coefs, gate, se_m, se_a1,
                           dmix, abk, abk + tnc, abk + 2 * tnc, seg);
    }
    if (any_dz || ds_out) {
        const int V = (d.c % 4 == 0) ? 4 : 1;
        dim3 grid(node_grid(d.hw * (d.c / V), d.n), d.n);
        if (V == 4) hipLaunchKernelGGL((node_apply_kernel<4>), grid, dim3(256), 0, st, d.hw, d.c, d.nterms, d.n, zt, dy, dys, y, mask8, d.relu, abk, abk + tnc, abk + 2 * tnc, dzt, ds_out);
        else hipLaunchKernelGGL((node_apply_kernel<1>), grid, dim3(256), 0, st, d.hw, d.c, d.nterms, d.n, zt, dy, dys, y, mask8, d.relu, abk, abk + tnc, abk + 2 * tnc, dzt, ds_out);
    }
    return launch_status("node_bwd");
}
