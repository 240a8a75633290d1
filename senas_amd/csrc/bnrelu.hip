// Batched BatchNorm2d + ReLU over k INDEPENDENT tensors of one shape (DepSepConv's depthwise half: depthwise conv ->
// BatchNorm2d(c_in) -> ReLU, utils/operations.py:107-115).  In a search cell the six depthwise outputs that leave one
// state (3 edges x dep_sep_conv_3/5) each need their own normalise + activate pass; as single-term cell nodes that was
// 1 forward + 3 backward launches EACH at launch-bound tensor sizes.  Here the k tensors share the launches:
// forward 1, backward 2, for any k <= SENAS_MAX_BNRELU.  blockIdx.z = tensor.
//
// Same arithmetic as the fused cell node (node.hip): statistics in fp64 from the producer's per-image sums, biased
// variance for normalisation, unbiased for running_var, ReLU mask as one byte per 16-byte piece of the output.
#include "common.h"

namespace senas {

namespace {

struct BnItems {
    senas_bnrelu_item it[SENAS_MAX_BNRELU];
};

constexpr int kMaxC = 64;      // channels per tensor (c % 4 == 0)

// per-channel block reduction helper: every thread holds 4 (S1) + 4 (S2) partial sums of its channel quad q; threads with
// the same q are `lanes` apart in steps of Q.  red: [256][8] doubles.
__device__ __forceinline__ void fold_quads(double (&s1)[4], double (&s2)[4], double* red, int Q, int lanes) {
    double* mine = red + (size_t)threadIdx.x * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) { mine[j] = s1[j]; mine[4 + j] = s2[j]; }
    __syncthreads();
    if ((int)threadIdx.x < Q) {
        for (int l = 1; l < lanes; ++l) {
            const double* o = red + (size_t)(threadIdx.x + l * Q) * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[j] += o[j]; s2[j] += o[4 + j]; }
        }
    }
}

// grid = (pixel chunks, n, k); block 256.  LDS: scale[c], shift[c].
__global__ __launch_bounds__(256) void bnrelu_multi_fwd_kernel(BnItems items, int nimg, long hw, int c, long chunk, int training,
                                                               float momentum, float eps) {
    __shared__ float sc[2 * kMaxC];
    const senas_bnrelu_item& it = items.it[blockIdx.z];
    const int n = blockIdx.y;
    const bool writer = blockIdx.x == 0 && n == 0;
    if ((int)threadIdx.x < c) {
        const int ch = threadIdx.x;
        const float gam = it.gamma[ch], bet = it.beta[ch];
        float mean, invstd;
        if (training) {
            double s = 0.0, q = 0.0;
            for (int i = 0; i < nimg; ++i) { s += it.stats[((size_t)i * c + ch) * 2]; q += it.stats[((size_t)i * c + ch) * 2 + 1]; }
            const double mm = (double)nimg * (double)hw, mu = s / mm;
            double var = q / mm - mu * mu;
            if (var < 0.0) var = 0.0;
            mean = (float)mu;
            invstd = (float)(1.0 / sqrt(var + (double)eps));
            if (writer && it.running_mean != nullptr) {
                const double unbiased = mm > 1.0 ? var * mm / (mm - 1.0) : var;
                it.running_mean[ch] = (1.f - momentum) * it.running_mean[ch] + momentum * mean;
                it.running_var[ch] = (1.f - momentum) * it.running_var[ch] + momentum * (float)unbiased;
            }
        } else {
            mean = it.running_mean[ch];
            invstd = 1.f / sqrtf(it.running_var[ch] + eps);
        }
        const float scale = gam * invstd;
        sc[ch] = scale;
        sc[c + ch] = bet - mean * scale;
        if (writer) { it.mean_invstd[ch] = mean; it.mean_invstd[c + ch] = invstd; }
    }
    if (writer && threadIdx.x == 0 && training && it.num_batches_tracked != nullptr) *it.num_batches_tracked += 1;
    __syncthreads();
    const int Q = c >> 2, q = threadIdx.x % Q, pl = threadIdx.x / Q, lanes = 256 / Q;
    const float4 s4 = *reinterpret_cast<const float4*>(sc + 4 * q), b4 = *reinterpret_cast<const float4*>(sc + c + 4 * q);
    long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk;
    if (p1 > hw) p1 = hw;
    if (pl >= lanes) return;
    const size_t img = (size_t)n * hw;
    for (long p = p0 + pl; p < p1; p += lanes) {
        const size_t o4 = (img + p) * Q + q;
        const float4 z = reinterpret_cast<const float4*>(it.z)[o4];
        float4 y;
        y.x = fmaf(z.x, s4.x, b4.x); y.y = fmaf(z.y, s4.y, b4.y); y.z = fmaf(z.z, s4.z, b4.z); y.w = fmaf(z.w, s4.w, b4.w);
        const unsigned mk = (y.x > 0.f ? 1u : 0u) | (y.y > 0.f ? 2u : 0u) | (y.z > 0.f ? 4u : 0u) | (y.w > 0.f ? 8u : 0u);
        y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
        reinterpret_cast<float4*>(it.y)[o4] = y;
        if (it.mask8 != nullptr) it.mask8[o4] = (uint8_t)mk;
    }
}

// S1 = sum ds, S2 = sum ds * z per (image, channel), ds = dy where the ReLU let the value through; added into it.sums
__global__ __launch_bounds__(256) void bnrelu_multi_bwd_reduce_kernel(BnItems items, long hw, int c, long chunk) {
    extern __shared__ __attribute__((aligned(16))) double red[];      // [256][8]
    const senas_bnrelu_item& it = items.it[blockIdx.z];
    const int n = blockIdx.y;
    const int Q = c >> 2, q = threadIdx.x % Q, pl = threadIdx.x / Q, lanes = 256 / Q;
    const int dq = (int)(it.dy_pixel_stride >> 2);
    long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk;
    if (p1 > hw) p1 = hw;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    const size_t img = (size_t)n * hw;
    if (pl < lanes) {
        for (long p = p0 + pl; p < p1; p += lanes) {
            const size_t o4 = (img + p) * Q + q;
            const float4 d = reinterpret_cast<const float4*>(it.dy)[(img + p) * dq + q];
            const float4 z = reinterpret_cast<const float4*>(it.z)[o4];
            const unsigned mk = it.mask8[o4];
            const float ds[4] = {(mk & 1u) ? d.x : 0.f, (mk & 2u) ? d.y : 0.f, (mk & 4u) ? d.z : 0.f, (mk & 8u) ? d.w : 0.f};
            const float zz[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[j] += (double)ds[j]; s2[j] += (double)ds[j] * (double)zz[j]; }
        }
    }
    fold_quads(s1, s2, red, Q, lanes);
    if ((int)threadIdx.x < Q) {
        double* dst = it.sums + ((size_t)n * c + 4 * q) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) { atomicAdd(dst + 2 * j, s1[j]); atomicAdd(dst + 2 * j + 1, s2[j]); }
    }
}

// dz = g * (ds - mean(ds) - xhat * mean(ds * xhat)) written as A*ds + B*z + K per channel; block (0, 0, t) also writes
// d gamma = sum ds * xhat and d beta = sum ds
__global__ __launch_bounds__(256) void bnrelu_multi_bwd_apply_kernel(BnItems items, int nimg, long hw, int c, long chunk) {
    __shared__ float abk[3 * kMaxC];
    const senas_bnrelu_item& it = items.it[blockIdx.z];
    const int n = blockIdx.y;
    const bool writer = blockIdx.x == 0 && n == 0;
    if ((int)threadIdx.x < c) {
        const int ch = threadIdx.x;
        double S1 = 0.0, S2 = 0.0;
        for (int i = 0; i < nimg; ++i) { S1 += it.sums[((size_t)i * c + ch) * 2]; S2 += it.sums[((size_t)i * c + ch) * 2 + 1]; }
        const double mu = (double)it.mean_invstd[ch], is = (double)it.mean_invstd[c + ch], gam = (double)it.gamma[ch];
        const double mm = (double)nimg * (double)hw;
        const double g = gam * is, m1 = S1 / mm, m2 = (S2 - mu * S1) * is * is / mm;
        abk[ch] = (float)g;
        abk[c + ch] = (float)(-g * m2);
        abk[2 * c + ch] = (float)(g * (mu * m2 - m1));
        if (writer) {
            if (it.dgamma != nullptr) it.dgamma[ch] = (float)((S2 - mu * S1) * is);
            if (it.dbeta != nullptr) it.dbeta[ch] = (float)S1;
        }
    }
    __syncthreads();
    if (it.dz == nullptr) return;
    const int Q = c >> 2, q = threadIdx.x % Q, pl = threadIdx.x / Q, lanes = 256 / Q;
    if (pl >= lanes) return;
    const int dq = (int)(it.dy_pixel_stride >> 2);
    const float4 a4 = *reinterpret_cast<const float4*>(abk + 4 * q), b4 = *reinterpret_cast<const float4*>(abk + c + 4 * q),
                 k4 = *reinterpret_cast<const float4*>(abk + 2 * c + 4 * q);
    long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk;
    if (p1 > hw) p1 = hw;
    const size_t img = (size_t)n * hw;
    for (long p = p0 + pl; p < p1; p += lanes) {
        const size_t o4 = (img + p) * Q + q;
        const float4 d = reinterpret_cast<const float4*>(it.dy)[(img + p) * dq + q];
        const float4 z = reinterpret_cast<const float4*>(it.z)[o4];
        const unsigned mk = it.mask8[o4];
        float4 r;
        r.x = fmaf(a4.x, (mk & 1u) ? d.x : 0.f, fmaf(b4.x, z.x, k4.x));
        r.y = fmaf(a4.y, (mk & 2u) ? d.y : 0.f, fmaf(b4.y, z.y, k4.y));
        r.z = fmaf(a4.z, (mk & 4u) ? d.z : 0.f, fmaf(b4.z, z.z, k4.z));
        r.w = fmaf(a4.w, (mk & 8u) ? d.w : 0.f, fmaf(b4.w, z.w, k4.w));
        reinterpret_cast<float4*>(it.dz)[o4] = r;
    }
}

static bool shape_ok(int k, int n, int64_t hw, int c) {
    return k >= 1 && k <= SENAS_MAX_BNRELU && n >= 1 && n <= 65535 && hw >= 1 && c >= 4 && c <= kMaxC && c % 4 == 0 &&
           256 % (c / 4) == 0 && hw * n * (int64_t)c < 0x7fffffffLL;
}

static long pick_chunk(int64_t hw, int n, int k) {
    // enough blocks to fill the chip, at most 64 per image (the reduce ends in atomics on the image's accumulators)
    long blocks = 1024 / ((long)n * k);
    if (blocks < 1) blocks = 1;
    if (blocks > 64) blocks = 64;
    long chunk = (hw + blocks - 1) / blocks;
    if (chunk < 64) chunk = 64;
    return chunk;
}

}  // namespace
}  // namespace senas

extern "C" int senas_bnrelu_multi_fwd(const senas_bnrelu_item* items, int k, int n, int64_t hw, int c, int training, float momentum,
                                      float eps, void* stream) {
    using namespace senas;
    SENAS_REQUIRE(items && shape_ok(k, n, hw, c), "bnrelu_multi_fwd: bad argument (1 <= k <= 8, c % 4 == 0, c <= 64)");
    BnItems b{};
    for (int t = 0; t < k; ++t) {
        b.it[t] = items[t];
        SENAS_REQUIRE(b.it[t].z && b.it[t].y && b.it[t].gamma && b.it[t].beta && b.it[t].mean_invstd, "bnrelu_multi_fwd: null pointer");
        SENAS_REQUIRE(training ? b.it[t].stats != nullptr : (b.it[t].running_mean && b.it[t].running_var),
                      "bnrelu_multi_fwd: statistics missing");
    }
    const long chunk = pick_chunk(hw, n, k);
    dim3 grid((unsigned)((hw + chunk - 1) / chunk), n, k);
    hipLaunchKernelGGL(bnrelu_multi_fwd_kernel, grid, dim3(256), 0, as_stream(stream), b, n, (long)hw, c, chunk, training, momentum, eps);
    return launch_status("bnrelu_multi_fwd");
}

extern "C" int senas_bnrelu_multi_bwd(const senas_bnrelu_item* items, int k, int n, int64_t hw, int c, void* stream) {
    using namespace senas;
    SENAS_REQUIRE(items && shape_ok(k, n, hw, c), "bnrelu_multi_bwd: bad argument (1 <= k <= 8, c % 4 == 0, c <= 64)");
    BnItems b{};
    for (int t = 0; t < k; ++t) {
        b.it[t] = items[t];
        SENAS_REQUIRE(b.it[t].z && b.it[t].dy && b.it[t].mask8 && b.it[t].sums && b.it[t].gamma && b.it[t].mean_invstd,
                      "bnrelu_multi_bwd: null pointer");
        if (b.it[t].dy_pixel_stride <= 0) b.it[t].dy_pixel_stride = c;
        SENAS_REQUIRE(b.it[t].dy_pixel_stride >= c && b.it[t].dy_pixel_stride % 4 == 0 &&
                      (reinterpret_cast<uintptr_t>(b.it[t].dy) & 15) == 0, "bnrelu_multi_bwd: dy must keep 16-byte alignment");
    }
    const long chunk = pick_chunk(hw, n, k);
    dim3 grid((unsigned)((hw + chunk - 1) / chunk), n, k);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(bnrelu_multi_bwd_reduce_kernel, grid, dim3(256), 256 * 8 * sizeof(double), st, b, (long)hw, c, chunk);
    hipLaunchKernelGGL(bnrelu_multi_bwd_apply_kernel, grid, dim3(256), 0, st, b, n, (long)hw, c, chunk);
    return launch_status("bnrelu_multi_bwd");
}
