// The second half of the DepSepConv candidates of a search cell (utils/operations.py:107-115:
//   depthwise conv -> BatchNorm2d(c_in) -> ReLU -> 1x1 conv (c_in -> c_out) -> BatchNorm2d(c_out))
// as ONE pass per direction over the depthwise output z1, for the k candidates that leave one state (blockIdx.z = problem):
//
//   forward  : z2 = W * relu(BN1(z1))            -- BN1 + ReLU applied on load, the activated tensor is never stored;
//              producer-side statistics of z2 for the BatchNorm2d that follows (applied by the cell node)
//   backward : with dmid = W^T dz2 (recomputed per pixel: 8 x 4 FMAs), ds = dmid where BN1(z1) > 0:
//     reduce : S1 = sum ds, S2 = sum ds * z1 per (image, channel)  AND  dW = sum dz2 (x) relu(BN1(z1))  (same operands),
//              dW accumulated in fp64 (one atomic per element and block), rounded to fp32 by the apply launch
//     apply  : dz1 = A * ds + B * z1 + K  (batch-norm backward), d gamma1, d beta1
//
// It replaces three forward launches' worth of traffic by one (bnrelu_multi_fwd wrote and pw_multi_fwd re-read a c_in-wide
// tensor) and six backward launches by two: pw_multi dgrad / wgrad part / wgrad sum, bnrelu_multi reduce / apply.
// At 4 x 32 x H x W these passes are launch- and latency-bound; the supernet step runs 120 such groups.
#include "common.h"

namespace senas {

namespace {

struct DsItems {
    senas_dstail_item it[SENAS_MAX_DSTAIL];
};

constexpr int kMaxCin = 64, kMaxCout = 8;
constexpr int kDwSlots = 1;     // weight-gradient accumulators per image (more, dealt by chunk index, bought nothing: 17.5 us against 17.2)

// scale / shift of BN1 into LDS (sc[0..c) scale, sc[c..2c) shift); training: from the producer-side sums, eval: running
__device__ __forceinline__ void bn1_coefficients(const senas_dstail_item& it, int nimg, long hw, int c, int training, float momentum,
                                                 float eps, bool writer, bool first_pass, float* sc) {
    if ((int)threadIdx.x < c) {
        const int ch = threadIdx.x;
        const float gam = it.gamma1[ch], bet = it.beta1[ch];
        float mean, invstd;
        if (!first_pass) {                                   // backward: what the forward pass saved
            mean = it.mean_invstd[ch];
            invstd = it.mean_invstd[c + ch];
        } else if (training) {
            double s = 0.0, q = 0.0;
            for (int i = 0; i < nimg; ++i) { s += it.stats1[((size_t)i * c + ch) * 2]; q += it.stats1[((size_t)i * c + ch) * 2 + 1]; }
            const double mm = (double)nimg * (double)hw, mu = s / mm;
            double var = q / mm - mu * mu;
            if (var < 0.0) var = 0.0;
            mean = (float)mu;
            invstd = (float)(1.0 / sqrt(var + (double)eps));
            if (writer && it.running_mean1 != nullptr) {
                const double unbiased = mm > 1.0 ? var * mm / (mm - 1.0) : var;
                it.running_mean1[ch] = (1.f - momentum) * it.running_mean1[ch] + momentum * mean;
                it.running_var1[ch] = (1.f - momentum) * it.running_var1[ch] + momentum * (float)unbiased;
            }
        } else {
            mean = it.running_mean1[ch];
            invstd = 1.f / sqrtf(it.running_var1[ch] + eps);
        }
        const float scale = gam * invstd;
        sc[ch] = scale;
        sc[c + ch] = bet - mean * scale;
        if (first_pass && writer) { it.mean_invstd[ch] = mean; it.mean_invstd[c + ch] = invstd; }
    }
}

// grid = (pixel chunks of one image, n, k).  thread = ONE pixel, all COUT outputs: a lane reads its 4 * Q-float row once
// (consecutive lanes, consecutive rows), applies BN1 + ReLU and multiplies by W from LDS (broadcast reads).
template <int COUT>
__global__ __launch_bounds__(256) void dstail_fwd_kernel(DsItems items, int nimg, long hw, int cin, long chunk, int training,
                                                         float momentum, float eps) {
    __shared__ __attribute__((aligned(16))) float wl[kMaxCout * kMaxCin];
    __shared__ __attribute__((aligned(16))) float sc[2 * kMaxCin];
    __shared__ double red[16][2 * kMaxCout];
    const senas_dstail_item& it = items.it[blockIdx.z];
    const int n = blockIdx.y;
    const bool writer = blockIdx.x == 0 && n == 0;
    bn1_coefficients(it, nimg, hw, cin, training, momentum, eps, writer, true, sc);
    if (writer && threadIdx.x == 0 && training && it.num_batches_tracked1 != nullptr) *it.num_batches_tracked1 += 1;
    for (int i = threadIdx.x; i < COUT * cin; i += 256) wl[i] = it.w[i];
    __syncthreads();
    long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk;
    if (p1 > hw) p1 = hw;
    double s1[COUT], s2[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) s1[co] = s2[co] = 0.0;
    const size_t img = (size_t)n * hw;
    const int Q = cin >> 2;
    for (long p = p0 + threadIdx.x; p < p1; p += 256) {
        const float4* xp = reinterpret_cast<const float4*>(it.z1 + (img + p) * cin);
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
        for (int c4 = 0; c4 < Q; ++c4) {
            float4 xv = xp[c4];
            const float4 s4 = *reinterpret_cast<const float4*>(sc + 4 * c4), b4 = *reinterpret_cast<const float4*>(sc + cin + 4 * c4);
            xv.x = fmaxf(fmaf(xv.x, s4.x, b4.x), 0.f); xv.y = fmaxf(fmaf(xv.y, s4.y, b4.y), 0.f);
            xv.z = fmaxf(fmaf(xv.z, s4.z, b4.z), 0.f); xv.w = fmaxf(fmaf(xv.w, s4.w, b4.w), 0.f);
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float4 wv = *reinterpret_cast<const float4*>(wl + co * cin + 4 * c4);
                acc[co] = fmaf(xv.x, wv.x, fmaf(xv.y, wv.y, fmaf(xv.z, wv.z, fmaf(xv.w, wv.w, acc[co]))));
            }
        }
        float* yp = it.z2 + (img + p) * COUT;
#pragma unroll
        for (int co = 0; co < COUT; co += 4) stv<4>(yp + co, reinterpret_cast<float(&)[4]>(acc[co]));
#pragma unroll
        for (int co = 0; co < COUT; ++co) { s1[co] += (double)acc[co]; s2[co] += (double)acc[co] * (double)acc[co]; }
    }
    if (it.stats2 == nullptr) return;                              // block-uniform
    // per-image channel sums of z2: 16-lane rows with DPP (VALU speed), the 16 rows of the block through LDS, one fp64 atomic
    // pair per channel and block
    const int row = threadIdx.x >> 4;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const double a = row_strided_sum(s1[co], 1), b = row_strided_sum(s2[co], 1);
        if ((threadIdx.x & 15) == 0) { red[row][2 * co] = a; red[row][2 * co + 1] = b; }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * COUT) {
        double tot = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) tot += red[r][threadIdx.x];
        atomicAdd(it.stats2 + (size_t)n * COUT * 2 + threadIdx.x, tot);
    }
}

// ds quad of one pixel: dmid = W^T dz2 for the thread's 4 input channels, masked by BN1(z1) > 0; mid = relu(BN1(z1))
template <int COUT>
__device__ __forceinline__ void ds_quad(const float* __restrict__ dp, bool vec, const float* wl, int cin, int q, const float4& z, const float4& s4,
                                        const float4& b4, float (&d)[COUT], float (&ds)[4], float (&mid)[4]) {
    if (vec) {                                                   // (block-uniform: dz2 rows are 16-byte aligned)
#pragma unroll
        for (int co = 0; co < COUT; co += 4) ldv<4>(dp + co, reinterpret_cast<float(&)[4]>(d[co]));
    } else {
#pragma unroll
        for (int co = 0; co < COUT; ++co) d[co] = dp[co];
    }
    float dm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float4 wv = *reinterpret_cast<const float4*>(wl + co * cin + 4 * q);
        dm[0] = fmaf(d[co], wv.x, dm[0]); dm[1] = fmaf(d[co], wv.y, dm[1]); dm[2] = fmaf(d[co], wv.z, dm[2]); dm[3] = fmaf(d[co], wv.w, dm[3]);
    }
    const float pre[4] = {fmaf(z.x, s4.x, b4.x), fmaf(z.y, s4.y, b4.y), fmaf(z.z, s4.z, b4.z), fmaf(z.w, s4.w, b4.w)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool pos = pre[j] > 0.f;
        ds[j] = pos ? dm[j] : 0.f;
        mid[j] = pos ? pre[j] : 0.f;
    }
}

// grid = (pixel chunks, n, k).  thread = (pixel lane, input-channel quad).  WG: also the weight gradient of the 1x1.
template <int COUT, bool WG>
__global__ __launch_bounds__(256) void dstail_bwd_reduce_kernel(DsItems items, long hw, int cin, long chunk) {
    extern __shared__ __attribute__((aligned(16))) double red[];      // [16 rows][Q][8] doubles, [16][Q][4 * COUT] floats
    __shared__ __attribute__((aligned(16))) float wl[kMaxCout * kMaxCin];
    __shared__ __attribute__((aligned(16))) float sc[2 * kMaxCin];
    const senas_dstail_item& it = items.it[blockIdx.z];
    const int n = blockIdx.y;
    SENAS_PHASE(0);
    bn1_coefficients(it, 0, hw, cin, 1, 0.f, 0.f, false, false, sc);
    for (int i = threadIdx.x; i < COUT * cin; i += 256) wl[i] = it.w[i];
    __syncthreads();
    SENAS_PHASE(1);
    const int Q = cin >> 2, q = threadIdx.x % Q, pl = threadIdx.x / Q, lanes = 256 / Q;
    const int dst = (int)it.dz2_pixel_stride;
    const bool vec = (dst & 3) == 0 && (reinterpret_cast<uintptr_t>(it.dz2) & 15) == 0;
    const float4 s4 = *reinterpret_cast<const float4*>(sc + 4 * q), b4 = *reinterpret_cast<const float4*>(sc + cin + 4 * q);
    long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk;
    if (p1 > hw) p1 = hw;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    float wacc[COUT][4];
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int j = 0; j < 4; ++j) wacc[co][j] = 0.f;
    const size_t img = (size_t)n * hw;
    for (long p = p0 + pl; p < p1; p += lanes) {
        const float4 z = reinterpret_cast<const float4*>(it.z1)[(img + p) * Q + q];
        float d[COUT], ds[4], mid[4];
        ds_quad<COUT>(it.dz2 + (img + p) * dst, vec, wl, cin, q, z, s4, b4, d, ds, mid);
        const float zz[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[j] += (double)ds[j]; s2[j] += (double)ds[j] * (double)zz[j]; }
        if (WG) {
#pragma unroll
            for (int co = 0; co < COUT; ++co)
#pragma unroll
                for (int j = 0; j < 4; ++j) wacc[co][j] = fmaf(d[co], mid[j], wacc[co][j]);
        }
    }
    SENAS_PHASE(2);
    // ---- S1 / S2 and dW: the pixel lanes of a 16-lane row that share q folded in registers (DPP), the 16 rows of the block
    // through LDS, then one fp64 atomic per value and block.  (The first form had thread q walk the other 256 / Q - 1 threads'
    // partial sums in LDS: 4 us of a 9.5 us launch on the small maps.)
    // No device-scope fence anywhere: on this multi-XCD part a release fence writes the whole L2 back, once per block -- the
    // last-block-folds-the-partials form of this kernel measured 107 us instead of 25.  The apply launch rounds the dW
    // accumulators (double[n][kDwSlots][COUT][cin], zero on entry) to the fp32 gradient; fp64 accumulation makes the summation order immaterial
    // at fp32 precision.
    // 16 partials per value: (wave, 16-lane row).  S1 / S2 stay in fp64 throughout; the per-thread dW sums are fp32 already and
    // are folded in fp32 inside a row (one DPP add per step: the fp64 form, two moves and an add per step plus shuffles, cost
    // 4.5 us for the 32 values), in fp64 from there on.
    const int part = threadIdx.x >> 4, rl = threadIdx.x & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[j] = row_strided_sum(s1[j], Q); s2[j] = row_strided_sum(s2[j], Q); }
    double* reds = red;                                                       // [16][Q][8] doubles
    float* redw = reinterpret_cast<float*>(red + 16 * Q * 8);                 // [16][Q][COUT * 4] floats
    if (rl < Q) {
        double* mine = reds + (size_t)(part * Q + rl) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) { mine[j] = s1[j]; mine[4 + j] = s2[j]; }
    }
    SENAS_PHASE(5);
    if (WG) {
#pragma unroll
        for (int co = 0; co < COUT; ++co)
#pragma unroll
            for (int j = 0; j < 4; ++j) wacc[co][j] = row_strided_sum(wacc[co][j], Q);
        if (rl < Q) {
            float* mine = redw + (size_t)(part * Q + rl) * (COUT * 4);
#pragma unroll
            for (int co = 0; co < COUT; ++co) stv<4>(mine + co * 4, wacc[co]);
        }
    }
    SENAS_PHASE(6);
    __syncthreads();
    SENAS_PHASE(7);
    if ((int)threadIdx.x < Q * 8) {
        const int qq = threadIdx.x >> 3, j8 = threadIdx.x & 7;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += reds[(size_t)(w * Q + qq) * 8 + j8];
        atomicAdd(it.sums + ((size_t)n * cin + 4 * qq + (j8 & 3)) * 2 + (j8 >> 2), v);
    }
    SENAS_PHASE(3);
    if (!WG) return;
    const int nel = COUT * cin;
    for (int e = threadIdx.x; e < nel; e += 256) {
        const int co = e / cin, ci = e - co * cin;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += (double)redw[(size_t)(w * Q + (ci >> 2)) * (COUT * 4) + co * 4 + (ci & 3)];
        // (an accumulator per image: the 256 blocks of a 256 x 256 problem on ONE address cost 12 - 20 us per launch)
        atomicAdd(it.dw_acc + ((size_t)n * kDwSlots + (blockIdx.x % kDwSlots)) * nel + e, v);
    }
    SENAS_PHASE(4);
}

template <int COUT>
__global__ __launch_bounds__(256) void dstail_bwd_apply_kernel(DsItems items, int nimg, long hw, int cin, long chunk) {
    __shared__ __attribute__((aligned(16))) float wl[kMaxCout * kMaxCin];
    __shared__ __attribute__((aligned(16))) float sc[2 * kMaxCin];
    __shared__ __attribute__((aligned(16))) float abk[3 * kMaxCin];
    const senas_dstail_item& it = items.it[blockIdx.z];
    const int n = blockIdx.y;
    const bool writer = blockIdx.x == 0 && n == 0;
    bn1_coefficients(it, 0, hw, cin, 1, 0.f, 0.f, false, false, sc);
    if ((int)threadIdx.x < cin) {
        const int ch = threadIdx.x;
        double S1 = 0.0, S2 = 0.0;
        for (int i = 0; i < nimg; ++i) { S1 += it.sums[((size_t)i * cin + ch) * 2]; S2 += it.sums[((size_t)i * cin + ch) * 2 + 1]; }
        const double mu = (double)it.mean_invstd[ch], is = (double)it.mean_invstd[cin + ch], gam = (double)it.gamma1[ch];
        const double mm = (double)nimg * (double)hw;
        const double g = gam * is, m1 = S1 / mm, m2 = (S2 - mu * S1) * is * is / mm;
        abk[ch] = (float)g;
        abk[cin + ch] = (float)(-g * m2);
        abk[2 * cin + ch] = (float)(g * (mu * m2 - m1));
        if (writer) {
            if (it.dgamma1 != nullptr) it.dgamma1[ch] = (float)((S2 - mu * S1) * is);
            if (it.dbeta1 != nullptr) it.dbeta1[ch] = (float)S1;
        }
    }
    if (writer && it.dw != nullptr)                              // the reduce launch's fp64 accumulators (kDwSlots per image) -> the fp32 gradient
        for (int i = threadIdx.x; i < COUT * cin; i += 256) {
            double v = 0.0;
#pragma unroll 8
            for (int im = 0; im < nimg * kDwSlots; ++im) v += it.dw_acc[(size_t)im * COUT * cin + i];       // (fixed order)
            it.dw[i] = (float)v;
        }
    for (int i = threadIdx.x; i < COUT * cin; i += 256) wl[i] = it.w[i];
    __syncthreads();
    if (it.dz1 == nullptr) return;
    const int Q = cin >> 2, q = threadIdx.x % Q, pl = threadIdx.x / Q, lanes = 256 / Q;
    const int dst = (int)it.dz2_pixel_stride;
    const bool vec = (dst & 3) == 0 && (reinterpret_cast<uintptr_t>(it.dz2) & 15) == 0;
    const float4 s4 = *reinterpret_cast<const float4*>(sc + 4 * q), b4 = *reinterpret_cast<const float4*>(sc + cin + 4 * q);
    const float4 a4 = *reinterpret_cast<const float4*>(abk + 4 * q), bb4 = *reinterpret_cast<const float4*>(abk + cin + 4 * q),
                 k4 = *reinterpret_cast<const float4*>(abk + 2 * cin + 4 * q);
    long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk;
    if (p1 > hw) p1 = hw;
    const size_t img = (size_t)n * hw;
    for (long p = p0 + pl; p < p1; p += lanes) {
        const size_t o4 = (img + p) * Q + q;
        const float4 z = reinterpret_cast<const float4*>(it.z1)[o4];
        float d[COUT], ds[4], mid[4];
        ds_quad<COUT>(it.dz2 + (img + p) * dst, vec, wl, cin, q, z, s4, b4, d, ds, mid);
        float4 r;
        r.x = fmaf(a4.x, ds[0], fmaf(bb4.x, z.x, k4.x));
        r.y = fmaf(a4.y, ds[1], fmaf(bb4.y, z.y, k4.y));
        r.z = fmaf(a4.z, ds[2], fmaf(bb4.z, z.z, k4.z));
        r.w = fmaf(a4.w, ds[3], fmaf(bb4.w, z.w, k4.w));
        reinterpret_cast<float4*>(it.dz1)[o4] = r;
    }
}

bool ds_ok(int k, int n, int64_t hw, int cin, int cout) {
    const int cv = cin / 4;
    return k >= 1 && k <= SENAS_MAX_DSTAIL && n >= 1 && n <= 65535 && hw >= 1 && cin % 4 == 0 && cin >= 4 && cin <= kMaxCin &&
           (cv & (cv - 1)) == 0 && (cout == 4 || cout == 8) && (int64_t)n * hw * cin < 0x7fffffffLL;
}

long ds_chunk(int64_t hw, int n, int k) {
    // enough blocks to fill the chip, at most 64 per image (the reduce ends in atomics on the image's accumulators and the
    // last block of a problem folds one partial weight gradient per block)
    long blocks = 1024 / ((long)n * k);
    if (blocks < 1) blocks = 1;
    if (blocks > 64) blocks = 64;
    long chunk = (hw + blocks - 1) / blocks;
    if (chunk < 64) chunk = 64;
    return chunk;
}

}  // namespace

SENAS_PHASE_READER(dstail)

}  // namespace senas

extern "C" int senas_dstail_fwd(const senas_dstail_item* items, int k, int n, int64_t hw, int cin, int cout, int training,
                                float momentum, float eps, void* stream) {
    using namespace senas;
    if (!ds_ok(k, n, hw, cin, cout)) return SENAS_EUNSUPPORTED;
    SENAS_REQUIRE(items, "dstail_fwd: null pointer");
    DsItems b{};
    bool want = true;
    for (int t = 0; t < k; ++t) {
        b.it[t] = items[t];
        SENAS_REQUIRE(b.it[t].z1 && b.it[t].gamma1 && b.it[t].beta1 && b.it[t].mean_invstd && b.it[t].w && b.it[t].z2, "dstail_fwd: null pointer");
        SENAS_REQUIRE(training ? b.it[t].stats1 != nullptr : (b.it[t].running_mean1 && b.it[t].running_var1), "dstail_fwd: statistics missing");
        want = want && b.it[t].stats2 != nullptr;
    }
    if (!want) for (int t = 0; t < k; ++t) b.it[t].stats2 = nullptr;
    // one pixel per thread; enough blocks to fill the chip, at most 64 per image (every block ends in 2 * cout atomics)
    long blocks = 2048 / ((long)n * k);
    if (blocks < 1) blocks = 1;
    if (blocks > 64) blocks = 64;
    long chunk = (hw + blocks - 1) / blocks;
    chunk = (chunk + 255) / 256 * 256;
    dim3 grid((unsigned)((hw + chunk - 1) / chunk), n, k);
    if (cout == 8) hipLaunchKernelGGL((dstail_fwd_kernel<8>), grid, dim3(256), 0, as_stream(stream), b, n, (long)hw, cin, chunk, training, momentum, eps);
    else hipLaunchKernelGGL((dstail_fwd_kernel<4>), grid, dim3(256), 0, as_stream(stream), b, n, (long)hw, cin, chunk, training, momentum, eps);
    return launch_status("dstail_fwd");
}

extern "C" int64_t senas_dstail_ws_bytes(int k, int n, int64_t hw, int cin, int cout) {
    using namespace senas;
    if (!ds_ok(k, n, hw, cin, cout)) return 0;
    return (int64_t)n * kDwSlots * cin * cout * sizeof(double);     // the fp64 weight-gradient accumulators of one problem: kDwSlots per image
}

extern "C" int senas_dstail_bwd(const senas_dstail_item* items, int k, int n, int64_t hw, int cin, int cout, void* stream) {
    using namespace senas;
    if (!ds_ok(k, n, hw, cin, cout)) return SENAS_EUNSUPPORTED;
    SENAS_REQUIRE(items, "dstail_bwd: null pointer");
    DsItems b{};
    bool wg = false;
    for (int t = 0; t < k; ++t) {
        b.it[t] = items[t];
        SENAS_REQUIRE(b.it[t].z1 && b.it[t].gamma1 && b.it[t].beta1 && b.it[t].mean_invstd && b.it[t].w && b.it[t].dz2 && b.it[t].sums,
                      "dstail_bwd: null pointer");
        if (b.it[t].dz2_pixel_stride <= 0) b.it[t].dz2_pixel_stride = cout;
        SENAS_REQUIRE(b.it[t].dz2_pixel_stride >= cout, "dstail_bwd: dz2 pixel stride smaller than c_out");
        wg = wg || b.it[t].dw != nullptr;
    }
    for (int t = 0; t < k; ++t)
        SENAS_REQUIRE(!wg || (b.it[t].dw && b.it[t].dw_acc), "dstail_bwd: weight gradients want dw and dw_acc for every problem");
    const long chunk = ds_chunk(hw, n, k);
    dim3 grid((unsigned)((hw + chunk - 1) / chunk), n, k);
    hipStream_t st = as_stream(stream);
    const size_t lds = (size_t)16 * (cin / 4) * (8 * sizeof(double) + 4 * cout * sizeof(float));     // [16 rows][Q] x (8 doubles, 4 * cout floats)
#define SENAS_DS(CO)                                                                                                   \
    do {                                                                                                               \
        if (wg) hipLaunchKernelGGL((dstail_bwd_reduce_kernel<CO, true>), grid, dim3(256), lds, st, b, (long)hw, cin, chunk);   \
        else hipLaunchKernelGGL((dstail_bwd_reduce_kernel<CO, false>), grid, dim3(256), lds, st, b, (long)hw, cin, chunk);     \
        hipLaunchKernelGGL((dstail_bwd_apply_kernel<CO>), grid, dim3(256), 0, st, b, n, (long)hw, cin, chunk);                 \
    } while (0)
    if (cout == 8) SENAS_DS(8);
    else SENAS_DS(4);
#undef SENAS_DS
    return launch_status("dstail_bwd");
}
