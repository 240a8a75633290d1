// Dice + cross-entropy loss and the segmentation metric as device kernels.
//
// Reference: utils/loss/loss.py:45-70,124-228 (DiceCrossEntropyLoss = nn.CrossEntropyLoss + SoftDiceLoss over
// batch+space, background dropped, smooth 1e-5, denominator + 1e-8) builds a CPU one-hot, copies it to the
// device and materialises tp/fp/fn tensors; utils/metrics.py:127-173 syncs with the host three times per step.
// Here: one pass over the logits per direction, no intermediate tensors, no host round trip.
//   forward : per pixel softmax -> sum of -log p[target], and per class  S_c = sum p_c, T_c = sum p_c [t = c],
//             N_c = sum [t = c]   (tp = T, fp = S - T, fn = N - T), fp64 block partials -> fp64 atomics;
//             a one-thread epilogue turns them into the loss and into d loss / d (tp, fp, fn)
//   backward: softmax recomputed from the logits, gradient through softmax + CE written in one pass
// logits: [pixels][c] (NHWC), c <= 8; target: int64 [pixels].
#include "common.h"

namespace senas {

constexpr int kMaxClass = 8;

__device__ __forceinline__ void softmax_c(const float* __restrict__ lp, int c, float (&p)[kMaxClass], float& lse) {
    float m = lp[0];
    for (int j = 1; j < c; ++j) m = fmaxf(m, lp[j]);
    float s = 0.f;
    for (int j = 0; j < c; ++j) { p[j] = expf(lp[j] - m); s += p[j]; }
    const float inv = 1.f / s;
    for (int j = 0; j < c; ++j) p[j] *= inv;
    lse = m + logf(s);
}

// acc: double[1 + 3c] = { sum of -log p[t],  S[c], T[c], N[c] }
__global__ __launch_bounds__(256) void dice_ce_reduce_kernel(long npix, int c, const float* __restrict__ logits,
                                                             const int64_t* __restrict__ target, double* __restrict__ acc) {
    __shared__ double red[4][1 + 3 * kMaxClass];
    double ce = 0.0, S[kMaxClass], T[kMaxClass], N[kMaxClass];
#pragma unroll
    for (int j = 0; j < kMaxClass; ++j) S[j] = T[j] = N[j] = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
        float p[kMaxClass], lse;
        const float* lp = logits + i * c;
        softmax_c(lp, c, p, lse);
        const int t = (int)target[i];
#pragma unroll
        for (int j = 0; j < kMaxClass; ++j) {
            if (j < c) {
                S[j] += (double)p[j];
                if (j == t) { T[j] += (double)p[j]; N[j] += 1.0; ce += (double)(lse - lp[j]); }
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    ce = wave_sum(ce);
    if (lane == 0) red[wave][0] = ce;
#pragma unroll
    for (int j = 0; j < kMaxClass; ++j) {
        if (j < c) {
            const double s = wave_sum(S[j]), t = wave_sum(T[j]), n = wave_sum(N[j]);
            if (lane == 0) { red[wave][1 + j] = s; red[wave][1 + c + j] = t; red[wave][1 + 2 * c + j] = n; }
        }
    }
    __syncthreads();
    if (threadIdx.x < 1 + 3 * c)
        atomicAdd(&acc[threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// loss[0] = w_ce * ce / npix + w_dice * (1 - mean_c dc_c);   coef = { g_tp[c], g_fp[c], g_fn[c], w_ce / npix }
__global__ void dice_ce_finalize_kernel(long npix, int c, const double* __restrict__ acc, float w_ce, float w_dice, float smooth,
                                        int do_bg, float* __restrict__ loss, float* __restrict__ coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int c0 = do_bg ? 0 : 1, nc = c - c0;
    double dsum = 0.0;
    for (int j = 0; j < c; ++j) {
        const double tp = acc[1 + c + j], fp = acc[1 + j] - tp, fn = acc[1 + 2 * c + j] - tp;
        const double num = 2.0 * tp + smooth, den = 2.0 * tp + fp + fn + smooth + 1e-8;
        double gtp = 0.0, gfp = 0.0, gfn = 0.0;
        if (j >= c0 && nc > 0) {
            dsum += num / den;
            // loss_dice = 1 - (1/nc) sum dc:  d/dtp = -(1/nc) (2 den - 2 num) / den^2,  d/dfp = d/dfn = (1/nc) num / den^2
            const double k = (double)w_dice / nc;
            gtp = -k * (2.0 * den - 2.0 * num) / (den * den);
            gfp = k * num / (den * den);
            gfn = gfp;
        }
        coef[j] = (float)gtp; coef[c + j] = (float)gfp; coef[2 * c + j] = (float)gfn;
    }
    coef[3 * c] = w_ce / (float)npix;
    const double dice = nc > 0 ? 1.0 - dsum / nc : 0.0;
    loss[0] = (float)((double)w_ce * acc[0] / (double)npix + (double)w_dice * dice);
}

__global__ __launch_bounds__(256) void dice_ce_bwd_kernel(long npix, int c, const float* __restrict__ logits,
                                                          const int64_t* __restrict__ target, const float* __restrict__ coef,
                                                          const float* __restrict__ dloss, float* __restrict__ dlogits) {
    __shared__ float cf[3 * kMaxClass + 1];
    if (threadIdx.x < 3 * c + 1) cf[threadIdx.x] = coef[threadIdx.x];
    __syncthreads();
    const float up = dloss != nullptr ? dloss[0] : 1.f;
    const float kce = cf[3 * c];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
        float p[kMaxClass], lse, dp[kMaxClass];
        const float* lp = logits + i * c;
        softmax_c(lp, c, p, lse);
        const int t = (int)target[i];
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxClass; ++j) {
            if (j < c) {
                dp[j] = j == t ? cf[j] - cf[2 * c + j] : cf[c + j];        // d loss_dice / d p_j
                dot = fmaf(p[j], dp[j], dot);
            }
        }
#pragma unroll
        for (int j = 0; j < kMaxClass; ++j)
            if (j < c) dlogits[i * c + j] = up * (p[j] * (dp[j] - dot) + kce * (p[j] - (j == t ? 1.f : 0.f)));
    }
}

// ---- metric: arg-max, per-image (labelled, correct) pixels, per-class tp / fp / fn counts (utils/metrics.py:127-173)
// part: int64 [n][2] then [c-1][3], zeroed.  "correct" follows the reference literally: sum of (argmax & (target > 0)),
// i.e. the low bit of the arg-max index on foreground pixels (metrics.py:133-139) -- the intersection for 2 classes.
__global__ __launch_bounds__(256) void seg_metric_count_kernel(long hw, int c, const float* __restrict__ logits,
                                                               const int64_t* __restrict__ target,
                                                               unsigned long long* __restrict__ part, int nimg) {
    __shared__ unsigned long long red[2 + 3 * (kMaxClass - 1)];
    const int n = blockIdx.y;
    if (threadIdx.x < 2 + 3 * (c - 1)) red[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned labeled = 0, correct = 0, cnt[3 * (kMaxClass - 1)];
#pragma unroll
    for (int j = 0; j < 3 * (kMaxClass - 1); ++j) cnt[j] = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
        const float* lp = logits + ((long)n * hw + i) * c;
        int best = 0;
        float bv = lp[0];
        for (int j = 1; j < c; ++j) if (lp[j] > bv) { bv = lp[j]; best = j; }     // first maximum, as torch.argmax
        const int t = (int)target[(long)n * hw + i];
        if (t > 0) { labeled += 1; correct += (unsigned)(best & 1); }
#pragma unroll
        for (int k = 1; k < kMaxClass; ++k) {
            if (k < c) {
                const bool pk = best == k, tk = t == k;
                cnt[3 * (k - 1)] += pk && tk;
                cnt[3 * (k - 1) + 1] += pk && !tk;
                cnt[3 * (k - 1) + 2] += !pk && tk;
            }
        }
    }
    atomicAdd(&red[0], (unsigned long long)labeled);
    atomicAdd(&red[1], (unsigned long long)correct);
#pragma unroll
    for (int j = 0; j < 3 * (kMaxClass - 1); ++j)
        if (j < 3 * (c - 1)) atomicAdd(&red[2 + j], (unsigned long long)cnt[j]);
    __syncthreads();
    if (threadIdx.x < 2) atomicAdd(&part[(size_t)n * 2 + threadIdx.x], red[threadIdx.x]);
    else if (threadIdx.x < 2 + 3 * (c - 1)) atomicAdd(&part[(size_t)nimg * 2 + threadIdx.x - 2], red[threadIdx.x]);
}

// counts[c-1][3] += this batch;  acc_sum[0] += mean_i (correct_i + eps) / (labeled_i + eps)  (float32 arithmetic, as the reference)
__global__ void seg_metric_finalize_kernel(int nimg, int c, const unsigned long long* __restrict__ part, float eps,
                                           long long* __restrict__ counts, double* __restrict__ acc_sum) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s = 0.f;
    for (int i = 0; i < nimg; ++i) s += ((float)part[2 * i + 1] + eps) / ((float)part[2 * i] + eps);
    acc_sum[0] += (double)(s / (float)nimg);
    for (int j = 0; j < 3 * (c - 1); ++j) counts[j] += (long long)part[(size_t)nimg * 2 + j];
}

}  // namespace senas

using namespace senas;

extern "C" int senas_dice_ce_fwd(int64_t npix, int c, const float* logits, const int64_t* target, float w_ce, float w_dice,
                                 float smooth, int do_bg, double* acc, float* loss, float* coef, void* stream) {
    SENAS_REQUIRE(npix > 0 && c >= 1 && c <= kMaxClass, "dice_ce_fwd: 1..8 classes");
    SENAS_REQUIRE(logits && target && acc && loss && coef, "dice_ce_fwd: null pointer");
    hipStream_t st = as_stream(stream);
    long blocks = (npix + 1023) / 1024;               // 4 pixels per thread
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(dice_ce_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (long)npix, c, logits, target, acc);
    hipLaunchKernelGGL(dice_ce_finalize_kernel, dim3(1), dim3(64), 0, st, (long)npix, c, acc, w_ce, w_dice, smooth, do_bg, loss, coef);
    return launch_status("dice_ce_fwd");
}

extern "C" int senas_dice_ce_bwd(int64_t npix, int c, const float* logits, const int64_t* target, const float* coef,
                                 const float* dloss, float* dlogits, void* stream) {
    SENAS_REQUIRE(npix > 0 && c >= 1 && c <= kMaxClass, "dice_ce_bwd: 1..8 classes");
    SENAS_REQUIRE(logits && target && coef && dlogits, "dice_ce_bwd: null pointer");
    long blocks = (npix + 511) / 512;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dice_ce_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (long)npix, c, logits, target, coef,
                       dloss, dlogits);
    return launch_status("dice_ce_bwd");
}

extern "C" int senas_seg_metric_update(int n, int64_t hw, int c, const float* logits, const int64_t* target, float eps,
                                       void* part_zeroed, int64_t* counts, double* acc_sum, void* stream) {
    SENAS_REQUIRE(n > 0 && n <= 65535 && hw > 0 && c >= 2 && c <= kMaxClass, "seg_metric_update: bad sizes");
    SENAS_REQUIRE(logits && target && part_zeroed && counts && acc_sum, "seg_metric_update: null pointer");
    hipStream_t st = as_stream(stream);
    long blocks = (hw + 1023) / 1024;
    if (blocks > 256) blocks = 256;
    unsigned long long* part = reinterpret_cast<unsigned long long*>(part_zeroed);
    hipLaunchKernelGGL(seg_metric_count_kernel, dim3((unsigned)blocks, n), dim3(256), 0, st, (long)hw, c, logits, target, part, n);
    hipLaunchKernelGGL(seg_metric_finalize_kernel, dim3(1), dim3(64), 0, st, n, c, part, eps, reinterpret_cast<long long*>(counts), acc_sum);
    return launch_status("seg_metric_update");
}
