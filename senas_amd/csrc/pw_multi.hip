// k independent pointwise (1x1) convolutions of one shape in shared launches: the c_in -> c_out <= 8 second halves of the
// DepSepConv candidates (utils/operations.py:107-115: ... -> ReLU -> 1x1 conv -> BatchNorm2d) of the edges that leave one
// state of a search cell.  Each has its own input (its own depthwise output), its own weights and its own output;
// blockIdx.y = problem.  At 4 x 8 x H x W these are launch-bound: one launch forward (with the producer-side batch-norm
// statistics), one for the data gradients, two for the weight gradients, for all k <= SENAS_MAX_PWMULTI of them.
// All three are single passes over the fat (c_in-channel) tensor with the 8 x c_in weights in LDS.
#include "common.h"

namespace senas {

namespace {

struct PwTab {
    const float* a[SENAS_MAX_PWMULTI];      // forward: x_p; data gradient: dy_p; weight gradient: x_p
    const float* b[SENAS_MAX_PWMULTI];      // forward / data gradient: w_p; weight gradient: dy_p
    float* out[SENAS_MAX_PWMULTI];          // y_p / dx_p / partial rows of dw_p
    double* stats[SENAS_MAX_PWMULTI];
    float* dw[SENAS_MAX_PWMULTI];
};

constexpr int kMaxW = 8 * 64;               // c_out * c_in floats of one problem

// y[pix][co] = sum_ci x[pix][ci] * w[co][ci]; thread = 4 output channels of one pixel; block b owns P chunks of 256 elements
__global__ __launch_bounds__(256) void pw_multi_fwd_kernel(PwTab tab, int nimg, long hw, int cin, int cout, long total, int P) {
    __shared__ __attribute__((aligned(16))) float wl[kMaxW];
    const int p = blockIdx.y;
    const float* __restrict__ x = tab.a[p];
    float* __restrict__ y = tab.out[p];
    double* __restrict__ stats = tab.stats[p];
    for (int i = threadIdx.x; i < cout * cin; i += 256) wl[i] = tab.b[p][i];
    __syncthreads();
    Stats4 acc_st;
    stats_init4(acc_st);
    const bool uniform = P > 0;
    const int chunks = uniform ? P : 1, cv = cout >> 2;
    int n_blk = 0, ch_thr = 0;
    for (int kk = 0; kk < chunks; ++kk) {
        long idx = ((long)blockIdx.x * chunks + kk) * 256 + threadIdx.x;
        const bool active = idx < total;
        if (!active) idx = total - 1;
        const int ch = (int)(idx % cv) * 4;
        const long pix = idx / cv;
        const int n = (int)(pix / hw);
        n_blk = n; ch_thr = ch;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float4* xp = reinterpret_cast<const float4*>(x + (size_t)pix * cin);
        for (int c4 = 0; c4 < (cin >> 2); ++c4) {
            const float4 xv = xp[c4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 wv = *reinterpret_cast<const float4*>(wl + (ch + j) * cin + 4 * c4);
                acc[j] = fmaf(xv.x, wv.x, fmaf(xv.y, wv.y, fmaf(xv.z, wv.z, fmaf(xv.w, wv.w, acc[j]))));
            }
        }
        if (active) stv<4>(y + (size_t)pix * cout + ch, acc);
        stats_accumulate4(acc_st, stats, uniform, n, cout, ch, acc, active);
    }
    stats_flush4(acc_st, stats, uniform, n_blk, cout, ch_thr);
}

// dx[pix][ci] = sum_co dy[pix][co] * w[co][ci]; thread = 4 input channels of one pixel
__global__ __launch_bounds__(256) void pw_multi_dgrad_kernel(PwTab tab, int cin, int cout, long total) {
    __shared__ __attribute__((aligned(16))) float wl[kMaxW];
    const int p = blockIdx.y;
    const float* __restrict__ dy = tab.a[p];
    float* __restrict__ dx = tab.out[p];
    for (int i = threadIdx.x; i < cout * cin; i += 256) wl[i] = tab.b[p][i];
    __syncthreads();
    if (dx == nullptr) return;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cv = cin >> 2, c4 = (int)(idx % cv);
    const long pix = idx / cv;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* dp = dy + (size_t)pix * cout;
    for (int co = 0; co < cout; ++co) {
        const float d = dp[co];
        const float4 wv = *reinterpret_cast<const float4*>(wl + co * cin + 4 * c4);
        acc[0] = fmaf(d, wv.x, acc[0]); acc[1] = fmaf(d, wv.y, acc[1]); acc[2] = fmaf(d, wv.z, acc[2]); acc[3] = fmaf(d, wv.w, acc[3]);
    }
    stv<4>(dx + (size_t)pix * cin + 4 * c4, acc);
}

// partial dw[co][ci] over a chunk of pixels: thread = (pixel lane, 4 input channels), 8 x 4 sums in registers, lanes folded
// through LDS -> part[block][co][ci]; the sum launch adds the blocks in a fixed order
__global__ __launch_bounds__(256) void pw_multi_wgrad_part_kernel(PwTab tab, int cin, int cout, long npix, long chunk) {
    extern __shared__ __attribute__((aligned(16))) float red[];       // [256][32]
    const int p = blockIdx.y;
    const float* __restrict__ x = tab.a[p];
    const float* __restrict__ dy = tab.b[p];
    const int cv = cin >> 2, c4 = threadIdx.x % cv, pl = threadIdx.x / cv, lanes = 256 / cv;
    long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk;
    if (p1 > npix) p1 = npix;
    float acc[8][4];
#pragma unroll
    for (int co = 0; co < 8; ++co)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[co][j] = 0.f;
    for (long pix = p0 + pl; pix < p1; pix += lanes) {
        const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)pix * cin + 4 * c4);
        const float* dp = dy + (size_t)pix * cout;
#pragma unroll
        for (int co = 0; co < 8; ++co) {
            const float d = co < cout ? dp[co] : 0.f;
            acc[co][0] = fmaf(d, xv.x, acc[co][0]); acc[co][1] = fmaf(d, xv.y, acc[co][1]);
            acc[co][2] = fmaf(d, xv.z, acc[co][2]); acc[co][3] = fmaf(d, xv.w, acc[co][3]);
        }
    }
#pragma unroll
    for (int co = 0; co < 8; ++co)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(size_t)threadIdx.x * 32 + co * 4 + j] = acc[co][j];
    __syncthreads();
    float* part = tab.out[p] + (size_t)blockIdx.x * cout * cin;
    for (int e = threadIdx.x; e < cout * cin; e += 256) {
        const int co = e / cin, ci = e - co * cin;
        float v = 0.f;
        for (int l = 0; l < lanes; ++l) v += red[(size_t)(l * cv + (ci >> 2)) * 32 + co * 4 + (ci & 3)];
        part[e] = v;
    }
}

__global__ __launch_bounds__(256) void pw_multi_wgrad_sum_kernel(PwTab tab, int n_elem, int n_blocks) {
    const float* __restrict__ part = tab.out[blockIdx.y];
    float* __restrict__ dw = tab.dw[blockIdx.y];
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n_elem) return;
    float v = 0.f;
    for (int b = lane; b < n_blocks; b += 64) v += part[(size_t)b * n_elem + i];
    v = wave_sum(v);
    if (lane == 0) dw[i] = v;
}

bool pw_ok(int k, int n, int64_t hw, int cin, int cout) {
    const int cv = cin / 4, qv = cout / 4;
    return k >= 1 && k <= SENAS_MAX_PWMULTI && n >= 1 && hw >= 1 && cin % 4 == 0 && cin >= 4 && cin <= 64 && (cv & (cv - 1)) == 0 &&
           cout % 4 == 0 && cout >= 4 && cout <= 8 && (qv & (qv - 1)) == 0 && (int64_t)n * hw * cin < 0x7fffffffLL;
}

long pw_wgrad_blocks(int n, int64_t hw, int cin, long* chunk) {
    const long npix = (long)n * hw, lanes = 256 / (cin / 4);
    long nblk = (npix + 4 * lanes - 1) / (4 * lanes);           // ~4 pixels per thread
    if (nblk > 512) nblk = 512;
    if (nblk < 1) nblk = 1;
    *chunk = (npix + nblk - 1) / nblk;
    return (npix + *chunk - 1) / *chunk;
}

}  // namespace
}  // namespace senas

extern "C" int senas_pw_multi_fwd(int k, int n, int64_t hw, int cin, int cout, const float* const* x, const float* const* w,
                                  float* const* y, double* const* stats, void* stream) {
    using namespace senas;
    if (!pw_ok(k, n, hw, cin, cout)) return SENAS_EUNSUPPORTED;
    SENAS_REQUIRE(x && w && y, "pw_multi_fwd: null pointer");
    PwTab tab{};
    bool want = stats != nullptr;
    for (int p = 0; p < k; ++p) {
        SENAS_REQUIRE(x[p] && w[p] && y[p], "pw_multi_fwd: null pointer");
        tab.a[p] = x[p]; tab.b[p] = w[p]; tab.out[p] = y[p]; tab.stats[p] = want ? stats[p] : nullptr;
        want = want && tab.stats[p] != nullptr;
    }
    const long per_img = hw * (cout / 4), total = per_img * n;
    const int P = want ? stats_chunks_per_block(per_img, cout, total) : 0;
    dim3 grid((unsigned)((total + 256L * (P > 0 ? P : 1) - 1) / (256L * (P > 0 ? P : 1))), k);
    hipLaunchKernelGGL(pw_multi_fwd_kernel, grid, dim3(256), 0, as_stream(stream), tab, n, (long)hw, cin, cout, total, P);
    return launch_status("pw_multi_fwd");
}

extern "C" int senas_pw_multi_bwd_data(int k, int n, int64_t hw, int cin, int cout, const float* const* dy, const float* const* w,
                                       float* const* dx, void* stream) {
    using namespace senas;
    if (!pw_ok(k, n, hw, cin, cout)) return SENAS_EUNSUPPORTED;
    SENAS_REQUIRE(dy && w && dx, "pw_multi_bwd_data: null pointer");
    PwTab tab{};
    for (int p = 0; p < k; ++p) {
        SENAS_REQUIRE(dy[p] && w[p], "pw_multi_bwd_data: null pointer");
        tab.a[p] = dy[p]; tab.b[p] = w[p]; tab.out[p] = dx[p];            // dx[p] may be NULL: skipped
    }
    const long total = (long)n * hw * (cin / 4);
    hipLaunchKernelGGL(pw_multi_dgrad_kernel, dim3((unsigned)((total + 255) / 256), k), dim3(256), 0, as_stream(stream), tab, cin, cout, total);
    return launch_status("pw_multi_bwd_data");
}

extern "C" int64_t senas_pw_multi_ws_bytes(int k, int n, int64_t hw, int cin, int cout) {
    using namespace senas;
    if (!pw_ok(k, n, hw, cin, cout)) return 0;
    long chunk;
    return (int64_t)k * pw_wgrad_blocks(n, hw, cin, &chunk) * cin * cout * sizeof(float) + 256;
}

extern "C" int senas_pw_multi_bwd_weight(int k, int n, int64_t hw, int cin, int cout, const float* const* x, const float* const* dy,
                                         float* const* dw, void* ws, void* stream) {
    using namespace senas;
    if (!pw_ok(k, n, hw, cin, cout)) return SENAS_EUNSUPPORTED;
    SENAS_REQUIRE(x && dy && dw && ws, "pw_multi_bwd_weight: null pointer");
    long chunk;
    const long nblk = pw_wgrad_blocks(n, hw, cin, &chunk);
    PwTab tab{};
    for (int p = 0; p < k; ++p) {
        SENAS_REQUIRE(x[p] && dy[p] && dw[p], "pw_multi_bwd_weight: null pointer");
        tab.a[p] = x[p]; tab.b[p] = dy[p]; tab.dw[p] = dw[p];
        tab.out[p] = reinterpret_cast<float*>(ws) + (size_t)p * nblk * cin * cout;
    }
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(pw_multi_wgrad_part_kernel, dim3((unsigned)nblk, k), dim3(256), 256 * 32 * sizeof(float), st, tab, cin, cout,
                       (long)n * hw, chunk);
    const int n_elem = cin * cout;
    hipLaunchKernelGGL(pw_multi_wgrad_sum_kernel, dim3((n_elem + 3) / 4, k), dim3(256), 0, st, tab, n_elem, (int)nblk);
    return launch_status("pw_multi_bwd_weight");
}
