// Persistent, software-pipelined form of the stride-1 LDS-window convolution (conv_lds.hip) for maps that fill the chip
// with 8 x 32 tiles: the 5x5 dilated / 3x3 layers of the derived network at 128x128 and 256x256, forward and data gradient.
//
// What conv_lds.hip leaves on the table there (phase stamps, tools/phase_probe.py, 8 x 32 x 256 x 256, 5x5 dilation 3):
// a block spends 3-8 us per 16-channel pass waiting for its window (HBM round trips of 2-4 us, four pieces in flight per
// thread) and 7 us in its store epilogue, against 2 x 11 us of MFMA work -- the matrix pipe idles a third of the time
// and a second resident block does not fill the holes (both blocks share the pipe while they compute, so they drift
// into the same rhythm).
//
// Here a block keeps going over tiles (grid = 2 blocks per CU) and nothing it waits for is issued late:
//   * the window of the NEXT step (next channel pass, or pass 0 of the next tile) is requested piece by piece during
//     the first taps of the current step, parked in registers, and written to LDS between the two barriers that
//     separate the steps -- by then it has had most of a tap loop (5-20 us) to arrive;
//   * weight fragments run K - 1 taps ahead in a register ring (vmcnt retires in order: a weight fetch issued after
//     a window piece cannot be waited for without waiting for that piece, so EVERY load is issued at least K - 1
//     taps before its first use and the waits are counted ones, never vmcnt(0), inside the tap loop -- the loop is fully
//     unrolled so that the compiler can count);
//   * results leave straight from the accumulators (lane = channel: 128-byte runs per pixel), no LDS transpose, no
//     barrier between the last tap and the stores;
//   * next to no vector instructions besides the MFMAs (see the note at the kernel).
// Fragment conventions, the packed weight image and the XCD-contiguous tile order are conv_lds.hip's.
#include "common.h"

namespace senas {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int CH = 16, PST = 20, P4 = PST / 4;
constexpr int TW = 32, MT = 2, TH = 4 * MT;
constexpr int NT = 256, XL = NT / 4;

__device__ __forceinline__ f32x16 mfma32p(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ constexpr int acc_row_p(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

// On gfx950 the f32 MFMA runs at the vector rate and does NOT overlap with other vector instructions of its SIMD
// (tools/mfma_probe.hip: 16 integer instructions per 16 MFMAs take a bare loop from 154 to 115 TFLOP/s; LDS reads and
// memory instructions are free): every VALU cycle in this kernel is an MFMA cycle lost.  Hence the compile-time window
// geometry (K, DIL: all LDS offsets are instruction immediates), per-thread piece offsets computed once per block,
// uniform (scalar) base pointers with 32-bit lane offsets for every load and store, and a fast path for interior tiles.
template <bool TG, int K, int DIL>
__global__ __launch_bounds__(256, 2) void conv_pipe_kernel(GatherGeom g, const float* __restrict__ in, const float* __restrict__ wp,
                                                           float* __restrict__ out, int in_relu, const float* __restrict__ mask,
                                                           double* __restrict__ stats, int gx, int gy, int ntiles) {
    constexpr int T = K * K, D = K - 1, NB = K;           // weight ring: NB slots, fetched D taps ahead
    constexpr int HALO = DIL * (K / 2), TILE_W = TW + 2 * HALO, TILE_H = TH + 2 * HALO, WPIX = TILE_H * TILE_W;
    constexpr int NP = (WPIX + XL - 1) / XL;              // 16-byte window pieces per thread
    constexpr int DTY = XL / TILE_W, DTX = XL % TILE_W;
    constexpr int PF_TAPS = (T * 3) / 5;                  // the pieces of the next step are requested during the first 3/5 of the taps
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double red[4 * 32 * 2];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = g.hin, W = g.win, cin = g.cin, cout = g.cout;
    const int ngroups = cin >> 3, npass = cin / CH;

    // this block's tiles: block b runs on XCD b & 7 and takes every (gridDim.x / 8)-th tile of that XCD's contiguous eighth of
    // the tile list, so that the blocks of an XCD work on neighbouring tiles at any one time (their windows overlap by the halo)
    const unsigned xcd = blockIdx.x & 7u, stride_j = gridDim.x >> 3;
    const unsigned base = (unsigned)ntiles >> 3, rem = (unsigned)ntiles & 7u;
    const unsigned first = xcd * base + (xcd < rem ? xcd : rem), cnt = base + (xcd < rem ? 1u : 0u);
    unsigned j = blockIdx.x >> 3;
    if (j >= cnt) return;                                 // block-uniform

    int n, cot, oy0, ox0;
    auto decode = [&](unsigned jj, int& n_, int& cot_, int& oy_, int& ox_) {
        const unsigned lp = first + jj;
        const unsigned bx = lp % (unsigned)gx, by = (lp / (unsigned)gx) % (unsigned)gy, bz = lp / (unsigned)(gx * gy);
        n_ = (int)(bz % (unsigned)g.n);
        cot_ = (int)(bz / (unsigned)g.n);
        oy_ = (int)by * TH;
        ox_ = (int)bx * TW;
    };
    decode(j, n, cot, oy0, ox0);
    int pass = 0;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;

    float4* lds4 = reinterpret_cast<float4*>(lds);
    const float4* a_base = lds4 + ((MT * wave) * TILE_W + r) * P4 + h;   // this lane's pixel for tap offset (0, 0), sub-tile 0

    // ---- window pieces: slot k of this thread = window pixel k * 64 + (tid >> 2), 16-byte piece tid & 3.  Its offset from
    // the window's origin pixel (floats; the origin may lie outside the image, the sum never does for a piece that is used)
    const int sq = threadIdx.x & 3, spl = threadIdx.x >> 2;
    const int ty0 = spl / TILE_W, tx0 = spl - ty0 * TILE_W;
    const int rel_safe = (HALO * W + HALO) * cin + sq * 4;           // the tile's first output pixel: always inside the image
    int rel[NP];
    {
        int ty = ty0, tx = tx0;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            rel[k] = k * XL + spl < WPIX ? (ty * W + tx) * cin + sq * 4 : rel_safe;
            ty += DTY; tx += DTX;
            if (tx >= TILE_W) { tx -= TILE_W; ++ty; }
        }
    }
    float4 pf[NP];
    unsigned pf_ok = 0;                                               // bit k: piece k lies inside the image
    bool p_interior = true;                                           // the whole window lies inside the image (uniform)
    int p_oy = 0, p_ox = 0;
    const float* p_src = in;                                          // uniform: the window's origin pixel, current channel pass
    auto pf_begin = [&](int n_, int oy_, int ox_, int pass_) {
        p_oy = oy_ - HALO; p_ox = ox_ - HALO;
        p_interior = p_oy >= 0 && p_ox >= 0 && p_oy + TILE_H <= H && p_ox + TILE_W <= W;
        pf_ok = p_interior ? 0xffffffffu : 0u;
        p_src = in + (((long)n_ * H + p_oy) * W + p_ox) * cin + pass_ * CH;
    };
    auto pf_issue = [&](int k) {
        int off = rel[k];
        if (!p_interior) {                                            // border tiles only (uniform branch)
            const int wq = k * XL + spl, ty = wq / TILE_W, tx = wq - ty * TILE_W;
            const bool inb = (unsigned)(p_oy + ty) < (unsigned)H && (unsigned)(p_ox + tx) < (unsigned)W;
            off = inb ? off : rel_safe;
            pf_ok |= (inb ? 1u : 0u) << k;
        }
        pf[k] = *reinterpret_cast<const float4*>(p_src + off);
    };
    auto commit = [&](bool interior) {
        float4* dst = lds4 + spl * P4 + sq;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            float4 v = pf[k];
            if (in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (!interior && !((pf_ok >> k) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((k + 1) * XL <= WPIX || k * XL + spl < WPIX) dst[k * XL * P4] = v;
        }
    };

    // ---- weight ring: uniform fragment pointers, the lane's 16 bytes as a 32-bit offset
    float4 bq[NB][2];
    const int lane4 = lane * 4;
    const size_t wstep = (size_t)ngroups * 256;
    auto wptr = [&](int cot_, int pass_) { return wp + ((size_t)cot_ * T * ngroups + pass_ * 2) * 256; };
    auto load_b = [&](const float* wt, int slot) {
        bq[slot][0] = *reinterpret_cast<const float4*>(wt + lane4);
        bq[slot][1] = *reinterpret_cast<const float4*>(wt + 256 + lane4);
    };

    // prologue: the first step's window and its first D weight fragments, in the open
    pf_begin(n, oy0, ox0, 0);
#pragma unroll
    for (int k = 0; k < NP; ++k) pf_issue(k);
    {
        const float* wt = wptr(cot, 0);
#pragma unroll
        for (int u = 0; u < D; ++u) load_b(wt + (size_t)u * wstep, u);
    }

    int sidx = 0;                                                     // (phase stamps only)
    SENAS_PHASE(0);
    for (;;) {
        // the step after this one
        int n_n = n, cot_n = cot, oy_n = oy0, ox_n = ox0, pass_n = pass + 1;
        unsigned j_n = j;
        bool have_next = true;
        if (pass_n == npass) {
            pass_n = 0;
            j_n = j + stride_j;
            have_next = j_n < cnt;
            if (have_next) decode(j_n, n_n, cot_n, oy_n, ox_n);
        }
        __syncthreads();                                              // the previous step's readers are done
        if (sidx < 12) SENAS_PHASE(1 + sidx * 5);
        commit(p_interior);
        if (sidx < 12) SENAS_PHASE(2 + sidx * 5);
        __syncthreads();
        if (sidx < 12) SENAS_PHASE(3 + sidx * 5);
#ifdef SENAS_PHASES
        if (sidx == 1 && blockIdx.x == 0 && threadIdx.x == 0) senas_phase_buf[61] = __builtin_amdgcn_s_memtime();
#endif
        // without a next step the prefetches still run (on this tile's pass 0, never committed): no data-dependent branch
        // inside the unrolled tap loop, so the counted waits stay exact
        pf_begin(n_n, oy_n, ox_n, pass_n);
        const float* wt_cur = wptr(cot, pass);
        const float* wt_nxt = wptr(cot_n, pass_n);

        float4 af[2][2][MT];
        auto load_a = [&](int t, int slot) {
            const int ky = t / K, kx = t - ky * K;
            // plain gather: window row = oy_local + ky*d ; transposed (stride 1): oy_local + (k-1-ky)*d
            const int dy = (TG ? (K - 1 - ky) : ky) * DIL, dx = (TG ? (K - 1 - kx) : kx) * DIL;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int m = 0; m < MT; ++m) af[slot][c2][m] = a_base[(m * TILE_W + dy * TILE_W + dx) * P4 + c2 * 2];
        };
        load_a(0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int as = t & 1, bs = t % NB;
            if (t + 1 < T) load_a(t + 1, as ^ 1);
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma32p(af[as][c2][m].x, bq[bs][c2].x, acc[m]);
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    acc[m] = mfma32p(af[as][c2][m].y, bq[bs][c2].y, acc[m]);
                    acc[m] = mfma32p(af[as][c2][m].z, bq[bs][c2].z, acc[m]);
                    acc[m] = mfma32p(af[as][c2][m].w, bq[bs][c2].w, acc[m]);
                }
            __builtin_amdgcn_sched_barrier(0);
            // (t + D) % NB == (t - 1) % NB: the slot of the tap before this one takes the fragment D taps ahead
            if (t + D < T) load_b(wt_cur + (size_t)(t + D) * wstep, (t + D) % NB);
            else load_b(wt_nxt + (size_t)(t + D - T) * wstep, (t + D) % NB);
#pragma unroll
            for (int k = 0; k < NP; ++k)
                if (k * PF_TAPS / NP == t) pf_issue(k);
            __builtin_amdgcn_sched_barrier(0);
        }

        if (sidx < 12) SENAS_PHASE(4 + sidx * 5);
#ifdef SENAS_PHASES
        if (sidx == 1 && blockIdx.x == 0 && threadIdx.x == 0) senas_phase_buf[62] = __builtin_amdgcn_s_memtime();
#endif
        if (pass == npass - 1) {
            // ---- epilogue: lane = output channel, register v = pixel (row MT*wave + m, column acc_row(v, h)); the stores
            // of a row share one uniform base pointer, the lane adds (4 h) * cout + r
            const int co = cot * 32 + r;
            const bool cok = co < cout;
            const bool full = oy0 + TH <= g.hout && ox0 + TW <= g.wout && cot * 32 + 32 <= cout;      // uniform
            const int lane_o = 4 * h * cout + r;
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int oy = oy0 + MT * wave + m;
                const size_t row = (((size_t)n * g.hout + oy) * g.wout + ox0) * cout + cot * 32;
                float* orow = out + row;
                const float* mrow = mask + row;
                if (full) {
                    float mk[16];
                    if (mask != nullptr) {
#pragma unroll
                        for (int v = 0; v < 16; ++v) mk[v] = (mrow + acc_row_p(v, 0) * cout)[lane_o];
                    }
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        float val = acc[m][v];
                        acc[m][v] = 0.f;
                        if (mask != nullptr && !(mk[v] > 0.f)) val = 0.f;
                        s += val;
                        q += (double)val * val;
                        (orow + acc_row_p(v, 0) * cout)[lane_o] = val;
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int ox = ox0 + acc_row_p(v, h);
                        float val = acc[m][v];
                        acc[m][v] = 0.f;
                        if (oy < g.hout && ox < g.wout && cok) {
                            const int o = acc_row_p(v, 0) * cout + lane_o;
                            if (mask != nullptr && !(mrow[o] > 0.f)) val = 0.f;
                            s += val;
                            q += (double)val * val;
                            orow[o] = val;
                        }
                    }
                }
            }
            if (sidx == 1) SENAS_PHASE(50);
            if (stats != nullptr) {                                   // block-level reduction: 2 atomics per channel per tile
                s += __shfl_xor(s, 32, 64);
                q += __shfl_xor(q, 32, 64);
                if (h == 0) { red[(wave * 32 + r) * 2] = s; red[(wave * 32 + r) * 2 + 1] = q; }
                if (sidx == 1) SENAS_PHASE(51);
                __syncthreads();
                if (sidx == 1) SENAS_PHASE(52);
                if (wave == 0 && h == 0 && cok) {
                    for (int w = 1; w < 4; ++w) { s += red[(w * 32 + r) * 2]; q += red[(w * 32 + r) * 2 + 1]; }
                    double* st = stats + ((size_t)n * cout + co) * 2;
                    atomicAdd(st, s);
                    atomicAdd(st + 1, q);
                }
            }
        }
        if (sidx < 12) SENAS_PHASE(5 + sidx * 5);
        ++sidx;
        if (!have_next) break;
        n = n_n; cot = cot_n; oy0 = oy_n; ox0 = ox_n; pass = pass_n; j = j_n;
    }
}

template <bool TG, int K, int DIL>
int launch_pipe_variant(const GatherGeom& g, const float* in, const float* wp, float* out, int in_relu, const float* mask,
                        double* stats, hipStream_t st) {
    constexpr int HALO = DIL * (K / 2);
    constexpr size_t bytes = (size_t)(TH + 2 * HALO) * (TW + 2 * HALO) * PST * sizeof(float);
    static_assert(bytes <= 78 * 1024, "two blocks per CU");
    static bool attr_set = false;
    if (bytes > 64 * 1024 && !attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pipe_kernel<TG, K, DIL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024);
        if (e != hipSuccess) { set_error("conv_pipe: raising the dynamic LDS limit", e); return SENAS_ELAUNCH; }
        attr_set = true;
    }
    const int gx = (g.wout + TW - 1) / TW, gy = (g.hout + TH - 1) / TH;
    const long ntiles = (long)gx * gy * g.n * ((g.cout + 31) / 32);
    hipLaunchKernelGGL((conv_pipe_kernel<TG, K, DIL>), dim3(512), dim3(NT), bytes, st, g, in, wp, out, in_relu, mask, stats, gx, gy,
                       (int)ntiles);
    return launch_status("conv_pipe");
}

}  // namespace

// stride 1, "same" padding, 3x3 (dilation 1) or 5x5 (dilation 1-3), 16-channel passes, enough 8 x 32 tiles for two blocks per CU
bool pipe_gather_ok(const GatherGeom& g) {
    if (g.stride != 1 || g.cin % CH != 0 || g.cin < CH) return false;
    if (g.kh != g.kw || g.pad != g.dil * (g.kh / 2) || g.hout != g.hin || g.wout != g.win) return false;
    if (!((g.kh == 3 && g.dil == 1) || (g.kh == 5 && g.dil >= 1 && g.dil <= 3))) return false;
    if (g.wout < 32 || g.hout < 8) return false;
    const long ntiles = (long)((g.wout + TW - 1) / TW) * ((g.hout + TH - 1) / TH) * g.n * ((g.cout + 31) / 32);
    if (ntiles < 512 || ntiles > 0x3fffffffL) return false;
    return (long)g.n * g.hin * g.win * g.cin < 0x7fffffffL && (long)g.n * g.hout * g.wout * g.cout < 0x7fffffffL;
}

template <bool TG>
int launch_pipe_gather(const GatherGeom& g, const float* in, const float* wp, float* out, int in_relu, const float* mask,
                       double* stats, hipStream_t st) {
    if (g.kh == 3) return launch_pipe_variant<TG, 3, 1>(g, in, wp, out, in_relu, mask, stats, st);
    if (g.dil == 1) return launch_pipe_variant<TG, 5, 1>(g, in, wp, out, in_relu, mask, stats, st);
    if (g.dil == 2) return launch_pipe_variant<TG, 5, 2>(g, in, wp, out, in_relu, mask, stats, st);
    return launch_pipe_variant<TG, 5, 3>(g, in, wp, out, in_relu, mask, stats, st);
}

SENAS_PHASE_READER(conv_pipe)

// the kernel symbol launch_pipe_gather picks (for senas_conv2d_kernel_name)
void pipe_gather_name(const GatherGeom& g, bool tg, char* buf, int len) {
    snprintf(buf, len, "conv_pipe_kernel<%s, %d, %d>", tg ? "true" : "false", g.kh, g.kh == 3 ? 1 : g.dil);
}

template int launch_pipe_gather<false>(const GatherGeom&, const float*, const float*, float*, int, const float*, double*, hipStream_t);
template int launch_pipe_gather<true>(const GatherGeom&, const float*, const float*, float*, int, const float*, double*, hipStream_t);

}  // namespace senas
