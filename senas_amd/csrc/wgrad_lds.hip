// Weight gradient of a stride-1 "same" convolution with both operands staged in LDS, gfx950.
//
//   dW[tap][a][b] = sum over pixels p of  X[p + off(tap)][a] * G[p][b]        (a: in-channel, b: out-channel)
//
// GEMM view per tap: M = 32 in-channels, N = 32 out-channels, K = pixels; v_mfma_f32_32x32x2_f32 eats two
// pixels per instruction.  A block (8 waves) walks 8x32 (or 4x32) pixel tiles persistently:
//   - the X window (tile + halo, zero-filled borders, ReLU on load) and the G tile go to LDS once per tile,
//     so the k*k taps re-read X from LDS instead of L1 and the tap loop has no bounds checks at all;
//   - the work units (tap, 32-channel slice of a) are dealt round-robin to the 8 waves; each wave keeps one
//     32x32 accumulator per unit in registers ACROSS tiles and adds it to the result once, at the end --
//     atomic traffic is (units x 4 KiB) per block instead of per 128 pixels.
//   - K order inside a 32-pixel row is permuted (lane half h takes pixels 16h .. 16h+15) so that both
//     operands advance by one pixel per MFMA with compile-time LDS offsets.
#include "common.h"

namespace senas {

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

constexpr int TW = 32;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

}  // namespace

// A: in-channels (multiple of 32, <= 128); UW: accumulators per wave; RS: row split -- the 8 waves form
// 8/RS unit groups x RS row lanes: wave (ug, rs) owns units ug, ug + 8/RS, ... and the tile rows rs, rs + RS, ...
// (few taps -> large RS, so a 1x1 or 3x3 kernel still keeps all 8 waves on the matrix cores; the row lanes'
// partial sums meet in LDS before the one atomic pass).  grid = persistent blocks of 512 threads.
// PFX: 16-byte pieces of the NEXT tile's X window each thread requests before the K loop of the current tile
// and parks in registers until the loop is done (the G tile always travels that way), so the HBM/L2 latency
// of staging hides behind the MFMAs; pieces beyond PFX*512 are fetched after the loop.
// dynamic LDS: X window [(th + 2*halo) * (32 + 2*halo)][A] floats, then G tile [th * 32][32] floats.
template <int A, int UW, int RS, int PFX>
__global__ __launch_bounds__(512) void wgrad_lds_kernel(WgradGeom g, const float* __restrict__ X,
                                                        const float* __restrict__ G, float* __restrict__ dwp,
                                                        int x_relu, int th, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int UG = 8 / RS;
    constexpr int PP = A / 4;                    // 16-byte pieces per pixel of X
    constexpr int XL = 512 / PP;                 // pixels of the window covered by one slot
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform
    const int rs = wave % RS, ug = wave / RS;
    const int r = lane & 31, h = lane >> 5;
    const int halo = g.pad;
    const int tile_w = TW + 2 * halo, tile_h = th + 2 * halo;
    float* xs = lds;
    float* gs = lds + tile_h * tile_w * A;
    float4* xs4 = reinterpret_cast<float4*>(xs);
    float4* gs4 = reinterpret_cast<float4*>(gs);
    constexpr int a_tiles = A / 32;
    const int taps = g.kh * g.kw;
    const int units = taps * a_tiles;

    // this wave's units: u = ug + UG*t; slots past the last unit reuse unit 0's operands
    int uoff[UW];                 // LDS float offset of the unit's tap shift + channel slice
#pragma unroll
    for (int t = 0; t < UW; ++t) {
        const int u = ug + UG * t;
        const int uc = u < units ? u : 0;
        const int tap = uc / a_tiles, at = uc - tap * a_tiles;
        const int ky = tap / g.kw, kx = tap - ky * g.kw;
        uoff[t] = ((ky * g.dil) * tile_w + kx * g.dil) * A + at * 32 + r;
    }
    f32x16 acc[UW];
#pragma unroll
    for (int t = 0; t < UW; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

    // staging geometry of this thread (tile-independent): slot k holds window pixel k*XL + xpl, piece xq
    const int xq = threadIdx.x % PP, xpl = threadIdx.x / PP;       // threads with xpl >= XL idle while X is staged (A = 96)
    const int xslot = xpl * PP + xq;
    const int ty0 = xpl / tile_w, tx0 = xpl - ty0 * tile_w;
    const int wpix = tile_h * tile_w;            // pixels in the X window
    const int gpix = th * TW;                    // pixels in the G tile
    const int gq = threadIdx.x & 7, gpl = threadIdx.x >> 3;      // G: 8 pieces per pixel, 64 pixels per slot
    const int per_img = tiles_x * tiles_y;
    const int ntiles = g.n * per_img;
    float4 px[PFX > 0 ? PFX : 1], pg[4];

    // request the first PFX slots of X and the whole G tile of `tile` (loads only; borders resolved in commit)
    auto issue = [&](int tile) {
        const int n = tile / per_img, tr = tile - n * per_img;
        const int oy0 = (tr / tiles_x) * th, ox0 = (tr % tiles_x) * TW;
        const float* src = X + (size_t)n * g.hi * g.wi * A + xq * 4;
        int ty = ty0, tx = tx0;
#pragma unroll
        for (int k = 0; k < PFX; ++k) {
            const int iy = oy0 - halo + ty, ix = ox0 - halo + tx;
            const bool inb = xpl < XL && k * XL + xpl < wpix && iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi;
            px[k] = *reinterpret_cast<const float4*>(src + (inb ? ((size_t)iy * g.wi + ix) * A : 0));
            tx += XL;
            { const int w1 = tx >= tile_w; tx -= w1 ? tile_w : 0; ty += w1; const int w2 = tx >= tile_w; tx -= w2 ? tile_w : 0; ty += w2; }   // XL <= 64 <= 2*tile_w
        }
        if (g.B == 32) {
            const float* gsrc = G + (size_t)n * g.hg * g.wg * 32 + gq * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pix = k * 64 + gpl;
                const int gy = oy0 + (pix >> 5), gx = ox0 + (pix & 31);
                const bool inb = pix < gpix && gy < g.hg && gx < g.wg;
                pg[k] = *reinterpret_cast<const float4*>(gsrc + (inb ? ((size_t)gy * g.wg + gx) * 32 : 0));
            }
        }
    };
    // registers -> LDS (ReLU, zero borders), then whatever did not fit in the prefetch slots
    auto commit = [&](int tile) {
        const int n = tile / per_img, tr = tile - n * per_img;
        const int oy0 = (tr / tiles_x) * th, ox0 = (tr % tiles_x) * TW;
        int ty = ty0, tx = tx0;
#pragma unroll
        for (int k = 0; k < PFX; ++k) {
            const int iy = oy0 - halo + ty, ix = ox0 - halo + tx;
            const bool live = xpl < XL && k * XL + xpl < wpix;
            const bool inb = live && iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi;
            float4 v = px[k];
            if (x_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (!inb) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) xs4[k * XL * PP + xslot] = v;
            tx += XL;
            { const int w1 = tx >= tile_w; tx -= w1 ? tile_w : 0; ty += w1; const int w2 = tx >= tile_w; tx -= w2 ? tile_w : 0; ty += w2; }   // XL <= 64 <= 2*tile_w
        }
        const float* src = X + (size_t)n * g.hi * g.wi * A + xq * 4;
        for (int k0 = PFX; k0 * XL < wpix; k0 += 4) {           // remainder, 4 loads in flight
            float4 v[4];
            bool ok[4], lv[4];
            int tyy = ty, txx = tx;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int iy = oy0 - halo + tyy, ix = ox0 - halo + txx;
                lv[u] = xpl < XL && (k0 + u) * XL + xpl < wpix;
                ok[u] = lv[u] && iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi;
                v[u] = *reinterpret_cast<const float4*>(src + (ok[u] ? ((size_t)iy * g.wi + ix) * A : 0));
                txx += XL;
                { const int w1 = txx >= tile_w; txx -= w1 ? tile_w : 0; tyy += w1; const int w2 = txx >= tile_w; txx -= w2 ? tile_w : 0; tyy += w2; }
            }
            ty = tyy; tx = txx;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (x_relu) { v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f); v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f); }
                if (!ok[u]) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lv[u]) xs4[(k0 + u) * XL * PP + xslot] = v[u];
            }
        }
        if (g.B == 32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pix = k * 64 + gpl;
                const int gy = oy0 + (pix >> 5), gx = ox0 + (pix & 31);
                float4 v = pg[k];
                if (!(gy < g.hg && gx < g.wg)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pix < gpix) gs4[k * 512 + threadIdx.x] = v;
            }
        } else {
            const float* gsrc = G + (size_t)n * g.hg * g.wg * g.B;
            for (int idx = threadIdx.x; idx < gpix * 32; idx += 512) {   // B < 32: zero-pad the columns
                const int pix = idx >> 5, b = idx & 31;
                const int gy = oy0 + (pix >> 5), gx = ox0 + (pix & 31);
                float v = 0.f;
                if (b < g.B && gy < g.hg && gx < g.wg) v = gsrc[((size_t)gy * g.wg + gx) * g.B + b];
                gs[idx] = v;
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                  // previous tile's readers are done
        commit(tile);
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);     // in flight during the K loop
        // ---- K loop: rows of the tile, 16 MFMA steps per row (lane half h covers pixels 16h .. 16h+15).
        // Straight-line code: slots past the last unit compute on unit 0's operands and are dropped in the
        // epilogue (they sit in the shadow of the waves that own a real unit there), and the operands of step
        // s+1 are requested before the MFMAs of step s are issued.
        for (int row = rs; row < th; row += RS) {
            const float* gp = gs + (row * TW + 16 * h) * 32 + r;
            const float* xp = xs + (row * tile_w + 16 * h) * A;
            float b = gp[0], a[UW];
#pragma unroll
            for (int t = 0; t < UW; ++t) a[t] = xp[uoff[t]];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float bn = 0.f, an[UW];
                if (s < 15) {
                    bn = gp[(s + 1) * 32];
#pragma unroll
                    for (int t = 0; t < UW; ++t) an[t] = xp[uoff[t] + (s + 1) * A];
                }
#pragma unroll
                for (int t = 0; t < UW; ++t) acc[t] = mfma32(a[t], b, acc[t]);
                if (s < 15) {
                    b = bn;
#pragma unroll
                    for (int t = 0; t < UW; ++t) a[t] = an[t];
                }
            }
        }
    }
    // ---- one atomic pass per block: dwp[tap][a][32]
    if (RS == 1) {
        if (r < g.B) {
#pragma unroll
            for (int t = 0; t < UW; ++t) {
                const int u = ug + UG * t;
                if (u < units) {
                    const int tap = u / a_tiles, abase = (u - tap * a_tiles) * 32;
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        atomicAdd(&dwp[((size_t)tap * A + abase + acc_row(v, h)) * 32 + r], acc[t][v]);
                }
            }
        }
    } else {
        // the RS row lanes of a unit group fold their accumulators pairwise through LDS (the staging buffers are
        // free now): log2(RS) rounds of "upper half stores, lower half adds", 16-byte conflict-free accesses
        // ([slot][quad][lane] float4), then row lane 0 issues the atomics from registers.
        float4* fold = reinterpret_cast<float4*>(lds);
#pragma unroll
        for (int step = RS / 2; step >= 1; step >>= 1) {
            __syncthreads();
            if (rs >= step && rs < 2 * step) {
                const int slot = ug * step + (rs - step);
#pragma unroll
                for (int t = 0; t < UW; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        fold[((slot * UW + t) * 4 + q) * 64 + lane] =
                            make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
            }
            __syncthreads();
            if (rs < step) {
                const int slot = ug * step + rs;
#pragma unroll
                for (int t = 0; t < UW; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 p = fold[((slot * UW + t) * 4 + q) * 64 + lane];
                        acc[t][4 * q] += p.x; acc[t][4 * q + 1] += p.y; acc[t][4 * q + 2] += p.z; acc[t][4 * q + 3] += p.w;
                    }
            }
        }
        if (rs == 0 && r < g.B) {
#pragma unroll
            for (int t = 0; t < UW; ++t) {
                const int u = ug + UG * t;
                if (u < units) {
                    const int tap = u / a_tiles, abase = (u - tap * a_tiles) * 32;
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        atomicAdd(&dwp[((size_t)tap * A + abase + acc_row(v, h)) * 32 + r], acc[t][v]);
                }
            }
        }
    }
}

static size_t wgrad_lds_bytes(const WgradGeom& g, int th) {
    return ((size_t)(th + 2 * g.pad) * (TW + 2 * g.pad) * g.A + (size_t)th * TW * 32) * sizeof(float);
}

// (accumulators per wave, row split) for a unit count; 0 = not covered
static void wgrad_lds_shape(int units, int& uw, int& rs) {
    if (units <= 1) { uw = 1; rs = 8; }
    else if (units <= 2) { uw = 2; rs = 8; }
    else if (units <= 4) { uw = 4; rs = 8; }
    else if (units <= 10) { uw = 5; rs = 4; }
    else if (units <= 20) { uw = 5; rs = 2; }
    else if (units <= 28) { uw = 7; rs = 2; }
    else if (units <= 36) { uw = 9; rs = 2; }
    else { uw = 0; rs = 0; }
}

bool lds_wgrad_ok(const WgradGeom& g) {
    if (g.stride != 1 || g.B > 32 || g.A % 32 != 0 || g.A > 128) return false;
    if (g.kh != g.kw || g.pad != g.dil * (g.kh / 2) || g.hg != g.hi || g.wg != g.wi) return false;
    if (g.wg < TW || g.hg < 8) return false;
    const int units = g.kh * g.kw * (g.A / 32);
    int uw, rs;
    wgrad_lds_shape(units, uw, rs);
    if (uw == 0) return false;
    return wgrad_lds_bytes(g, 4) <= 150 * 1024 && (long)g.n * g.hi * g.wi * g.A < 0x7fffffffL;
}

template <int A, int UW, int RS, int PFX>
static int launch_one(const WgradGeom& g, const float* X, const float* G, float* ws, int x_relu, int th, hipStream_t st) {
    size_t bytes = wgrad_lds_bytes(g, th);
    const size_t fold = RS > 1 ? (size_t)4 * UW * 4096 : 0;      // epilogue: 4 storing waves x UW accumulators x 4 KiB
    if (fold > bytes) bytes = fold;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_lds_kernel<A, UW, RS, PFX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) { set_error("wgrad_lds: raising the dynamic LDS limit", e); return SENAS_ELAUNCH; }
        attr_set = true;
    }
    const int tiles_x = (g.wg + TW - 1) / TW, tiles_y = (g.hg + th - 1) / th;
    const int ntiles = g.n * tiles_x * tiles_y;
    const int blocks = ntiles < 256 ? ntiles : 256;            // one persistent block per CU
    hipLaunchKernelGGL((wgrad_lds_kernel<A, UW, RS, PFX>), dim3(blocks), dim3(512), bytes, st, g, X, G, ws, x_relu, th, tiles_x, tiles_y);
    return launch_status("wgrad_lds");
}

// the instantiations: (A, units) pairs that exist are A in {32, 64, 96, 128} x odd square kernels 1, 3, 5 with
// <= 36 units; PFX is what the register budget of the shape leaves (256 VGPRs at 2 waves per SIMD)
#define SENAS_WGRAD_LDS_SHAPES(X_)   \
    X_(32, 1, 8, 4)                  \
    X_(32, 5, 4, 6)                  \
    X_(32, 7, 2, 10)                 \
    X_(64, 2, 8, 8)                  \
    X_(64, 5, 2, 11)                 \
    X_(96, 4, 8, 12)                 \
    X_(96, 7, 2, 10)                 \
    X_(128, 4, 8, 16)                \
    X_(128, 9, 2, 4)

// ws: zeroed float[taps][A][32]; the caller unpacks it into the torch layout afterwards
int launch_lds_wgrad(const WgradGeom& g, const float* X, const float* G, float* ws, int x_relu, hipStream_t st) {
    // tallest tile that fits in LDS; on small maps shrink it until every CU has a tile (the kernel is
    // critical-path-bound there: a shorter tile is a shorter serial K loop per block)
    int th = wgrad_lds_bytes(g, 8) <= 150 * 1024 ? 8 : 4;
    const int tiles_x = (g.wg + TW - 1) / TW;
    while (th > 2 && (long)g.n * tiles_x * ((g.hg + th - 1) / th) < 256) th >>= 1;
    const int units = g.kh * g.kw * (g.A / 32);
    int uw, rs;
    wgrad_lds_shape(units, uw, rs);
#define SENAS_CASE(A_, UW_, RS_, PF_) \
    if (g.A == A_ && uw == UW_ && rs == RS_) return launch_one<A_, UW_, RS_, PF_>(g, X, G, ws, x_relu, th, st);
    SENAS_WGRAD_LDS_SHAPES(SENAS_CASE)
#undef SENAS_CASE
    set_error_msg("wgrad_lds: no kernel for this (channels, taps) pair");
    return SENAS_EINVAL;
}

// the kernel symbol launch_lds_wgrad picks (for senas_conv2d_kernel_name)
void lds_wgrad_name(const WgradGeom& g, char* buf, int len) {
    int uw, rs;
    wgrad_lds_shape(g.kh * g.kw * (g.A / 32), uw, rs);
    int pf = 0;
#define SENAS_CASE(A_, UW_, RS_, PF_) if (g.A == A_ && uw == UW_ && rs == RS_) pf = PF_;
    SENAS_WGRAD_LDS_SHAPES(SENAS_CASE)
#undef SENAS_CASE
    snprintf(buf, len, "wgrad_lds_kernel<%d, %d, %d, %d>", g.A, uw, rs, pf);
}

}  // namespace senas
