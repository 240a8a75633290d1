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

// A: in-channels (multiple of 32, <= 128); UW: accumulators per wave; grid = persistent blocks of 512 threads.
// dynamic LDS: X window [(th + 2*halo) * (32 + 2*halo)][A] floats, then G tile [th * 32][32] floats.
template <int A, int UW>
__global__ __launch_bounds__(512) void wgrad_lds_kernel(WgradGeom g, const float* __restrict__ X,
                                                        const float* __restrict__ G, float* __restrict__ dwp,
                                                        int x_relu, int th, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: unit tests below are scalar branches
    const int r = lane & 31, h = lane >> 5;
    const int halo = g.pad;
    const int tile_w = TW + 2 * halo, tile_h = th + 2 * halo;
    float* xs = lds;
    float* gs = lds + tile_h * tile_w * A;
    constexpr int a_tiles = A / 32;
    const int taps = g.kh * g.kw;
    const int units = taps * a_tiles;

    // this wave's units: u = wave + 8*t
    int uoff[UW];                 // LDS float offset of the unit's tap shift + channel slice
    bool uok[UW];
#pragma unroll
    for (int t = 0; t < UW; ++t) {
        const int u = wave + 8 * t;
        uok[t] = u < units;
        const int uc = uok[t] ? u : 0;
        const int tap = uc / a_tiles, at = uc - tap * a_tiles;
        const int ky = tap / g.kw, kx = tap - ky * g.kw;
        uoff[t] = ((ky * g.dil) * tile_w + kx * g.dil) * A + at * 32 + r;
    }
    f32x16 acc[UW];
#pragma unroll
    for (int t = 0; t < UW; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

    const int per_img = tiles_x * tiles_y;
    const int ntiles = g.n * per_img;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / per_img, tr = tile - n * per_img;
        const int oy0 = (tr / tiles_x) * th, ox0 = (tr % tiles_x) * TW;
        __syncthreads();                                  // previous tile's readers are done
        // ---- stage X window (A/4 16-byte pieces per pixel) and G tile (8 pieces per pixel, B padded to 32)
        {
            const float* src = X + (size_t)n * g.hi * g.wi * A;
            constexpr int PP = A / 4;
            const int pieces = tile_h * tile_w * PP;
            float4* xs4 = reinterpret_cast<float4*>(xs);
            for (int base = 0; base < pieces; base += 512 * 4) {
                float4 v[4];
                int dst[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = base + u * 512 + threadIdx.x;
                    const int pix = idx / PP, q = idx - pix * PP;
                    const int ty = pix / tile_w, tx = pix - ty * tile_w;
                    const int iy = oy0 - halo + ty, ix = ox0 - halo + tx;
                    const bool inb = idx < pieces && iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi;
                    dst[u] = idx < pieces ? idx : -1;
                    v[u] = *reinterpret_cast<const float4*>(src + (inb ? ((size_t)iy * g.wi + ix) * A + q * 4 : 0));
                    if (!inb) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (x_relu) { v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f); v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f); }
                    if (dst[u] >= 0) xs4[dst[u]] = v[u];
                }
            }
            const float* gsrc = G + (size_t)n * g.hg * g.wg * g.B;
            if (g.B == 32) {                               // full rows: 8 16-byte pieces per pixel, 4 in flight
                float4* gs4 = reinterpret_cast<float4*>(gs);
                const int gpieces = th * TW * 8;
                for (int base = 0; base < gpieces; base += 512 * 4) {
                    float4 v[4];
                    int dst[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = base + u * 512 + threadIdx.x;
                        const int pix = idx >> 3, q = idx & 7;
                        const int py = pix / TW, px = pix - py * TW;
                        const int gy = oy0 + py, gx = ox0 + px;
                        const bool inb = idx < gpieces && gy < g.hg && gx < g.wg;
                        dst[u] = idx < gpieces ? idx : -1;
                        v[u] = *reinterpret_cast<const float4*>(gsrc + (inb ? ((size_t)gy * g.wg + gx) * 32 + q * 4 : 0));
                        if (!inb) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (dst[u] >= 0) gs4[dst[u]] = v[u];
                }
            } else {
                for (int idx = threadIdx.x; idx < th * TW * 32; idx += 512) {   // B < 32: zero-pad the columns
                    const int pix = idx >> 5, b = idx & 31;
                    const int py = pix / TW, px = pix - py * TW;
                    const int gy = oy0 + py, gx = ox0 + px;
                    float v = 0.f;
                    if (b < g.B && gy < g.hg && gx < g.wg) v = gsrc[((size_t)gy * g.wg + gx) * g.B + b];
                    gs[idx] = v;
                }
            }
        }
        __syncthreads();
        // ---- K loop: rows of the tile, 16 MFMA steps per row (lane half h covers pixels 16h .. 16h+15)
        for (int row = 0; row < th; ++row) {
            const float* gp = gs + (row * TW + 16 * h) * 32 + r;
            const float* xp = xs + (row * tile_w + 16 * h) * A;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float b = gp[s * 32];
#pragma unroll
                for (int t = 0; t < UW; ++t) {
                    const float a = xp[uoff[t] + s * A];
                    if (uok[t]) acc[t] = mfma32(a, b, acc[t]);           // wave-uniform
                }
            }
        }
    }
    // ---- one atomic pass per block: dwp[tap][a][32]
    if (r < g.B) {
#pragma unroll
        for (int t = 0; t < UW; ++t) {
            const int u = wave + 8 * t;
            if (u < units) {
                const int tap = u / a_tiles, abase = (u - tap * a_tiles) * 32;
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    atomicAdd(&dwp[((size_t)tap * A + abase + acc_row(v, h)) * 32 + r], acc[t][v]);
            }
        }
    }
}

static size_t wgrad_lds_bytes(const WgradGeom& g, int th) {
    return ((size_t)(th + 2 * g.pad) * (TW + 2 * g.pad) * g.A + (size_t)th * TW * 32) * sizeof(float);
}

bool lds_wgrad_ok(const WgradGeom& g) {
    if (g.stride != 1 || g.B > 32 || g.A % 32 != 0 || g.A > 128) return false;
    if (g.kh != g.kw || g.pad != g.dil * (g.kh / 2) || g.hg != g.hi || g.wg != g.wi) return false;
    if (g.wg < TW || g.hg < 8) return false;
    const int units = g.kh * g.kw * (g.A / 32);
    if (units > 40) return false;
    return wgrad_lds_bytes(g, 4) <= 150 * 1024 && (long)g.n * g.hi * g.wi * g.A < 0x7fffffffL;
}

template <int A, int UW>
static int launch_one(const WgradGeom& g, const float* X, const float* G, float* ws, int x_relu, int th, hipStream_t st) {
    const size_t bytes = wgrad_lds_bytes(g, th);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_lds_kernel<A, UW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) { set_error("wgrad_lds: raising the dynamic LDS limit", e); return SENAS_ELAUNCH; }
        attr_set = true;
    }
    const int tiles_x = (g.wg + TW - 1) / TW, tiles_y = (g.hg + th - 1) / th;
    const int ntiles = g.n * tiles_x * tiles_y;
    const int blocks = ntiles < 256 ? ntiles : 256;            // one persistent block per CU
    hipLaunchKernelGGL((wgrad_lds_kernel<A, UW>), dim3(blocks), dim3(512), bytes, st, g, X, G, ws, x_relu, th, tiles_x, tiles_y);
    return launch_status("wgrad_lds");
}

// ws: zeroed float[taps][A][32]; the caller unpacks it into the torch layout afterwards
int launch_lds_wgrad(const WgradGeom& g, const float* X, const float* G, float* ws, int x_relu, hipStream_t st) {
    // tallest tile that fits in LDS; on small maps shrink it until every CU has a tile (the kernel is
    // critical-path-bound there: a shorter tile is a shorter serial K loop per block)
    int th = wgrad_lds_bytes(g, 8) <= 150 * 1024 ? 8 : 4;
    const int tiles_x = (g.wg + TW - 1) / TW;
    while (th > 2 && (long)g.n * tiles_x * ((g.hg + th - 1) / th) < 256) th >>= 1;
    const int units = g.kh * g.kw * (g.A / 32);
    const int uw = (units + 7) / 8;
#define SENAS_WG(AA)                                                                      \
    do {                                                                                  \
        if (uw <= 2) return launch_one<AA, 2>(g, X, G, ws, x_relu, th, st);               \
        if (uw <= 4) return launch_one<AA, 4>(g, X, G, ws, x_relu, th, st);               \
        return launch_one<AA, 5>(g, X, G, ws, x_relu, th, st);                            \
    } while (0)
    switch (g.A) {
        case 32: SENAS_WG(32);
        case 64: SENAS_WG(64);
        case 96: SENAS_WG(96);
        default: SENAS_WG(128);
    }
#undef SENAS_WG
}

}  // namespace senas
