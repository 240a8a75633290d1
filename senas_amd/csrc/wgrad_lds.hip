// Weight gradient of a stride-1 "same" convolution with both operands staged in LDS, gfx950.
//
//   dW[tap][a][b] = sum over pixels p of  X[p + off(tap)][a] * G[p][b]        (a: in-channel, b: out-channel)
//
// GEMM view per tap: M = 32 in-channels, N = 32 out-channels, K = pixels; v_mfma_f32_32x32x2_f32 eats two
// pixels per instruction.  A block (8 waves) walks 8x32 (or 4x32 / 2x32; 16- and 8-wide on narrow maps) pixel tiles
// persistently:
//   - the X window (tile + halo, zero-filled borders, ReLU on load) and the G tile go to LDS once per tile,
//     so the k*k taps re-read X from LDS instead of L1 and the tap loop has no bounds checks at all; the NEXT tile
//     is requested into registers before the K loop of the current one, so staging hides behind the MFMAs;
//   - the work units (tap, 32-channel slice of a) are dealt to the 8 waves as Q full units + REM row-shared units
//     (perfect balance for any unit count); each wave keeps one 32x32 accumulator per unit in registers ACROSS
//     tiles and writes it out once, at the end, into this block's slice of a partial image -- no atomics; the
//     second launch adds the blocks' slices in a fixed order while transposing to the torch layout;
//   - the K loop is straight-line ds_read2 + MFMA code (no wave-uniform branches inside);
//   - K order inside a 32-pixel row is permuted (lane half h takes pixels 16h .. 16h+15) so that both
//     operands advance by one pixel per MFMA with compile-time LDS offsets.
// Measured (tools/conv_bench.py, sum launch included): 114-122 TFLOP/s = 73-78 % of the fp32-MFMA peak on
// 8x32->32x256x256 5x5 (was 92 with atomics, unbalanced units and branches in the K loop).
#include "common.h"

namespace senas {

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {


__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

}  // namespace

// A: in-channels (multiple of 32, <= 128).  The work units (tap, 32-channel slice of a) are dealt to the 8 waves as
//   Q   "full" units per wave (unit w + 8t, all rows of every tile), and
//   REM "shared" units (units 8Q .. 8Q+REM-1), whose tile rows are dealt round-robin: wave w takes row r of shared
//       unit j when (r + j) % 8 == w
// so every wave issues the same number of MFMAs whatever the unit count is (25 units = 3 full + 1 shared, a 1x1
// kernel = 1 shared: 8 row lanes).  Accumulators stay in registers ACROSS tiles; at the end the shared units' 8
// partials meet in LDS and every unit is written ONCE, without atomics, to this block's slice of the partial
// image part[block][unit][32][32] -- the unpack launch adds the blocks' slices in a fixed order (bitwise
// reproducible weight gradients) while it transposes to the torch layout.
// PFX: 16-byte pieces of the NEXT tile's X window each thread requests before the K loop of the current tile
// and parks in registers until the loop is done (the G tile always travels that way), so the HBM/L2 latency
// of staging hides behind the MFMAs; pieces beyond PFX*512 are fetched after the loop.
// dynamic LDS: X window [(th + 2*halo) * (32 + 2*halo)][A] floats, then G tile [th * 32][32] floats.
// TWL: tile width in pixels (32, 16, 8): the 32 K-pixels of an MFMA row are 32/TWL image rows of TWL columns, so
// 16x16 and 8x8 maps run here too; `th` counts MFMA rows (th * 32/TWL image rows per tile).
// S: stride of the convolution (1, or 2: X lives on the fine grid, G on the coarse one -- Conv2d stride 2 and
// ConvTranspose2d stride 2 alike); the tile is laid out over G, the X window covers S times as many rows and columns.
template <int A, int Q, int REM, int PFX, int TWL, int S>
__global__ __launch_bounds__(512) void wgrad_lds_kernel(WgradGeom g, const float* __restrict__ X,
                                                        const float* __restrict__ G1, float* __restrict__ part1,
                                                        int x_relu, int th, int tiles_x, int tiles_y, WPair2 second) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // blockIdx.y == 1: the second problem of a pair launch (same X and shapes; its own dy, partial images, dilation)
    const float* __restrict__ G = G1;
    float* __restrict__ part = part1;
    if (blockIdx.y != 0) { G = second.G; part = second.part; g.dil = second.dil; g.pad = second.pad; }
    constexpr int UW = Q + REM;
    constexpr int RPM = 32 / TWL;                // image rows per MFMA row
    constexpr int PP = A / 4;                    // 16-byte pieces per pixel of X
    constexpr int XL = 512 / PP;                 // pixels of the window covered by one slot
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform
    const int r = lane & 31, h = lane >> 5;
    const int halo = g.pad;
    const int tile_w = S * TWL + 2 * halo, tile_h = S * th * RPM + 2 * halo;
    float* xs = lds;
    float* gs = lds + tile_h * tile_w * A;
    float4* xs4 = reinterpret_cast<float4*>(xs);
    float4* gs4 = reinterpret_cast<float4*>(gs);
    constexpr int a_tiles = A / 32;

    // accumulator t < Q: full unit wave + 8t; t >= Q: shared unit 8Q + (t - Q)
    int uoff[UW];                 // LDS float offset of the unit's tap shift + channel slice
#pragma unroll
    for (int t = 0; t < UW; ++t) {
        const int u = t < Q ? wave + 8 * t : 8 * Q + (t - Q);
        const int tap = u / a_tiles, at = u - tap * a_tiles;
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
    const int dty = XL / tile_w, dtx = XL - dty * tile_w;        // slot-to-slot step in (row, column)
    const int wpix = tile_h * tile_w;            // pixels in the X window
    const int gpix = th * 32;                    // pixels in the G tile (stored linearly: TWL-wide rows back to back)
    const int gq = threadIdx.x & 7, gpl = threadIdx.x >> 3;      // G: 8 pieces per pixel, 64 pixels per slot
    const int per_img = tiles_x * tiles_y;
    const int ntiles = g.n * per_img;
    float4 px[PFX > 0 ? PFX : 1], pg[4];

    // request the first PFX slots of X and the whole G tile of `tile` (loads only; borders resolved in commit)
    auto issue = [&](int tile) {
        const int n = tile / per_img, tr = tile - n * per_img;
        const int oy0 = (tr / tiles_x) * th * RPM, ox0 = (tr % tiles_x) * TWL;
        const float* src = X + (size_t)n * g.hi * g.wi * A + xq * 4;
        int ty = ty0, tx = tx0;
#pragma unroll
        for (int k = 0; k < PFX; ++k) {
            const int iy = S * oy0 - halo + ty, ix = S * ox0 - halo + tx;
            const bool inb = xpl < XL && k * XL + xpl < wpix && iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi;
            px[k] = *reinterpret_cast<const float4*>(src + (inb ? ((size_t)iy * g.wi + ix) * A : 0));
            ty += dty; tx += dtx;
            if (tx >= tile_w) { tx -= tile_w; ++ty; }
        }
        if (g.B == 32) {
            const float* gsrc = G + (size_t)n * g.hg * g.wg * 32 + gq * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pix = k * 64 + gpl;
                const int gy = oy0 + pix / TWL, gx = ox0 + pix % TWL;
                const bool inb = pix < gpix && gy < g.hg && gx < g.wg;
                pg[k] = *reinterpret_cast<const float4*>(gsrc + (inb ? ((size_t)gy * g.wg + gx) * 32 : 0));
            }
        }
    };
    // registers -> LDS (ReLU, zero borders), then whatever did not fit in the prefetch slots
    auto commit = [&](int tile) {
        const int n = tile / per_img, tr = tile - n * per_img;
        const int oy0 = (tr / tiles_x) * th * RPM, ox0 = (tr % tiles_x) * TWL;
        int ty = ty0, tx = tx0;
#pragma unroll
        for (int k = 0; k < PFX; ++k) {
            const int iy = S * oy0 - halo + ty, ix = S * ox0 - halo + tx;
            const bool live = xpl < XL && k * XL + xpl < wpix;
            const bool inb = live && iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi;
            float4 v = px[k];
            if (x_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (!inb) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) xs4[k * XL * PP + xslot] = v;
            ty += dty; tx += dtx;
            if (tx >= tile_w) { tx -= tile_w; ++ty; }
        }
        const float* src = X + (size_t)n * g.hi * g.wi * A + xq * 4;
        for (int k0 = PFX; k0 * XL < wpix; k0 += 4) {           // remainder, 4 loads in flight
            float4 v[4];
            bool ok[4], lv[4];
            int tyy = ty, txx = tx;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int iy = S * oy0 - halo + tyy, ix = S * ox0 - halo + txx;
                lv[u] = xpl < XL && (k0 + u) * XL + xpl < wpix;
                ok[u] = lv[u] && iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi;
                v[u] = *reinterpret_cast<const float4*>(src + (ok[u] ? ((size_t)iy * g.wi + ix) * A : 0));
                tyy += dty; txx += dtx;
                if (txx >= tile_w) { txx -= tile_w; ++tyy; }
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
                const int gy = oy0 + pix / TWL, gx = ox0 + pix % TWL;
                float4 v = pg[k];
                if (!(gy < g.hg && gx < g.wg)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pix < gpix) gs4[k * 512 + threadIdx.x] = v;
            }
        } else {
            const float* gsrc = G + (size_t)n * g.hg * g.wg * g.B;
            for (int idx = threadIdx.x; idx < gpix * 32; idx += 512) {   // B < 32: zero-pad the columns
                const int pix = idx >> 5, b = idx & 31;
                const int gy = oy0 + pix / TWL, gx = ox0 + pix % TWL;
                float v = 0.f;
                if (b < g.B && gy < g.hg && gx < g.wg) v = gsrc[((size_t)gy * g.wg + gx) * g.B + b];
                gs[idx] = v;
            }
        }
    };

    // XCD-aware tile order: workgroups b and b + 8 share an XCD (round-robin placement), so within every sweep of
    // gridDim.x tiles an XCD takes a CONTIGUOUS eighth -- neighbouring tiles (overlapping X windows) meet in one L2.
    // Still a bijection per sweep, so every tile is visited exactly once.
    int tile = blockIdx.x;
    if ((gridDim.x & 7u) == 0) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                  // previous tile's readers are done
        commit(tile);
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);     // in flight during the K loop
        // ---- K loop: rows of the tile, 16 MFMA steps per row (lane half h covers pixels 16h .. 16h+15); straight-line
        // per (row, unit group): the operands of step s+1 are requested before the MFMAs of step s are issued
        const int rowstep = tile_w * A;
        auto koff = [&](int s) { return S * (TWL == 8 ? (s >> 3) * rowstep + (s & 7) * A : s * A); };     // s is a compile-time index
        for (int row = 0; row < th; ++row) {
            const float* gp = gs + (row * 32 + 16 * h) * 32 + r;
            // K-pixel k = 16h + s of this MFMA row sits at image row k / TWL, column k % TWL of the tile
            const float* xp = xs + S * (TWL == 32 ? (row * tile_w + 16 * h) : (TWL == 16 ? (row * 2 + h) * tile_w : (row * 4 + 2 * h) * tile_w)) * A;
            if (Q > 0) {
                float b = gp[0], a[Q > 0 ? Q : 1];
#pragma unroll
                for (int t = 0; t < Q; ++t) a[t] = xp[uoff[t]];
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    float bn = 0.f, an[Q > 0 ? Q : 1];
                    if (s < 15) {
                        bn = gp[(s + 1) * 32];
#pragma unroll
                        for (int t = 0; t < Q; ++t) an[t] = xp[uoff[t] + koff(s + 1)];
                    }
#pragma unroll
                    for (int t = 0; t < Q; ++t) acc[t] = mfma32(a[t], b, acc[t]);
                    if (s < 15) {
                        b = bn;
#pragma unroll
                        for (int t = 0; t < Q; ++t) a[t] = an[t];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < REM; ++j) {
                if (((row + j) & 7) == wave) {                    // wave-uniform: this row of shared unit j is mine
                    float b = gp[0], a = xp[uoff[Q + j]];
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        float bn = 0.f, an = 0.f;
                        if (s < 15) { bn = gp[(s + 1) * 32]; an = xp[uoff[Q + j] + koff(s + 1)]; }
                        acc[Q + j] = mfma32(a, b, acc[Q + j]);
                        b = bn; a = an;
                    }
                }
            }
        }
    }
    // ---- shared units: the 8 waves' partial sums fold pairwise through LDS (the staging buffers are free now),
    // 16-byte conflict-free accesses ([slot][unit][quad][lane] float4); wave 0 ends up with the totals
    if (REM > 0) {
        float4* fold = reinterpret_cast<float4*>(lds);
#pragma unroll
        for (int step = 4; step >= 1; step >>= 1) {
            __syncthreads();
            if (wave >= step && wave < 2 * step) {
                const int slot = wave - step;
#pragma unroll
                for (int j = 0; j < REM; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        fold[((slot * REM + j) * 4 + q) * 64 + lane] =
                            make_float4(acc[Q + j][4 * q], acc[Q + j][4 * q + 1], acc[Q + j][4 * q + 2], acc[Q + j][4 * q + 3]);
            }
            __syncthreads();
            if (wave < step) {
#pragma unroll
                for (int j = 0; j < REM; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 p = fold[((wave * REM + j) * 4 + q) * 64 + lane];
                        acc[Q + j][4 * q] += p.x; acc[Q + j][4 * q + 1] += p.y; acc[Q + j][4 * q + 2] += p.z; acc[Q + j][4 * q + 3] += p.w;
                    }
            }
        }
    }
    // ---- this block's slice of the partial image: part[block][unit][a (32)][b (32)], every element written once
    float* mine = part + (size_t)blockIdx.x * (8 * Q + REM) * 1024;
#pragma unroll
    for (int t = 0; t < UW; ++t) {
        if (t < Q || wave == 0) {
            const int u = t < Q ? wave + 8 * t : 8 * Q + (t - Q);
#pragma unroll
            for (int v = 0; v < 16; ++v) mine[(size_t)u * 1024 + acc_row(v, h) * 32 + r] = acc[t][v];
        }
    }
}

// part[block][unit = tap*a_tiles + at][32 a][32 b] summed over the blocks -> torch layout dw[b][a][tap].
// grid = units * 32 blocks (one row `a` of one unit each); thread = (b, k): the 8 k-groups each add every 8th block's
// slice in index order, then meet in LDS in a fixed order -- bitwise reproducible, 128-byte coalesced reads.
__global__ __launch_bounds__(256) void wgrad_lds_sum_kernel(const float* __restrict__ part, float* __restrict__ dw, int A, int B,
                                                            int taps, int nblk) {
    __shared__ float red[8][32];
    const int b = threadIdx.x & 31, k = threadIdx.x >> 5;
    const int unit = blockIdx.x >> 5, arow = blockIdx.x & 31;
    const int a_tiles = A / 32;
    const size_t per_blk = (size_t)taps * a_tiles * 1024;
    const float* p = part + ((size_t)unit * 32 + arow) * 32 + b;
    float s = 0.f;
    int j = k;
    for (; j + 24 < nblk; j += 32) {                 // 4 independent loads in flight
        const float v0 = p[(size_t)j * per_blk], v1 = p[(size_t)(j + 8) * per_blk], v2 = p[(size_t)(j + 16) * per_blk],
                    v3 = p[(size_t)(j + 24) * per_blk];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; j < nblk; j += 8) s += p[(size_t)j * per_blk];
    red[k][b] = s;
    __syncthreads();
    if (k == 0 && b < B) {
        float tot = red[0][b];
#pragma unroll
        for (int q = 1; q < 8; ++q) tot += red[q][b];
        const int tap = unit / a_tiles, a = (unit - tap * a_tiles) * 32 + arow;
        dw[((size_t)b * A + a) * taps + tap] = tot;
    }
}

// stride 2: the X window is twice as large per G pixel, so the tile is at most 16 wide
static int wgrad_tile_width(const WgradGeom& g) {
    if (g.stride == 2) return g.wg >= 16 ? 16 : 8;
    return g.wg >= 32 ? 32 : (g.wg >= 16 ? 16 : 8);
}

// th MFMA rows = th * 32/twl rows of G; the X window covers stride times as many rows and columns, plus the halo
static size_t wgrad_lds_bytes(const WgradGeom& g, int th, int twl) {
    return ((size_t)(g.stride * th * (32 / twl) + 2 * g.pad) * (g.stride * twl + 2 * g.pad) * g.A + (size_t)th * 32 * 32) * sizeof(float);
}

// the (A, units) pairs that exist: A in {32, 64, 96, 128} x odd square kernels 1, 3, 5 with <= 36 units.
// X_(A, Q, REM, PFX): units = 8*Q + REM; PFX is what the register budget leaves (256 VGPRs at 2 waves per SIMD)
#define SENAS_WGRAD_LDS_SHAPES(X_)   \
    X_(32, 0, 1, 4)                  \
    X_(32, 1, 1, 6)                  \
    X_(32, 3, 1, 14)                 \
    X_(64, 0, 2, 8)                  \
    X_(64, 2, 2, 11)                 \
    X_(96, 0, 3, 12)                 \
    X_(96, 3, 3, 12)                 \
    X_(128, 0, 4, 16)                \
    X_(128, 4, 4, 6)
// narrow maps (16 / 8 pixels wide) exist for the 32- and 128-channel shapes only
static bool wgrad_lds_narrow_ok(int A) { return A == 32 || A == 128; }

static bool wgrad_lds_has_shape(int A, int units) {
#define SENAS_CASE(A_, Q_, REM_, PF_) if (A == A_ && units == 8 * Q_ + REM_) return true;
    SENAS_WGRAD_LDS_SHAPES(SENAS_CASE)
#undef SENAS_CASE
    return false;
}

bool lds_wgrad_ok(const WgradGeom& g) {
    if ((g.stride != 1 && g.stride != 2) || g.B > 32 || g.A % 32 != 0 || g.A > 128) return false;
    if (g.kh != g.kw || g.pad != g.dil * (g.kh / 2) || g.hi != g.stride * g.hg || g.wi != g.stride * g.wg) return false;
    if (g.stride == 2 && g.A != 32) return false;             // strided forms are instantiated for 32 channels only
    if (g.wg < 8 || g.hg < 4 || (wgrad_tile_width(g) < 32 && !wgrad_lds_narrow_ok(g.A))) return false;
    if (wgrad_tile_width(g) == 32 && g.hg < 8) return false;
    if (!wgrad_lds_has_shape(g.A, g.kh * g.kw * (g.A / 32))) return false;
    return wgrad_lds_bytes(g, wgrad_tile_width(g) == 32 ? 4 : (g.stride == 2 ? 1 : 2), wgrad_tile_width(g)) <= 150 * 1024 &&
           (long)g.n * g.hi * g.wi * g.A < 0x7fffffffL;
}

static int wgrad_lds_tile_rows(const WgradGeom& g) {
    // tallest tile that fits in LDS; on small maps shrink it until every CU has a tile (the kernel is
    // critical-path-bound there: a shorter tile is a shorter serial K loop per block)
    const int twl = wgrad_tile_width(g), rpm = 32 / twl;
    int th = 8;
    while (th > 1 && wgrad_lds_bytes(g, th, twl) > 150 * 1024) th >>= 1;
    const int tiles_x = (g.wg + twl - 1) / twl;
    const int th_min = twl == 32 ? 2 : 1;
    while (th > th_min && (long)g.n * tiles_x * ((g.hg + th * rpm - 1) / (th * rpm)) < 256) th >>= 1;
    return th;
}

static int wgrad_lds_blocks(const WgradGeom& g) {
    const int th = wgrad_lds_tile_rows(g), twl = wgrad_tile_width(g), rows = th * (32 / twl);
    const long ntiles = (long)g.n * ((g.wg + twl - 1) / twl) * ((g.hg + rows - 1) / rows);
    return (int)(ntiles < 256 ? ntiles : 256);                 // one persistent block per CU
}

// bytes of the partial image the launch writes: blocks x units x 32 x 32 floats
int64_t lds_wgrad_ws_bytes(const WgradGeom& g) {
    return (int64_t)wgrad_lds_blocks(g) * g.kh * g.kw * (g.A / 32) * 1024 * sizeof(float);
}

template <int A, int Q, int REM, int PFX, int TWL, int S>
static int launch_one(const WgradGeom& g, const float* X, const float* G, float* part, int x_relu, hipStream_t st,
                      const WPair2& second = WPair2{}) {
    const int th = wgrad_lds_tile_rows(g);
    size_t bytes = wgrad_lds_bytes(g, th, TWL);
    if (second.on) {                                           // (lds_wgrad_pair_ok: both problems take this th)
        WgradGeom g2 = g;
        g2.dil = second.dil; g2.pad = second.pad;
        const size_t b2 = wgrad_lds_bytes(g2, th, TWL);
        if (b2 > bytes) bytes = b2;
    }
    const size_t fold = (size_t)4 * REM * 4096;                // epilogue: 4 storing waves x REM accumulators x 4 KiB
    if (fold > bytes) bytes = fold;
    if (int rc = raise_lds_limit(reinterpret_cast<const void*>(&wgrad_lds_kernel<A, Q, REM, PFX, TWL, S>), 150 * 1024,
                                 "wgrad_lds: raising the dynamic LDS limit")) return rc;
    const int rows = th * (32 / TWL);
    const int tiles_x = (g.wg + TWL - 1) / TWL, tiles_y = (g.hg + rows - 1) / rows;
    hipLaunchKernelGGL((wgrad_lds_kernel<A, Q, REM, PFX, TWL, S>), dim3(wgrad_lds_blocks(g), second.on ? 2 : 1), dim3(512), bytes, st, g, X, G,
                       part, x_relu, th, tiles_x, tiles_y, second);
    return launch_status("wgrad_lds");
}

static int launch_lds_wgrad_any(const WgradGeom& g, const float* X, const float* G, float* part, float* dw, int x_relu, senas_sum_item* defer,
                                hipStream_t st, const WPair2& second, float* dw2, senas_sum_item* defer2) {
    const int units = g.kh * g.kw * (g.A / 32);
    const int twl = wgrad_tile_width(g);
    int rc = SENAS_EINVAL;
    bool found = false;
#define SENAS_CASE(A_, Q_, REM_, PF_) \
    if (!found && g.A == A_ && units == 8 * Q_ + REM_) {                                                          \
        found = true;                                                                                             \
        if (g.stride == 2) {                                                                                      \
            if constexpr (A_ == 32) {                                                                             \
                rc = twl == 16 ? launch_one<A_, Q_, REM_, PF_, 16, 2>(g, X, G, part, x_relu, st, second)                  \
                               : launch_one<A_, Q_, REM_, PF_, 8, 2>(g, X, G, part, x_relu, st, second);                  \
            }                                                                                                     \
        } else if (twl == 32) rc = launch_one<A_, Q_, REM_, PF_, 32, 1>(g, X, G, part, x_relu, st, second);              \
        else if constexpr (A_ == 32 || A_ == 128) {                                                               \
            rc = twl == 16 ? launch_one<A_, Q_, REM_, PF_, 16, 1>(g, X, G, part, x_relu, st, second)                      \
                           : launch_one<A_, Q_, REM_, PF_, 8, 1>(g, X, G, part, x_relu, st, second);                      \
        }                                                                                                         \
    }
    SENAS_WGRAD_LDS_SHAPES(SENAS_CASE)
#undef SENAS_CASE
    if (!found) { set_error_msg("wgrad_lds: no kernel for this (channels, taps) pair"); return SENAS_EINVAL; }
    if (rc != SENAS_OK) return rc;
    const int nblk = wgrad_lds_blocks(g);
    if (defer != nullptr) {              // the caller folds the partial images later, together with other convolutions' (senas_wgrad_sum_batched)
        *defer = senas_sum_item{part, dw, 2, g.A, g.B, g.kh * g.kw, 0, nblk};
        if (second.on) *defer2 = senas_sum_item{second.part, dw2, 2, g.A, g.B, g.kh * g.kw, 0, nblk};
        return SENAS_OK;
    }
    hipLaunchKernelGGL(wgrad_lds_sum_kernel, dim3(units * 32), dim3(256), 0, st, part, dw, g.A, g.B, g.kh * g.kw, nblk);
    if (second.on) hipLaunchKernelGGL(wgrad_lds_sum_kernel, dim3(units * 32), dim3(256), 0, st, second.part, dw2, g.A, g.B, g.kh * g.kw, nblk);
    return launch_status("wgrad_lds sum");
}

// part: lds_wgrad_ws_bytes(g) of scratch (need not be zeroed); dw: torch layout, overwritten
int launch_lds_wgrad(const WgradGeom& g, const float* X, const float* G, float* part, float* dw, int x_relu, senas_sum_item* defer,
                     hipStream_t st) {
    return launch_lds_wgrad_any(g, X, G, part, dw, x_relu, defer, st, WPair2{}, nullptr, nullptr);
}

// Two problems that differ in dilation / padding only: the tile list (a function of the map and of what fits in LDS) must be
// the same for both, so that one grid serves both and each partial image has the block count its workspace was sized for.
bool lds_wgrad_pair_ok(const WgradGeom& g, const WgradGeom& g2) {
    return lds_wgrad_ok(g) && lds_wgrad_ok(g2) && g.stride == 1 && g2.stride == 1 && wgrad_lds_tile_rows(g) == wgrad_lds_tile_rows(g2) &&
           wgrad_lds_blocks(g) == wgrad_lds_blocks(g2);
}

int launch_lds_wgrad_pair(const WgradGeom& g, const WgradGeom& g2, const float* X, const float* G, const float* G2, float* part, float* part2,
                          float* dw, float* dw2, int x_relu, senas_sum_item* defer, senas_sum_item* defer2, hipStream_t st) {
    if ((defer == nullptr) != (defer2 == nullptr)) { set_error_msg("wgrad_lds pair: both sums are deferred or neither"); return SENAS_EINVAL; }
    return launch_lds_wgrad_any(g, X, G, part, dw, x_relu, defer, st, WPair2{G2, part2, g2.dil, g2.pad, 1}, dw2, defer2);
}

// the kernel symbol launch_lds_wgrad picks (for senas_conv2d_kernel_name)
void lds_wgrad_name(const WgradGeom& g, char* buf, int len) {
    const int units = g.kh * g.kw * (g.A / 32);
    int q = 0, rem = 0, pf = 0;
#define SENAS_CASE(A_, Q_, REM_, PF_) if (g.A == A_ && units == 8 * Q_ + REM_) { q = Q_; rem = REM_; pf = PF_; }
    SENAS_WGRAD_LDS_SHAPES(SENAS_CASE)
#undef SENAS_CASE
    snprintf(buf, len, "wgrad_lds_kernel<%d, %d, %d, %d, %d, %d>", g.A, q, rem, pf, wgrad_tile_width(g), g.stride);
}

}  // namespace senas
