// Weight gradient of a stride-1 "same" convolution on the bf16 matrix pipe (fp32 accumulation): the split-operand twin
// of wgrad_lds.hip, same work decomposition --
//
//   dW[tap][a][b] = sum over pixels p of  X[p + off(tap)][a] * G[p][b]        (a: in-channel, b: out-channel)
//
// GEMM view per (tap, 32-channel slice of a) "unit": M = 32 in-channels, N = 32 out-channels, K = pixels, 16 of them per
// v_mfma_f32_32x32x16_bf16.  Both operands have the reduction index (the pixel) on the SLOW axis of their NHWC tiles, so
// the fragments are fetched with gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads 4 pixels x 16
// channels and every lane receives 4 pixels of ITS channel): no transposed copy of either operand exists anywhere.
// LDS images: per bf16 plane a dense [pixel][32 channels] array (64 B per pixel), so the 4 pixels x 2 channel blocks a
// 32-lane half reads in one instruction are 256 contiguous bytes -- all 64 banks once, whatever the tap shift.
// Operands are split into NS bf16 planes as in conv_bf.hip (NS = 1 bf16, 2 bf16x3, 3 bf16x6) while they are staged.
// A block (8 waves) walks th x 32 pixel tiles persistently; the units are dealt to the waves as Q full + REM row-shared
// ones (wgrad_lds.hip); the NEXT tile is requested into registers before the K loop of the current one (the K loop itself
// issues no global loads, so the in-order vmcnt queue never delays it); per-block partial images, summed by the batched
// second stage in block order (bitwise reproducible).
#include "common.h"

namespace senas {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ f32x16 mfma_bf(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

template <int NS>
__device__ __forceinline__ void split4(const float4& x, uint2 (&pl)[NS]) {
    float r[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int p = 0; p < NS; ++p) {
        const unsigned u0 = pk_bf16(r[0], r[1]), u1 = pk_bf16(r[2], r[3]);
        pl[p] = make_uint2(u0, u1);
        if (p + 1 < NS) { r[0] -= bf_lo(u0); r[1] -= bf_hi(u0); r[2] -= bf_lo(u1); r[3] -= bf_hi(u1); }
    }
}

// 8 K-elements (pixels) of this lane's channel: two transposing reads, 4 pixels each, 256 bytes apart
__device__ __forceinline__ uint4 tr_frag(const unsigned char* p) {
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 256));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

template <int NS> struct WProd;
template <> struct WProd<1> { static constexpr int N = 1; static constexpr int a[1] = {0}; static constexpr int b[1] = {0}; };
template <> struct WProd<2> { static constexpr int N = 3; static constexpr int a[3] = {1, 0, 0}; static constexpr int b[3] = {0, 1, 0}; };
template <> struct WProd<3> { static constexpr int N = 6; static constexpr int a[6] = {1, 2, 0, 1, 0, 0}; static constexpr int b[6] = {1, 0, 2, 0, 1, 0}; };

}  // namespace

// dynamic LDS: X planes [a tile][plane][window pixel][32 ch] bf16, then G planes [plane][tile pixel][32 ch] bf16
// GB: the gradient operand G is a bf16 tensor in HBM (the stored gradient of a bf16-stored convolution output): staged by a copy
template <int A, int Q, int REM, int PFX, int NS, bool GB = false>
__global__ __launch_bounds__(512) void wgrad_bf_kernel(WgradGeom g, const float* __restrict__ X, const float* __restrict__ G,
                                                       float* __restrict__ part, int x_relu, int th, int tiles_x, int tiles_y) {
    static_assert(!GB || NS == 1, "a bf16-stored gradient goes with the plain bf16 products");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int UW = Q + REM;
    constexpr int PP = A / 4;                    // 16-byte fp32 pieces per pixel of X
    constexpr int XL = 512 / PP;                 // window pixels per staging sweep
    constexpr int a_tiles = A / 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int halo = g.pad;
    const int tile_w = 32 + 2 * halo, tile_h = th + 2 * halo;
    const int wpix = tile_h * tile_w;
    const int gpix = th * 32;
    const int xplane = wpix * 64;                // bytes of one X plane
    const int gplane = gpix * 64;
    unsigned char* xs = lds;
    unsigned char* gs = lds + (size_t)a_tiles * NS * xplane;

    // this lane's place in a transposing read: group gi = lane / 16 reads channel block cb = gi & 1, pixels 8 * (gi >> 1) ..;
    // lane 4q + p of the group supplies the address of pixel q, channels 4p .. 4p + 3 of the block
    const int li = lane & 15, tq = li >> 2, tp = li & 3, cb = (lane >> 4) & 1;
    const int lane_off = (8 * h + tq) * 64 + cb * 32 + tp * 8;

    int uoff[UW];                                // byte offset of the unit's tap shift + channel-slice planes
#pragma unroll
    for (int t = 0; t < UW; ++t) {
        const int u = t < Q ? wave + 8 * t : 8 * Q + (t - Q);
        const int tap = u / a_tiles, at = u - tap * a_tiles;
        const int ky = tap / g.kw, kx = tap - ky * g.kw;
        uoff[t] = ((ky * g.dil) * tile_w + kx * g.dil) * 64 + at * NS * xplane;
    }
    f32x16 acc[UW];
#pragma unroll
    for (int t = 0; t < UW; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

    const int xq = threadIdx.x % PP, xpl = threadIdx.x / PP;
    const int ty0 = xpl / tile_w, tx0 = xpl - ty0 * tile_w;
    const int dty = XL / tile_w, dtx = XL - dty * tile_w;
    const int x_at = (4 * xq) / 32, x_ch = (4 * xq) % 32;          // this thread's channel slice / first channel in it
    const int gq = threadIdx.x & 7, gpl = threadIdx.x >> 3;
    const int per_img = tiles_x * tiles_y;
    const int ntiles = g.n * per_img;
    float4 px[PFX > 0 ? PFX : 1], pg[4];

    auto issue = [&](int tile) {
        const int n = tile / per_img, tr = tile - n * per_img;
        const int oy0 = (tr / tiles_x) * th, ox0 = (tr % tiles_x) * 32;
        const float* src = X + (size_t)n * g.hi * g.wi * A + xq * 4;
        int ty = ty0, tx = tx0;
#pragma unroll
        for (int k = 0; k < PFX; ++k) {
            const int iy = oy0 - halo + ty, ix = ox0 - halo + tx;
            const bool inb = xpl < XL && k * XL + xpl < wpix && iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi;
            px[k] = *reinterpret_cast<const float4*>(src + (inb ? ((size_t)iy * g.wi + ix) * A : 0));     // (zeroed in commit)
            ty += dty; tx += dtx;
            if (tx >= tile_w) { tx -= tile_w; ++ty; }
        }
        const size_t gbase = (size_t)n * g.hg * g.wg * 32 + gq * 4;
        const float* gsrc = G + gbase;
        const unsigned short* gsrch = reinterpret_cast<const unsigned short*>(G) + gbase;       // (GB: the same element offsets, 2-byte elements)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pix = k * 64 + gpl;
            const int gy = oy0 + (pix >> 5), gx = ox0 + (pix & 31);
            const bool inb = pix < gpix && gy < g.hg && gx < g.wg;
            const size_t o = inb ? ((size_t)gy * g.wg + gx) * 32 : 0;
            if constexpr (GB) {
                const uint2 raw = *reinterpret_cast<const uint2*>(gsrch + o);
                pg[k] = make_float4(__builtin_bit_cast(float, raw.x), __builtin_bit_cast(float, raw.y), 0.f, 0.f);
            } else {
                pg[k] = *reinterpret_cast<const float4*>(gsrc + o);
            }
        }
    };
    auto put_x = [&](int slot_pix, float4 v) {
        if (x_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        uint2 pl[NS];
        split4<NS>(v, pl);
        unsigned char* dst = xs + (size_t)x_at * NS * xplane + (size_t)slot_pix * 64 + x_ch * 2;
#pragma unroll
        for (int p = 0; p < NS; ++p) *reinterpret_cast<uint2*>(dst + (size_t)p * xplane) = pl[p];
    };
    auto commit = [&](int tile) {
        const int n = tile / per_img, tr = tile - n * per_img;
        const int oy0 = (tr / tiles_x) * th, ox0 = (tr % tiles_x) * 32;
        int ty = ty0, tx = tx0;
#pragma unroll
        for (int k = 0; k < PFX; ++k) {
            const int iy = oy0 - halo + ty, ix = ox0 - halo + tx;
            const bool live = xpl < XL && k * XL + xpl < wpix;
            float4 v = px[k];
            if (!(live && iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) put_x(k * XL + xpl, v);
            ty += dty; tx += dtx;
            if (tx >= tile_w) { tx -= tile_w; ++ty; }
        }
        const float* src = X + (size_t)n * g.hi * g.wi * A + xq * 4;
        for (int k0 = PFX; k0 * XL < wpix; k0 += 4) {               // what did not fit in the prefetch registers
            float4 v[4];
            bool lv[4], ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int iy = oy0 - halo + ty, ix = ox0 - halo + tx;
                lv[u] = xpl < XL && (k0 + u) * XL + xpl < wpix;
                ok[u] = lv[u] && iy >= 0 && iy < g.hi && ix >= 0 && ix < g.wi;
                v[u] = *reinterpret_cast<const float4*>(src + (ok[u] ? ((size_t)iy * g.wi + ix) * A : 0));
                ty += dty; tx += dtx;
                if (tx >= tile_w) { tx -= tile_w; ++ty; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!ok[u]) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lv[u]) put_x((k0 + u) * XL + xpl, v[u]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pix = k * 64 + gpl;
            const int gy = oy0 + (pix >> 5), gx = ox0 + (pix & 31);
            if (pix < gpix) {
                uint2 pl[NS];
                float4 v = pg[k];
                if (!(gy < g.hg && gx < g.wg)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (GB) pl[0] = make_uint2(__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y));
                else split4<NS>(v, pl);
#pragma unroll
                for (int p = 0; p < NS; ++p) *reinterpret_cast<uint2*>(gs + (size_t)p * gplane + (size_t)pix * 64 + gq * 8) = pl[p];
            }
        }
    };

    int tile = blockIdx.x;
    if ((gridDim.x & 7u) == 0) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (tile < ntiles) issue(tile);
    const int rowstep = tile_w * 64;
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                  // previous tile's readers are done
        commit(tile);
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);     // in flight during the K loop
        // ---- K loop: (row, 16-pixel chunk) steps; the fragments of step s + 1 are requested before the MFMAs of step s
        const unsigned char* xl = xs + lane_off;
        const unsigned char* gl = gs + lane_off;
        auto load_b = [&](int step, uint4 (&bf)[NS]) {
            const unsigned char* p = gl + (step >> 1) * (32 * 64) + (step & 1) * 1024;
#pragma unroll
            for (int q = 0; q < NS; ++q) bf[q] = tr_frag(p + (size_t)q * gplane);
        };
        auto load_a = [&](int step, int t, uint4 (&af)[NS]) {
            const unsigned char* p = xl + uoff[t] + (step >> 1) * rowstep + (step & 1) * 1024;
#pragma unroll
            for (int q = 0; q < NS; ++q) af[q] = tr_frag(p + (size_t)q * xplane);
        };
        const int nsteps = 2 * th;
        if (Q > 0) {
            // items (step, unit) in sequence; the A fragments of the next item and the B fragments of the next step are in
            // flight while the current item's MFMAs issue (two-slot rings, indexed statically: 2 * Q items per trip)
            uint4 bf[2][NS], af[2][NS];
            load_b(0, bf[0]);
            load_a(0, 0, af[0]);
            for (int s = 0; s < nsteps; s += 2) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (s + e + 1 < nsteps) load_b(s + e + 1, bf[e ^ 1]);
#pragma unroll
                    for (int t = 0; t < Q; ++t) {
                        const int cur = (e * Q + t) & 1;
                        if (t + 1 < Q) load_a(s + e, t + 1, af[cur ^ 1]);
                        else if (s + e + 1 < nsteps) load_a(s + e + 1, 0, af[cur ^ 1]);
                        __builtin_amdgcn_sched_barrier(0);                  // the requests stay in front of these MFMAs
#pragma unroll
                        for (int i = 0; i < WProd<NS>::N; ++i) acc[t] = mfma_bf(af[cur][WProd<NS>::a[i]], bf[e][WProd<NS>::b[i]], acc[t]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < REM; ++j) {
            for (int row = 0; row < th; ++row) {
                if (((row + j) & 7) != wave) continue;                // wave-uniform: this row of shared unit j is mine
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    uint4 bf[NS], af[NS];
                    load_b(2 * row + c, bf);
                    load_a(2 * row + c, Q + j, af);
#pragma unroll
                    for (int i = 0; i < WProd<NS>::N; ++i) acc[Q + j] = mfma_bf(af[WProd<NS>::a[i]], bf[WProd<NS>::b[i]], acc[Q + j]);
                }
            }
        }
    }
    // ---- shared units: the 8 waves' partial sums fold pairwise through LDS; wave 0 ends up with the totals
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
    // ---- this block's slice of the partial image: part[block][unit][a (32)][b (32)]
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

static size_t wgrad_bf_bytes(const WgradGeom& g, int th, int ns) {
    const size_t x = (size_t)(th + 2 * g.pad) * (32 + 2 * g.pad) * 64 * ns * (g.A / 32);
    const size_t gt = (size_t)th * 32 * 64 * ns;
    return x + gt;
}

static int wgrad_bf_rows(const WgradGeom& g, int ns) {
    int th = 8;
    while (th > 1 && wgrad_bf_bytes(g, th, ns) > 158 * 1024) th >>= 1;
    const int tiles_x = (g.wg + 31) / 32;
    while (th > 2 && (long)g.n * tiles_x * ((g.hg + th - 1) / th) < 256) th >>= 1;
    return th;
}

static int wgrad_bf_blocks(const WgradGeom& g, int ns) {
    const int th = wgrad_bf_rows(g, ns);
    const long ntiles = (long)g.n * ((g.wg + 31) / 32) * ((g.hg + th - 1) / th);
    return (int)(ntiles < 256 ? ntiles : 256);
}

// (A, Q, REM, PFX): units = taps * A / 32 = 8 * Q + REM, as wgrad_lds.hip
#define SENAS_WGRAD_BF_SHAPES(X_)   \
    X_(32, 1, 1, 6)                 \
    X_(32, 3, 1, 14)                \
    X_(64, 2, 2, 11)                \
    X_(96, 3, 3, 8)                 \
    X_(128, 4, 4, 4)

bool bf_wgrad_ok(const WgradGeom& g, int terms) {
    if (terms != 1 && terms != 3 && terms != 6) return false;
    const int ns = terms == 1 ? 1 : (terms == 3 ? 2 : 3);
    if (g.stride != 1 || g.B != 32 || g.A % 32 != 0 || g.A > 128) return false;
    if (g.kh != g.kw || g.pad != g.dil * (g.kh / 2) || g.hi != g.hg || g.wi != g.wg) return false;
    if (g.wg < 32 || g.hg < 8 || g.pad > 6) return false;
    const int units = g.kh * g.kw * (g.A / 32);
    bool found = false;
#define SENAS_CASE(A_, Q_, REM_, PF_) if (g.A == A_ && units == 8 * Q_ + REM_) found = true;
    SENAS_WGRAD_BF_SHAPES(SENAS_CASE)
#undef SENAS_CASE
    if (!found) return false;
    const int th = wgrad_bf_rows(g, ns);
    return wgrad_bf_bytes(g, th, ns) <= 158 * 1024 && (long)g.n * g.hi * g.wi * g.A < 0x7fffffffL;
}

int64_t bf_wgrad_ws_bytes(const WgradGeom& g, int terms) {
    const int ns = terms == 1 ? 1 : (terms == 3 ? 2 : 3);
    return (int64_t)wgrad_bf_blocks(g, ns) * g.kh * g.kw * (g.A / 32) * 1024 * sizeof(float);
}

template <int A, int Q, int REM, int PFX, int NS, bool GB = false>
static int launch_wbf(const WgradGeom& g, const float* X, const float* G, float* part, int x_relu, hipStream_t st) {
    const int th = wgrad_bf_rows(g, NS);
    size_t bytes = wgrad_bf_bytes(g, th, NS);
    const size_t fold = (size_t)4 * REM * 4096;
    if (fold > bytes) bytes = fold;
    if (bytes > 64 * 1024)
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(&wgrad_bf_kernel<A, Q, REM, PFX, NS, GB>), 160 * 1024,
                                     "wgrad_bf: raising the dynamic LDS limit")) return rc;
    const int tiles_x = (g.wg + 31) / 32, tiles_y = (g.hg + th - 1) / th;
    hipLaunchKernelGGL((wgrad_bf_kernel<A, Q, REM, PFX, NS, GB>), dim3(wgrad_bf_blocks(g, NS)), dim3(512), bytes, st, g, X, G, part, x_relu, th,
                       tiles_x, tiles_y);
    return launch_status("wgrad_bf");
}

// prefetch registers the split forms can afford next to their accumulators and fragment rings (no spills: checked with
// -Rpass-analysis=kernel-resource-usage)
static constexpr int bf_pfx(int pf, int uw, int ns) {
    const int cap = ns == 2 ? (uw >= 8 ? 0 : (uw >= 6 ? 2 : (uw >= 4 ? 8 : pf))) : (uw >= 6 ? 0 : (uw >= 4 ? 4 : 6));
    return pf < cap ? pf : cap;
}

// part: bf_wgrad_ws_bytes(g, terms) of scratch; defer: the sum of the per-block images (senas_wgrad_sum_batched, kind 2)
int launch_bf_wgrad(const WgradGeom& g, int terms, const float* X, const float* G, float* part, float* dw, int x_relu, senas_sum_item* defer,
                    hipStream_t st) {
    const int units = g.kh * g.kw * (g.A / 32);
    const int ns = terms == 1 ? 1 : (terms == 3 ? 2 : 3);
    int rc = SENAS_EINVAL;
    bool found = false;
#define SENAS_CASE(A_, Q_, REM_, PF_)                                                                      \
    if (!found && g.A == A_ && units == 8 * Q_ + REM_) {                                                   \
        found = true;                                                                                      \
        if (ns == 1) rc = launch_wbf<A_, Q_, REM_, PF_, 1>(g, X, G, part, x_relu, st);                     \
        else if (ns == 2) rc = launch_wbf<A_, Q_, REM_, bf_pfx(PF_, Q_ + REM_, 2), 2>(g, X, G, part, x_relu, st);   \
        else rc = launch_wbf<A_, Q_, REM_, bf_pfx(PF_, Q_ + REM_, 3), 3>(g, X, G, part, x_relu, st);       \
    }
    SENAS_WGRAD_BF_SHAPES(SENAS_CASE)
#undef SENAS_CASE
    if (!found) { set_error_msg("wgrad_bf: no kernel for this (channels, taps) pair"); return SENAS_EINVAL; }
    if (rc != SENAS_OK) return rc;
    *defer = senas_sum_item{part, dw, 2, g.A, g.B, g.kh * g.kw, 0, wgrad_bf_blocks(g, ns)};
    return SENAS_OK;
}

// "bf16s": the same with G a bf16 tensor (plain bf16 products); X stays fp32
int launch_bf_wgrad_stored(const WgradGeom& g, const float* X, const void* G, float* part, float* dw, int x_relu, senas_sum_item* defer,
                           hipStream_t st) {
    const int units = g.kh * g.kw * (g.A / 32);
    const float* Gf = reinterpret_cast<const float*>(G);
    int rc = SENAS_EINVAL;
    bool found = false;
#define SENAS_CASE(A_, Q_, REM_, PF_)                                                                      \
    if (!found && g.A == A_ && units == 8 * Q_ + REM_) {                                                   \
        found = true;                                                                                      \
        rc = launch_wbf<A_, Q_, REM_, PF_, 1, true>(g, X, Gf, part, x_relu, st);                           \
    }
    SENAS_WGRAD_BF_SHAPES(SENAS_CASE)
#undef SENAS_CASE
    if (!found) { set_error_msg("wgrad_bf: no kernel for this (channels, taps) pair"); return SENAS_EINVAL; }
    if (rc != SENAS_OK) return rc;
    *defer = senas_sum_item{part, dw, 2, g.A, g.B, g.kh * g.kw, 0, wgrad_bf_blocks(g, 1)};
    return SENAS_OK;
}

void bf_wgrad_name(const WgradGeom& g, int terms, char* buf, int len) {
    const int units = g.kh * g.kw * (g.A / 32);
    const int ns = terms == 1 ? 1 : (terms == 3 ? 2 : 3);
    int q = 0, rem = 0, pf = 0;
#define SENAS_CASE(A_, Q_, REM_, PF_) if (g.A == A_ && units == 8 * Q_ + REM_) { q = Q_; rem = REM_; pf = PF_; }
    SENAS_WGRAD_BF_SHAPES(SENAS_CASE)
#undef SENAS_CASE
    if (ns > 1) pf = bf_pfx(pf, q + rem, ns);
    snprintf(buf, len, "wgrad_bf_kernel<%d, %d, %d, %d, %d>", g.A, q, rem, pf, ns);
}

}  // namespace senas
