// Stride-1 dense convolution (forward, and data-gradient of a stride-1 conv) with the input tile
// staged in LDS -- the main kernel of the derived network's 5x5 dilated / 3x3 layers.
//
// Why LDS here: the direct-global kernel (conv_mfma.hip) re-reads every input pixel once per tap as
// 16-byte fragments, 32 cache lines per wave-instruction, and ends up L1/TA-bound at ~30% of the
// fp32-MFMA rate.  Here a block loads its (TH + 2*halo) x (32 + 2*halo) input window ONCE, in full
// 64-byte runs (ReLU applied on the way in, borders zero-filled so the tap loop has no bounds
// checks), and all k*k taps read their A fragments from LDS with conflict-free ds_read_b128.
//
// Tile (big maps): TH x 32 output pixels per 256-thread block, TH = 4*MT; wave w owns rows MT*w .. MT*w+MT-1 (MT
// 32-pixel MFMA sub-tiles).  MT = 2 for maps that fill the chip, MT = 1 when that would leave CUs idle
// (twice the blocks, half the critical path).  Channels go through LDS 16 at a time (pixel stride 80 B =
// 64 B data + 16 B pad: the 16 lanes of a ds_read_b128 group land on 16 distinct 4-bank slots), so the
// window costs 880 px * 80 B = 69 KiB for 5x5 dilation 3 at MT = 2 -- two blocks per CU (three at dilation 2), one
// loading or storing while the others compute: measured 71-81 % of the fp32-MFMA peak at 256x256.
// B fragments (weights) stream from the packed image in L2.
//
// Small maps (template parameters KS, RW, TWL; lds_gather_shape picks them): the taps of a tile are dealt to KS
// groups of waves whose partial accumulators meet in LDS; when even 4-row tiles cannot give every CU a block the
// block shrinks to ONE 32-pixel MFMA row with its taps on 4 waves (one per SIMD); maps 16 or 8 pixels wide fold 2 or
// 4 image rows into that MFMA row.  Those variants keep their weight fragments in registers (requested before the
// window is staged).  Epilogue: full 32-channel tiles are transposed through LDS and stored 16 bytes per lane.
#include "common.h"

namespace senas {

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

constexpr int TW = 32, CH = 16, PST = 20;     // PST: pixel stride in floats (16 data + 4 pad)
constexpr int P4 = PST / 4;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

#ifndef SENAS_XCD_ORDER
#define SENAS_XCD_ORDER 1
#endif

template <int MT>
struct Frag {
    float4 a[2][MT];     // [channel group within the pass][sub-tile]
    float4 b[2];
};

}  // namespace

// grid = (tiles_x, tiles_y, n * co_tiles); dynamic LDS = max(tile_h * tile_w * PST floats, fold scratch)
// MAXT (KS > 1 only): taps per wave group, rounded up -- those variants keep ALL their weight fragments of a channel
// pass in registers, requested before the window is staged, so the short tap loop of a small-map block never waits
// for L2 (with 6-7 taps per wave there is not enough MFMA work per tap to hide a weight fetch behind).
// PF: 16-byte pieces of the NEXT channel pass's window each thread requests before the tap loop of the current pass
// and parks in registers until the loop is done -- the HBM/L2 latency of staging hides behind the MFMAs (only the
// first pass of a block is staged in the open; pieces beyond PF per thread are fetched after the loop).
// RW: row waves per tap group (4, or 1 for maps so small that even 4-row tiles leave most CUs idle: the block is then
// ONE 32-pixel MFMA row whose taps are dealt to the 4 waves, one wave per SIMD -- the shortest critical path there is).
// TWL: tile width in pixels (32, 16, 8): a 32-pixel MFMA row covers 32/TWL image rows, so 16x16 and 8x8 maps use the
// same kernel.  Block = 64 * RW * KS threads; tile = RW * MT * (32/TWL) image rows x TWL columns.
// S: stride of a PLAIN gather (1, or 2: Conv2d stride-2 forward, ConvTranspose2d stride-2 data gradient): output pixel
// (oy, ox) reads window pixel (S*oy + ky*d, S*ox + kx*d), so the window covers S times the tile plus the halo.
// EPI: the inference epilogue (common.h Epi) instead of the raw store; training instantiations ignore `epi`.
template <bool TG, int MT, int KS, int MAXT, int PF, int RW, int TWL, int S = 1, bool EPI = false>
__global__ __launch_bounds__(64 * RW * KS) void conv_lds_kernel(GatherGeom g, const float* __restrict__ in,
                                                       const float* __restrict__ wp, float* __restrict__ out,
                                                       int in_relu, const float* __restrict__ mask,
                                                       double* __restrict__ stats, Epi epi, Pair2 second) {
    constexpr int RPM = 32 / TWL;                        // image rows per MFMA row
    constexpr int TH = RW * MT * RPM;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 64 * RW * KS;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wv % RW;                            // tile row group
    const int kg = wv / RW;                              // tap group (wave-uniform)
    const int r = lane & 31, h = lane >> 5;
    const int pr = r / TWL, px = r % TWL;                // this lane's pixel inside its MFMA row
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (linear id b and b + 8 share an L2), so
    // the k-th workgroup of an XCD takes the k-th tile of that XCD's CONTIGUOUS eighth of the tile list -- neighbouring
    // tiles, whose windows overlap by the halo, then hit the same L2 instead of fetching the halo once per XCD
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (SENAS_XCD_ORDER) {
        const unsigned gx = gridDim.x, gy = gridDim.y, nblk = gx * gy * gridDim.z;
        const unsigned lin = bx + gx * (by + gy * bz);
        const unsigned xcd = lin & 7u, base = nblk >> 3, rem = nblk & 7u;
        const unsigned lp = xcd * base + (xcd < rem ? xcd : rem) + (lin >> 3);
        bx = lp % gx;
        by = (lp / gx) % gy;
        bz = lp / (gx * gy);
    }
    if (second.nz != 0 && bz >= (unsigned)second.nz) {   // the launch's second problem (block-uniform)
        bz -= second.nz;
        in = second.in; wp = second.w; out = second.out; mask = second.mask; stats = second.stats;
        g.dil = second.dil; g.pad = second.pad;
    }
    const int n = bz % g.n, cot = bz / g.n;
    const int co = cot * 32 + r;
    const int oy0 = by * TH, ox0 = bx * TWL;
    const int halo = g.pad;                              // = dil * (k / 2) on this path
    static_assert(S == 1 || !TG, "the strided form is a plain gather");
    const int tile_w = S * TWL + 2 * halo, tile_h = S * TH + 2 * halo;
    const int ngroups = g.cin >> 3, npass = g.cin / CH;
    const int taps = g.kh * g.kw;
    wp += (size_t)cot * taps * ngroups * 256 + lane * 4;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;

    // LDS offset (in 16-byte units, so the accesses are provably aligned -> ds_read_b128) of this lane's
    // pixel for tap (0,0), channel chunk h, sub-tile 0; sub-tile m is tile_w * P4 * m further
    float4* lds4 = reinterpret_cast<float4*>(lds);
    const int lbase = (S * (MT * wave * RPM + pr) * tile_w + S * px) * P4 + h;
    const int lrow = S * RPM * tile_w * P4;

    // staging geometry of this thread (pass-independent): slot k holds window pixel k*(NT/4) + (tid>>2), piece tid&3
    constexpr int XL = NT / 4;
    const int sq = threadIdx.x & 3, spl = threadIdx.x >> 2;
    const int ty0 = spl / tile_w, tx0 = spl - ty0 * tile_w;
    const int dty = XL / tile_w, dtx = XL - dty * tile_w;             // slot-to-slot step in (row, column)
    const int wpix = tile_h * tile_w;
    const float* src0 = in + (size_t)n * g.hin * g.win * g.cin + sq * 4;
    float4 pf[PF > 0 ? PF : 1];
    auto issue = [&](int pass) {                                       // loads only; borders resolved in commit
        const float* src = src0 + pass * CH;
        int ty = ty0, tx = tx0;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int iy = S * oy0 - halo + ty, ix = S * ox0 - halo + tx;
            const bool inb = k * XL + spl < wpix && iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win;
            pf[k] = *reinterpret_cast<const float4*>(src + (inb ? ((size_t)iy * g.win + ix) * g.cin : 0));
            ty += dty; tx += dtx;
            if (tx >= tile_w) { tx -= tile_w; ++ty; }
        }
    };
    auto commit = [&](int pass) {                                      // registers -> LDS, then the remainder
        const float* src = src0 + pass * CH;
        int ty = ty0, tx = tx0;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int iy = S * oy0 - halo + ty, ix = S * ox0 - halo + tx;
            const bool live = k * XL + spl < wpix;
            const bool inb = live && iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win;
            float4 v = pf[k];
            if (in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (!inb) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) lds4[(k * XL + spl) * P4 + sq] = v;
            ty += dty; tx += dtx;
            if (tx >= tile_w) { tx -= tile_w; ++ty; }
        }
        for (int k0 = PF; k0 * XL < wpix; k0 += 4) {                   // 4 loads in flight per thread
            float4 v[4];
            bool ok[4], lv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int iy = S * oy0 - halo + ty, ix = S * ox0 - halo + tx;
                lv[u] = (k0 + u) * XL + spl < wpix;
                ok[u] = lv[u] && iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win;
                v[u] = *reinterpret_cast<const float4*>(src + (ok[u] ? ((size_t)iy * g.win + ix) * g.cin : 0));
                ty += dty; tx += dtx;
                if (tx >= tile_w) { tx -= tile_w; ++ty; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (in_relu) { v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f); v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f); }
                if (!ok[u]) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lv[u]) lds4[((k0 + u) * XL + spl) * P4 + sq] = v[u];
            }
        }
    };

    SENAS_PHASE(0);
    issue(0);
    for (int pass = 0; pass < npass; ++pass) {
        float4 bfr[KS > 1 ? MAXT : 1][2];                // KS > 1: this wave's weight fragments of the pass
        if (KS > 1) {
#pragma unroll
            for (int i = 0; i < MAXT; ++i) {
                const int t = kg + KS * i;
                const float* wt = wp + ((size_t)(t < taps ? t : 0) * ngroups + pass * 2) * 256;
                bfr[i][0] = *reinterpret_cast<const float4*>(wt);
                bfr[i][1] = *reinterpret_cast<const float4*>(wt + 256);
                if (t >= taps) bfr[i][0] = bfr[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);      // padding slot: adds nothing
            }
        }
        __syncthreads();                                 // previous pass's readers are done
        SENAS_PHASE(1 + pass * 4);
        commit(pass);
        SENAS_PHASE(2 + pass * 4);
        __syncthreads();
        SENAS_PHASE(3 + pass * 4);
        if (pass + 1 < npass) issue(pass + 1);           // in flight during the tap loop

        // ---- taps, software-pipelined over two alternating fragment sets (no register copies):
        // while tap t's MFMAs issue, tap t+1's A fragments (LDS) and B fragments (L2) are in flight
        auto tap_ptrs = [&](int t, int& toff, const float*& wt) {
            const int ky = t / g.kw, kx = t - ky * g.kw;
            // plain gather: window row = oy_local + ky*d ; transposed (stride 1): oy_local + (k-1-ky)*d
            const int dy = (TG ? (g.kh - 1 - ky) : ky) * g.dil, dx = (TG ? (g.kw - 1 - kx) : kx) * g.dil;
            toff = (dy * tile_w + dx) * P4;
            wt = wp + ((size_t)t * ngroups + pass * 2) * 256;
        };
        auto load_a = [&](int toff, Frag<MT>& f) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int m = 0; m < MT; ++m) f.a[c2][m] = lds4[lbase + m * lrow + toff + c2 * 2];
        };
        auto load_b = [&](const float* wt, Frag<MT>& f) {
            f.b[0] = *reinterpret_cast<const float4*>(wt);
            f.b[1] = *reinterpret_cast<const float4*>(wt + 256);
        };
        // One tap = 8*MT MFMAs (64 cycles each).  hipcc waits vmcnt(0) in front of the first MFMA that needs
        // a weight fragment, so (a) both fragments of the tap are consumed by the first 2*MT MFMAs, while only
        // they are outstanding, and (b) the NEXT tap's weight loads are issued right after those
        // (pinned by sched_barrier): by the time the next tap starts they have had 6*MT*64 cycles to land.
        auto step = [&](const Frag<MT>& cur, Frag<MT>& nxt, bool more, int toff_n, const float* wt_n) {
            if (more) load_a(toff_n, nxt);
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma32(cur.a[c2][m].x, cur.b[c2].x, acc[m]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) load_b(wt_n, nxt);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    acc[m] = mfma32(cur.a[c2][m].y, cur.b[c2].y, acc[m]);
                    acc[m] = mfma32(cur.a[c2][m].z, cur.b[c2].z, acc[m]);
                    acc[m] = mfma32(cur.a[c2][m].w, cur.b[c2].w, acc[m]);
                }
        };
        if (KS == 1) {
            Frag<MT> f0, f1;
            int toff;
            const float* wt;
            tap_ptrs(0, toff, wt);
            load_b(wt, f0);
            load_a(toff, f0);
            for (int t = 0;;) {
                bool more = t + 1 < taps;
                if (more) tap_ptrs(t + 1, toff, wt);
                step(f0, f1, more, toff, wt);
                if (++t >= taps) break;
                more = t + 1 < taps;
                if (more) tap_ptrs(t + 1, toff, wt);
                step(f1, f0, more, toff, wt);
                if (++t >= taps) break;
            }
        } else {
            // straight-line: weights are in registers, A fragments come from LDS with compile-time trip counts
#pragma unroll
            for (int i = 0; i < MAXT; ++i) {
                const int t = kg + KS * i;
                int toff;
                const float* unused;
                tap_ptrs(t < taps ? t : 0, toff, unused);
                float4 fa[2][MT];
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int m = 0; m < MT; ++m) fa[c2][m] = lds4[lbase + m * lrow + toff + c2 * 2];
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        acc[m] = mfma32(fa[c2][m].x, bfr[i][c2].x, acc[m]);
                        acc[m] = mfma32(fa[c2][m].y, bfr[i][c2].y, acc[m]);
                        acc[m] = mfma32(fa[c2][m].z, bfr[i][c2].z, acc[m]);
                        acc[m] = mfma32(fa[c2][m].w, bfr[i][c2].w, acc[m]);
                    }
            }
        }
        SENAS_PHASE(4 + pass * 4);
    }

    SENAS_PHASE(40);
    if (KS > 1) {            // pairwise fold of the tap groups' accumulators: [slot][sub-tile][quad][lane] float4
        float4* fold = reinterpret_cast<float4*>(lds);
#pragma unroll
        for (int sgrp = KS / 2; sgrp >= 1; sgrp >>= 1) {
            __syncthreads();
            if (kg >= sgrp && kg < 2 * sgrp) {
                const int slot = (kg - sgrp) * RW + wave;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        fold[((slot * MT + m) * 4 + q4) * 64 + lane] =
                            make_float4(acc[m][4 * q4], acc[m][4 * q4 + 1], acc[m][4 * q4 + 2], acc[m][4 * q4 + 3]);
            }
            __syncthreads();
            if (kg < sgrp) {
                const int slot = kg * RW + wave;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const float4 pv = fold[((slot * MT + m) * 4 + q4) * 64 + lane];
                        acc[m][4 * q4] += pv.x; acc[m][4 * q4 + 1] += pv.y; acc[m][4 * q4 + 2] += pv.z; acc[m][4 * q4 + 3] += pv.w;
                    }
            }
        }
    }

    SENAS_PHASE(41);
    // ---- epilogue: lane = output channel, register v = pixel (row MT*wave + m, column acc_row(v, h))
    const bool cok = co < g.cout;
    double s = 0.0, q = 0.0;
    if ((g.cout & 31) == 0) {
        // full 32-channel tile: transpose each wave's sub-tiles through LDS ([pixel][36 floats]: rows stay 16-byte aligned,
        // 4-float pad against bank conflicts) so that a lane stores 16 contiguous bytes -- 4x fewer, fully coalesced
        // store (and mask load) instructions than one dword per (pixel, channel)
        if (stats != nullptr && kg == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int pq = acc_row(v, h);
                    const int oy = oy0 + (MT * wave + m) * RPM + pq / TWL, ox = ox0 + pq % TWL;
                    if (oy < g.hout && ox < g.wout) { s += acc[m][v]; q += (double)acc[m][v] * acc[m][v]; }
                }
        }
        __syncthreads();                                 // the window (or the fold scratch) is dead
        float* tp = lds + (size_t)wave * MT * 32 * 36;
        if (kg == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int v = 0; v < 16; ++v) tp[(m * 32 + acc_row(v, h)) * 36 + r] = acc[m][v];
        }
        __syncthreads();
        if (kg == 0) {
            const int prow = lane >> 3, c4 = lane & 7;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int pq = k * 8 + prow;
                    const int oy = oy0 + (MT * wave + m) * RPM + pq / TWL, ox = ox0 + pq % TWL;
                    if (oy < g.hout && ox < g.wout) {
                        float4 val = *reinterpret_cast<const float4*>(tp + (m * 32 + pq) * 36 + c4 * 4);
                        const size_t o = out_offset(g, ((size_t)n * g.hout + oy) * g.wout + ox, cot * 32 + c4 * 4);      // (mask / epilogue forms: interleaved only)
                        if (mask != nullptr) {
                            const float4 mk = *reinterpret_cast<const float4*>(mask + o);
                            if (!(mk.x > 0.f)) val.x = 0.f;
                            if (!(mk.y > 0.f)) val.y = 0.f;
                            if (!(mk.z > 0.f)) val.z = 0.f;
                            if (!(mk.w > 0.f)) val.w = 0.f;
                        }
                        if constexpr (EPI) {
                            const int pc = n * g.cout + cot * 32 + c4 * 4;
                            const float4 sc = *reinterpret_cast<const float4*>(epi.scale + pc);
                            const float4 bi = *reinterpret_cast<const float4*>(epi.bias + pc);
                            val.x = fmaf(val.x, sc.x, bi.x); val.y = fmaf(val.y, sc.y, bi.y);
                            val.z = fmaf(val.z, sc.z, bi.z); val.w = fmaf(val.w, sc.w, bi.w);
                            if (epi.addend != nullptr) {
                                const float4 ad = *reinterpret_cast<const float4*>(epi.addend + o);
                                float4 as = make_float4(1.f, 1.f, 1.f, 1.f);
                                if (epi.add_scale != nullptr) as = *reinterpret_cast<const float4*>(epi.add_scale + pc);
                                val.x = fmaf(ad.x, as.x, val.x); val.y = fmaf(ad.y, as.y, val.y);
                                val.z = fmaf(ad.z, as.z, val.z); val.w = fmaf(ad.w, as.w, val.w);
                            }
                            if (epi.relu) {
                                val.x = fmaxf(val.x, 0.f); val.y = fmaxf(val.y, 0.f);
                                val.z = fmaxf(val.z, 0.f); val.w = fmaxf(val.w, 0.f);
                            }
                        }
                        *reinterpret_cast<float4*>(out + o) = val;
                    }
                }
        }
    } else {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int pq = acc_row(v, h);                // pixel of the MFMA row this register holds
                const int oy = oy0 + (MT * wave + m) * RPM + pq / TWL, ox = ox0 + pq % TWL;
                float val = acc[m][v];
                if (kg == 0 && oy < g.hout && ox < g.wout && cok) {
                    const size_t o = out_offset(g, ((size_t)n * g.hout + oy) * g.wout + ox, co);
                    if (mask != nullptr && !(mask[o] > 0.f)) val = 0.f;
                    s += val;
                    q += (double)val * val;
                    if constexpr (EPI) {
                        const int pc = n * g.cout + co;
                        val = fmaf(val, epi.scale[pc], epi.bias[pc]);
                        if (epi.addend != nullptr) val = fmaf(epi.addend[o], epi.add_scale != nullptr ? epi.add_scale[pc] : 1.f, val);
                        if (epi.relu) val = fmaxf(val, 0.f);
                    }
                    out[o] = val;
                }
            }
        }
    }
    SENAS_PHASE(42);
    if (stats != nullptr) {                              // block-level reduction: 2 atomics per channel per block
        __syncthreads();
        double* red = reinterpret_cast<double*>(lds);    // [RW waves][32 channels][2]
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        if (kg == 0 && h == 0) { red[(wave * 32 + r) * 2] = s; red[(wave * 32 + r) * 2 + 1] = q; }
        __syncthreads();
        if (kg == 0 && wave == 0 && h == 0 && cok) {
            for (int w = 1; w < RW; ++w) { s += red[(w * 32 + r) * 2]; q += red[(w * 32 + r) * 2 + 1]; }
            double* st = stats + ((size_t)n * g.cout + co) * 2;
            atomicAdd(st, s);
            atomicAdd(st + 1, q);
        }
    }
    SENAS_PHASE(43);
}

SENAS_PHASE_READER(conv_lds)

static int lds_tile_width(const GatherGeom& g) { return g.wout >= 32 ? 32 : (g.wout >= 16 ? 16 : 8); }

bool lds_gather_ok(const GatherGeom& g) {
    // stride 1, "same" padding, 16-channel passes, maps at least 8 wide and 4 high
    if (g.stride != 1 || g.cin % CH != 0) return false;
    if (g.kh != g.kw || g.pad != g.dil * (g.kh / 2) || g.hout != g.hin || g.wout != g.win) return false;
    if (g.wout < 8 || g.hout < 4) return false;
    const size_t bytes = (size_t)(8 + 2 * g.pad) * (TW + 2 * g.pad) * PST * sizeof(float);
    return bytes <= 150 * 1024 && (long)g.n * g.hin * g.win * g.cin < 0x7fffffffL;
}

static size_t conv_lds_bytes(const GatherGeom& g, int th, int twl, int mt, int ks, int rw, int s = 1) {
    size_t bytes = (size_t)(s * th + 2 * g.pad) * (s * twl + 2 * g.pad) * PST * sizeof(float);
    const size_t fold = ks > 1 ? (size_t)(ks / 2) * rw * mt * 4096 : 0;
    if (bytes < 4 * 32 * 2 * sizeof(double)) bytes = 4 * 32 * 2 * sizeof(double);      // statistics scratch
    const size_t tr = (size_t)rw * mt * 32 * 36 * sizeof(float);                      // epilogue transpose
    if (bytes < tr) bytes = tr;
    return fold > bytes ? fold : bytes;
}

template <bool TG, int MT, int KS, int MAXT, int PF, int RW, int TWL, int S = 1, bool EPI = false>
static int launch_lds_variant(const GatherGeom& g, const float* in, const float* wp, float* out, int in_relu,
                              const float* mask, double* stats, hipStream_t st, const Epi& epi = Epi{}, const Pair2& pr0 = Pair2{}) {
    constexpr int TH = RW * MT * (32 / TWL);
    GatherGeom gmax = g;                                 // the wider halo of the two problems sizes the window
    Pair2 pr = pr0;
    const int nz1 = g.n * ((g.cout + 31) / 32);
    if (pr.nz != 0) { pr.nz = nz1; if (pr.pad > gmax.pad) gmax.pad = pr.pad; }
    const size_t bytes = conv_lds_bytes(gmax, TH, TWL, MT, KS, RW, S);
    if (bytes > 64 * 1024)
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(&conv_lds_kernel<TG, MT, KS, MAXT, PF, RW, TWL, S, EPI>), 150 * 1024,
                                     "conv_lds: raising the dynamic LDS limit")) return rc;
    dim3 grid((g.wout + TWL - 1) / TWL, (g.hout + TH - 1) / TH, pr.nz != 0 ? 2 * nz1 : nz1);
    hipLaunchKernelGGL((conv_lds_kernel<TG, MT, KS, MAXT, PF, RW, TWL, S, EPI>), grid, dim3(64 * RW * KS), bytes, st, g, in, wp, out, in_relu,
                       mask, stats, epi, pr);
    return launch_status("conv_lds");
}

// (MT, KS, RW, TWL) for a geometry.  Maps >= 32 wide: 8x32 tiles when they still give every CU two blocks; otherwise
// 4x32 tiles, and when even those leave the chip under-filled the taps of a tile are dealt to 2 or 4 groups of waves.
// When 4-row tiles would not even give half the CUs a block (32x32 maps and below) a block is ONE 32-pixel MFMA row
// with its taps on 4 waves (RW = 1): small maps are critical-path-bound, not throughput-bound.
void lds_gather_shape(const GatherGeom& g, int& mt, int& ks, int& rw, int& twl) {
    twl = lds_tile_width(g);
    const int taps = g.kh * g.kw;
    const long cot = (g.cout + 31) / 32, tx = (g.wout + twl - 1) / twl;
    rw = 4;
    if (twl < 32) {                                      // 16- and 8-wide maps: always the single-row form
        mt = 1; rw = 1; ks = taps >= 9 && taps <= 25 ? 4 : 1;
        return;
    }
    const long blocks8 = tx * ((g.hout + 7) / 8) * g.n * cot, blocks4 = tx * ((g.hout + 3) / 4) * g.n * cot;
    mt = (blocks8 >= 512 && g.hout >= 8) ? 2 : 1;
    ks = 1;
    if (mt == 1 && taps >= 9 && taps <= 25) {
        ks = blocks4 <= 256 ? 4 : (blocks4 <= 512 ? 2 : 1);
        if (blocks4 <= 128) rw = 1;
    }
}

// every (MT, KS, MAXT, RW, TWL) lds_gather_shape can pick; LV(MT, KS, MAXT, RW, TWL) returns from the caller
#define SENAS_LDS_DISPATCH(LV)                                                                                          \
    if (rw == 1) {                                                                                                      \
        if (twl == 32) { if (taps <= 9) LV(1, 4, 3, 1, 32); LV(1, 4, 7, 1, 32); }                                       \
        if (twl == 16) { if (ks == 1) LV(1, 1, 1, 1, 16); if (taps <= 9) LV(1, 4, 3, 1, 16); LV(1, 4, 7, 1, 16); }      \
        if (ks == 1) LV(1, 1, 1, 1, 8);                                                                                 \
        if (taps <= 9) LV(1, 4, 3, 1, 8);                                                                               \
        LV(1, 4, 7, 1, 8);                                                                                              \
    }                                                                                                                   \
    if (mt == 2) LV(2, 1, 1, 4, 32);                                                                                    \
    if (ks == 4 && taps <= 9) LV(1, 4, 3, 4, 32);                                                                       \
    if (ks == 4 && taps <= 25) LV(1, 4, 7, 4, 32);                                                                      \
    if (ks == 2 && taps <= 9) LV(1, 2, 5, 4, 32);                                                                       \
    if (ks == 2 && taps <= 25) LV(1, 2, 13, 4, 32);                                                                     \
    LV(1, 1, 1, 4, 32);

template <bool TG>
int launch_lds_gather(const GatherGeom& g, const float* in, const float* wp, float* out, int in_relu,
                      const float* mask, double* stats, hipStream_t st, const Pair2& pr) {
    int mt, ks, rw, twl;
    GatherGeom gs = g;
    if (pr.nz != 0) gs.n *= 2;                           // (tile choice by the blocks of the whole launch)
    lds_gather_shape(gs, mt, ks, rw, twl);
    const int taps = g.kh * g.kw;
#define SENAS_LV(MT_, KS_, MAXT_, RW_, TWL_) return launch_lds_variant<TG, MT_, KS_, MAXT_, 0, RW_, TWL_>(g, in, wp, out, in_relu, mask, stats, st, Epi{}, pr)
    SENAS_LDS_DISPATCH(SENAS_LV)
#undef SENAS_LV
}

// forward convolution with the inference epilogue: same tile choice, EPI twin of the kernel
int launch_lds_gather_epi(const GatherGeom& g, const float* in, const float* wp, float* out, int in_relu, const Epi& epi,
                          hipStream_t st) {
    int mt, ks, rw, twl;
    lds_gather_shape(g, mt, ks, rw, twl);
    const int taps = g.kh * g.kw;
#define SENAS_LV(MT_, KS_, MAXT_, RW_, TWL_) \
    return launch_lds_variant<false, MT_, KS_, MAXT_, 0, RW_, TWL_, 1, true>(g, in, wp, out, in_relu, nullptr, nullptr, st, epi)
    SENAS_LDS_DISPATCH(SENAS_LV)
#undef SENAS_LV
}

// the kernel symbol launch_lds_gather picks (for senas_conv2d_kernel_name)
void lds_gather_name(const GatherGeom& g, bool tg, char* buf, int len) {
    int mt, ks, rw, twl;
    lds_gather_shape(g, mt, ks, rw, twl);
    const int taps = g.kh * g.kw;
    const int maxt = ks == 1 ? 1 : (taps <= 9 ? (ks == 4 ? 3 : 5) : (ks == 4 ? 7 : 13));
    snprintf(buf, len, "conv_lds_kernel<%s, %d, %d, %d, 0, %d, %d, 1, false>", tg ? "true" : "false", mt, ks, maxt, rw, twl);
}

// ---- stride-2 plain gather: one 32-pixel MFMA row per block (2 output rows x 16, or 4 x 8), taps on 4 waves
bool lds_gather_s2_ok(const GatherGeom& g) {
    if (g.stride != 2 || g.cin % CH != 0) return false;
    if (g.kh != g.kw || g.pad != g.dil * (g.kh / 2) || g.hin != 2 * g.hout || g.win != 2 * g.wout) return false;
    const int taps = g.kh * g.kw;
    if (taps != 9 && taps != 25) return false;
    if (g.wout < 8 || g.hout < 4) return false;
    return (long)g.n * g.hin * g.win * g.cin < 0x7fffffffL;
}

int launch_lds_gather_s2(const GatherGeom& g, const float* in, const float* wp, float* out, int in_relu, const float* mask,
                         double* stats, hipStream_t st, const Pair2& pr) {
    const int taps = g.kh * g.kw;
    if (g.wout >= 16) {
        if (taps <= 9) return launch_lds_variant<false, 1, 4, 3, 0, 1, 16, 2>(g, in, wp, out, in_relu, mask, stats, st, Epi{}, pr);
        return launch_lds_variant<false, 1, 4, 7, 0, 1, 16, 2>(g, in, wp, out, in_relu, mask, stats, st, Epi{}, pr);
    }
    if (taps <= 9) return launch_lds_variant<false, 1, 4, 3, 0, 1, 8, 2>(g, in, wp, out, in_relu, mask, stats, st, Epi{}, pr);
    return launch_lds_variant<false, 1, 4, 7, 0, 1, 8, 2>(g, in, wp, out, in_relu, mask, stats, st, Epi{}, pr);
}

template int launch_lds_gather<false>(const GatherGeom&, const float*, const float*, float*, int, const float*, double*, hipStream_t, const Pair2&);
template int launch_lds_gather<true>(const GatherGeom&, const float*, const float*, float*, int, const float*, double*, hipStream_t, const Pair2&);

}  // namespace senas
