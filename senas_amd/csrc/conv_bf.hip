// Stride-1 "same" dense convolution (forward, and the data gradient of one) on the bf16 matrix pipe with fp32
// accumulation -- the split-operand forms of the fp32 convolution, and the plain bf16-operand form:
//
//   NS = 1  "bf16"   : both operands rounded to bf16 (RNE), one v_mfma_f32_32x32x16_bf16 per 16 reduction channels
//   NS = 2  "bf16x3" : x = hi + lo (hi = bf16(x), lo = bf16(x - hi)); products hi*hi + hi*lo + lo*hi -- every product is
//                      exact in the fp32 accumulator, what is dropped (lo*lo and the residues x - hi - lo) is ~2^-17 of |x||w|
//   NS = 3  "bf16x6" : x = hi + mid + lo exactly (3 x 8 significant bits = fp32's 24); products hi*hi + hi*mid + mid*hi +
//                      hi*lo + lo*hi + mid*mid; dropped terms are <= 2^-25 of |x||w| -- below fp32's own rounding
//
// The fp32-input MFMA (conv_lds.hip) runs at the vector rate, 64 FLOP/clk/SIMD; the bf16 form does 1024.  Three of them
// per 16 channels are 5.3x, six 2.7x the fp32 pipe -- and unlike the fp32 form they leave 24 of every 32 cycles of the
// SIMD's issue port to other instructions (MI355X_MICROARCH.md, cycle constants), so the staging work overlaps.
//
// Structure (one persistent 512-thread workgroup per CU, roles fixed per wave):
//   waves 0-3  consumers : wave w owns rows MT*w .. MT*w + MT-1 of a (4*MT) x 32 output tile; per tap it reads its A
//                          fragments (pixels x 16 channels, 16 B per lane and plane) from the LDS window with
//                          ds_read_b128 and its B fragments (weights, pre-split once per step) from the packed image in L2,
//                          three taps ahead in a register ring; MFMAs back to back.
//   waves 4-7  producers : load the NEXT work item's fp32 input window (tile + halo, 16 or 32 channels) from HBM/L2,
//                          apply the ReLU-on-load, split into bf16 planes and write them to the OTHER window buffer.
// A work item is (tile, channel pass); one __syncthreads per item swaps the buffers.  Producer loads never sit in front
// of a consumer's weight loads in the in-order vmcnt queue -- the reason for the role split: with one role per wave the
// window prefetch (HBM latency) would stall every weight fragment wait behind it.
//
// LDS window: [pixel][plane][16 channels] bf16, pixel stride 80 B (112 B for NS = 3): 64 (96) B data + 16 B pad, so the 16
// lanes of a ds_read_b128 group (MI355X_MICROARCH.md section LDS) land on 16 distinct 16-byte slots.
// Epilogue: lane = output channel, so a store instruction writes two full 128-byte rows; per-image channel statistics
// (fp64) stay in registers across the tiles of an image and leave as one atomic pair per channel and wave.
#include "common.h"
#include <stdlib.h>

namespace senas {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ f32x16 mfma_bf(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

// two floats -> two bf16 (round to nearest even), a in the low half
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// x -> NS bf16 planes whose sum is x up to 2^-9 (1), 2^-18 (2) relative, or exactly (3); planes[p] = 4 packed bf16
template <int NS>
__device__ __forceinline__ void split4(const float4& x, uint2 (&pl)[NS]) {
    float r[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int p = 0; p < NS; ++p) {
        const unsigned u0 = pk_bf16(r[0], r[1]), u1 = pk_bf16(r[2], r[3]);
        pl[p] = make_uint2(u0, u1);
        if (p + 1 < NS) {                       // the residue is exact in fp32
            r[0] -= bf_lo(u0); r[1] -= bf_hi(u0); r[2] -= bf_lo(u1); r[3] -= bf_hi(u1);
        }
    }
}

struct BfArgs {
    GatherGeom g;
    const float* in;
    const uint4* wimg;
    float* out;
    const float* mask;
    double* stats;
    int in_relu;
    int tiles_x, tiles_y, ntiles, cot;      // ntiles = cot * n * tiles_y * tiles_x
    int probe;                              // tuning build only (make probe, -DSENAS_BF_PROBE): bit 0 no staging, 1 no taps, 2 no epilogue, 3 no LDS fragment reads inside the tap loop, 4 no weight-fragment loads
};

// The shipped library compiles the probe tests away: no launch can skip work.
#ifdef SENAS_BF_PROBE
#define SENAS_BF_SKIP(a, bit) (((a).probe >> (bit)) & 1)
#else
#define SENAS_BF_SKIP(a, bit) false
#endif

}  // namespace

// products of an NS-plane split, smallest first: (plane of A, plane of B)
template <int NS> struct Prod;
template <> struct Prod<1> { static constexpr int N = 1; static constexpr int a[1] = {0}; static constexpr int b[1] = {0}; };
template <> struct Prod<2> { static constexpr int N = 3; static constexpr int a[3] = {1, 0, 0}; static constexpr int b[3] = {0, 1, 0}; };
template <> struct Prod<3> { static constexpr int N = 6; static constexpr int a[6] = {1, 2, 0, 1, 0, 0}; static constexpr int b[6] = {1, 0, 2, 0, 1, 0}; };

// largest halo an instantiation serves: 3x3 kernels up to dilation 2, 5x5 up to dilation 3
constexpr int bf_max_halo(int ksz) { return ksz == 3 ? 2 : 6; }
// staging sweeps of the producers' 256 threads over the largest window: (4 MT + 2 halo) x (32 + 2 halo) pixels, 256 / Q per sweep
constexpr int bf_pieces(int mt, int ns, int ksz) {
    const int halo = bf_max_halo(ksz), xl = ns == 1 ? 32 : 64;
    return ((4 * mt + 2 * halo) * (32 + 2 * halo) + xl - 1) / xl;
}

// IB: the input tensor is bf16 in HBM (a stored gradient of a bf16-stored convolution output: the data gradient's operand) -- its
// staging is a copy; OB: the output tensor is written as bf16 (fp32 accumulators rounded to nearest even on the way out; the
// producer-side statistics are taken from the accumulators, before the rounding).  NS = 1 only ("bf16s": section 4c of DESIGN.md).
template <bool TG, int MT, int NS, int KSZ, bool IB = false, bool OB = false>
__global__ __launch_bounds__(512) void conv_bf_kernel(BfArgs a) {
    static_assert(!(IB || OB) || NS == 1, "bf16-stored operands go with the plain bf16 products");
    constexpr int CP = NS == 1 ? 32 : 16;           // channels per pass
    constexpr int Q = CP / 4;                       // float4 pieces per pixel and pass
    constexpr int XL = 256 / Q;                     // window pixels per staging sweep of the 4 producer waves
    constexpr int U2 = NS == 3 ? 3 : 2;             // 16-byte fragments per lane, tap and operand
    constexpr int P16 = NS == 3 ? 7 : 5;            // pixel stride in 16-byte units
    constexpr int TAPS = KSZ * KSZ;
    constexpr int TH = 4 * MT;
    constexpr int PF = bf_pieces(MT, NS, KSZ);      // staging sweeps: the largest window of this instantiation
    constexpr int RING_WANT = NS == 1 ? 12 : (NS == 2 ? 10 : 7);
    constexpr int RING = RING_WANT < KSZ * KSZ ? RING_WANT : KSZ * KSZ;      // (the look-ahead never reaches past the NEXT item)
    // fragments RING - 1 taps ahead of the MFMAs that use them (the epilogue's stores sit in the same in-order vmcnt queue:
    // the look-ahead has to outlast them)
    constexpr int AHEAD = RING - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const GatherGeom& g = a.g;
    const int tid = threadIdx.x & 255;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wv >= 4;
    const int wave = wv & 3;
    const int r = lane & 31, h = lane >> 5;
    const int halo = g.pad;
    const int tile_w = 32 + 2 * halo, tile_h = TH + 2 * halo;
    const int wpix = tile_w * tile_h;
    const int wbytes = PF * XL * P16 * 16;          // (padding slots behind the window take the stores of idle pieces)
    const int npass = g.cin / CP;

    // this block's tiles: XCD x (blocks b with b % 8 == x share an L2) owns a contiguous eighth of the tile list, and a block
    // a contiguous run inside it -- neighbouring tiles (overlapping halos) meet in one L2, and a block stays inside one image
    const int G = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, idx = b >> 3;
    const int gx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    const int x_lo = (int)(((long)a.ntiles * xcd) >> 3), x_hi = (int)(((long)a.ntiles * (xcd + 1)) >> 3);
    const int xn = x_hi - x_lo;
    const int t0 = x_lo + (int)(((long)xn * idx) / gx), t1 = x_lo + (int)(((long)xn * (idx + 1)) / gx);
    const int items = (t1 - t0) * npass;
    if (items <= 0) return;

    auto decode = [&](int tile, int& n, int& cot, int& oy0, int& ox0) {
        const int tx = tile % a.tiles_x;
        int q = tile / a.tiles_x;
        const int ty = q % a.tiles_y;
        q /= a.tiles_y;
        n = q % g.n;
        cot = q / g.n;
        oy0 = ty * TH;
        ox0 = tx * 32;
    };

    if (producer) {
        // ------------------------------------------------------------------------------------------------ producers
        // Rules this section follows (each one measured on this kernel):
        //  - NO control flow between the loads of an item, nor between its LDS stores: across a branch the compiler waits for
        //    every outstanding load (vmcnt(0)), which serialises them.  Pieces beyond the window load element 0 and are
        //    stored into the padding slots behind the window (the buffers hold PF * XL pixels); pieces outside the image
        //    are zeroed by a select.
        //  - the loads of item j + 2 are issued right after item j + 1 has been converted and stored, one whole phase before
        //    they are needed: the HBM / L2 round trip is never on the critical path of a phase.
        //  - the window position (ty, tx) of every piece is item-independent: packed once into one register per piece;
        //    offsets are 32-bit (the launcher checks that the tensor has fewer than 2^31 elements).
        const int sq = tid % Q, spl = tid / Q;
        int pos[PF];
        {
            int ty = spl / tile_w, tx = spl - ty * tile_w;
            const int dty = XL / tile_w, dtx = XL - dty * tile_w;
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                pos[k] = (k * XL + spl < wpix) ? ((ty << 16) | tx) : -1;       // -1: a padding slot
                ty += dty; tx += dtx;
                if (tx >= tile_w) { tx -= tile_w; ++ty; }
            }
        }
        const int rowc = g.win * g.cin;
        const float lo_clamp = a.in_relu ? 0.f : -__builtin_huge_valf();       // max(v, lo): the ReLU on load, or nothing
        auto issue = [&](int j, float4 (&pf)[PF], unsigned& ok) {
            int n, cot, oy0, ox0;
            decode(t0 + j / npass, n, cot, oy0, ox0);
            const size_t base = (size_t)n * g.hin * g.win * g.cin + (j % npass) * CP + sq * 4;
            const float* src = a.in + base;
            const unsigned short* srch = reinterpret_cast<const unsigned short*>(a.in) + base;      // (IB: the same element offsets, 2-byte elements)
            const int by = oy0 - halo, bx = ox0 - halo;
            ok = 0;                                     // bit k: piece k of the item lies inside the image
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int iy = by + (pos[k] >> 16), ix = bx + (pos[k] & 0xffff);
                const bool inb = pos[k] >= 0 && (unsigned)iy < (unsigned)g.hin && (unsigned)ix < (unsigned)g.win;
                ok |= inb ? (1u << k) : 0u;
                const int off = (iy * rowc + ix * g.cin) & -(int)inb;          // (arithmetic, not a select: no branch)
                if constexpr (IB) {
                    const uint2 raw = *reinterpret_cast<const uint2*>(srch + off);      // four bf16: the plane itself
                    pf[k] = make_float4(__builtin_bit_cast(float, raw.x), __builtin_bit_cast(float, raw.y), 0.f, 0.f);
                } else {
                    pf[k] = *reinterpret_cast<const float4*>(src + off);
                }
            }
        };
        auto commit = [&](int j, const float4 (&pf)[PF], unsigned ok) {
            unsigned char* buf = lds_raw + (size_t)(j & 1) * wbytes + spl * (P16 * 16) + sq * 8;
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                float4 v = pf[k];
                const bool inb = (ok >> k) & 1u;
                if constexpr (IB) {                     // already bf16 (no ReLU on load on this path: the launcher checks): a copy
                    const uint2 raw = make_uint2(inb ? __builtin_bit_cast(unsigned, v.x) : 0u, inb ? __builtin_bit_cast(unsigned, v.y) : 0u);
                    *reinterpret_cast<uint2*>(buf + k * (XL * P16 * 16)) = raw;
                    continue;
                }
                v.x = inb ? fmaxf(v.x, lo_clamp) : 0.f; v.y = inb ? fmaxf(v.y, lo_clamp) : 0.f;
                v.z = inb ? fmaxf(v.z, lo_clamp) : 0.f; v.w = inb ? fmaxf(v.w, lo_clamp) : 0.f;
                uint2 pl[NS];
                split4<NS>(v, pl);
#pragma unroll
                for (int p = 0; p < NS; ++p) *reinterpret_cast<uint2*>(buf + k * (XL * P16 * 16) + p * 32) = pl[p];
            }
        };
        if (SENAS_BF_SKIP(a, 0)) {
            for (int j = 0; j < items; ++j) __syncthreads();
            return;
        }
        if constexpr (NS == 1) {
            // one register set (28 pieces of 16 bytes): the loads of item j + 1 fly while the consumers work on item j - 1
            float4 pf[PF];
            unsigned ok;
            issue(0, pf, ok);
            commit(0, pf, ok);
            if (items > 1) issue(1, pf, ok);
            __syncthreads();                            // item 0 is staged
            for (int j = 1; j < items; ++j) {           // the consumers work on item j - 1
                commit(j, pf, ok);
                if (j + 1 < items) issue(j + 1, pf, ok);
                __syncthreads();
            }
        } else {
            // two register sets: an item's loads are issued TWO phases before they are stored -- on the small maps a phase
            // (75 .. 150 MFMAs) is shorter than an HBM round trip under load
            float4 pa[PF], pb[PF];
            unsigned oka, okb = 0;
            issue(0, pa, oka);
            if (items > 1) issue(1, pb, okb);
            commit(0, pa, oka);
            if (items > 2) issue(2, pa, oka);
            __syncthreads();                            // item 0 is staged
            for (int j = 1; j < items; j += 2) {
                commit(j, pb, okb);
                if (j + 2 < items) issue(j + 2, pb, okb);
                __syncthreads();
                if (j + 1 < items) {
                    commit(j + 1, pa, oka);
                    if (j + 3 < items) issue(j + 3, pa, oka);
                    __syncthreads();
                }
            }
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- consumers
    f32x16 acc[MT];
    uint4 bq[RING][U2];
    double s_sum = 0.0, q_sum = 0.0;
    int stat_n = -1, stat_cot = 0;
    const uint4* wbase = a.wimg + lane;
    auto wptr = [&](int cot, int pass, int t) { return wbase + ((size_t)(cot * TAPS + t) * npass + pass) * (U2 * 64); };
    auto flush_stats = [&]() {
        if (a.stats != nullptr && stat_n >= 0) {
            const double s = s_sum + __shfl_xor(s_sum, 32, 64), q = q_sum + __shfl_xor(q_sum, 32, 64);
            if (h == 0) {
                double* st = a.stats + ((size_t)stat_n * g.cout + stat_cot * 32 + r) * 2;
                atomicAdd(st, s);
                atomicAdd(st + 1, q);
            }
        }
        s_sum = q_sum = 0.0;
    };
    {   // weight fragments of the first AHEAD taps of the first item
        int n, cot, oy0, ox0;
        decode(t0, n, cot, oy0, ox0);
#pragma unroll
        for (int i = 0; i < AHEAD; ++i)
#pragma unroll
            for (int f = 0; f < U2; ++f) bq[i][f] = wptr(cot, 0, i < TAPS ? i : 0)[f * 64];
    }
    // LDS offset (16-byte units) of this lane's pixel for tap offset 0, sub-tile 0; sub-tile m is one window row further
    const int lbase = ((MT * wave) * tile_w + r) * P16 + h;
    const int lrow = tile_w * P16;
    __syncthreads();                                    // item 0's window is in buffer 0
    for (int j = 0; j < items; ++j) {
        int n, cot, oy0, ox0;
        decode(t0 + j / npass, n, cot, oy0, ox0);
        const int pass = j % npass;
        // where the two fragments past this item's last tap come from: the next item's taps 0 and 1
        int n2, cot2, oy2, ox2, pass2 = pass + 1;
        decode(t0 + (j + 1 < items ? j + 1 : j) / npass, n2, cot2, oy2, ox2);
        if (pass2 >= npass) pass2 = 0;
        if (j + 1 >= items) { cot2 = cot; pass2 = pass; }
        const uint4* lds4 = reinterpret_cast<const uint4*>(lds_raw + (size_t)(j & 1) * wbytes);
        if (pass == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;
        }
        auto tap_off = [&](int t) {
            const int ky = t / KSZ, kx = t - ky * KSZ;
            const int dy = (TG ? (KSZ - 1 - ky) : ky) * g.dil, dx = (TG ? (KSZ - 1 - kx) : kx) * g.dil;
            return (dy * tile_w + dx) * P16;
        };
        uint4 aq[2][MT][U2];
        {
            const int toff = tap_off(0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int f = 0; f < U2; ++f) aq[0][m][f] = lds4[lbase + m * lrow + toff + f * 2];
        }
        if (!SENAS_BF_SKIP(a, 1))
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            // weight fragments AHEAD taps ahead (the ring holds taps t .. t + AHEAD), LDS fragments one tap ahead; the
            // scheduling barriers keep the requests IN FRONT of this tap's MFMAs (left alone, the compiler sinks the loads
            // to their first use and every tap waits out a full L2 round trip)
            {
                const int tn = t + AHEAD;
                const uint4* wp = tn < TAPS ? wptr(cot, pass, tn) : wptr(cot2, pass2, tn - TAPS);
                if (!SENAS_BF_SKIP(a, 4))
#pragma unroll
                for (int f = 0; f < U2; ++f) bq[(t + AHEAD) % RING][f] = wp[f * 64];
            }
            if (t + 1 < TAPS && !SENAS_BF_SKIP(a, 3)) {
                const int toff = tap_off(t + 1);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int f = 0; f < U2; ++f) aq[(t + 1) & 1][m][f] = lds4[lbase + m * lrow + toff + f * 2];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (NS == 1) {
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = mfma_bf(aq[t & 1][m][f], bq[t % RING][f], acc[m]);
            } else {
#pragma unroll
                for (int i = 0; i < Prod<NS>::N; ++i)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = mfma_bf(aq[t & 1][m][Prod<NS>::a[i]], bq[t % RING][Prod<NS>::b[i]], acc[m]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the ring's look-ahead slots become taps 0 .. AHEAD - 1 of the next item
        {
            uint4 nx[AHEAD][U2];
#pragma unroll
            for (int i = 0; i < AHEAD; ++i)
#pragma unroll
                for (int f = 0; f < U2; ++f) nx[i][f] = bq[(TAPS + i) % RING][f];
#pragma unroll
            for (int i = 0; i < AHEAD; ++i)
#pragma unroll
                for (int f = 0; f < U2; ++f) bq[i][f] = nx[i][f];
        }
        if (pass == npass - 1 && !SENAS_BF_SKIP(a, 2)) {
            // ---- epilogue: lane = output channel co, register v = pixel (row MT*wave + m, column acc_row(v, h))
            if (a.stats != nullptr && (n != stat_n || cot != stat_cot)) {
                flush_stats();
                stat_n = n; stat_cot = cot;
            }
            const int co = cot * 32 + r;
            float* __restrict__ outp = a.out;
            const float* __restrict__ maskp = a.mask;
            const bool full = oy0 + TH <= g.hout && ox0 + 32 <= g.wout;          // block-uniform: no per-element bounds tests
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int oy = oy0 + MT * wave + m;
                const size_t row = (((size_t)n * g.hout + oy) * g.wout + ox0) * g.cout + co;
                if (full) {
                    if (maskp != nullptr) {                                    // all 16 mask values in flight, then the stores
                        float mk[16];
#pragma unroll
                        for (int v = 0; v < 16; ++v) mk[v] = maskp[row + (size_t)acc_row(v, h) * g.cout];
#pragma unroll
                        for (int v = 0; v < 16; ++v)
                            if (!(mk[v] > 0.f)) acc[m][v] = 0.f;
                    }
                    if constexpr (OB) {
                        __bf16* __restrict__ outh = reinterpret_cast<__bf16*>(a.out);
#pragma unroll
                        for (int v = 0; v < 16; ++v) outh[row + (size_t)acc_row(v, h) * g.cout] = (__bf16)acc[m][v];
                    } else {
#pragma unroll
                        for (int v = 0; v < 16; ++v) outp[row + (size_t)acc_row(v, h) * g.cout] = acc[m][v];
                    }
                    if (a.stats != nullptr) {               // 16 values in fp32, then into the fp64 running sums
                        float s16 = 0.f, q16 = 0.f;
#pragma unroll
                        for (int v = 0; v < 16; ++v) { s16 += acc[m][v]; q16 = fmaf(acc[m][v], acc[m][v], q16); }
                        s_sum += (double)s16;
                        q_sum += (double)q16;
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int ox = ox0 + acc_row(v, h);
                        float val = acc[m][v];
                        if (oy < g.hout && ox < g.wout) {
                            const size_t o = row + (size_t)acc_row(v, h) * g.cout;
                            if (maskp != nullptr && !(maskp[o] > 0.f)) val = 0.f;
                            if (a.stats != nullptr) { s_sum += val; q_sum += (double)val * val; }
                            if constexpr (OB) reinterpret_cast<__bf16*>(a.out)[o] = (__bf16)val;
                            else outp[o] = val;
                        }
                    }
                }
            }
        }
        // barrier j + 1 of the block: item j is consumed (its buffer may be refilled) and item j + 1 is staged; the producers
        // run exactly `items` barriers (one after each fill), the consumers one before item 0 and one after every item but the last
        if (j + 1 < items) __syncthreads();
    }
    flush_stats();
}

bool bf_gather_ok(const GatherGeom& g, int terms) {
    if (terms != 1 && terms != 3 && terms != 6) return false;
    if (g.stride != 1 || g.kh != g.kw || (g.kh != 3 && g.kh != 5)) return false;
    if (g.pad != g.dil * (g.kh / 2) || g.hout != g.hin || g.wout != g.win) return false;
    if (g.cin % (terms == 1 ? 32 : 16) != 0 || g.cout % 32 != 0) return false;
    if (g.wout < 32 || g.hout < 8 || g.pad > bf_max_halo(g.kh)) return false;
    return (long)g.n * g.hin * g.win * g.cin < 0x7fffffffL && (long)g.n * g.hout * g.wout * g.cout < 0x7fffffffL;
}

// rows per wave: 2 (8 x 32 tiles) while that still gives every CU two tiles, else 1; bf16x6 windows only fit with 1
static int bf_mt(const GatherGeom& g, int terms) {
    if (terms == 6) return 1;
    const long tiles8 = (long)((g.wout + 31) / 32) * ((g.hout + 7) / 8) * g.n * (g.cout / 32);
    return tiles8 >= 512 ? 2 : 1;            // (measured: 8-row tiles from two tiles per CU on, tools/bf_probe2.py)
}

template <bool TG, int MT, int NS, int KSZ, bool IB = false, bool OB = false>
static int launch_bf(const GatherGeom& g, const float* in, const void* wimg, float* out, int in_relu, const float* mask, double* stats,
                     hipStream_t st) {
    constexpr int TH = 4 * MT, P16 = NS == 3 ? 7 : 5;
    BfArgs a;
    a.g = g; a.in = in; a.wimg = reinterpret_cast<const uint4*>(wimg); a.out = out; a.mask = mask; a.stats = stats; a.in_relu = in_relu;
    a.tiles_x = (g.wout + 31) / 32;
    a.tiles_y = (g.hout + TH - 1) / TH;
    a.cot = g.cout / 32;
    a.ntiles = a.cot * g.n * a.tiles_y * a.tiles_x;
#ifdef SENAS_BF_PROBE
    static const int probe = getenv("SENAS_BF_PROBE") ? atoi(getenv("SENAS_BF_PROBE")) : 0;
    a.probe = probe;
#else
    a.probe = 0;
#endif
    constexpr int XL = NS == 1 ? 32 : 64;
    const size_t wbytes = (size_t)bf_pieces(MT, NS, KSZ) * XL * P16 * 16;
    const size_t bytes = 2 * wbytes;
    if (bytes > 160 * 1024) { set_error_msg("conv_bf: window does not fit in LDS"); return SENAS_EUNSUPPORTED; }
    if (bytes > 64 * 1024)
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(&conv_bf_kernel<TG, MT, NS, KSZ, IB, OB>), 160 * 1024,
                                     "conv_bf: raising the dynamic LDS limit")) return rc;
    int blocks = a.ntiles < 256 ? a.ntiles : 256;
    hipLaunchKernelGGL((conv_bf_kernel<TG, MT, NS, KSZ, IB, OB>), dim3(blocks), dim3(512), bytes, st, a);
    return launch_status("conv_bf");
}

// "bf16s": the forward pass writes a bf16 tensor (OB), the data gradient reads a bf16 gradient tensor (IB); plain bf16 products
int launch_bf_gather_stored(const GatherGeom& g, bool tg, const void* in, const void* wimg, void* out, int in_relu, const float* mask,
                            double* stats, hipStream_t st) {
    const int mt = bf_mt(g, 1);
    const float* inf = reinterpret_cast<const float*>(in);
    float* outf = reinterpret_cast<float*>(out);
    if (tg) {
        if (in_relu) { set_error_msg("conv_bf (bf16-stored gradient): no ReLU on load on this path"); return SENAS_EINVAL; }
        if (mt == 2) { if (g.kh == 3) return launch_bf<true, 2, 1, 3, true, false>(g, inf, wimg, outf, 0, mask, stats, st);
                       return launch_bf<true, 2, 1, 5, true, false>(g, inf, wimg, outf, 0, mask, stats, st); }
        if (g.kh == 3) return launch_bf<true, 1, 1, 3, true, false>(g, inf, wimg, outf, 0, mask, stats, st);
        return launch_bf<true, 1, 1, 5, true, false>(g, inf, wimg, outf, 0, mask, stats, st);
    }
    if (mt == 2) { if (g.kh == 3) return launch_bf<false, 2, 1, 3, false, true>(g, inf, wimg, outf, in_relu, mask, stats, st);
                   return launch_bf<false, 2, 1, 5, false, true>(g, inf, wimg, outf, in_relu, mask, stats, st); }
    if (g.kh == 3) return launch_bf<false, 1, 1, 3, false, true>(g, inf, wimg, outf, in_relu, mask, stats, st);
    return launch_bf<false, 1, 1, 5, false, true>(g, inf, wimg, outf, in_relu, mask, stats, st);
}

template <bool TG>
int launch_bf_gather(const GatherGeom& g, int terms, const float* in, const void* wimg, float* out, int in_relu, const float* mask,
                     double* stats, hipStream_t st) {
    const int mt = bf_mt(g, terms);
#define SENAS_BF(MT_, NS_)                                                                                        \
    do {                                                                                                          \
        if (g.kh == 3) return launch_bf<TG, MT_, NS_, 3>(g, in, wimg, out, in_relu, mask, stats, st);             \
        return launch_bf<TG, MT_, NS_, 5>(g, in, wimg, out, in_relu, mask, stats, st);                            \
    } while (0)
    if (terms == 1) { if (mt == 2) SENAS_BF(2, 1); SENAS_BF(1, 1); }
    if (terms == 3) { if (mt == 2) SENAS_BF(2, 2); SENAS_BF(1, 2); }
    SENAS_BF(1, 3);
#undef SENAS_BF
}

template int launch_bf_gather<false>(const GatherGeom&, int, const float*, const void*, float*, int, const float*, double*, hipStream_t);
template int launch_bf_gather<true>(const GatherGeom&, int, const float*, const void*, float*, int, const float*, double*, hipStream_t);

void bf_gather_name(const GatherGeom& g, int terms, bool tg, char* buf, int len) {
    snprintf(buf, len, "conv_bf_kernel<%s, %d, %d, %d>", tg ? "true" : "false", bf_mt(g, terms), terms == 1 ? 1 : (terms == 3 ? 2 : 3), g.kh);
}

// ---- weight image: [co tile][tap][reduction channels / 16][plane][lane 64] x 8 bf16; lane (r, h) element j is
// W[reduction channel 16*kc + 8*h + j][output channel 32*nt + r] of the tap, split like the activations
struct BfPackArgs {
    const float* src;
    uint4* dst;
    int d0, d1, taps, swap, ns;
    long frags;                 // nt * taps * (A / 16) * 64 lanes
};

__device__ __forceinline__ void bf_pack_one(const BfPackArgs& it, long i) {
    const int A = it.swap ? it.d1 : it.d0, B = it.swap ? it.d0 : it.d1;
    const int lane = (int)(i & 63);
    const long rest = i >> 6;
    const int kc = (int)(rest % (A / 16));
    const int t = (int)((rest / (A / 16)) % it.taps), nt = (int)(rest / ((long)(A / 16) * it.taps));
    const int r = lane & 31, h = lane >> 5;
    const int j = nt * 32 + r;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = kc * 16 + 8 * h + e;
        v[e] = 0.f;
        if (j < B) {
            const int s0 = it.swap ? j : ch, s1 = it.swap ? ch : j;
            v[e] = it.src[((size_t)s0 * it.d1 + s1) * it.taps + t];
        }
    }
    uint4* dst = it.dst + ((size_t)rest * it.ns) * 64 + lane;
    for (int p = 0; p < it.ns; ++p) {
        unsigned u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u[e] = pk_bf16(v[2 * e], v[2 * e + 1]);
            v[2 * e] -= bf_lo(u[e]);
            v[2 * e + 1] -= bf_hi(u[e]);
        }
        dst[(size_t)p * 64] = make_uint4(u[0], u[1], u[2], u[3]);
    }
}

__global__ void bf_pack_kernel(BfPackArgs it) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < it.frags) bf_pack_one(it, i);
}

int64_t bf_image_bytes(int A, int B, int taps, int terms) {
    const int ns = terms == 1 ? 1 : (terms == 3 ? 2 : 3);
    return (int64_t)((B + 31) / 32) * taps * (A / 16) * ns * 1024;
}

void launch_bf_pack(const float* w, void* img, int d0, int d1, int taps, int swap, int terms, hipStream_t st) {
    BfPackArgs it;
    it.src = w; it.dst = reinterpret_cast<uint4*>(img); it.d0 = d0; it.d1 = d1; it.taps = taps; it.swap = swap;
    it.ns = terms == 1 ? 1 : (terms == 3 ? 2 : 3);
    const int A = swap ? d1 : d0, B = swap ? d0 : d1;
    it.frags = (long)((B + 31) / 32) * taps * (A / 16) * 64;
    hipLaunchKernelGGL(bf_pack_kernel, dim3((unsigned)((it.frags + 255) / 256)), dim3(256), 0, st, it);
}

// batched form: items on blockIdx.y (senas_pack_item with the term count in bits 8.. of `swap`)
struct BfPackItem {
    const float* src;
    float* dst;
    int d0, d1, taps, swap;
    long elems;
};

__global__ void bf_pack_batched_kernel(const BfPackItem* __restrict__ items) {
    const BfPackItem raw = items[blockIdx.y];
    const int terms = raw.swap >> 8;
    if (terms == 0) return;                                     // an fp32 image: pack_weights_batched_kernel's item
    BfPackArgs it;
    it.src = raw.src; it.dst = reinterpret_cast<uint4*>(raw.dst); it.d0 = raw.d0; it.d1 = raw.d1; it.taps = raw.taps; it.swap = raw.swap & 1;
    it.ns = terms == 1 ? 1 : (terms == 3 ? 2 : 3);
    const int A = it.swap ? it.d1 : it.d0, B = it.swap ? it.d0 : it.d1;
    it.frags = (long)((B + 31) / 32) * it.taps * (A / 16) * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < it.frags; i += (long)gridDim.x * blockDim.x) bf_pack_one(it, i);
}

int launch_bf_pack_batched(const void* items_dev, int n, int64_t max_elems, hipStream_t st) {
    long blocks = (max_elems / 32 + 255) / 256;                 // one thread per fragment lane (8 weights x planes)
    if (blocks < 1) blocks = 1;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(bf_pack_batched_kernel, dim3((unsigned)blocks, n), dim3(256), 0, st, reinterpret_cast<const BfPackItem*>(items_dev));
    return launch_status("bf_pack_batched");
}

}  // namespace senas
