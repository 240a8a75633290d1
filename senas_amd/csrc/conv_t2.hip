// Transposed gather at stride 2 with the input window in LDS: ConvTranspose2d(stride 2) forward (the UP candidates of a
// search / derived cell: se_conv_3, dil_3_conv_5, dil_2_conv_5, utils/operations.py:58-60,118-130) and the data gradient of a
// stride-2 Conv2d (the DOWN candidates).  out[oy][ox] = sum over the taps with (oy + pad - ky*d) and (ox + pad - kx*d) even of
// in[(oy + pad - ky*d) / 2][(ox + pad - kx*d) / 2] * W[tap].
//
// The direct-global form (conv_mfma_kernel<true, ...>) re-reads every input pixel once per tap as 16-byte fragments and is
// L1/TA-bound (18 TFLOP/s at 4 x 32 x 64 x 64 -> 128 x 128).  Here the four output phases (oy & 1, ox & 1) of a tile of
// input-grid positions share ONE staged window (halo <= 3 pixels): each phase is a small stride-1 convolution over it with
// its own subset of the taps (dilation 3: 9 + 6 + 6 + 4 of the 25; dilation 2: all 25 in phase (0, 0), the other three
// phases are zero).  Fragment conventions are conv_lds.hip's: 16-channel passes, pixel stride 20 floats, weights from the
// packed image [co-tile][tap][ci/8][2][32][4].
//
// Block = 4 waves on a tile of 2 x 16 input-grid positions (one 32-row MFMA tile, 4 x 32 output pixels), the whole c_in
// window staged once (pixel stride c_in + 4 floats).  Odd dilation: one wave per phase (rotated with the block index so
// that the 9-tap phase does not always land on the same SIMD).  Even dilation: the four waves deal phase (0, 0)'s taps
// among themselves, meet in LDS, and the other three phases are stored as zeros.  One accumulator tile per wave keeps the
// kernel under 96 registers, so several blocks share a CU and hide each other's load latency.
#include "common.h"

namespace senas {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int CH = 16;
constexpr int TQW = 16, TQH = 2;

__device__ __forceinline__ f32x16 mfma32t(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row_t(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

// STEP = 1: odd dilation, wave = phase; STEP = 4: even dilation, wave = every fourth tap of phase (0, 0)
template <int STEP>
__global__ __launch_bounds__(256) void conv_t2_lds_kernel(GatherGeom g, const float* __restrict__ in, const float* __restrict__ wp,
                                                          float* __restrict__ out, double* __restrict__ stats, int hq) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int tap_list[4][25];                                   // per wave: (tap index << 16) | window offset in 16-byte pieces
    __shared__ double red[4 * 2 * 32 * 2];                            // [wave][h][channel][sum, sum of squares]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int co_tiles = (g.cout + 31) / 32;
    const int n = blockIdx.z / co_tiles, cot = blockIdx.z - n * co_tiles;
    const int qy0 = blockIdx.y * TQH, qx0 = blockIdx.x * TQW;
    const int WW = TQW + 2 * hq, WH = TQH + 2 * hq;
    const int taps = g.kh * g.kw, ngroups = g.cin >> 3, npass = g.cin / CH;
    const int P4 = (g.cin + 4) >> 2;                                  // window pixel stride in 16-byte pieces
    const int phase = STEP == 1 ? ((wave + blockIdx.x + blockIdx.y) & 3) : 0;   // the phase this wave computes
    const int py = phase >> 1, px = phase & 1;

    // this wave's taps: lane t tests tap t, the valid ones are compacted in order
    int cnt;
    {
        const int ky = lane / g.kw, kx = lane - ky * g.kw;
        const int ty = py + g.pad - ky * g.dil, tx = px + g.pad - kx * g.dil;
        const bool valid = lane < taps && !(ty & 1) && !(tx & 1);
        const unsigned long long m = __ballot(valid);
        // position q reads input (q + ty / 2, q + tx / 2): window pixel (local q + hq + ty / 2, ...)   (ty may be negative: >> is floor)
        if (valid) tap_list[wave][__popcll(m & ((1ull << lane) - 1ull))] = (lane << 16) | (((hq + (ty >> 1)) * WW + hq + (tx >> 1)) * P4);
        cnt = __popcll(m);
    }

    float4* lds4 = reinterpret_cast<float4*>(lds);
    {
        const float* src = in + (size_t)n * g.hin * g.win * g.cin;
        const int pieces = g.cin >> 2, total = WH * WW * pieces;
        for (int i = threadIdx.x; i < total; i += 256) {
            const int px_ = i / pieces, sq = i - px_ * pieces;
            const int wy = px_ / WW, wx = px_ - wy * WW;
            const int iy = qy0 - hq + wy, ix = qx0 - hq + wx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win)
                v = *reinterpret_cast<const float4*>(src + ((size_t)iy * g.win + ix) * g.cin + sq * 4);
            lds4[px_ * P4 + sq] = v;
        }
    }
    __syncthreads();

    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    wp += (size_t)cot * taps * ngroups * 256 + lane * 4;
    const int lbase = ((r / TQW) * WW + (r % TQW)) * P4 + h;          // this lane's position, tap offset (0, 0) = window origin

    const int first = STEP == 1 ? 0 : wave;
    if (first < cnt) {
        // two alternating fragment sets over the flattened (tap, 16-channel pass) steps: the next step's loads are in
        // flight while the current one's MFMAs issue
        float4 a0[2], b0[2], a1[2], b1[2];
        auto load = [&](int i, int pass, float4 (&a)[2], float4 (&b)[2]) {
            const int e = tap_list[wave][i];
            const int t = e >> 16, off = (e & 0xffff) + pass * 4;
            const float* wt = wp + ((size_t)t * ngroups + pass * 2) * 256;
            b[0] = *reinterpret_cast<const float4*>(wt);
            b[1] = *reinterpret_cast<const float4*>(wt + 256);
            a[0] = lds4[lbase + off];
            a[1] = lds4[lbase + off + 2];
        };
        auto mac = [&](const float4 (&a)[2], const float4 (&b)[2]) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                acc = mfma32t(a[c2].x, b[c2].x, acc);
                acc = mfma32t(a[c2].y, b[c2].y, acc);
                acc = mfma32t(a[c2].z, b[c2].z, acc);
                acc = mfma32t(a[c2].w, b[c2].w, acc);
            }
        };
        int i = first, pass = 0;
        auto advance = [&]() {                                        // wave-uniform
            if (++pass == npass) { pass = 0; i += STEP; }
            return i < cnt;
        };
        load(i, pass, a0, b0);
        bool more = advance();
        while (true) {
            if (more) load(i, pass, a1, b1);
            mac(a0, b0);
            if (!more) break;
            more = advance();
            if (more) load(i, pass, a0, b0);
            mac(a1, b1);
            if (!more) break;
            more = advance();
        }
    }

    if (STEP != 1) {                                                  // the four tap groups meet in wave 0; waves 1..3 store the zero phases
        __syncthreads();                                              // the window is dead
        if (wave > 0) {
#pragma unroll
            for (int v = 0; v < 16; ++v) lds[((wave - 1) * 16 + v) * 64 + lane] = acc[v];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll 1
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[v] += lds[(k * 16 + v) * 64 + lane];
        } else {
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        }
    }
    // epilogue: lane = output channel, register v = position acc_row(v, h) of the 2 x 16 positions
    const int sp = STEP == 1 ? phase : wave;                          // the phase this wave stores
    const int spy = sp >> 1, spx = sp & 1;
    const int co = cot * 32 + r;
    const bool cok = co < g.cout;
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int i = acc_row_t(v, h);
        const int qy = qy0 + i / TQW, qx = qx0 + i % TQW;
        if (cok && qy < g.hin && qx < g.win) {
            out[out_offset(g, ((size_t)n * g.hout + 2 * qy + spy) * g.wout + 2 * qx + spx, co)] = acc[v];
            s += (double)acc[v];
            q += (double)acc[v] * (double)acc[v];
        }
    }
    if (stats == nullptr) return;                                     // block-uniform
    red[((wave * 2 + h) * 32 + r) * 2] = s;
    red[((wave * 2 + h) * 32 + r) * 2 + 1] = q;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int ch = threadIdx.x >> 1, which = threadIdx.x & 1;
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) tot += red[(k * 32 + ch) * 2 + which];
        if (cot * 32 + ch < g.cout) atomicAdd(stats + ((size_t)n * g.cout + cot * 32 + ch) * 2 + which, tot);
    }
}

size_t t2_lds_bytes(const GatherGeom& g) {
    const int hq = (g.pad + 1) / 2;
    const size_t window = (size_t)(TQH + 2 * hq) * (TQW + 2 * hq) * (g.cin + 4) * sizeof(float), fold = (size_t)3 * 16 * 64 * sizeof(float);
    return window > fold ? window : fold;
}

}  // namespace

// transposed gather, stride 2, output exactly twice the input, 3x3 / 5x5 with "same" padding, c_in in 16-channel passes
bool t2_lds_ok(const GatherGeom& g) {
    if (g.stride != 2 || g.kh != g.kw || (g.kh != 3 && g.kh != 5) || g.dil < 1 || g.dil > 3 || g.pad != g.dil * (g.kh / 2)) return false;
    if (g.hout != 2 * g.hin || g.wout != 2 * g.win || g.cin % CH != 0 || g.cin < CH || g.cout < 1) return false;
    if (t2_lds_bytes(g) > 60 * 1024) return false;
    return g.n >= 1 && (long)g.n * ((g.cout + 31) / 32) <= 65535 && (g.hin + 1) / 2 <= 65535 && (long)g.n * g.hout * g.wout * g.cout < 0x7fffffffL;
}

int launch_t2_lds(const GatherGeom& g, const float* in, const float* wp, float* out, double* stats, hipStream_t st) {
    const int hq = (g.pad + 1) / 2;
    const int co_tiles = (g.cout + 31) / 32;
    dim3 grid((g.win + TQW - 1) / TQW, (g.hin + TQH - 1) / TQH, g.n * co_tiles);
    if (g.dil & 1) hipLaunchKernelGGL(conv_t2_lds_kernel<1>, grid, dim3(256), t2_lds_bytes(g), st, g, in, wp, out, stats, hq);
    else hipLaunchKernelGGL(conv_t2_lds_kernel<4>, grid, dim3(256), t2_lds_bytes(g), st, g, in, wp, out, stats, hq);
    return launch_status("conv_t2_lds");
}

}  // namespace senas
