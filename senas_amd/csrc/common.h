// Shared helpers for the gfx950 kernels of libsenas_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/senas_hip.h"

namespace senas {

void set_error(const char* what, hipError_t e);
void set_error_msg(const char* what);

// Raise a kernel's dynamic-LDS limit above 64 KiB, once per (kernel, device) -- abi.hip
int raise_lds_limit(const void* kernel, int bytes, const char* what);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Every launcher ends with this: report (not swallow) launch failures.
inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(what, e); return SENAS_ELAUNCH; }
    return SENAS_OK;
}

#define SENAS_REQUIRE(cond, msg)                                     \
    do { if (!(cond)) { ::senas::set_error_msg(msg); return SENAS_EINVAL; } } while (0)

constexpr int kWave = 64;   // gfx950 wavefront

// Phase timestamps for kernel tuning: only in the debug library (`make phases` -> libsenas_hip_phases.so, used by
// tools/phase_probe.py); the shipped library compiles PHASE() to nothing.  100 MHz wall clock, block 0 / thread 0.
#ifdef SENAS_PHASES
static __device__ unsigned long long senas_phase_buf[64];          // one per translation unit
#define SENAS_PHASE_READER(name)                                                                          \
    extern "C" int senas_debug_read_phases_##name(unsigned long long* host64) {                           \
        if (hipDeviceSynchronize() != hipSuccess) return -1;                                              \
        return hipMemcpyFromSymbol(host64, HIP_SYMBOL(senas::senas_phase_buf), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1; \
    }
#define SENAS_PHASE(k)                                                                                   \
    do {                                                                                                 \
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)                   \
            senas_phase_buf[k] = wall_clock64();                                                         \
    } while (0)
#else
#define SENAS_PHASE(k) do { } while (0)
#define SENAS_PHASE_READER(name)
#endif

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    return v;
}

// ---- strided sums inside a 16-lane row with DPP (VALU speed; ds_bpermute-based __shfl_xor costs ~60 ns each here,
// which dominated every "100 partial sums x log2(lanes) steps" epilogue).  row_strided_sum(v, s): every lane ends up with
// the sum over the lanes of ITS 16-lane row that are congruent to it modulo s (s = 1, 2, 4, 8; s >= 16: unchanged).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffLL), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
template <typename T>
__device__ __forceinline__ T row_strided_sum(T v, int stride) {        // stride is wave-uniform
    if (stride <= 8) v += dpp_mov<0x128>(v);     // row_ror:8
    if (stride <= 4) v += dpp_mov<0x124>(v);     // row_ror:4
    if (stride <= 2) v += dpp_mov<0x122>(v);     // row_ror:2
    if (stride <= 1) v += dpp_mov<0x121>(v);     // row_ror:1
    return v;
}
// group_sum(v, q): sum over the aligned group of q consecutive lanes (q = 1, 2, 4, 8, 16), result in every lane of it
template <typename T>
__device__ __forceinline__ T group_sum(T v, int q) {                   // q is wave-uniform
    if (q >= 2) v += dpp_mov<0xB1>(v);           // quad_perm [1,0,3,2]
    if (q >= 4) v += dpp_mov<0x4E>(v);           // quad_perm [2,3,0,1]
    if (q >= 8) v += dpp_mov<0x141>(v);          // row_half_mirror
    if (q >= 16) v += dpp_mov<0x140>(v);         // row_mirror
    return v;
}
// After row_strided_sum, the lanes of a wave that still hold distinct partial sums of the same residue class are one per
// "slot": slot = lane / max(stride, 16); there are wave_slots(stride) of them (4 for stride <= 16, 2 for 32, 1 for 64).
__device__ __forceinline__ int wave_slots(int stride) { return stride <= 16 ? 4 : 64 / stride; }
__device__ __forceinline__ int lane_slot(int lane, int stride) { return lane / (stride < 16 ? 16 : stride); }
__device__ __forceinline__ bool lane_holds_partial(int lane, int stride) { return (lane & 15) < (stride < 16 ? stride : 16); }

// Producer-side batch-norm statistics for the "thread = 4 channels of one output pixel" kernels (pool, resample,
// depthwise).  A block owns `per_thread` consecutive chunks of 256 flat elements (same channel group in every chunk,
// because 256 % (c/4) == 0); each thread adds its values to fp64 registers with stats_accumulate4 and the block
// flushes ONCE with stats_flush4: shuffles over the pixel lanes of a wave, LDS over the 4 waves, one fp64 atomic
// pair per channel -- so a 256x256 map costs ~64 atomics per (image, channel), not 2048.
// `uniform` (block lies inside one image, c = 4*cv with cv a power of two <= 64) is decided on the host side of the
// launch and checked per block; otherwise the block keys its sums by (image, channel) in LDS (stats_flush4).
struct Stats4 {
    double s[4], q[4];
    int n;              // off the uniform layout: the image of this thread's (single) element, -1 = none
};
__device__ __forceinline__ void stats_init4(Stats4& a) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a.s[j] = a.q[j] = 0.0;
    a.n = -1;
}
__device__ __forceinline__ void stats_accumulate4(Stats4& a, double* __restrict__ stats, bool uniform, int n, int c, int ch,
                                                  const float (&v)[4], bool active) {
    if (stats == nullptr || !active) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a.s[j] += (double)v[j]; a.q[j] += (double)v[j] * (double)v[j]; }
    if (!uniform) a.n = n;                                          // (one chunk per block there: one image per thread)
}
// every thread of the block calls it once, after its last element (n: the block's image, ch: the thread's first channel)
__device__ __forceinline__ void stats_flush4(Stats4& a, double* __restrict__ stats, bool uniform, int n, int c, int ch) {
    if (stats == nullptr) return;                                   // block-uniform
    __shared__ double red_stats[2048];                              // [wave * slots + slot][channel][2]: 4 * slots * 2c <= 2048
    if (!uniform) {
        // maps so small that a block of 256 elements straddles images (or c / 4 is not a power of two): the block's sums
        // are keyed by (image, channel) in LDS and leave as one atomic pair per key -- per-element global atomics made
        // the 8x8 maps of the deepest cells the slowest depthwise launches of the search step (62 us for 2 blocks)
        __shared__ int n_lo, n_hi;
        if (threadIdx.x == 0) { n_lo = 0x7fffffff; n_hi = -1; }
        __syncthreads();
        if (a.n >= 0) { atomicMin(&n_lo, a.n); atomicMax(&n_hi, a.n); }
        __syncthreads();
        const int lo = n_lo, span = n_hi - lo + 1;                  // block-uniform; span <= 0: no active thread
        if (span <= 0) return;
        if ((long)span * 2 * c > 2048) {                            // (cannot happen with >= c/4 elements per image; kept safe)
            if (a.n >= 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double* st = stats + ((size_t)a.n * c + ch + j) * 2;
                    atomicAdd(st, a.s[j]);
                    atomicAdd(st + 1, a.q[j]);
                }
            }
            return;
        }
        for (int i = threadIdx.x; i < span * 2 * c; i += blockDim.x) red_stats[i] = 0.0;
        __syncthreads();
        if (a.n >= 0) {
            double* dst = red_stats + (size_t)(a.n - lo) * 2 * c;
#pragma unroll
            for (int j = 0; j < 4; ++j) { atomicAdd(dst + (ch + j) * 2, a.s[j]); atomicAdd(dst + (ch + j) * 2 + 1, a.q[j]); }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < span * 2 * c; i += blockDim.x) atomicAdd(stats + (size_t)lo * c * 2 + i, red_stats[i]);
        return;
    }
    const int cv = c >> 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a.s[j] = row_strided_sum(a.s[j], cv);
        a.q[j] = row_strided_sum(a.q[j], cv);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slots = wave_slots(cv);
    if (lane_holds_partial(lane, cv)) {                             // ch is this lane's first channel (threads are laid out c-fastest)
        double* dst = red_stats + (size_t)(wave * slots + lane_slot(lane, cv)) * 2 * c;
#pragma unroll
        for (int j = 0; j < 4; ++j) { dst[(ch + j) * 2] = a.s[j]; dst[(ch + j) * 2 + 1] = a.q[j]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * c; i += 256) {
        double t = 0.0;
        for (int w = 0; w < 4 * slots; ++w) t += red_stats[(size_t)w * 2 * c + i];
        atomicAdd(stats + (size_t)n * c * 2 + i, t);
    }
}
// host side: chunks of 256 elements per block such that a block never straddles two images; 0 = no uniform layout
inline int stats_chunks_per_block(long per_img_elems, int c, long total_elems) {
    const int cv = c >> 2;
    if (c % 4 != 0 || (cv & (cv - 1)) != 0 || cv > 64) return 0;
    for (int p = 8; p >= 1; p >>= 1)          // fat blocks only while >= 2048 of them remain: small maps stay latency-bound
        if (per_img_elems % (256L * p) == 0 && (p == 1 || total_elems / (256L * p) >= 2048)) return p;
    return 0;
}

// Geometry of one "gather" pass: out[n,oy,ox,:] reads in[n, f(oy,ky), f(ox,kx), :].
//   plain      : iy = oy*stride - pad + ky*dil                       (conv fwd, convT dgrad)
//   transposed : iy = (oy + pad - ky*dil)/stride when divisible      (convT fwd, conv dgrad)
struct GatherGeom {
    int n, hin, win, cin, hout, wout, cout;
    int kh, kw, stride, pad, dil;
    // Planar output groups (0: the usual interleaved NHWC).  A stacked convolution of a search cell (search/cell.py:100-106: the
    // same-named candidates of the edges that leave one state, weights stacked along c_out) is read back one 8-channel edge at a
    // time by DIFFERENT node kernels: interleaved, every such read takes 32 bytes of each 128-byte pixel and the rest of the line
    // is fetched again by the next node (counters: the node family moved 3x its algorithmic bytes).  With oplane != 0 channel ch
    // of pixel p goes to out[(ch / 8) * oplane + p * 8 + ch % 8]: every edge's slice is a dense [n][h][w][8] tensor.
    long oplane;
};

// offset of (pixel index over n*h*w, channel) in a forward output
__device__ __forceinline__ size_t out_offset(const GatherGeom& g, size_t pixel, int ch) {
    return g.oplane == 0 ? pixel * (size_t)g.cout + ch : (size_t)(ch >> 3) * (size_t)g.oplane + pixel * 8 + (ch & 7);
}

template <bool TG>
__device__ __forceinline__ bool tap_src(const GatherGeom& g, int o, int k, int lim, int& i) {
    if (!TG) {
        i = o * g.stride - g.pad + k * g.dil;
    } else {
        int t = o + g.pad - k * g.dil;
        if (t < 0) return false;
        if (g.stride == 2) { if (t & 1) return false; i = t >> 1; }
        else i = t;
    }
    return i >= 0 && i < lim;
}

// XCD-aware block coordinates of an (nx, ny) grid whose x walks a tensor in memory order and whose y counts problems that read
// the same tensor (1 for single-problem launches).  Workgroups are placed on the 8 XCDs round-robin in dispatch order, and each
// XCD has its own L2: with the plain order the rows a stencil shares with its neighbours -- and the input the problems share --
// are fetched by several XCDs (counters: 3x the input for a 3x3 stencil, 16x for six depthwise problems of 3x3 / 5x5 taps).
// Here XCD k takes the CONTIGUOUS eighth [k nx/8, (k+1) nx/8) of x, and walks it problem-fastest: the blocks that run together on
// an XCD read the same rows.  A bijection of the grid whenever nx is a multiple of 8; the plain coordinates otherwise.
struct VBlock { unsigned x, y; };
__device__ __forceinline__ VBlock xcd_block() {
    const unsigned nx = gridDim.x, ny = gridDim.y;
    if ((nx & 7u) != 0u) return VBlock{blockIdx.x, blockIdx.y};
    const unsigned lin = blockIdx.x + nx * blockIdx.y, k = lin & 7u, j = lin >> 3;
    return VBlock{k * (nx >> 3) + j / ny, j % ny};
}

// Weight-gradient geometry: G lives on the coarse grid (hg x wg, B channels), I on the fine grid
// (hi x wi, A channels); dW[b][a][tap] = sum_{n,p} I[n, p*stride - pad + k*dil][a] * G[n, p][b].
struct WgradGeom {
    int n, hg, wg, B, hi, wi, A, kh, kw, stride, pad, dil, chunk;
};

// A second problem riding in the same launch (search cell: dil_3_conv_5 and dil_2_conv_5 of the same edges, utils/operations.py:
// 69-72, differ in nothing but the dilation): grid.z is doubled, blocks with blockIdx.z >= nz work on problem 2.
struct Pair2 {
    const float* in;
    const float* w;           // packed image (conv_lds) / torch weights (conv_c8)
    float* out;
    const float* mask;
    double* stats;
    int dil, pad;
    int nz;                   // grid.z of ONE problem; 0: no second problem
};

// A second weight-gradient problem riding in the same launch (blockIdx.y == 1): the same X, shapes and kernel size; its own
// upstream gradient, partial images, dilation and padding (the two dilated 5x5 candidates of the same edges; two derived-cell
// candidates that read one state).
struct WPair2 {
    const float* G;
    float* part;
    int dil, pad;
    int on;                   // 0: no second problem
};

// conv_mfma.hip
bool mfma_gather_ok(const GatherGeom& g, bool tg);
template <bool TG>
int launch_mfma_gather(const GatherGeom& g, const float* in, const float* wp, float* out, int in_relu,
                       const float* mask, double* stats, hipStream_t st);
void launch_pack_mfma(const float* w, float* wp, int d0, int d1, int taps, int swap, hipStream_t st);
bool mfma_wgrad_ok(const WgradGeom& g);
int launch_mfma_wgrad(WgradGeom g, const float* I, const float* G, float* dw, float* ws, int i_relu, int g_relu,
                      int ws_is_zero, hipStream_t st);

// wgrad_lds.hip (stride-1 "same" weight gradient, persistent, both operands in LDS); part: per-block partial image
bool lds_wgrad_ok(const WgradGeom& g);
int64_t lds_wgrad_ws_bytes(const WgradGeom& g);
int launch_lds_wgrad(const WgradGeom& g, const float* X, const float* G, float* part, float* dw, int x_relu, senas_sum_item* defer,
                     hipStream_t st);
// two problems (g2: g with the second dilation / padding) in one launch; false: the two do not share a kernel and tile height
bool lds_wgrad_pair_ok(const WgradGeom& g, const WgradGeom& g2);
int launch_lds_wgrad_pair(const WgradGeom& g, const WgradGeom& g2, const float* X, const float* G, const float* G2, float* part, float* part2,
                          float* dw, float* dw2, int x_relu, senas_sum_item* defer, senas_sum_item* defer2, hipStream_t st);
void lds_wgrad_name(const WgradGeom& g, char* buf, int len);
void launch_unpack_wgrad(const float* ws, float* dw, int A, int B, int taps, hipStream_t st);

// Inference epilogue of a forward convolution (eval-mode batch-norm folded into the producer, the node sum and ReLU
// folded into the last producer):  y = act( scale[n][c] * acc + bias[n][c] + add_scale[n][c] * addend[n,p,c] )
struct Epi {
    const float* scale;       // [n][cout]
    const float* bias;        // [n][cout]
    const float* addend;      // NHWC like the output, or nullptr
    const float* add_scale;   // [n][cout], or nullptr (= 1)
    int relu;
};

// conv_lds.hip (stride-1 "same" convolutions with the input window staged in LDS)
bool lds_gather_ok(const GatherGeom& g);
int launch_lds_gather_epi(const GatherGeom& g, const float* in, const float* wp, float* out, int in_relu, const Epi& epi,
                          hipStream_t st);
void lds_gather_name(const GatherGeom& g, bool tg, char* buf, int len);
bool lds_gather_s2_ok(const GatherGeom& g);       // stride-2 plain gather (Conv2d forward, ConvTranspose2d data gradient)
int launch_lds_gather_s2(const GatherGeom& g, const float* in, const float* wp, float* out, int in_relu, const float* mask,
                         double* stats, hipStream_t st, const Pair2& pr = Pair2{});
template <bool TG>
int launch_lds_gather(const GatherGeom& g, const float* in, const float* wp, float* out, int in_relu,
                      const float* mask, double* stats, hipStream_t st, const Pair2& pr = Pair2{});

// conv_bf.hip (the same convolutions on the bf16 matrix pipe: operands split into 1, 2 or 3 bf16 planes, fp32 accumulation;
// terms = 1 plain bf16, 3 "bf16x3", 6 "bf16x6")
bool bf_gather_ok(const GatherGeom& g, int terms);
template <bool TG>
int launch_bf_gather(const GatherGeom& g, int terms, const float* in, const void* wimg, float* out, int in_relu, const float* mask,
                     double* stats, hipStream_t st);
void bf_gather_name(const GatherGeom& g, int terms, bool tg, char* buf, int len);
// "bf16s" (bf16-STORED convolution outputs and their gradients, plain bf16 products): forward x fp32 -> y bf16; data gradient
// dy bf16 -> dx fp32
int launch_bf_gather_stored(const GatherGeom& g, bool tg, const void* in, const void* wimg, void* out, int in_relu, const float* mask,
                            double* stats, hipStream_t st);
int64_t bf_image_bytes(int A, int B, int taps, int terms);
void launch_bf_pack(const float* w, void* img, int d0, int d1, int taps, int swap, int terms, hipStream_t st);
int launch_bf_pack_batched(const void* items_dev, int n, int64_t max_elems, hipStream_t st);

// wgrad_bf.hip (the weight gradient of the same convolutions on the bf16 pipe; partial images as wgrad_lds.hip)
bool bf_wgrad_ok(const WgradGeom& g, int terms);
int64_t bf_wgrad_ws_bytes(const WgradGeom& g, int terms);
int launch_bf_wgrad(const WgradGeom& g, int terms, const float* X, const float* G, float* part, float* dw, int x_relu, senas_sum_item* defer,
                    hipStream_t st);
int launch_bf_wgrad_stored(const WgradGeom& g, const float* X, const void* G, float* part, float* dw, int x_relu, senas_sum_item* defer,
                           hipStream_t st);
void bf_wgrad_name(const WgradGeom& g, int terms, char* buf, int len);

// conv_stem.hip (stem forward: 1..4 input channels, (tap, channel) on the K axis of the fp32 MFMA)
bool stem_mfma_ok(const GatherGeom& g);
int launch_stem_mfma(const GatherGeom& g, const float* in, const float* w, float* out, int in_relu, double* stats, hipStream_t st);
// ... and its weight gradient (x window + dy tile in LDS, per-block partial rows in the torch layout)
bool stem_wgrad_ok(const WgradGeom& g);
int64_t stem_wgrad_ws_bytes(const WgradGeom& g);
int launch_stem_wgrad(const WgradGeom& g, const float* X, const float* G, float* part, int x_relu, int* nblk_out, hipStream_t st);

// conv_thin.hip (one side of the GEMM view has <= 4 channels: HBM-bound single-pass kernels; weights in torch layout)
bool thin_k_ok(const GatherGeom& g);
bool thin_k3_ok(const GatherGeom& g);      // ... and its straight-line 3x3 form (2 or 4 input channels)
bool thin_k4_ok(const GatherGeom& g);     // stride-1 plain gather, 4 output columns per thread
int launch_thin_k4(const GatherGeom& g, const float* in, const float* w, int d1, int swap, int flip, float* out, int in_relu,
                   double* stats, hipStream_t st);
template <bool TG>
int launch_thin_k(const GatherGeom& g, const float* in, const float* w, int d1, int swap, float* out, int in_relu,
                  const float* mask, double* stats, hipStream_t st);
// conv_t2.hip (transposed gather at stride 2 with the input window in LDS: ConvTranspose2d forward, stride-2 Conv2d data gradient)
bool t2_lds_ok(const GatherGeom& g);
int launch_t2_lds(const GatherGeom& g, const float* in, const float* wp, float* out, double* stats, hipStream_t st);
// conv_c8.hip (the 8-channel inner-edge convolutions of the search cell on 16 x 16 x 4 fp32 MFMA tiles)
bool c8_mfma_ok(const GatherGeom& g);
int c8_mfma_tiles_per_wave(const GatherGeom& g);
int launch_c8_mfma(const GatherGeom& g, const float* in, const float* w, int d1, int swap, int flip, float* out, double* stats,
                   hipStream_t st, const Pair2& pr = Pair2{});
bool c8_mfma_wgrad_ok(const WgradGeom& g);
int64_t c8_mfma_wgrad_ws_bytes(const WgradGeom& g);
int launch_c8_mfma_wgrad(const WgradGeom& g, const float* X, const float* G, float* part, int* nblk_out, hipStream_t st,
                         const WPair2& second = WPair2{});
bool thin_n_ok(const GatherGeom& g);
bool thin_n3_ok(const GatherGeom& g);      // ... and its straight-line 3x3 form (at most 4 output channels)
template <bool TG>
int launch_thin_n(const GatherGeom& g, const float* in, const float* w, int d1, int swap, float* out, int in_relu,
                  double* stats, hipStream_t st);
bool thin_n_wgrad_ok(const WgradGeom& g);
int64_t thin_n_wgrad_ws_bytes(const WgradGeom& g);
int launch_thin_n_wgrad(WgradGeom g, const float* I, const float* G, float* part, int i_relu, int g_relu, int* nblk_out,
                        hipStream_t st);

bool wgrad_c8_ok(const WgradGeom& g);
int64_t wgrad_c8_ws_bytes(const WgradGeom& g);
int launch_wgrad_c8(WgradGeom g, const float* I, const float* G, float* part, int i_relu, int g_relu, int* nblk_out, hipStream_t st);

// V consecutive floats (V == 4: one 16-byte access; the caller guarantees 16-byte alignment)
template <int V>
__device__ __forceinline__ void ldv(const float* __restrict__ p, float (&v)[V]) {
    if constexpr (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = p[j];
    }
}
template <int V>
__device__ __forceinline__ void stv(float* __restrict__ p, const float (&v)[V]) {
    if constexpr (V == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int j = 0; j < V; ++j) p[j] = v[j];
    }
}

}  // namespace senas
