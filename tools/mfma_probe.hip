// What the chip sustains on bare f32 MFMA loops (operands in registers, 2 waves per SIMD, every CU busy): FLOP/s by wall
// clock and the in-kernel shader clock (s_memtime / s_memrealtime), for both f32 shapes, on random and on zero operands.
// Tuning aid (MI355X_MICROARCH.md, "DVFS give-back"): tells a kernel that is short of MFMA issue slots from one that
// is at the power wall.     hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/_build/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// 32x32x2 with NV integer VALU instructions (v_mad_u32_u24-class, independent of the MFMAs) per 16 MFMAs, and optionally
// LDS: 4 ds_read_b128 per 16 MFMAs whose results feed the next round's A operands
// 32x32x2 with NG global_load_dwordx4 (L2-resident, 1 KiB per wave-instruction) per 16 MFMAs; the data feeds the next round
template <int NG>
__global__ __launch_bounds__(256) void probe_vmem(const float* __restrict__ src, float* __restrict__ dst, int iters, unsigned long long* stamps) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = src[(tid * 16 + i) & 0xfffff]; b[i] = src[(tid * 16 + 8 + i) & 0xfffff]; }
    unsigned long long c0 = 0, r0 = 0;
    if (threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    f32x16 acc[2];
    for (int m = 0; m < 2; ++m) for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;
    const float4* s4 = reinterpret_cast<const float4*>(src) + (threadIdx.x & 63);
    float4 g[NG > 0 ? NG : 1];
    for (int it = 0; it < iters; ++it) {
        const float4* p = s4 + ((it & 63) << 6) * 4;                  // 64 KiB-wide slice of the buffer, shared by all waves: L2 hits
#pragma unroll
        for (int k = 0; k < NG; ++k) g[k] = p[k * 64];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[i], a[i], acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NG; ++k) { a[k & 7] = g[k].x; b[k & 7] = g[k].w; }
    }
    float sum = 0.f;
    for (int m = 0; m < 2; ++m) for (int v = 0; v < 16; ++v) sum += acc[m][v];
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - c0;
        stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    dst[tid] = sum;
}

template <int NV, int LDSR>
__global__ __launch_bounds__(256) void probe_mix(const float* __restrict__ src, float* __restrict__ dst, int iters, unsigned long long* stamps) {
    __shared__ float4 sh[1024];
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = src[(tid * 16 + i) & 0xfffff]; b[i] = src[(tid * 16 + 8 + i) & 0xfffff]; }
    for (int i = threadIdx.x; i < 1024; i += 256) sh[i] = make_float4(a[0], a[1], a[2], a[3]);
    __syncthreads();
    unsigned long long c0 = 0, r0 = 0;
    if (threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    f32x16 acc[2];
    for (int m = 0; m < 2; ++m) for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;
    unsigned x0 = tid, x1 = tid * 3 + 1, x2 = tid ^ 0x55, x3 = tid + 7;
    for (int it = 0; it < iters; ++it) {
        float4 l0, l1, l2, l3;
        if (LDSR) {
            const int o = (threadIdx.x + it) & 255;
            l0 = sh[o]; l1 = sh[o + 256]; l2 = sh[o + 512]; l3 = sh[o + 768];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[i], a[i], acc[1], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV / 8; ++k) {
                x0 = x0 * 5u + x1; x1 = x1 * 3u + x2; x2 = x2 * 7u + x3; x3 = x3 * 9u + x0;
            }
        }
        if (LDSR) { a[0] = l0.x; a[1] = l1.y; a[2] = l2.z; a[3] = l3.w; }
    }
    float sum = (float)(x0 + x1 + x2 + x3);
    for (int m = 0; m < 2; ++m) for (int v = 0; v < 16; ++v) sum += acc[m][v];
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - c0;
        stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    dst[tid] = sum;
}

template <int SHAPE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, float* __restrict__ dst, int iters, unsigned long long* stamps) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = src[(tid * 16 + i) & 0xfffff]; b[i] = src[(tid * 16 + 8 + i) & 0xfffff]; }
    unsigned long long c0 = 0, r0 = 0;
    if (threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    float sum = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[2];
        for (int m = 0; m < 2; ++m) for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[i], a[i], acc[1], 0, 0, 0);
            }
        }
        for (int m = 0; m < 2; ++m) for (int v = 0; v < 16; ++v) sum += acc[m][v];
    } else {
        f32x4 acc[8];
        for (int m = 0; m < 8; ++m) for (int v = 0; v < 4; ++v) acc[m][v] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {               // 4 x (16x16x4) = the FLOPs of 2 x (32x32x2)
                acc[(2 * i) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc[(2 * i) & 7], 0, 0, 0);
                acc[(2 * i + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i], a[i], acc[(2 * i + 1) & 7], 0, 0, 0);
                acc[(2 * i + 4) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], a[i], acc[(2 * i + 4) & 7], 0, 0, 0);
                acc[(2 * i + 5) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i], b[i], acc[(2 * i + 5) & 7], 0, 0, 0);
            }
        }
        for (int m = 0; m < 8; ++m) for (int v = 0; v < 4; ++v) sum += acc[m][v];
    }
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - c0;
        stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    dst[tid] = sum;
}

template <typename KERNEL>
static void run_k(KERNEL kern, const char* name, const float* src, float* dst, unsigned long long* stamps, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, src, dst, iters, stamps);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, src, dst, iters, stamps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 2);
    hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double clk = 0.0;
    for (int i = 0; i < blocks; ++i) clk += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;      // MHz
    const double flops = (double)reps * blocks * 4 * iters * 16 * 4096.0;
    printf("%-58s %8.3f ms/launch  %7.1f TFLOP/s  in-kernel clock %6.0f MHz\n", name, ms / reps, flops / (ms * 1e-3) / 1e12, clk / blocks);
}

template <int SHAPE>
static void run(const char* name, const float* src, float* dst, unsigned long long* stamps, int blocks, int iters) {
    run_k(probe<SHAPE>, name, src, dst, stamps, blocks, iters);
}

int main() {
    const int blocks = 512, iters = 20000;
    const size_t nsrc = 1 << 20;
    std::vector<float> h(nsrc);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *src, *zero, *dst;
    unsigned long long* stamps;
    hipMalloc(&src, nsrc * 4); hipMalloc(&zero, nsrc * 4); hipMalloc(&dst, blocks * 256 * 4); hipMalloc(&stamps, blocks * 16);
    hipMemcpy(src, h.data(), nsrc * 4, hipMemcpyHostToDevice);
    hipMemset(zero, 0, nsrc * 4);
    for (int round = 0; round < 1; ++round) {
        run<32>("32x32x2  f32, random operands", src, dst, stamps, blocks, iters);
        run<16>("16x16x4  f32, random operands", src, dst, stamps, blocks, iters);
        run<32>("32x32x2  f32, zero operands", zero, dst, stamps, blocks, iters);
        run<16>("16x16x4  f32, zero operands", zero, dst, stamps, blocks, iters);
    }
    run<32>("32x32x2 random, ONE wave per SIMD (256 blocks)", src, dst, stamps, 256, iters);
    run<16>("16x16x4 random, ONE wave per SIMD (256 blocks)", src, dst, stamps, 256, iters);
    run_k(probe_mix<0, 0>, "32x32x2 random, 2 waves/SIMD, mix kernel, no extras", src, dst, stamps, blocks, iters);
    run_k(probe_mix<16, 0>, "32x32x2 random, 2 waves/SIMD, +16 int VALU per 16 MFMA", src, dst, stamps, blocks, iters);
    run_k(probe_mix<32, 0>, "32x32x2 random, 2 waves/SIMD, +32 int VALU per 16 MFMA", src, dst, stamps, blocks, iters);
    run_k(probe_mix<64, 0>, "32x32x2 random, 2 waves/SIMD, +64 int VALU per 16 MFMA", src, dst, stamps, blocks, iters);
    run_k(probe_mix<0, 1>, "32x32x2 random, 2 waves/SIMD, +4 ds_read_b128 per 16 MFMA", src, dst, stamps, blocks, iters);
    run_k(probe_mix<32, 1>, "32x32x2 random, 2 waves/SIMD, +32 VALU +4 ds_read_b128", src, dst, stamps, blocks, iters);
    run_k(probe_mix<32, 1>, "32x32x2 random, 1 wave/SIMD,  +32 VALU +4 ds_read_b128", src, dst, stamps, 256, iters);
    run_k(probe_vmem<0>, "32x32x2 random, 1 wave/SIMD, vmem kernel, no loads", src, dst, stamps, 256, iters);
    run_k(probe_vmem<3>, "32x32x2 random, 1 wave/SIMD, +3 global_load_dwordx4 per 16 MFMA", src, dst, stamps, 256, iters);
    run_k(probe_vmem<6>, "32x32x2 random, 1 wave/SIMD, +6 global_load_dwordx4 per 16 MFMA", src, dst, stamps, 256, iters);
    run_k(probe_vmem<3>, "32x32x2 random, 2 waves/SIMD, +3 global_load_dwordx4 per 16 MFMA", src, dst, stamps, 512, iters);
    run_k(probe_mix<0, 1>, "32x32x2 random, 1 wave/SIMD, +4 ds_read_b128 per 16 MFMA", src, dst, stamps, 256, iters);
    {   // short launches: what the clock governor does with 0.25 ms dispatches
        run_k(probe<32>, "32x32x2 random, 1 wave/SIMD, 0.25 ms launches", src, dst, stamps, 256, 580);
    }
    return 0;
}
