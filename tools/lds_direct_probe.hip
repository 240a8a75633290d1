// Does `global_load_lds_dwordx4` (gfx950: 16 bytes per lane, global -> LDS without a register in between) put lane i's piece at
// M0 base + 16 i?  hipcc --offload-arch=gfx950 -O3 tools/lds_direct_probe.hip -o tools/_build/lds_direct_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* __restrict__ in, float* __restrict__ out, int masked) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6;
    for (int j = 0; j < 4; ++j) lds[threadIdx.x * 4 + j] = -1.f;
    __syncthreads();
    const float* src = in + (size_t)((threadIdx.x * 7) % 256) * 4;
    if (!masked || (threadIdx.x & 3) != 1)        // lanes switched off leave their slot alone
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = lds[threadIdx.x * 4 + j];
}
int main() {
    std::vector<float> h(1024), o(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *din, *dout;
    hipMalloc(&din, 4096); hipMalloc(&dout, 4096);
    hipMemcpy(din, h.data(), 4096, hipMemcpyHostToDevice);
    int bad = 0;
    for (int masked = 0; masked < 2; ++masked) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, din, dout, masked);
        hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
        for (int t = 0; t < 256; ++t)
            for (int j = 0; j < 4; ++j) {
                const float want = (masked && (t & 3) == 1) ? -1.f : h[((t * 7) % 256) * 4 + j];
                if (o[t * 4 + j] != want) { if (bad < 8) printf("masked %d thread %d piece %d: got %g want %g\n", masked, t, j, o[t * 4 + j], want); ++bad; }
            }
    }
    printf(bad ? "MISMATCHES: %d\n" : "global_load_lds_dwordx4: lane i -> M0 + 16 i, masked lanes untouched (OK)\n", bad);
    return bad != 0;
}
