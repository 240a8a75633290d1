// clip_grad_norm_ + SGD(momentum, weight decay) over ALL parameter tensors in two launches
// (experiments/train_model.py:284-289, experiments/search_arc.py:280-285 do this with one torch kernel per
// tensor list chunk: ~110 launches per step on the 807 tensors of the derived network).
//   pass 1: sum of squares of every gradient   -> one fp64 partial per block (plain stores), folded in a FIXED order by
//           one block (no atomics: the clip coefficient is bit-identical on every data-parallel rank)
//   pass 2: coef = min(1, max_norm / (sqrt(total) + 1e-6));  g *= coef (written back, as clip_grad_norm_ does);
//           d = g + wd * p;  buf = first ? d : momentum * buf + (1 - dampening) * d;
//           p -= lr * (nesterov ? d + momentum * buf : buf)
#include "common.h"

namespace senas {

struct SgdItem {                 // == senas_sgd_item
    float* param;
    float* grad;
    float* buf;
    int64_t numel;
    int64_t first;               // != 0: buf is uninitialised -- this step sets it to the gradient (torch's first step)
};

__global__ __launch_bounds__(256) void sgd_sqnorm_kernel(const SgdItem* __restrict__ items, double* __restrict__ partial) {
    __shared__ double red[4];
    const SgdItem it = items[blockIdx.y];
    const int64_t base = (int64_t)blockIdx.x * 1024;
    if (base >= it.numel) return;                                   // block-uniform (its slot stays zero)
    double s = 0.0;
    if (it.grad != nullptr) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = base + k * 256 + threadIdx.x;
            if (i < it.numel) { const float g = it.grad[i]; s += (double)g * (double)g; }
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[1 + (size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// partial[0] = sum of partial[1 .. count] in an order that depends on nothing but count: 1024 threads, four independent
// strided chains each (the loads of a chain do not wait for each other), then a fixed tree
__global__ __launch_bounds__(1024) void sgd_sqnorm_fold_kernel(double* __restrict__ partial, long count) {
    __shared__ double red[1024];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    long i = threadIdx.x;
    for (; i + 3072 < count; i += 4096) {
        s0 += partial[1 + i]; s1 += partial[1 + i + 1024]; s2 += partial[1 + i + 2048]; s3 += partial[1 + i + 3072];
    }
    for (; i < count; i += 1024) s0 += partial[1 + i];
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[0] = red[0];
}

__global__ __launch_bounds__(256) void sgd_step_kernel(const SgdItem* __restrict__ items, const double* __restrict__ partial,
                                                       float max_norm, float lr, float momentum, float dampening,
                                                       float weight_decay, int nesterov, int first, float* __restrict__ norm_out) {
    const SgdItem it = items[blockIdx.y];
    const int64_t base = (int64_t)blockIdx.x * 1024;
    if (base >= it.numel) return;
    float coef = 1.f;
    if (max_norm > 0.f) {
        const float total_norm = (float)sqrt(partial[0]);
        coef = fminf(max_norm / (total_norm + 1e-6f), 1.f);
        if (norm_out != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *norm_out = total_norm;
    }
    if (it.grad == nullptr) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i >= it.numel) continue;
        const float g = it.grad[i] * coef;
        it.grad[i] = g;
        float p = it.param[i];
        float d = weight_decay != 0.f ? fmaf(weight_decay, p, g) : g;
        if (momentum != 0.f) {
            const float b = (first || it.first) ? d : fmaf(momentum, it.buf[i], (1.f - dampening) * d);
            it.buf[i] = b;
            d = nesterov ? fmaf(momentum, b, d) : b;
        }
        it.param[i] = fmaf(-lr, d, p);
    }
}

}  // namespace senas

static_assert(sizeof(senas_sgd_item) == sizeof(senas::SgdItem), "senas_sgd_item layout");

extern "C" int senas_sgd_clip_step(const senas_sgd_item* items_dev, int n, int64_t max_numel, double* partial64,
                                   float max_norm, float lr, float momentum, float dampening, float weight_decay,
                                   int nesterov, int first_step, float* total_norm_out, void* stream) {
    SENAS_REQUIRE(items_dev && n > 0 && n <= 65535 && max_numel > 0, "sgd_clip_step: bad argument");
    SENAS_REQUIRE(max_norm <= 0.f || partial64, "sgd_clip_step: clipping needs the partial-sum buffer");
    hipStream_t st = senas::as_stream(stream);
    dim3 grid((unsigned)((max_numel + 1023) / 1024), n);
    const senas::SgdItem* items = reinterpret_cast<const senas::SgdItem*>(items_dev);
    if (max_norm > 0.f) {
        const long count = (long)grid.x * n;
        hipError_t e = hipMemsetAsync(partial64, 0, (size_t)(1 + count) * sizeof(double), st);
        if (e != hipSuccess) { senas::set_error("sgd_clip_step: memset", e); return SENAS_ELAUNCH; }
        hipLaunchKernelGGL(senas::sgd_sqnorm_kernel, grid, dim3(256), 0, st, items, partial64);
        hipLaunchKernelGGL(senas::sgd_sqnorm_fold_kernel, dim3(1), dim3(1024), 0, st, partial64, count);
    }
    hipLaunchKernelGGL(senas::sgd_step_kernel, grid, dim3(256), 0, st, items, partial64, max_norm, lr, momentum, dampening,
                       weight_decay, nesterov, first_step, total_norm_out);
    return senas::launch_status("sgd_clip_step");
}
