// K10 + K11: parameter-arena kernels.  The student (and teacher) parameters live in ONE flat fp32 buffer
// (desed_task_amd/arena.py), so the EMA teacher update (sed_trainer.py:187-199) and Adam
// (torch.optim.Adam defaults, train_sed.py:199-201) are single streaming launches over 1,112,420 floats
// (float4 per lane) instead of 62 / 124 tiny per-tensor ops.  HBM-bound: 12 B/param (EMA), 28 B/param (Adam).
#include "sed_common.h"

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ teacher, const float* __restrict__ student, size_t n4,
                                                  size_t n, float alpha, float one_minus_alpha,
                                                  const float* __restrict__ alpha_dev) {
    if (alpha_dev) { alpha = alpha_dev[0]; one_minus_alpha = alpha_dev[1]; }   // device-resident (hipGraph replays)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        float4 t = ((float4*)teacher)[i];
        const float4 s = ((const float4*)student)[i];
        t.x = t.x * alpha + one_minus_alpha * s.x; t.y = t.y * alpha + one_minus_alpha * s.y;
        t.z = t.z * alpha + one_minus_alpha * s.z; t.w = t.w * alpha + one_minus_alpha * s.w;
        ((float4*)teacher)[i] = t;
    }
    if (i == 0) for (size_t j = n4 * 4; j < n; ++j) teacher[j] = teacher[j] * alpha + one_minus_alpha * student[j];
}
// teacher <- alpha * teacher + (1 - alpha) * student over n floats (16-byte aligned buffers)
SED_API int sed_ema_update(float* teacher, const float* student, long long n, float alpha, float one_minus_alpha,
                              const float* alpha_dev, void* stream) {
    if (n <= 0) return SED_OK;
    const size_t n4 = (size_t)n / 4;
    const int grid = (int)((n4 + 255) / 256) + (n4 == 0 ? 1 : 0);
    SED_LAUNCH(ema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, teacher, student, n4, (size_t)n, alpha, one_minus_alpha,
               alpha_dev);
    return sed_check_launch();
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float b1, float b2, float eps,
                                                   float step_size, float inv_bc2_sqrt, float grad_scale,
                                                   const float* __restrict__ hyper_dev) {
    if (hyper_dev) { step_size = hyper_dev[0]; inv_bc2_sqrt = hyper_dev[1]; }   // device-resident (hipGraph replays)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i] * grad_scale;
        const float mi = m[i] * b1 + (1.0f - b1) * gi;
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}
// torch.optim.Adam (no weight decay, no amsgrad): step_size = lr / (1 - b1^t), inv_bc2_sqrt = 1 / sqrt(1 - b2^t).
// grad_scale folds the data-parallel 1/world_size averaging into the update.
SED_API int sed_adam_step(float* p, const float* g, float* m, float* v, long long n, float b1, float b2, float eps,
                             float step_size, float inv_bc2_sqrt, float grad_scale, const float* hyper_dev, void* stream) {
    if (n <= 0) return SED_OK;
    int grid = (int)(((size_t)n + 255) / 256);
    if (grid > 2048) grid = 2048;
    SED_LAUNCH(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (size_t)n, b1, b2, eps, step_size,
               inv_bc2_sqrt, grad_scale, hyper_dev);
    return sed_check_launch();
}

// Zero up to four (small) accumulator buffers in one launch; null / 0 entries are skipped.
SED_API int sed_zero_buffers(float* p0, long long n0, float* p1, long long n1, float* p2, long long n2, float* p3, long long n3,
                                void* stream) {
    sed_zero4((hipStream_t)stream, p0, (int)n0, p1, (int)n1, p2, (int)n2, p3, (int)n3);
    return sed_check_launch();
}
