// Hardware self-test of the MFMA fragment maps every other kernel relies on (sed_common.h).
// C[32][32] = A[32][K] * B[K][32] via v_mfma_f32_32x32x2_f32, and the 16x16x4 analogue.
#include "sed_common.h"

int sed_tuning[SED_TUNE_COUNT] = {0};
SED_API int sed_set_tuning(int key, int value) {
    if (key < 0 || key >= SED_TUNE_COUNT) return SED_ERR_ARG;
    sed_tuning[key] = value;
    return SED_OK;
}

__global__ __launch_bounds__(64) void selftest_mfma32_kernel(const float* A, const float* Bm, float* C, int K) {
    const int lane = threadIdx.x;
    f32x16 acc = f32x16_zero();
    for (int k = 0; k < K; k += 2) {
        const float a = A[(lane & 31) * K + k + (lane >> 5)];
        const float b = Bm[(k + (lane >> 5)) * 32 + (lane & 31)];
        acc = mfma32(a, b, acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) C[mfma32_row(r, lane) * 32 + (lane & 31)] = acc[r];
}
__global__ __launch_bounds__(64) void selftest_mfma16_kernel(const float* A, const float* Bm, float* C, int K) {
    const int lane = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 4) {
        const float a = A[(lane & 15) * K + k + (lane >> 4)];
        const float b = Bm[(k + (lane >> 4)) * 16 + (lane & 15)];
        acc = mfma16(a, b, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) C[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];
}
// bf16 32x32x16: C[32][32] = A[32][K] * B[K][32] with operands rounded to bf16 (K multiple of 16)
__global__ __launch_bounds__(64) void selftest_mfma32_bf16_kernel(const float* A, const float* Bm, float* C, int K) {
    const int lane = threadIdx.x;
    f32x16 acc = f32x16_zero();
    for (int k = 0; k < K; k += 16) {
        s16x8 a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = k + 8 * (lane >> 5) + e;
            a[e] = (short)f32_to_bf16(A[(lane & 31) * K + kk]);
            b[e] = (short)f32_to_bf16(Bm[kk * 32 + (lane & 31)]);
        }
        acc = mfma32_bf16(a, b, acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) C[mfma32_row(r, lane) * 32 + (lane & 31)] = acc[r];
}
// bf16 16x16x32: C[16][16] = A[16][K] * B[K][16] with operands rounded to bf16 (K multiple of 32)
__global__ __launch_bounds__(64) void selftest_mfma16_bf16_kernel(const float* A, const float* Bm, float* C, int K) {
    const int lane = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 32) {
        s16x8 a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = k + 8 * (lane >> 4) + e;
            a[e] = (short)f32_to_bf16(A[(lane & 15) * K + kk]);
            b[e] = (short)f32_to_bf16(Bm[kk * 16 + (lane & 15)]);
        }
        acc = mfma16_bf16(a, b, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) C[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];
}
SED_API int sed_selftest_mfma(const float* A, const float* Bm, float* C, int K, int shape, void* stream) {
    if (shape == 32) SED_LAUNCH(selftest_mfma32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bm, C, K);
    else if (shape == 3216) SED_LAUNCH(selftest_mfma32_bf16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bm, C, K);
    else if (shape == 1632) SED_LAUNCH(selftest_mfma16_bf16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bm, C, K);
    else if (shape == 16) SED_LAUNCH(selftest_mfma16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bm, C, K);
    else return SED_ERR_ARG;
    return sed_check_launch();
}
