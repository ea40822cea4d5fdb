// DIAGNOSTICS ONLY -- not part of libsed_hip.so.  Which packed-fp32 instruction forms return wrong results when MFMA waves of another
// kernel are resident on the same CU?  (profiles/r06_mel_mechanism.md: the multi-frame mel kernel's wrong bins are
// v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0] delivering src0.lo + 0 in lanes 48-63.)  Every form is executed on values that two plain
// packed subtractions produced just before (as in the mel kernel's DFT-4), re-computed with scalar VALU instructions from the same
// registers and compared bitwise; mismatches are counted per form, 16-lane quarter and half.
//   build:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/mel_repro/pk_probe.hip -o tools/_pkprobe.so ; run: tools/mel_repro/pk_probe.py
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define NFORMS 18
#define CHK(F, r, elo, ehi)                                                                                   \
    {                                                                                                         \
        const bool bl = __float_as_uint(r.x) != __float_as_uint(elo), bh = __float_as_uint(r.y) != __float_as_uint(ehi); \
        if (bl) atomicAdd(&counts[((F) * 4 + (lane >> 4)) * 2 + 0], 1u);                                      \
        if (bh) atomicAdd(&counts[((F) * 4 + (lane >> 4)) * 2 + 1], 1u);                                      \
        if ((bl || bh) && (F) == 0) {                                                                         \
            const unsigned i = atomicAdd(&counts[NFORMS * 8], 1u);                                            \
            if (i < 64) { unsigned* e = counts + NFORMS * 8 + 8 + 8 * i; e[0] = __float_as_uint(a.x); e[1] = __float_as_uint(a.y); \
                e[2] = __float_as_uint(b.x); e[3] = __float_as_uint(b.y); e[4] = __float_as_uint(r.x); e[5] = __float_as_uint(r.y); e[6] = lane; e[7] = it; } \
        }                                                                                                     \
    }
__device__ __forceinline__ float sadd(float x, float y) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float ssub(float x, float y) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float smul(float x, float y) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float sfma(float x, float y, float z) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z)); return r; }

extern "C" __global__ __launch_bounds__(256, 2) void pk_probe_kernel(const float* __restrict__ in, unsigned* __restrict__ counts, int iters, int n) {
    __shared__ float pad[PROBE_LDS_FLOATS];
    const int gid = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    if (iters < 0) pad[threadIdx.x] = 1.f;
    f32x2 v0, v1, v2, v3;
    v0.x = in[gid]; v0.y = in[gid + n]; v1.x = in[gid + 2 * n]; v1.y = in[gid + 3 * n];
    v2.x = in[gid + 4 * n]; v2.y = in[gid + 5 * n]; v3.x = in[gid + 6 * n]; v3.y = in[gid + 7 * n];
    for (int it = 0; it < iters; ++it) {
        f32x2 a, b, c, r;
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(a) : "v"(v0), "v"(v2));      // a = v0 - v2
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(b) : "v"(v1), "v"(v3));      // b = v1 - v3
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(c) : "v"(v0), "v"(v3));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        CHK(1, r, sadd(a.x, b.y), ssub(a.y, b.x));
        f32x2 keep = r;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        CHK(0, r, ssub(a.x, b.y), sadd(a.y, b.x));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        CHK(2, r, sadd(a.x, b.x), sadd(a.y, b.y));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
        CHK(3, r, sadd(a.x, b.y), sadd(a.y, b.x));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        CHK(4, r, sadd(a.y, b.x), sadd(a.x, b.y));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
        CHK(5, r, smul(a.y, b.y), smul(a.y, b.x));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        CHK(6, r, sfma(a.x, b.x, -c.x), sfma(a.x, b.y, c.y));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        CHK(7, r, sfma(a.y, b.x, c.x), sfma(a.y, b.y, c.y));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        CHK(8, r, sfma(a.x, b.y, c.x), sfma(a.y, b.y, c.y));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
        CHK(9, r, sadd(a.x, b.x), sadd(a.y, b.x));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        CHK(10, r, sadd(a.x, b.y), sadd(a.y, b.y));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        CHK(11, r, smul(a.x, b.x), smul(a.y, b.y));
        // context variants of the SAME instruction (pk_add op_sel:[0,1] op_sel_hi:[1,0]):
        asm volatile("s_nop 7\n s_nop 7\n v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n s_nop 7\n s_nop 7" : "=v"(r) : "v"(a), "v"(b));   // idle before and after
        CHK(12, r, sadd(a.x, b.y), sadd(a.y, b.x));
        asm volatile("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));                    // same sum, swapped operand in src0
        CHK(13, r, sadd(b.y, a.x), sadd(b.x, a.y));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %0, %0, %0" : "=&v"(r) : "v"(a), "v"(b));  // consumed at once by a packed op
        { const float lo = sadd(a.x, b.y), hi = sadd(a.y, b.x); CHK(14, r, sadd(lo, lo), sadd(hi, hi)); }
        asm volatile("s_setprio 3\n v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n s_setprio 0" : "=v"(r) : "v"(a), "v"(b));
        CHK(15, r, sadd(a.x, b.y), sadd(a.y, b.x));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));                        // lo lane takes src2.hi
        CHK(16, r, sfma(a.x, b.x, c.y), sfma(a.y, b.y, c.y));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));                                      // mul, lo lane takes src1.hi
        CHK(17, r, smul(a.x, b.y), smul(a.y, b.y));
        // next inputs: a bounded rotation of what we have
        v0.x = 0.6f * keep.x + 0.3f; v0.y = 0.6f * keep.y - 0.2f;
        v1.x = 0.5f * a.y + 0.1f * v1.x; v1.y = 0.5f * b.x - 0.1f * v1.y;
        v2.x = 0.7f * c.y - 0.4f; v2.y = 0.7f * c.x + 0.5f;
        v3.x = 0.4f * b.y + 0.3f * v3.y; v3.y = 0.4f * a.x - 0.3f * v3.x;
    }
    if (v0.x == 12345.678f) counts[NFORMS * 8 + 1] = 1;      // keep the chain alive
}
extern "C" __attribute__((visibility("default"))) int pk_probe(const float* in, unsigned* counts, int iters, int n, int grid, void* stream) {
    hipLaunchKernelGGL(pk_probe_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, counts, iters, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
