// TEST INFRASTRUCTURE ONLY -- a single-threaded fiber emulator of the HIP execution model.
//
// The build container has no GPU and the round has 90 GPU-minutes, so the kernels under
// desed_task_amd/csrc/*.hip are also compiled as plain C++ against this header
// (-DSED_EMU, see tests/emu/build_emu.py) to check their index arithmetic, LDS staging,
// barrier structure and MFMA fragment maps on the CPU.  One fiber per GPU thread;
// __syncthreads(), wave shuffles and MFMA are rendezvous points.  The MFMA emulation follows
// the lane->element maps of /opt/skills/guides/cdna_hip_programming.md section 3 and computes a
// k-ordered fmaf chain (bitwise what v_mfma_f32_32x32x2_f32 does).
//
// Nothing under desed_task_amd/ includes this file; the product library is built by hipcc
// for gfx950 only (desed_task_amd/build.py).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
template <typename F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return 0; }

struct EmuFiber {
    dim3 tid;
    int lin;          // linear thread id in block
    void* sp;         // saved stack pointer
    char* stack;
    int state;        // 0 runnable, 1 at block barrier, 2 at wave sync, 3 done
};
extern thread_local EmuFiber* emu_cur;            // workgroups run on a pool of OS threads: per-thread block state
extern thread_local dim3 emu_blockIdx;
extern dim3 emu_blockDim, emu_gridDim;
extern thread_local char* emu_dyn_smem;
extern thread_local float emu_wave_xchg[16][64][8];   // [wave][lane][slot] exchange area for collectives

void emu_launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
void emu_block_barrier();
void emu_wave_sync();

#define threadIdx (emu_cur->tid)
#define blockIdx emu_blockIdx
#define blockDim emu_blockDim
#define gridDim emu_gridDim
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __syncthreads() emu_block_barrier()
#define __threadfence() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define SED_DYN_SMEM(name) char* name = emu_dyn_smem
#define SED_LAUNCH(kern, grid, block, smem, stream, ...) \
    emu_launch(grid, block, smem, [=]() { kern(__VA_ARGS__); })

static inline int emu_lane() { return emu_cur->lin & 63; }
static inline int emu_wave() { return emu_cur->lin >> 6; }

template <typename T>
static inline T emu_xchg(T v, int src_lane) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- and 64-bit shuffles only (HIP moves a double as two dwords)");
    float f[2] = {0.f, 0.f}; memcpy(f, &v, sizeof(T));
    T out;
    float r[2];
    for (int w = 0; w < (int)(sizeof(T) / 4); ++w) {
        emu_wave_xchg[emu_wave()][emu_lane()][0] = f[w];
        emu_wave_sync();
        r[w] = emu_wave_xchg[emu_wave()][src_lane & 63][0];
        emu_wave_sync();
    }
    memcpy(&out, r, sizeof(T));
    return out;
}
template <typename T> static inline T __shfl_xor(T v, int mask, int = 64) { return emu_xchg(v, emu_lane() ^ mask); }
template <typename T> static inline T __shfl_down(T v, int d, int = 64) { int l = emu_lane() + d; return emu_xchg(v, l > 63 ? emu_lane() : l); }
template <typename T> static inline T __shfl(T v, int src, int = 64) { return emu_xchg(v, src); }

template <typename T, typename U>
static inline T emu_atomic_fadd(T* p, T v) {      // CAS loop on the bit pattern (workgroups of one launch run concurrently)
    U* u = (U*)p;
    U o = __atomic_load_n(u, __ATOMIC_RELAXED);
    for (;;) {
        T f; memcpy(&f, &o, sizeof(T));
        f += v;
        U n; memcpy(&n, &f, sizeof(T));
        if (__atomic_compare_exchange_n(u, &o, n, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { T r; memcpy(&r, &o, sizeof(T)); return r; }
    }
}
static inline float atomicAdd(float* p, float v) { return emu_atomic_fadd<float, unsigned>(p, v); }
static inline double atomicAdd(double* p, double v) { return emu_atomic_fadd<double, unsigned long long>(p, v); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline int atomicMax(int* p, int v) { int o = __atomic_load_n(p, __ATOMIC_RELAXED); while (v > o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
static inline int atomicMin(int* p, int v) { int o = __atomic_load_n(p, __ATOMIC_RELAXED); while (v < o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }

static inline unsigned __brev(unsigned x) { unsigned r = 0; for (int i = 0; i < 32; ++i) { r = (r << 1) | (x & 1u); x >>= 1; } return r; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
template <typename T> static inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> static inline T max(T a, T b) { return a > b ? a : b; }

// ---- MFMA emulation ---------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));      // 8 bf16 bit patterns

static inline float emu_bf16_to_f32(short h) { unsigned u = ((unsigned)(unsigned short)h) << 16; float f; memcpy(&f, &u, 4); return f; }

// v_mfma_f32_32x32x16_bf16: A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31], e=0..7; D as the f32 32x32 map
static inline f32x16 emu_mfma_32x32x16_bf16(s16x8 a, s16x8 b, f32x16 c) {
    int w = emu_wave(), l = emu_lane();
    memcpy(&emu_wave_xchg[w][l][0], &a, 16);
    memcpy(&emu_wave_xchg[w][l][4], &b, 16);
    emu_wave_sync();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int kh = 0; kh < 2; ++kh) {
            s16x8 av, bv;
            memcpy(&av, &emu_wave_xchg[w][row + 32 * kh][0], 16);
            memcpy(&bv, &emu_wave_xchg[w][col + 32 * kh][4], 16);
            for (int e = 0; e < 8; ++e) acc += emu_bf16_to_f32(av[e]) * emu_bf16_to_f32(bv[e]);
        }
        c[r] = acc;
    }
    emu_wave_sync();
    return c;
}

// v_mfma_f32_16x16x32_bf16: A[i=l&15][k=8*(l>>4)+e], B[k=8*(l>>4)+e][j=l&15], e=0..7; D: col=l&15, row=4*(l>>4)+r
static inline f32x4 emu_mfma_16x16x32_bf16(s16x8 a, s16x8 b, f32x4 c) {
    int w = emu_wave(), l = emu_lane();
    memcpy(&emu_wave_xchg[w][l][0], &a, 16);
    memcpy(&emu_wave_xchg[w][l][4], &b, 16);
    emu_wave_sync();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int kg = 0; kg < 4; ++kg) {
            s16x8 av, bv;
            memcpy(&av, &emu_wave_xchg[w][row + 16 * kg][0], 16);
            memcpy(&bv, &emu_wave_xchg[w][col + 16 * kg][4], 16);
            for (int e = 0; e < 8; ++e) acc += emu_bf16_to_f32(av[e]) * emu_bf16_to_f32(bv[e]);
        }
        c[r] = acc;
    }
    emu_wave_sync();
    return c;
}

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
static inline f32x16 emu_mfma_32x32x2(float a, float b, f32x16 c) {
    int w = emu_wave(), l = emu_lane();
    emu_wave_xchg[w][l][0] = a;
    emu_wave_xchg[w][l][1] = b;
    emu_wave_sync();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k)
            acc = fmaf(emu_wave_xchg[w][row + 32 * k][0], emu_wave_xchg[w][col + 32 * k][1], acc);
        c[r] = acc;
    }
    emu_wave_sync();
    return c;
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15, row=(l>>4)*4+r
static inline f32x4 emu_mfma_16x16x4(float a, float b, f32x4 c) {
    int w = emu_wave(), l = emu_lane();
    emu_wave_xchg[w][l][0] = a;
    emu_wave_xchg[w][l][1] = b;
    emu_wave_sync();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k)
            acc = fmaf(emu_wave_xchg[w][row + 16 * k][0], emu_wave_xchg[w][col + 16 * k][1], acc);
        c[r] = acc;
    }
    emu_wave_sync();
    return c;
}
