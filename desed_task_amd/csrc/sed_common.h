// Common device helpers for the MI355X (gfx950 / CDNA4) SED kernels.
// Wave = 64 lanes.  All kernels are launched through SED_LAUNCH on the caller's stream and
// never allocate: the host (PyTorch) owns every buffer.
#pragma once

#ifdef SED_EMU
#include "hip_emu.h"   // tests/emu: CPU fiber emulator, test infrastructure only
#else
#include <hip/hip_runtime.h>
#define SED_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define SED_LAUNCH(kern, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kern, grid, block, smem, stream, __VA_ARGS__)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));      // 8 bf16 bit patterns (one MFMA operand fragment)
#endif

#include <stdint.h>

// The C-ABI entry points (include/sed_hip.h) are the ONLY symbols libsed_hip.so exports: the library is compiled with
// -fvisibility=hidden (kernel host stubs, template instantiations and helpers stay internal), the entry points opt back in.
#define SED_API extern "C" __attribute__((visibility("default")))

#define SED_MAX_SMEM(kern, bytes) \
    (void)hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))

// zero up to four small buffers in ONE launch (gradient accumulators that the kernels fill with atomics)
__global__ __launch_bounds__(256) static void sed_zero4_kernel(float* p0, int n0, float* p1, int n1, float* p2, int n2, float* p3, int n3) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n0) p0[i] = 0.f;
    if (i < n1) p1[i] = 0.f;
    if (i < n2) p2[i] = 0.f;
    if (i < n3) p3[i] = 0.f;
}
static inline void sed_zero4(hipStream_t s, float* p0, int n0, float* p1, int n1, float* p2, int n2, float* p3, int n3) {
    int n = n0 > n1 ? n0 : n1;
    n = n > n2 ? n : n2;
    n = n > n3 ? n : n3;
    if (n <= 0) return;
    SED_LAUNCH(sed_zero4_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p0, n0, p1, n1, p2, n2, p3, n3);
}

#define SED_OK 0
#define SED_ERR_ARG (-1)
#define SED_ERR_LAUNCH (-2)
#define SED_ERR_UNSUPPORTED (-3)

// Tuning overrides (tests and sweep tools only; all 0 = built-in choices).  Set explicitly through sed_set_tuning(): the entry
// points never read the process environment.
enum { SED_TUNE_GLU_GRID_CAP = 0, SED_TUNE_GLU_BWD128_SPLIT = 1, SED_TUNE_CONVB_CK = 2, SED_TUNE_CONVB_MP = 3, SED_TUNE_B0_NOCENTER = 4, SED_TUNE_GLU_FWD128 = 5, SED_TUNE_WGRAD_NARROW = 6, SED_TUNE_WGRAD_CAP = 7,
       SED_TUNE_ATTN_VALU = 8, SED_TUNE_WGRAD_WIDE = 9, SED_TUNE_GRU_LDS_KB = 10, SED_TUNE_MEL_TAPS_MEM = 11, SED_TUNE_CONVB_TPW = 12, SED_TUNE_MEL_WAVE = 13, SED_TUNE_GEMM_NTN = 14, SED_TUNE_LINEAR_TILES = 15, SED_TUNE_COUNT = 16 };
extern int sed_tuning[SED_TUNE_COUNT];

static inline int sed_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SED_OK : SED_ERR_LAUNCH;
}

// ---- MFMA wrappers (exact f32: a k-ordered fmaf chain, guide section 3) ---------------------
// 32x32x2: lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31];
//          acc[r] is D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
#ifdef SED_EMU
    return emu_mfma_32x32x2(a, b, c);
#else
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// 16x16x4: lane l supplies A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; acc[r] is D[row=(l>>4)*4+r][col=l&15].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
#ifdef SED_EMU
    return emu_mfma_16x16x4(a, b, c);
#else
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

// bf16 MFMA 32x32x16 (16x the f32 rate): lane l supplies A[i=l&31][k=8*(l>>5)+e] and B[k=8*(l>>5)+e][j=l&31], e=0..7;
// D uses the same map as mfma32.  Used for the split-bf16 ("bf16x3") paths: x = hi + lo with hi = bf16(x),
// lo = bf16(x - hi); a*b ~ hi*hi + hi*lo + lo*hi keeps ~16 mantissa bits (rel. error ~8e-6) at 3/16 of the f32 MFMA cost.
__device__ __forceinline__ f32x16 mfma32_bf16(s16x8 a, s16x8 b, f32x16 c) {
#ifdef SED_EMU
    return emu_mfma_32x32x16_bf16(a, b, c);
#else
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#endif
}
// bf16 MFMA 16x16x32 (same rate per flop): lane l supplies A[i=l&15][k=8*(l>>4)+e] and B[k=8*(l>>4)+e][j=l&15], e=0..7;
// acc[r] is D[row=4*(l>>4)+r][col=l&15] (the 16x16x4 f32 map).  A wave can own a 16-column slice of a wide output,
// which halves the per-wave weight-fragment registers of the 128-channel GLU backward.
__device__ __forceinline__ f32x4 mfma16_bf16(s16x8 a, s16x8 b, f32x4 c) {
#ifdef SED_EMU
    return emu_mfma_16x16x32_bf16(a, b, c);
#else
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#endif
}
// round-to-nearest-even fp32 -> bf16 bit pattern (finite inputs)
__device__ __forceinline__ unsigned short f32_to_bf16(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
// x -> (hi, lo) bf16 pair with x ~ hi + lo
__device__ __forceinline__ void bf16_split(float x, unsigned short& hi, unsigned short& lo) {
    hi = f32_to_bf16(x);
    lo = f32_to_bf16(x - bf16_to_f32(hi));
}

// ---- gfx950 hazard: packed-fp32 VALU with OP_SEL on src1, beside v_mfma_f32_32x32x16_bf16 waves (found in round 6) ---------------------
// A v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 whose LOW lane takes the HIGH dword of src1 (op_sel = [0,1] / [0,1,0]) intermittently
// computes lanes 48-63 of its low result with that operand read as 0 while waves of ANOTHER kernel that issues the gfx950 double-K bf16
// MFMA (v_mfma_f32_32x32x16_bf16: gemm_bf16x3_kernel, the split-bf16 convolutions) are resident on the same CU: 0.5 - 2.5 % of the
// executions under a saturating co-runner, transient (the same instruction on the same registers is right the next time), in eager
// launches as well as in hipGraph replays, not cured by s_nop / s_setprio / dependent or independent neighbours.  NOT affected: the
// same selection on src0 (op_sel = [1,0]) or src2, op_sel = [1,1], op_sel_hi on any source, no op_sel; no fault beside an fp32-MFMA
// GEMM, a rocBLAS GEMM or a VALU / LDS kernel.  (tools/mel_repro/pk_probe.{hip,py}: 1.3e8 executions per form; profiles/r06_mel_mechanism.md.
// It is what round 5 saw as "wrong bins in single frames of the wave-per-frame mel kernel inside graph replays".)
// RULE: no kernel of this library contains such an instruction -- hand-written packed arithmetic puts the operand whose halves are
// swapped into src0 (the adds and products commute: same bits), compiler-generated pairs are broken up where the compiler picked the
// form (sed_sadd, pinned scalars); tests/test_isa_audit.py::test_no_packed_f32_op_sel_on_src1 scans the ISA of every kernel.
// scalar fp32 add that the compiler cannot merge into a packed instruction
__device__ __forceinline__ float sed_sadd(float a, float b) {
#ifdef SED_EMU
    return a + b;
#else
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}

// scalar fp32 FMA that the compiler cannot merge into a packed instruction (same note)
__device__ __forceinline__ float sed_sfma(float a, float b, float c) {
#ifdef SED_EMU
    return fmaf(a, b, c);
#else
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#endif
}

// compiler scheduling fence: nothing moves across it (bounds the live ranges of hoisted LDS reads in fully unrolled MFMA chains)
__device__ __forceinline__ void sed_sched_fence() {
#ifndef SED_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// ({hi, lo} >> shift)[31:0], shift in 0..31 (v_alignbit_b32): moves bf16 elements across the dwords of an MFMA operand
__device__ __forceinline__ unsigned sed_alignbit(unsigned hi, unsigned lo, unsigned shift) {
#ifdef SED_EMU
    return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> shift);
#else
    return __builtin_amdgcn_alignbit(hi, lo, shift);
#endif
}

// Redefines a lane-dependent int opaquely: whatever is derived from it afterwards cannot be hoisted out of the enclosing loop
// (in fully unrolled tile loops LICM otherwise precomputes dozens of per-element addresses and they end up in scratch).
// "Pin" a value that was loaded from memory before a loop: the empty asm consumes and redefines the register, so the compiler's
// s_waitcnt for the load is placed HERE and the register is no longer "the result of a pending load".  Without it the waitcnt
// pass keeps (conservatively, through the loop's back edge) a vmcnt wait in front of the first use inside the loop -- free while
// nothing else is in flight, but it also drains whatever the loop itself issued meanwhile: in the BiGRU recurrences that was
// the next chunk's prefetch and the previous chunk's result stores, once per chunk, on the dependent chain.
template <class T_>
__device__ __forceinline__ void sed_pin(T_& x) {
#ifndef SED_EMU
    asm volatile("" : "+v"(x));
#else
    (void)x;
#endif
}
__device__ __forceinline__ void sed_opaque(int& x) {
#ifndef SED_EMU
    asm volatile("" : "+v"(x));
#else
    (void)x;
#endif
}

// A value that is the same in all 64 lanes of a wave (e.g. the wave's index in its workgroup), moved to a scalar register: what is
// derived from it -- loop counters, frame indices, base addresses, branch conditions -- then lives in SGPRs and on the scalar ALU.
__device__ __forceinline__ int sed_wave_uniform(int x) {
#ifdef SED_EMU
    return x;
#else
    return __builtin_amdgcn_readfirstlane(x);
#endif
}

// Rendezvous of the 64 lanes of ONE wave around an exchange through wave-private LDS (a lane reads what another lane of its
// own wave wrote).  The LDS pipeline executes a wave's DS instructions in issue order, so no hardware barrier is needed: the
// fence only keeps the compiler from moving the accesses across it.  (A workgroup barrier cannot be used where waves run
// different trip counts.)
__device__ __forceinline__ void sed_wave_sync() {
#ifdef SED_EMU
    emu_wave_sync();
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
}

// Two values at once -> packed bf16 pairs (low half = a): hi = (bf16(a), bf16(b)), lo = bf16 of the exact remainders.
// gfx950 converts a pair with one v_cvt_pk_bf16_f32 (RNE, same rounding as f32_to_bf16): 5 VALU ops per pair instead of ~24.
__device__ __forceinline__ void bf16_split2(float a, float b, unsigned& hi, unsigned& lo) {
#ifdef SED_EMU
    unsigned short h0, l0, h1, l1;
    bf16_split(a, h0, l0);
    bf16_split(b, h1, l1);
    hi = (unsigned)h0 | ((unsigned)h1 << 16);
    lo = (unsigned)l0 | ((unsigned)l1 << 16);
#else
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    const f32x2_t hf = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xFFFF0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v - hf, bf16x2_t));
#endif
}

__device__ __forceinline__ f32x16 f32x16_zero() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// ---- quad exchanges on the DPP path ------------------------------------------------------------------
// __shfl_xor(v, 1 | 2) compiles to ds_bpermute_b32 -- a round trip through the LDS crossbar (~100+ cycles of latency, LDS issue
// slots) -- even though the partner sits in the same quad.  quad_perm DPP moves the value inside the VALU: [1,0,3,2] = 0xB1 for
// lane ^ 1, [2,3,0,1] = 0x4E for lane ^ 2.  These sit on the dependent chain of every GRU step.
__device__ __forceinline__ float sed_quad_xor1(float v) {
#ifdef SED_EMU
    return __shfl_xor(v, 1);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
#endif
}
__device__ __forceinline__ float sed_quad_xor2(float v) {
#ifdef SED_EMU
    return __shfl_xor(v, 2);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
#endif
}
// value of another quad of the same 16-lane row (row_ror:4 / row_ror:8); with sed_quad_xor1/2 first, two of these complete an
// all-reduce over the 16 lanes of a row (the 16x16 MFMA accumulator keeps one matrix row in the 16 lanes of a lane row)
__device__ __forceinline__ float sed_row_ror4(float v) {
#ifdef SED_EMU
    return __shfl_xor(v, 4);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
#endif
}
__device__ __forceinline__ float sed_row_ror8(float v) {
#ifdef SED_EMU
    return __shfl_xor(v, 8);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
#endif
}
__device__ __forceinline__ float sed_row16_max(float v) {
    v = fmaxf(v, sed_quad_xor1(v)); v = fmaxf(v, sed_quad_xor2(v));
    v = fmaxf(v, sed_row_ror4(v)); v = fmaxf(v, sed_row_ror8(v));
    return v;
}
__device__ __forceinline__ float sed_row16_sum(float v) {
    v += sed_quad_xor1(v); v += sed_quad_xor2(v);
    v += sed_row_ror4(v); v += sed_row_ror8(v);
    return v;
}
// sum over the four lanes of a quad, result in all four
__device__ __forceinline__ float sed_quad_sum(float v) {
    v += sed_quad_xor1(v);
    v += sed_quad_xor2(v);
    return v;
}

// ---- wave / block reductions -------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// ---- counter-based dropout RNG ------------------------------------------------------------------
// keep(idx) is a pure function of (seed, idx): the backward pass regenerates the forward mask.
// murmur3 finaliser over idx mixed with the seed; the tests replicate it in numpy.
// Counter-based dropout mask: element idx of a tensor is kept when the top 24 bits of hash(idx, seed) reach the threshold.
// Version 2 (round 2): the SEED takes the full murmur3 finaliser -- it is uniform, so the compiler does that once per kernel on
// the scalar unit -- and the per-element part is one multiply-add, one xor-shift and one multiply (7 VALU issue slots instead
// of 14: a 32-bit multiply costs two, tools/imul_probe.py).  The hash sits in every GLU block's epilogue, forward and backward
// (409 M evaluations per step; 40 % of the fused block-0 forward's VALU instructions with version 1).  Quality on 4 M consecutive
// indices: keep rate within 5e-4 of p, |lag-k autocorrelation| <= 3e-3, no correlation between seeds s and s + 1.
// -DSED_HASH_V1 builds the three-multiply version for A/B runs (tools/build_variant.py).  tests/parity_cases.py::np_keep_mask
// is the host replica.
__device__ __forceinline__ uint32_t sed_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t sed_hash(uint32_t idx, uint32_t seed) {
#ifdef SED_HASH_V1
    return sed_mix32(idx * 0x9E3779B1u + seed);
#else
    uint32_t x = idx * 0x9E3779B1u + sed_mix32(seed);
    x ^= x >> 15;
    return x * 0x2C1B3C6Du;
#endif
}

// SpecAugment bands of clip b from the counter-based generator: u_k(clip i) = top 24 bits of sed_hash(4 i + k, seed) / 2^24, k = 0 / 1
// frequency-mask length / start, 2 / 3 time-mask length / start; torchaudio's mask_along_axis(_iid) float32 arithmetic
// (sed_feat.hip: specaug_bounds_seeded_kernel; sed_conv_bf16.hip: the fused CNN prologue).  n == 1: one draw for the whole batch.
__device__ __forceinline__ void sed_specaug_draw(int* __restrict__ bounds, int b, int n, int f_param, int n_freq, int t_param,
                                                 int n_time, uint32_t seed) {
    const int i = n == 1 ? 0 : b;
    const float inv = 1.0f / 16777216.0f;
    int f0 = 0, f1 = 0, t0 = 0, t1 = 0;
    if (f_param >= 1) {
        const float value = (float)(sed_hash(4u * i + 0u, seed) >> 8) * inv * (float)f_param;
        const float min_value = (float)(sed_hash(4u * i + 1u, seed) >> 8) * inv * ((float)n_freq - value);
        f0 = (int)min_value;
        f1 = f0 + (int)value;
    }
    if (t_param >= 1) {
        const float value = (float)(sed_hash(4u * i + 2u, seed) >> 8) * inv * (float)t_param;
        const float min_value = (float)(sed_hash(4u * i + 3u, seed) >> 8) * inv * ((float)n_time - value);
        t0 = (int)min_value;
        t1 = t0 + (int)value;
    }
    bounds[4 * b] = f0; bounds[4 * b + 1] = f1; bounds[4 * b + 2] = t0; bounds[4 * b + 3] = t1;
}
// threshold = round(p * 2^24): keep when the top 24 bits are >= threshold.
__device__ __forceinline__ bool sed_keep(uint32_t idx, uint32_t seed, uint32_t threshold24) {
    return (sed_hash(idx, seed) >> 8) >= threshold24;
}

__device__ __forceinline__ float sed_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// tanh via one exp: 1 - 2/(e^{2x}+1); abs error < 1e-7 (saturates correctly at +-1)
__device__ __forceinline__ float sed_tanh(float x) { return 1.0f - 2.0f / (expf(2.0f * x) + 1.0f); }

// Hardware-rate transcendentals for the latency-bound GRU chain: v_exp_f32 / v_rcp_f32 (1 ulp each), i.e.
// sigmoid/tanh to ~3e-7 absolute -- far inside the parity tolerance, ~10x fewer dependent instructions than
// the IEEE expf + division sequences.
__device__ __forceinline__ float sed_fast_exp(float x) {
#ifdef SED_EMU
    return exp2f(x * 1.44269504088896341f);
#else
    return __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
#endif
}
__device__ __forceinline__ float sed_fast_rcp(float x) {
#ifdef SED_EMU
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
__device__ __forceinline__ float sed_fast_sigmoid(float x) { return sed_fast_rcp(1.0f + sed_fast_exp(-x)); }
__device__ __forceinline__ float sed_fast_tanh(float x) { return 1.0f - 2.0f * sed_fast_rcp(sed_fast_exp(2.0f * x) + 1.0f); }

// Wave issue priority (s_setprio 0..3).  The latency-bound kernels that share the chip with a throughput kernel on another
// stream (BiGRU recurrences beside the prefetched mel front-end, the other model's tail, the weight-gradient GEMMs) raise it: a
// recurrence wave that is ready to issue then wins the SIMD's arbitration over a co-resident streaming wave.
__device__ __forceinline__ void sed_wave_prio_high() {
#if !defined(SED_EMU) && !defined(SED_NO_SETPRIO)
    __builtin_amdgcn_s_setprio(3);
#endif
}

// s_setprio 1 / 0 around an MFMA cluster (cdna_hip_programming.md T5): the wave that owns the matrix pipe keeps it while its sibling on
// the SIMD issues the loads and LDS traffic of the next phase
__device__ __forceinline__ void sed_mfma_prio(int on) {
#if !defined(SED_EMU) && !defined(SED_NO_SETPRIO)
    if (on) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#else
    (void)on;
#endif
}

// LDS-DMA: every lane copies 16 bytes from its own global address to lds_wave_base + 16 * lane, without a register in between
// (global_load_lds_dwordx4; the LDS base is wave-uniform, M0).  Counted on vmcnt; ordered for other waves' ds_reads only by the issuing
// wave's s_waitcnt vmcnt + a barrier (MI355X_MICROARCH.md item 7).
__device__ __forceinline__ void sed_dma16(const void* gptr, void* lds_wave_base) {
#ifdef SED_EMU
    memcpy((char*)lds_wave_base + 16 * emu_lane(), gptr, 16);
#else
    __builtin_amdgcn_global_load_lds(gptr, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}
// The same copy with NO vector register at all: `buffer_load_dwordx4 off, s[rsrc], soffset lds` on a buffer resource with ADD_TID_ENABLE and
// a stride of 16 -- lane l reads base + soffset + 16 l.  (word3: the DATA_FORMAT bits are the stride's high bits under ADD_TID_ENABLE and must
// be 0; tools/dma_probe/addtid_probe.hip is the hardware check.)  An MFMA-heavy CU takes ~90 cycles to issue a piece that reads a VGPR
// offset beside running MFMAs (profiles/r06_linear_diag.md); this form reads none.
#ifdef SED_EMU
struct sed_rsrc { const char* base; };
static inline sed_rsrc sed_make_rsrc_tid16(const void* p, unsigned) { return sed_rsrc{(const char*)p}; }
static inline void sed_dma16_tid(sed_rsrc r, unsigned soffset_bytes, void* lds_wave_base) {
    memcpy((char*)lds_wave_base + 16 * emu_lane(), r.base + soffset_bytes + 16 * emu_lane(), 16);
}
#else
typedef __amdgpu_buffer_rsrc_t sed_rsrc;
__device__ __forceinline__ sed_rsrc sed_make_rsrc_tid16(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)16, (int)bytes, 0x00007000 | (1 << 23));
}
__device__ __forceinline__ void sed_dma16_tid(sed_rsrc r, unsigned soffset_bytes, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, (int)soffset_bytes, 0, 0);
}
#endif
// s_waitcnt vmcnt(N) lgkmcnt(0) by hand (N a literal): the compiler neither counts LDS-DMA nor knows which stage a ds_read belongs to
#ifdef SED_EMU
#define SED_WAIT_VM_LDS(N) do { } while (0)
#else
#define SED_WAIT_VM_LDS(N) do { asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

// Raw workgroup barrier for hand-phased kernels (the two wave groups of linear_pp_kernel run one barrier apart): s_barrier with NO
// implied waitcnt -- global loads in flight survive it; LDS traffic is ordered by the explicit sed_wait_lds() calls.  The scheduler
// fences keep the compiler from moving LDS accesses or MFMAs across.
__device__ __forceinline__ void sed_phase_barrier() {
#ifdef SED_EMU
    emu_block_barrier();
#else
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
}
__device__ __forceinline__ void sed_wait_lds() {
#ifndef SED_EMU
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// ---------------------------------------------------------------------------------------------
// Halo tile of the single-channel first layer (conv0 / block 0): rows t0 - 1 .. t0 + ROWS of clip b, bins -1 .. F, staged into LDS at
// row pitch `pitch` (tile[i * pitch + j] <-> frame t0 - 1 + i, bin j - 1), minus `center`; zero padding and SpecAugment-masked
// bins / frames hold 0 - center.  Round 4: float4 row loads on a (row, bin quad) thread map -- the two halo COLUMNS are always zero
// padding and are written, never loaded; a row of F bins is F / 4 aligned 16-byte loads.  The scalar version it replaces walked
// the flattened (ROWS + 2) x (F + 2) tile one float at a time and paid an integer division by the runtime width, two clamps and
// five compares per element: ~ 60 instructions per load, ten loads per thread -- a quarter of conv0_kernel's instructions.
// All loads are unconditional (clamped row) and in flight before the first LDS store (tools/isa_exposed_loads.py).
// Needs F % 4 == 0 and F / 4 a power of two (every n_mels of the recipes); returns false otherwise (caller takes the scalar path).
// ---------------------------------------------------------------------------------------------
template <int ROWS, int NTHREADS, int MAXF, int NB = 8>       // NB: float4 loads in flight per thread (register budget of the caller)
__device__ __forceinline__ bool sed_stage_halo_f4(float* tile, const float* __restrict__ x, const int* __restrict__ bounds, int b, int t0,
                                                  int T, int F, int pitch, float center) {
    const int nq = F >> 2;
    if ((F & 3) || (nq & (nq - 1)) || nq < 1) return false;
    const int sh = 31 - __builtin_clz(nq), total = (ROWS + 2) << sh, tid = threadIdx.x;
    int mf0 = 0, mf1 = 0, mt0 = 0, mt1 = 0;
    if (bounds) { mf0 = bounds[4 * b]; mf1 = bounds[4 * b + 1]; mt0 = bounds[4 * b + 2]; mt1 = bounds[4 * b + 3]; }
    constexpr int NIT = ((ROWS + 2) * (MAXF / 4) + NTHREADS - 1) / NTHREADS, NBB = NIT < NB ? NIT : NB;
#pragma unroll
    for (int u0 = 0; u0 < NIT; u0 += NBB) {
        float4 v[NBB];
#pragma unroll
        for (int k = 0; k < NBB; ++k) {
            const int idx = tid + NTHREADS * (u0 + k), ic = idx < total ? idx : total - 1;
            const int i = ic >> sh, q = ic - (i << sh);
            const int t = t0 - 1 + i, tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
            v[k] = *(const float4*)(x + ((size_t)b * T + tc) * F + 4 * q);
        }
        sed_sched_fence();
#pragma unroll
        for (int k = 0; k < NBB; ++k) { sed_pin(v[k].x); sed_pin(v[k].y); sed_pin(v[k].z); sed_pin(v[k].w); }
#pragma unroll
        for (int k = 0; k < NBB; ++k) {
            const int idx = tid + NTHREADS * (u0 + k);
            if (idx < total) {
                const int i = idx >> sh, q = idx - (i << sh), t = t0 - 1 + i, f = 4 * q;
                const bool rowok = t >= 0 && t < T && !(t >= mt0 && t < mt1);
                float* d = tile + i * pitch + 1 + f;
                d[0] = ((rowok && !(f >= mf0 && f < mf1)) ? v[k].x : 0.f) - center;
                d[1] = ((rowok && !(f + 1 >= mf0 && f + 1 < mf1)) ? v[k].y : 0.f) - center;
                d[2] = ((rowok && !(f + 2 >= mf0 && f + 2 < mf1)) ? v[k].z : 0.f) - center;
                d[3] = ((rowok && !(f + 3 >= mf0 && f + 3 < mf1)) ? v[k].w : 0.f) - center;
            }
        }
    }
    if (tid < 2 * (ROWS + 2)) tile[(tid >> 1) * pitch + ((tid & 1) ? F + 1 : 0)] = 0.f - center;
    return true;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed fp32 FMA (v_pk_fma_f32): two lanes-worth of FMAs per VALU issue slot
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
