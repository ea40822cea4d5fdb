// DIAGNOSTICS ONLY -- not part of libsed_hip.so.  The round-5 MULTI-FRAME form of the wave-per-frame mel kernel (a wave transforms two
// or more frames per launch: runs of MEL_RUN frames per 4-wave workgroup, optionally persistent workgroups), kept as a reproducer of the
// hipGraph-replay fault described in profiles/r05_mel_graph_race.md / profiles/r06_mel_mechanism.md.  The product kernel
// (desed_task_amd/csrc/sed_mel.hip: mel_wave_kernel) gives every wave exactly ONE frame and has never shown the fault.
//   build:  bash tools/mel_repro/build.sh [name] [-DMEL_RUN=8 -DMEL_PERSISTENT -DMEL_DUMP ...]  ->  tools/_melrepro_<name>.so
//   run:    python tools/mel_repro/race.py tools/_melrepro_<name>.so [replays] [beside]
// MEL_DUMP: every frame also writes, into a debug buffer, (A) its 16 Z values per lane after pass 3 from the REGISTERS, (B) the same
// slots read back from the LDS buffer after the conjugate-exchange store, (C) the 16 magnitudes, and (D) where / when it ran
// (HW_ID, XCC_ID, s_memtime at frame start / end) -- so that a bad frame can be compared word by word with a good replay of the same
// input: which register / LDS word differs, in which lanes, on which CU / SIMD.
#include "sed_common.h"
#define MEL_NFFT 2048
#define MEL_M 1024
#define MEL_DUMP_FLOATS 5136      // per frame: 2048 (A) + 2048 (B) + 1024 (C) + 16 words (D)
#ifndef MEL_RUN
#define MEL_RUN 8
#endif
// MEL_PERSISTENT: workgroups walk several runs (grid capped at two workgroups per CU) -- the form in which the open issue above shows.
// Default: ONE run per workgroup (grid = all runs; the 26 KB of tables are re-read from L2 per run).
#define MEL_XPAD 1088          // 1024 + 64 padding slots (exchange 1)

// Complex arithmetic on the packed-fp32 pipe: a complex number is one 64-bit VGPR pair, and VOP3P's op_sel / neg modifiers pick and
// negate the halves, so that a rotation by +-i folds into the add and a complex product is two instructions (the compiler's own
// lowering of the float2 formulas spent 18 % of the frame loop on v_mov shuffles between scalar and packed forms).
#if defined(SED_EMU) || defined(MEL_PLAIN_MATH)
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cadd_mi(float2 a, float2 b) { return make_float2(a.x + b.y, a.y - b.x); }     // a - i b
__device__ __forceinline__ float2 cadd_pi(float2 a, float2 b) { return make_float2(a.x - b.y, a.y + b.x); }     // a + i b
__device__ __forceinline__ float2 cmulp(float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
__device__ __forceinline__ float2 cscale(float2 a, float2 w) { return make_float2(a.x * w.x, a.y * w.y); }      // elementwise
__device__ __forceinline__ float fast_sqrt(float x) { return sqrtf(x); }
#else
__device__ __forceinline__ f32x2 c_in(float2 a) { f32x2 r; r.x = a.x; r.y = a.y; return r; }
__device__ __forceinline__ float2 c_out(f32x2 a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return c_out(c_in(a) + c_in(b)); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return c_out(c_in(a) - c_in(b)); }
#ifdef MEL_SAFE_OPSEL
// MEL_SAFE_OPSEL: the product's fix applied to this reproducer -- the operand whose halves are swapped is src0 (op_sel:[1,0]), not src1
__device__ __forceinline__ float2 cadd_mi(float2 a, float2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(c_in(b)), "v"(c_in(a)));
    return c_out(r);
}
__device__ __forceinline__ float2 cadd_pi(float2 a, float2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(r) : "v"(c_in(b)), "v"(c_in(a)));
    return c_out(r);
}
#elif defined(MEL_CHECK)
// MEL_CHECK: every +-i packed add (the instruction the MEL_DUMP runs pointed at) is re-computed with two scalar VALU instructions from
// the SAME source registers and compared bitwise; a mismatch is logged: sources, the packed result, the scalar result, the packed
// instruction executed once more on the same sources (transient or repeatable?), lane, HW_ID, which of the two forms.
__device__ unsigned* mel_chk_buf;       // [0] = count, then 16 words per event
__device__ __forceinline__ void mel_chk(f32x2 r, float2 a, float2 b, int form) {
    float ex, ey;
    if (form == 0) {        // a - i b = (a.x + b.y, a.y - b.x)
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(ex) : "v"(a.x), "v"(b.y));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(ey) : "v"(a.y), "v"(b.x));
    } else {                // a + i b = (a.x - b.y, a.y + b.x)
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(ex) : "v"(a.x), "v"(b.y));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(ey) : "v"(a.y), "v"(b.x));
    }
    if (__float_as_uint(ex) != __float_as_uint(r.x) || __float_as_uint(ey) != __float_as_uint(r.y)) {
        f32x2 r2, av, bv;
        av.x = a.x; av.y = a.y; bv.x = b.x; bv.y = b.y;
        if (form == 0) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r2) : "v"(av), "v"(bv));
        else asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r2) : "v"(av), "v"(bv));
        unsigned* cb = mel_chk_buf;
        if (cb) {
            const unsigned i = atomicAdd(cb, 1u);
            if (i < 4000) {
                unsigned* e = cb + 16 + 16 * i;
                e[0] = __float_as_uint(a.x); e[1] = __float_as_uint(a.y); e[2] = __float_as_uint(b.x); e[3] = __float_as_uint(b.y);
                e[4] = __float_as_uint(r.x); e[5] = __float_as_uint(r.y); e[6] = __float_as_uint(ex); e[7] = __float_as_uint(ey);
                e[8] = __float_as_uint(r2.x); e[9] = __float_as_uint(r2.y); e[10] = threadIdx.x; e[11] = blockIdx.x;
                e[12] = __builtin_amdgcn_s_getreg((31 << 11) | 4); e[13] = (unsigned)form;
                e[14] = (unsigned)__builtin_amdgcn_s_memtime();
            }
        }
    }
}
__device__ __forceinline__ float2 cadd_mi(float2 a, float2 b) {
    f32x2 r;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(c_in(a)), "v"(c_in(b)));
    mel_chk(r, a, b, 0);
    return c_out(r);
}
__device__ __forceinline__ float2 cadd_pi(float2 a, float2 b) {
    f32x2 r;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(c_in(a)), "v"(c_in(b)));
    mel_chk(r, a, b, 1);
    return c_out(r);
}
#else
__device__ __forceinline__ float2 cadd_mi(float2 a, float2 b) {       // a - i b = (a.x + b.y, a.y - b.x)
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(c_in(a)), "v"(c_in(b)));
    return c_out(r);
}
__device__ __forceinline__ float2 cadd_pi(float2 a, float2 b) {       // a + i b = (a.x - b.y, a.y + b.x)
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(c_in(a)), "v"(c_in(b)));
    return c_out(r);
}
#endif
__device__ __forceinline__ float2 cmulp(float2 a, float2 w) {         // a w = a.x (w.x, w.y) + a.y (-w.y, w.x)
    f32x2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(c_in(a)), "v"(c_in(w)));         // (a.y w.y, a.y w.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(c_in(a)), "v"(c_in(w)), "v"(t));
    return c_out(r);
}
__device__ __forceinline__ float2 cscale(float2 a, float2 w) { return c_out(c_in(a) * c_in(w)); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }      // v_sqrt_f32: 1 ulp, no denormal fix-up
#endif
// forward DFT-4 in place: (v0, v1, v2, v3) -> (X0, X1, X2, X3), e^{-i pi / 2} = -i
__device__ __forceinline__ void dft4(float2& v0, float2& v1, float2& v2, float2& v3) {
    const float2 a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3), a3 = csub(v1, v3);
    v0 = cadd(a0, a2);
    v1 = cadd_mi(a1, a3);
    v2 = csub(a0, a2);
    v3 = cadd_pi(a1, a3);
}
// forward DFT-16 of v[0..15] in place, natural order in and out: n = n1 + 4 n2, K = 4 k1 + k2;
// A[n1][k2] = DFT4 over n2 of v[n1 + 4 n2];  A *= w16^(n1 k2);  X[4 k1 + k2] = DFT4 over n1 of A[n1][k2]
__device__ __forceinline__ void dft16(float2* v) {
    const float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) dft4(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);      // v[n1 + 4 k2] = A[n1][k2]
    // twiddles w16^(n1 k2) = (cos, -sin)(2 pi n1 k2 / 16): exponents 1 2 3 / 2 4 6 / 3 6 9
    v[1 + 4] = cmulp(v[1 + 4], make_float2(C1, -S1));
    v[1 + 8] = cmulp(v[1 + 8], make_float2(H, -H));
    v[1 + 12] = cmulp(v[1 + 12], make_float2(S1, -C1));
    v[2 + 4] = cmulp(v[2 + 4], make_float2(H, -H));
    v[2 + 8] = make_float2(v[2 + 8].y, -v[2 + 8].x);                                                  // w^4 = -i
    v[2 + 12] = cmulp(v[2 + 12], make_float2(-H, -H));
    v[3 + 4] = cmulp(v[3 + 4], make_float2(S1, -C1));
    v[3 + 8] = cmulp(v[3 + 8], make_float2(-H, -H));
    v[3 + 12] = cmulp(v[3 + 12], make_float2(-C1, S1));                                               // w^9
    // outer DFT4 over n1 for each k2: inputs v[n1 + 4 k2], outputs X[4 k1 + k2]
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) dft4(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);      // v[4 k2 + k1] = X[4 k1 + k2]
    // transpose the 4 x 4 register block to natural order (renaming only: everything is unrolled)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b) { const float2 t = v[4 * a + b]; v[4 * a + b] = v[4 * b + a]; v[4 * b + a] = t; }
}

// Workgroup = MEL_WAVES waves: the frame-invariant tables (window 8 KB, pass-2 twiddles 2 KB, the mel taps 16 KB) are shared by six
// waves, 8.5 KB of exchange buffer each -> 78 KB: two workgroups = twelve waves per CU = three per SIMD, at <= 168 VGPRs.
// Mel taps in LDS: band `lane` as MEL_GA and band 127 - lane as MEL_GB groups of four taps, the first group starting at the band's
// first bin rounded DOWN to a multiple of four (leading / trailing zeros), stored [group][lane] -- every lane reads its 16 bytes of group
// g at the same offset (conflict-free), and the magnitudes as aligned 16-byte reads too: 32 ds_read_b128 per frame instead of 120 b32.
#define MEL_WAVES 4
#ifndef MEL_OCC
#define MEL_OCC 2
#endif
#define MEL_GA 4
#define MEL_GB 12

template <bool LOG>
__global__ __launch_bounds__(64 * MEL_WAVES, MEL_OCC) void mel_wave_kernel(const float* __restrict__ audio, float* __restrict__ out,
                                                       int B, int N, int T, int hop, int n_mels,
                                                       const float* __restrict__ window, const float2* __restrict__ tw1024,
                                                       const float2* __restrict__ tw2048, const int* __restrict__ fb_start,
                                                       const int* __restrict__ fb_len, const float* __restrict__ fb_w, int fb_stride,
                                                       const float4* __restrict__ taps, int runs_per_clip, int segs_per_clip, float* __restrict__ dbg) {
    __shared__ float2 s_win[MEL_M];                             // (w[2n], w[2n + 1])
    __shared__ float2 s_tw16[15 * 16];                          // [r - 1][k]: e^{-2 pi i r k / 256}, r = 1..15, k < 16 (pass 2)
    __shared__ __attribute__((aligned(16))) float4 s_wa[MEL_GA][64];      // taps of band `lane`
    __shared__ __attribute__((aligned(16))) float4 s_wb[MEL_GB][64];      // taps of band 127 - lane
    __shared__ __attribute__((aligned(16))) float2 s_x[MEL_WAVES][MEL_XPAD];     // one exchange buffer per wave
#ifdef MEL_LDS_PAD
    __shared__ char s_pad[MEL_LDS_PAD];         // occupancy experiment: bytes nobody uses (kept alive by an impossible store)
    if (B < 0) s_pad[threadIdx.x] = 1;
#endif
    const int tid = threadIdx.x, wave = sed_wave_uniform(tid >> 6), lane = tid & 63;
    for (int i = tid; i < MEL_M; i += 64 * MEL_WAVES) s_win[i] = make_float2(window[2 * i], window[2 * i + 1]);
    if (tid < 240) s_tw16[tid] = tw1024[(4 * ((tid >> 4) + 1) * (tid & 15)) & (MEL_M - 1)];
    // (the tap tables come ready-made from sed_mel_taps: built per workgroup from fb_start / fb_len / fb_w they were eleven rounds of
    //  dependent gathers -- ~20 us in front of a workgroup's ~10 frames per wave)
    for (int i = tid; i < (MEL_GA + MEL_GB) * 64; i += 64 * MEL_WAVES) {
        if (i < MEL_GA * 64) s_wa[i >> 6][i & 63] = taps[i];
        else s_wb[(i >> 6) - MEL_GA][i & 63] = taps[i];
    }
    __syncthreads();                            // the only workgroup barrier

    float2* xb = s_x[wave];
    float* magb = reinterpret_cast<float*>(xb);
    // The only per-lane twiddle kept in registers is w2048^lane: the real-FFT twiddles w2048^(lane + 64 q) = w2048^lane w32^q and
    // the pass-3 twiddles w1024^(r (lane + 64 m)) = ((w2048^lane)^2 w16^m)^r are formed from it per frame (a dozen packed products
    // against eight more registers that the frame loop does not have)
    const float2 wl = tw2048[lane];
    // mel bands of this lane: A = lane, Bd = 127 - lane
    const int bandA = lane, bandB = 127 - lane;
    int sA = 0, lA = 0, sB = 0, lB = 0;
    if (bandA < n_mels) { sA = fb_start[bandA]; lA = fb_len[bandA]; }
    if (bandB < n_mels) { sB = fb_start[bandB]; lB = fb_len[bandB]; }
    const int gA0 = sA >> 2, gB0 = sB >> 2;     // first aligned group of four magnitudes of each band

    // XCD-aware walk: workgroup g sits on XCD g & 7 and takes the runs of the SEGMENTS s = (g & 7) (mod 8); a segment is a clip
    // (segs_per_clip = 1 at the recipes' batch sizes) or, for small batches, one of several stretches of consecutive runs of a clip
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, n_slots = (gridDim.x + 7 - xcd) >> 3;
    const int runs_per_seg = (runs_per_clip + segs_per_clip - 1) / segs_per_clip;
    const int segs_here = (B * segs_per_clip - xcd + 7) >> 3;
    const int total_runs = segs_here * runs_per_seg;
    float2 v[16];
    for (int run = slot; run < total_runs; run += n_slots) {
        const int seg = xcd + 8 * (run / runs_per_seg);
        const int b = seg / segs_per_clip;
        const int run_in_clip = (seg - b * segs_per_clip) * runs_per_seg + run % runs_per_seg;
        if (run_in_clip >= runs_per_clip) continue;
        const int t0 = run_in_clip * MEL_RUN;
        const float* clip = audio + (size_t)b * N;
        const int t_end = t0 + MEL_RUN < T ? t0 + MEL_RUN : T;

        auto load_frame = [&](int t_) {
            const int base = t_ * hop - MEL_NFFT / 2;
            if (base >= 0 && base + MEL_NFFT <= N && ((reinterpret_cast<uintptr_t>(clip + base) & 7) == 0)) {   // interior frame: 8-byte loads
                const float2* p = reinterpret_cast<const float2*>(clip + base);
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = p[lane + 64 * j];
            } else {                                // the first / last four frames of a clip: reflected indices, element by element
                int le = lane;
                sed_opaque(le);                     // (nothing of this cold path is worth a register outside it)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    int s0 = base + 2 * (le + 64 * j), s1 = s0 + 1;
                    if (s0 < 0) s0 = -s0;
                    if (s1 < 0) s1 = -s1;
                    if (s0 >= N) s0 = 2 * (N - 1) - s0;
                    if (s1 >= N) s1 = 2 * (N - 1) - s1;
                    v[j] = make_float2(clip[s0], clip[s1]);
                }
            }
        };
        int t = t0 + wave;
        if (t < t_end) load_frame(t);
        int frame_seq = 0;
        for (; t < t_end; t += MEL_WAVES, ++frame_seq) {
#ifdef MEL_DUMP
            const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
            float* dfr = dbg ? dbg + ((size_t)b * T + t) * MEL_DUMP_FLOATS : nullptr;
#endif
            // (the table reads below are frame-invariant: without an opaque index LICM keeps all of them live across the frame loop
            //  and the kernel spills)
            int ln = lane;
            sed_opaque(ln);
            float2 wl_ = wl;                    // (same for what is derived from the per-lane twiddle)
            sed_pin(wl_.x); sed_pin(wl_.y);
            // ---- window; pass 1: radix 16, Ns = 1: in[lane + 64 r] -> out[16 lane + r] ----
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = cscale(v[j], s_win[ln + 64 * j]);
            dft16(v);
            sed_wave_sync(); sed_sched_fence();                    // (the previous frame's mel stage has read its magnitudes from this buffer)
#pragma unroll
            for (int r = 0; r < 16; ++r) xb[17 * lane + r] = v[r];
            sed_wave_sync(); sed_sched_fence();
            // ---- pass 2: radix 16, Ns = 16: in[lane + 64 r] * w256^(r k), k = lane & 15 -> out[(lane - k) 16 + k + 16 r] ----
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = xb[lane + (lane >> 4) + 68 * r];        // e + (e >> 4), e = lane + 64 r
#pragma unroll
            for (int r = 1; r < 16; ++r) v[r] = cmulp(v[r], s_tw16[(r - 1) * 16 + (ln & 15)]);
            dft16(v);
            sed_wave_sync(); sed_sched_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) xb[((lane >> 4) << 8) + (lane & 15) + 16 * r] = v[r];
            sed_wave_sync(); sed_sched_fence();
            // ---- pass 3: radix 4, Ns = 256, four butterflies per lane: j = lane + 64 m, in[j + 256 r] * w1024^(r j) -> out[j + 256 r]
            const float2 wl2 = cmulp(wl_, wl_);                                          // w1024^lane
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float2 a0 = xb[lane + 64 * m], a1 = xb[lane + 64 * m + 256], a2 = xb[lane + 64 * m + 512], a3 = xb[lane + 64 * m + 768];
                const float a16 = 0.39269908169872414f * m;                              // w16^m = (cos, -sin)(2 pi m / 16): constants
                const float2 w1m = m == 0 ? wl2 : cmulp(wl2, make_float2(__builtin_cosf(a16), -__builtin_sinf(a16)));
                const float2 w2 = cmulp(w1m, w1m), w3 = cmulp(w2, w1m);
                a1 = cmulp(a1, w1m); a2 = cmulp(a2, w2); a3 = cmulp(a3, w3);
                dft4(a0, a1, a2, a3);
                v[m] = a0; v[m + 4] = a1; v[m + 8] = a2; v[m + 12] = a3;                  // v[q] = Z[lane + 64 q]
            }
#ifdef MEL_DUMP
            if (dfr) {                                   // (A) Z from the registers
#pragma unroll
                for (int q = 0; q < 16; ++q) reinterpret_cast<float2*>(dfr)[64 * q + lane] = v[q];
            }
#endif
            // ---- real-FFT step: X[k] = Xe + w2048^k Xo needs Z[1024 - k]: conjugate-pair exchange through the buffer ----
            sed_wave_sync(); sed_sched_fence();
#pragma unroll
            for (int q = 0; q < 16; ++q) xb[lane + 64 * q] = v[q];
            if (lane == 0) xb[MEL_M] = v[0];                                             // Z[1024] = Z[0]
            sed_wave_sync(); sed_sched_fence();
#ifdef MEL_DUMP
            if (dfr) {                                   // (B) the same slots read back from LDS
#pragma unroll
                for (int q = 0; q < 16; ++q) reinterpret_cast<float2*>(dfr + 2048)[64 * q + lane] = xb[lane + 64 * q];
            }
            sed_wave_sync(); sed_sched_fence();
#endif
            float mg[16], mag_nyq = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = ln + 64 * q;
                const float2 zk = v[q], zm = xb[MEL_M - k];
                const float2 xe = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                const float dr = zk.x - zm.x, di = zk.y + zm.y;                          // zk - conj(zm)
                const float2 xo = make_float2(0.5f * di, -0.5f * dr);                     // -i/2 (zk - conj(zm))
                // w2048^(lane + 64 q) = w2048^lane * w32^q, w32^q = (cos, -sin)(2 pi q / 32)
                const float ang = 0.19634954084936207f * q;                              // folded: q is a compile-time constant
                const float2 wq = cmulp(wl_, make_float2(__builtin_cosf(ang), -__builtin_sinf(ang)));
                const float2 wx = cmulp(wq, xo);
                const float re = xe.x + wx.x, im = xe.y + wx.y;
                mg[q] = fast_sqrt(re * re + im * im);
                if (q == 0) {                                                            // k = 0 (lane 0): DC and Nyquist are real
                    if (lane == 0) { mg[0] = fabsf(zk.x + zk.y); mag_nyq = fabsf(zk.x - zk.y); }
                }
            }
#ifdef MEL_DUMP
            if (dfr) {                                   // (C) magnitudes, (D) place and time
#pragma unroll
                for (int q = 0; q < 16; ++q) dfr[4096 + 64 * q + lane] = mg[q];
                if (lane == 0) {
                    const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
                    unsigned* d = reinterpret_cast<unsigned*>(dfr + 5120);
                    d[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID
                    d[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // XCC_ID
                    d[2] = (unsigned)ts0; d[3] = (unsigned)(ts0 >> 32); d[4] = (unsigned)ts1; d[5] = (unsigned)(ts1 >> 32);
                    d[6] = (unsigned)frame_seq; d[7] = (unsigned)blockIdx.x; d[8] = (unsigned)wave;
                }
            }
#endif
            sed_wave_sync(); sed_sched_fence();
#pragma unroll
            for (int q = 0; q < 16; ++q) magb[lane + 64 * q] = mg[q];
            if (lane == 0) magb[MEL_M] = mag_nyq;
            sed_wave_sync(); sed_sched_fence();
            // ---- the next frame's samples travel while the mel stage runs ----
            const int t_cur = t;
            if (t + MEL_WAVES < t_end) load_frame(t + MEL_WAVES);
            // ---- sparse HTK mel: bands `lane` and `127 - lane`; four partial sums per band ----
            const float4* mag4 = reinterpret_cast<const float4*>(magb);
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
#pragma unroll
            for (int g = 0; g < MEL_GA; ++g) {
                const float4 w = s_wa[g][ln], m = mag4[gA0 + g];
                a0 = fmaf(w.x, m.x, a0); a1 = fmaf(w.y, m.y, a1); a2 = fmaf(w.z, m.z, a2); a3 = fmaf(w.w, m.w, a3);
            }
            sed_sched_fence();
#pragma unroll
            for (int g = 0; g < MEL_GB; ++g) {
                const float4 w = s_wb[g][ln], m = mag4[gB0 + g];
                b0 = fmaf(w.x, m.x, b0); b1 = fmaf(w.y, m.y, b1); b2 = fmaf(w.z, m.z, b2); b3 = fmaf(w.w, m.w, b3);
                if ((g & 3) == 3) sed_sched_fence();        // at most four groups (32 VGPRs) of taps and magnitudes in flight
            }
            float accA = (a0 + a1) + (a2 + a3), accB = (b0 + b1) + (b2 + b3);
            // bands longer than the tables (no recipe has any): the remaining taps from memory
            for (int j = 4 * MEL_GA - (sA & 3); j < lA; ++j) accA = fmaf(fb_w[(size_t)bandA * fb_stride + j], magb[sA + j], accA);
            for (int j = 4 * MEL_GB - (sB & 3); j < lB; ++j) accB = fmaf(fb_w[(size_t)bandB * fb_stride + j], magb[sB + j], accB);
            if (LOG) {
                accA = fminf(fmaxf(20.0f * log10f(fmaxf(accA, 1e-5f)), -50.0f), 80.0f);
                accB = fminf(fmaxf(20.0f * log10f(fmaxf(accB, 1e-5f)), -50.0f), 80.0f);
            }
            float* o = out + ((size_t)b * T + t_cur) * n_mels;
            if (bandA < n_mels) o[bandA] = accA;
            if (bandB < n_mels) o[bandB] = accB;
        }
    }
}


SED_API int melrepro_set_check(unsigned* buf) {
#ifdef MEL_CHECK
    return hipMemcpyToSymbol(HIP_SYMBOL(mel_chk_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -2;
#else
    (void)buf;
    return -3;
#endif
}
SED_API int melrepro_dump_floats() {
#ifdef MEL_DUMP
    return MEL_DUMP_FLOATS;
#else
    return 0;
#endif
}
SED_API int melrepro_fwd_wave(const float* audio, float* out, int B, int N, int T, int n_fft, int hop, int n_mels,
                              const float* window, const float* tw1024, const float* tw2048, const int* fb_start,
                              const int* fb_len, const float* fb_w, int fb_stride, const float* taps, int apply_log, float* dbg, void* stream) {
    if (n_fft != MEL_NFFT || n_mels > 128 || n_mels < 1 || N < n_fft / 2 + 1 || T != 1 + N / hop || !taps) return SED_ERR_UNSUPPORTED;
    if (B <= 0) return SED_OK;
    const int runs_per_clip = (T + MEL_RUN - 1) / MEL_RUN;
    int segs_per_clip = B >= 32 ? 1 : 32 / B;
    if (segs_per_clip > runs_per_clip) segs_per_clip = runs_per_clip;
    long long runs = (long long)B * runs_per_clip;
#ifdef MEL_PERSISTENT
    int grid = runs < 512 ? (int)runs : 512;        // 2 resident workgroups on each of the 256 CUs
    grid = (grid + 7) & ~7;
#else
    const int rps_ = (runs_per_clip + segs_per_clip - 1) / segs_per_clip;
    const int grid = 8 * ((B * segs_per_clip + 7) / 8) * rps_;
    (void)runs;
#endif
    if (apply_log) return SED_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((mel_wave_kernel<false>), dim3(grid), dim3(64 * MEL_WAVES), 0, (hipStream_t)stream, audio, out, B, N, T, hop,
                       n_mels, window, (const float2*)tw1024, (const float2*)tw2048, fb_start, fb_len, fb_w, fb_stride,
                       (const float4*)taps, runs_per_clip, segs_per_clip, dbg);
    return hipGetLastError() == hipSuccess ? SED_OK : SED_ERR_LAUNCH;
}
