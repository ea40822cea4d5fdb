// K1: fused reflect-pad -> Hamming window -> 2048-point real FFT (LDS Stockham radix-4 on the
// packed 1024-point complex transform) -> magnitude -> sparse HTK mel (-> optional 20*log10 / clamp).
// Replaces torchaudio MelSpectrogram(n_fft=2048, hop=256, power=1, center=True, reflect) as built at
// recipes/dcase2023_task4_baseline/local/sed_trainer.py:80-91 and called at :282.
//
// Layout: audio (B, N) fp32 row-major; out (B, T, n_mels) fp32 -- frame-major ("NHWC with C=1"),
// so one frame's 128 mel values are one coalesced 512-byte store and conv0 reads rows directly.
// The Python side hands the reference's (B, n_mels, T) shape out as a transposed view.
//
// One 256-thread workgroup per frame (grid-stride over B*T frames).  LDS: two 8 KB ping-pong
// buffers (XOR-swizzled indices) + 8 KB per-pass twiddle tables + 4.1 KB magnitudes.  HBM: 4*N bytes in (each sample is re-read 8x by
// overlapping frames, served by L2) + 4*T*n_mels out per clip = 960,512 B/clip at the 2023 config.
#include "sed_common.h"

#define MEL_NFFT 2048
#define MEL_M 1024   // packed complex length

// LDS index swizzle of the two FFT buffers.  The Stockham passes read consecutive elements but WRITE with strides 4 and 16
// (Ns = 1, 4): as plain indices those ds_write_b64 hit every bank pair four times (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
// = 0.42 for the kernel).  XOR-ing bits 2..5 into bits 0..3 keeps the reads at 1.14x and makes every write conflict-free
// under the gfx950 lane groups (modelled over all five passes: 992 -> 640 LDS cycles per frame for the data buffers).
__device__ __forceinline__ int mel_sw(int i) { return i ^ ((i >> 2) & 15); }

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
// (the operand whose halves are swapped is src0, never src1: sed_common.h, "gfx950 hazard".  Round 5's form -- b as src1 with
//  op_sel:[0,1] -- is the instruction that returned a.x instead of a.x -+ b.y in lanes 48-63 beside the split-bf16 GEMM.)
__device__ __forceinline__ float2 cadd_mi(float2 a, float2 b) {       // a - i b = (b.y + a.x, -b.x + a.y)
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(c_in(b)), "v"(c_in(a)));
    return c_out(r);
}
__device__ __forceinline__ float2 cadd_pi(float2 a, float2 b) {       // a + i b = (-b.y + a.x, b.x + a.y)
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(r) : "v"(c_in(b)), "v"(c_in(a)));
    return c_out(r);
}
__device__ __forceinline__ float2 cmulp(float2 a, float2 w) {         // a w = a.x (w.x, w.y) + a.y (-w.y, w.x)
    f32x2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(c_in(a)), "v"(c_in(w)));         // (a.y w.y, a.y w.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(c_in(a)), "v"(c_in(w)), "v"(t));
    return c_out(r);
}
__device__ __forceinline__ float2 cscale(float2 a, float2 w) { return c_out(c_in(a) * c_in(w)); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }      // v_sqrt_f32: 1 ulp, no denormal fix-up
#endif

// WPT > 0: the thread's filterbank taps (every second tap of its band, zero-padded to WPT) live in registers for all its frames.
// With the weights read from memory inside the frame loop the mel stage was a chain of dependent loads -- start, length, then one
// weight per tap, each waited for before the next (tools/isa_exposed_loads.py): ~22 cache round trips per frame on the widest band.
template <bool LOG, int WPT>
__global__ __launch_bounds__(256, 4) void mel_kernel(const float* __restrict__ audio, float* __restrict__ out,
                                                  int B, int N, int T, int hop, int n_mels,
                                                  const float* __restrict__ window, const float2* __restrict__ tw1024,
                                                  const float2* __restrict__ tw2048, const int* __restrict__ fb_start,
                                                  const int* __restrict__ fb_len, const float* __restrict__ fb_w, int fb_stride, int S, int fps) {
    __shared__ float2 buf0[MEL_M];
    __shared__ float2 buf1[MEL_M];
    // per-pass twiddle tables [pass 1..4][j = 1..3][k < Ns]: w^(j k 256/Ns), contiguous in k.  (Indexing one 1024-entry table
    // with the stride 256/Ns put the 4 / 16 / 64 distinct twiddles of passes 1-3 on one, two and eight bank pairs.)
    __shared__ float2 stw[3 * (4 + 16 + 64 + 256)];
    __shared__ float mag[MEL_M + 8];
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * (4 + 16 + 64 + 256); i += 256) {
        int pass = 1, base = 0;
        while (i >= base + 3 * (1 << (2 * pass))) { base += 3 * (1 << (2 * pass)); ++pass; }
        const int Ns = 1 << (2 * pass), j = (i - base) / Ns, k = (i - base) - j * Ns;
        stw[i] = tw1024[((j + 1) * (256 / Ns) * k) & (MEL_M - 1)];
    }
    __syncthreads();

    // per-thread constants of every frame: its 8 window taps and 4 real-FFT twiddles
    float2 win[4], tw2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = tid + 256 * q;
        win[q] = make_float2(window[2 * n], window[2 * n + 1]);
        tw2[q] = tw2048[n];
    }
    // this thread's half of mel band tid / 2 (frame-invariant)
    const int mband = tid >> 1, mhalf = tid & 1;
    int fst = 0, fln = 0;
    if (mband < n_mels) { fst = fb_start[mband]; fln = fb_len[mband]; }
    float wr[WPT > 0 ? WPT : 1];
    if (WPT > 0) {
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int i = mhalf + 2 * j;
            wr[j] = (mband < n_mels && i < fln) ? fb_w[(size_t)mband * fb_stride + i] : 0.f;
        }
    }
    // the next frame's samples are fetched into registers while the current frame is transformed (a workgroup would
    // otherwise expose one L2/HBM latency per frame in front of ~10 short barrier-separated phases)
    float2 smp[4];
    auto load_frame = [&](int b_, int t_) {
        const float* clip = audio + (size_t)b_ * N;
        const int base = t_ * hop - MEL_NFFT / 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int s0 = base + 2 * (tid + 256 * q), s1 = s0 + 1;
            if (s0 < 0) s0 = -s0;
            if (s1 < 0) s1 = -s1;
            if (s0 >= N) s0 = 2 * (N - 1) - s0;
            if (s1 >= N) s1 = 2 * (N - 1) - s1;
            smp[q] = make_float2(clip[s0], clip[s1]);
        }
    };
    // Work list, XCD-aware: workgroup w runs on XCD w & 7 (round-robin dispatch), and the 8 frames that overlap one hop of audio
    // must meet in ONE L2 -- with frame = blockIdx + n * gridDim they sat on 8 XCDs, every L2 fetched every clip and the launch
    // read 246 MB for 31 MB of audio (profiles/r05f_pmc_mel.md).  Clips are cut into S segments of fps frames (S = 1 when 8 | B),
    // segment sg belongs to XCD sg & 7, and the workgroups of an XCD walk its segments' frames in order, slot-strided, so the
    // workgroups resident together read one moving window of a clip.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, n_slots = gridDim.x >> 3;
    const int n_segs = B * S;
    const int n_local = xcd < n_segs ? ((n_segs - xcd + 7) >> 3) * fps : 0;
    auto locate = [&](int i_, int& b_, int& t_) {
        const int j = i_ / fps, sg = xcd + 8 * j;
        b_ = sg / S;
        t_ = (sg - b_ * S) * fps + (i_ - j * fps);
        return t_ < T;
    };
    int li = slot, b = 0, t = 0;
    bool have = false;
    while (li < n_local && !(have = locate(li, b, t))) li += n_slots;
    if (have) load_frame(b, t);
    while (have) {
        // ---- reflect-padded, windowed samples packed as z[n] = x[2n] + i x[2n+1] ----
#pragma unroll
        for (int q = 0; q < 4; ++q) buf0[mel_sw(tid + 256 * q)] = make_float2(smp[q].x * win[q].x, smp[q].y * win[q].y);
        int nb = 0, nt = 0;
        bool nhave = false;
        li += n_slots;
        while (li < n_local && !(nhave = locate(li, nb, nt))) li += n_slots;
        if (nhave) load_frame(nb, nt);
        __syncthreads();
        // ---- 5 Stockham radix-4 passes, Ns = 1,4,16,64,256; result lands in buf1 ----
        float2* src = buf0;
        float2* dst = buf1;
#pragma unroll
        for (int pass = 0; pass < 5; ++pass) {
            const int Ns = 1 << (2 * pass);
            const int k = tid & (Ns - 1);
            float2 v0 = src[mel_sw(tid)], v1 = src[mel_sw(tid + 256)], v2 = src[mel_sw(tid + 512)], v3 = src[mel_sw(tid + 768)];
            if (pass > 0) {                              // twiddles exp(-2 pi i j k / (4 Ns)), j = 1, 2, 3
                const float2* tw = stw + (Ns - 4) + k;   // 3 * (4 + 16 + ... + Ns / 4) = Ns - 4
                v1 = cmulp(v1, tw[0]);
                v2 = cmulp(v2, tw[Ns]);
                v3 = cmulp(v3, tw[2 * Ns]);
            }
            // DFT-4 (forward, e^{-i pi/2} = -i).  (Hand-picked packed instructions, as in the wave kernel below: left to itself the
            // compiler paired these sums into v_pk_add_f32 ... op_sel:[0,1] -- the form sed_common.h's "gfx950 hazard" note forbids.)
            const float2 a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3), a3 = csub(v1, v3);
            const int d = ((tid - k) << 2) + k;
            dst[mel_sw(d)] = cadd(a0, a2);
            dst[mel_sw(d + Ns)] = cadd_mi(a1, a3);            // a1 - i a3
            dst[mel_sw(d + 2 * Ns)] = csub(a0, a2);
            dst[mel_sw(d + 3 * Ns)] = cadd_pi(a1, a3);        // a1 + i a3
            __syncthreads();
            float2* tmp = src; src = dst; dst = tmp;
        }
        const float2* Z = src;   // == buf1 after an odd number of swaps
        // ---- real-FFT post-processing + magnitude: bins 0..1024 ----
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = tid + 256 * q;
            float m;
            if (k == 0) {
                const float2 z0 = Z[mel_sw(0)];
                m = fabsf(z0.x + z0.y);
                mag[MEL_M] = fabsf(z0.x - z0.y);
            } else {
                const float2 zk = Z[mel_sw(k)], zm = Z[mel_sw(MEL_M - k)];
                // Xe = (zk + conj(zm)) / 2, Xo = -i (zk - conj(zm)) / 2, X[k] = Xe + w2048^k Xo -- scalar halves, one packed product
                const float ex = 0.5f * sed_sadd(zk.x, zm.x), ey = 0.5f * sed_sadd(zk.y, -zm.y);
                const float dr = sed_sadd(zk.x, -zm.x), di = sed_sadd(zk.y, zm.y);            // zk - conj(zm)
                const float2 wx = cmulp(tw2[q], make_float2(0.5f * di, -0.5f * dr));
                const float re = sed_sadd(ex, wx.x), im = sed_sadd(ey, wx.y);
                m = sqrtf(re * re + im * im);
            }
            mag[k] = m;
        }
        __syncthreads();
        // ---- sparse triangular mel: two threads per mel band (even/odd taps) ----
        {
            const int m = mband, half = mhalf;
            float acc = 0.f;
            if (WPT > 0) {                  // same taps in the same order; the padding taps add exact zeros
#pragma unroll
                for (int j = 0; j < WPT; ++j) {
                    const int idx = fst + half + 2 * j;
                    acc = fmaf(wr[j], mag[idx <= MEL_M ? idx : MEL_M], acc);
                }
            } else if (m < n_mels) {
                const float* w = fb_w + (size_t)m * fb_stride;
                for (int i = half; i < fln; i += 2) acc = fmaf(w[i], mag[fst + i], acc);
            }
            acc += sed_quad_xor1(acc);
            if (m < n_mels && half == 0) {
                float v = acc;
                if (LOG) {
                    v = 20.0f * log10f(fmaxf(v, 1e-5f));
                    v = fminf(fmaxf(v, -50.0f), 80.0f);
                }
                out[((size_t)b * T + t) * n_mels + m] = v;
            }
        }
        __syncthreads();
        b = nb; t = nt; have = nhave;
    }
}


// ---- rounds 5 / 6: one WAVE per frame, ONE frame per wave ------------------------------------------------------------------------------
// mel_kernel above gives a frame to a 256-thread workgroup: five radix-4 passes, ten workgroup barriers per frame (141 - 150 us at
// B = 48: wait_any 0.48, LDS conflict share 0.20 -- bound by its barrier-separated LDS passes, not by HBM).  Here a frame belongs to ONE
// wave: the 1024-point packed transform is 16 x 16 x 4 -- lane l holds 16 complex values, does a radix-16 DFT in registers (two radix-4
// stages), exchanges through a wave-private 8.5 KB LDS buffer, radix-16 again, exchanges, four radix-4 -- two exchanges instead of
// four, and NO workgroup barrier after the tables are staged: a wave's DS instructions execute in order, sed_wave_sync() only pins the
// compiler.  A workgroup = MEL_WAVES waves = MEL_WAVES consecutive frames of ONE clip ("a run"), and the runs of a clip stay on one XCD
// (workgroup g is dispatched to XCD g % 8; XCD x owns the clips c = x (mod 8)), so the 7/8 overlap of neighbouring frames is served by
// that XCD's L2 and HBM sees every sample once.  LDS exchange layout (MI355X_MICROARCH.md, LDS: ds_write_b64 is banked per 16
// contiguous lanes modulo 16 slots of 8 B, ds_read_b64 per 32 lanes modulo 32): exchange 1 writes element 16 l + r at slot e + (e >> 4)
// (17 l + r: distinct modulo 16 over 16 lanes) and reads l + 64 r there (contiguous; one 2-way pair); exchange 2 (writes
// 256 (l >> 4) + (l & 15) + 16 r, reads l + 64 m + 256 r) and the conjugate-pair exchange of the real-FFT step are contiguous per lane
// group as they are.  All LDS addresses are a per-lane base + an immediate.  Mel stage: lane l owns bands l and 127 - l (3 + 42 ...
// 13 + 13 taps: balanced), taps as aligned 16-byte LDS reads (MEL_GA + MEL_GB groups), magnitudes read from the wave's buffer; longer
// bands finish from memory (no recipe has any).
// WHY ONE FRAME PER WAVE (profiles/r05_mel_graph_race.md, profiles/r06_mel_mechanism.md): round 5's form of this kernel walked runs of
// 8 or 16 frames per workgroup, i.e. a wave transformed a SECOND frame in the same launch -- and replayed as a hipGraph node while
// gemm_bf16x3_kernel ran on another branch that second frame came back with a few neighbouring bins k and their mirror bins 1024 - k
// wrong in 0.2 - 7 % of the replays (never the first frame of a wave, never in eager launches, not cured by waits / fences / nop
// padding / -O1 / plain arithmetic).  With exactly one frame per wave and launch the fault has never been seen (round 5: 0 of 32 000
// replays beside the co-runners that break the multi-frame forms; this round: tests/test_gpu_parity.py::
// test_mel_in_graph_beside_{tails,gemm}, 3 000 replays each in every GPU session).  The multi-frame form is NOT part of the library
// any more; it lives on as a diagnostics-only reproducer (tools/mel_repro/).
#ifndef MEL_WAVES
#define MEL_WAVES 4            // waves = frames per workgroup; the 26 KB of tables are staged once per workgroup
#endif
#define MEL_XPAD 1088          // 1024 + 64 padding slots (exchange 1)

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

// Workgroup = MEL_WAVES waves: the frame-invariant LDS tables (pass-2 twiddles 2 KB, the mel taps 16 KB) are shared by its waves, 8.5 KB
// of exchange buffer each -> 52 KB at four waves: THREE workgroups = twelve waves per CU = three per SIMD (142 VGPRs: with one frame per
// wave there is no frame loop for LICM to hoist 47 table reads out of -- round 5's multi-frame form needed 184 and every attempt at
// <= 168 spilled).  The window (8 KB) is not staged: every tap is used by exactly one lane of a wave, once -- it travels L2 -> registers
// beside the samples (MEL_WIN_LDS: the staged form, 60 KB = two workgroups per CU; 104 vs 84 us at B = 48, profiles/r06_mel_variants.md).
// Mel taps in LDS: band `lane` as MEL_GA and band 127 - lane as MEL_GB groups of four taps, the first group starting at the band's
// first bin rounded DOWN to a multiple of four (leading / trailing zeros), stored [group][lane] -- every lane reads its 16 bytes of group
// g at the same offset (conflict-free), and the magnitudes as aligned 16-byte reads too: 32 ds_read_b128 per frame instead of 120 b32.
#ifndef MEL_OCC
#define MEL_OCC (12 / MEL_WAVES)
#endif
#define MEL_GA 4
#define MEL_GB 12

template <bool LOG>
__global__ __launch_bounds__(64 * MEL_WAVES, MEL_OCC) void mel_wave_kernel(const float* __restrict__ audio, float* __restrict__ out,
                                                       int B, int N, int T, int hop, int n_mels,
                                                       const float* __restrict__ window, const float2* __restrict__ tw1024,
                                                       const float2* __restrict__ tw2048, const int* __restrict__ fb_start,
                                                       const int* __restrict__ fb_len, const float* __restrict__ fb_w, int fb_stride,
                                                       const float4* __restrict__ taps, int runs_per_clip, int segs_per_clip) {
#ifdef MEL_WIN_LDS
    __shared__ float2 s_win[MEL_M];                             // (w[2n], w[2n + 1])
#endif
    __shared__ float2 s_tw16[15 * 16];                          // [r - 1][k]: e^{-2 pi i r k / 256}, r = 1..15, k < 16 (pass 2)
    __shared__ __attribute__((aligned(16))) float4 s_wa[MEL_GA][64];      // taps of band `lane`
    __shared__ __attribute__((aligned(16))) float4 s_wb[MEL_GB][64];      // taps of band 127 - lane
    __shared__ __attribute__((aligned(16))) float2 s_x[MEL_WAVES][MEL_XPAD];     // one exchange buffer per wave
    const int tid = threadIdx.x, wave = sed_wave_uniform(tid >> 6), lane = tid & 63;

    // XCD-aware placement: workgroup g sits on XCD g & 7 and takes ONE run of the SEGMENTS s = (g & 7) (mod 8); a segment is a clip
    // (segs_per_clip = 1 at the recipes' batch sizes) or, for small batches, one of several stretches of consecutive runs of a clip.
    // (All three exits below are taken by whole workgroups: nobody is left waiting at the barrier.)
    const int xcd = blockIdx.x & 7, run = blockIdx.x >> 3;
    const int runs_per_seg = (runs_per_clip + segs_per_clip - 1) / segs_per_clip;
    const int segs_here = (B * segs_per_clip - xcd + 7) >> 3;
    if (run >= segs_here * runs_per_seg) return;
    const int seg = xcd + 8 * (run / runs_per_seg);
    const int b = seg / segs_per_clip;
    const int run_in_clip = (seg - b * segs_per_clip) * runs_per_seg + run % runs_per_seg;
    if (run_in_clip >= runs_per_clip) return;
    const int t = run_in_clip * MEL_WAVES + wave;               // this wave's frame
    const bool have = t < T;
    const float* clip = audio + (size_t)b * N;

    // ---- the frame's samples leave for the registers BEFORE the tables are staged: one memory latency for both ----
    float2 v[16];
    if (have) {
        const int base = t * hop - MEL_NFFT / 2;
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
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = make_float2(0.f, 0.f);
    }
#ifndef MEL_WIN_LDS
    float2 wn[16];                              // every window tap is used once per wave: straight from L2 beside the samples
#pragma unroll
    for (int j = 0; j < 16; ++j) wn[j] = reinterpret_cast<const float2*>(window)[lane + 64 * j];
#else
    for (int i = tid; i < MEL_M; i += 64 * MEL_WAVES) s_win[i] = make_float2(window[2 * i], window[2 * i + 1]);
#endif
    if (tid < 240) s_tw16[tid] = tw1024[(4 * ((tid >> 4) + 1) * (tid & 15)) & (MEL_M - 1)];
    // (the tap tables come ready-made from sed_mel_taps: built per workgroup from fb_start / fb_len / fb_w they were eleven rounds of
    //  dependent gathers -- ~20 us in front of a workgroup's frames)
    for (int i = tid; i < (MEL_GA + MEL_GB) * 64; i += 64 * MEL_WAVES) {
        if (i < MEL_GA * 64) s_wa[i >> 6][i & 63] = taps[i];
        else s_wb[(i >> 6) - MEL_GA][i & 63] = taps[i];
    }
    // The only per-lane twiddle kept in registers is w2048^lane: the real-FFT twiddles w2048^(lane + 64 q) = w2048^lane w32^q and
    // the pass-3 twiddles w1024^(r (lane + 64 m)) = ((w2048^lane)^2 w16^m)^r are formed from it (a dozen packed products)
    const float2 wl = tw2048[lane];
    // mel bands of this lane: A = lane, Bd = 127 - lane
    const int bandA = lane, bandB = 127 - lane;
    int sA = 0, lA = 0, sB = 0, lB = 0;
    if (bandA < n_mels) { sA = fb_start[bandA]; lA = fb_len[bandA]; }
    if (bandB < n_mels) { sB = fb_start[bandB]; lB = fb_len[bandB]; }
    const int gA0 = sA >> 2, gB0 = sB >> 2;     // first aligned group of four magnitudes of each band
    __syncthreads();                            // the only workgroup barrier
    if (!have) return;                          // (the last run of a clip: T is not a multiple of MEL_WAVES)

    float2* xb = s_x[wave];
    float* magb = reinterpret_cast<float*>(xb);
    // ---- window; pass 1: radix 16, Ns = 1: in[lane + 64 r] -> out[16 lane + r] ----
#pragma unroll
#ifndef MEL_WIN_LDS
    for (int j = 0; j < 16; ++j) v[j] = cscale(v[j], wn[j]);
#else
    for (int j = 0; j < 16; ++j) v[j] = cscale(v[j], s_win[lane + 64 * j]);
#endif
    dft16(v);
    sed_sched_fence();
#pragma unroll
    for (int r = 0; r < 16; ++r) xb[17 * lane + r] = v[r];
    sed_wave_sync(); sed_sched_fence();
    // ---- pass 2: radix 16, Ns = 16: in[lane + 64 r] * w256^(r k), k = lane & 15 -> out[(lane - k) 16 + k + 16 r] ----
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = xb[lane + (lane >> 4) + 68 * r];        // e + (e >> 4), e = lane + 64 r
#pragma unroll
    for (int r = 1; r < 16; ++r) v[r] = cmulp(v[r], s_tw16[(r - 1) * 16 + (lane & 15)]);
    dft16(v);
    sed_wave_sync(); sed_sched_fence();
#pragma unroll
    for (int r = 0; r < 16; ++r) xb[((lane >> 4) << 8) + (lane & 15) + 16 * r] = v[r];
    sed_wave_sync(); sed_sched_fence();
    // ---- pass 3: radix 4, Ns = 256, four butterflies per lane: j = lane + 64 m, in[j + 256 r] * w1024^(r j) -> out[j + 256 r]
    const float2 wl2 = cmulp(wl, wl);                                            // w1024^lane
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
    // ---- real-FFT step: X[k] = Xe + w2048^k Xo needs Z[1024 - k]: conjugate-pair exchange through the buffer ----
    sed_wave_sync(); sed_sched_fence();
#pragma unroll
    for (int q = 0; q < 16; ++q) xb[lane + 64 * q] = v[q];
    if (lane == 0) xb[MEL_M] = v[0];                                             // Z[1024] = Z[0]
    sed_wave_sync(); sed_sched_fence();
    float mg[16], mag_nyq = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int k = lane + 64 * q;
        const float2 zk = v[q], zm = xb[MEL_M - k];
        const float2 xe = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        const float dr = zk.x - zm.x, di = zk.y + zm.y;                          // zk - conj(zm)
        const float2 xo = make_float2(0.5f * di, -0.5f * dr);                     // -i/2 (zk - conj(zm))
        // w2048^(lane + 64 q) = w2048^lane * w32^q, w32^q = (cos, -sin)(2 pi q / 32)
        const float ang = 0.19634954084936207f * q;                              // folded: q is a compile-time constant
        const float2 wq = cmulp(wl, make_float2(__builtin_cosf(ang), -__builtin_sinf(ang)));
        const float2 wx = cmulp(wq, xo);
        const float re = xe.x + wx.x, im = xe.y + wx.y;
        mg[q] = fast_sqrt(re * re + im * im);
        if (q == 0) {                                                            // k = 0 (lane 0): DC and Nyquist are real
            if (lane == 0) { mg[0] = fabsf(zk.x + zk.y); mag_nyq = fabsf(zk.x - zk.y); }
        }
    }
    sed_wave_sync(); sed_sched_fence();
#pragma unroll
    for (int q = 0; q < 16; ++q) magb[lane + 64 * q] = mg[q];
    if (lane == 0) magb[MEL_M] = mag_nyq;
    sed_wave_sync(); sed_sched_fence();
    // ---- sparse HTK mel: bands `lane` and `127 - lane`; four partial sums per band ----
    const float4* mag4 = reinterpret_cast<const float4*>(magb);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
#pragma unroll
    for (int g = 0; g < MEL_GA; ++g) {
        const float4 w = s_wa[g][lane], m = mag4[gA0 + g];
        a0 = fmaf(w.x, m.x, a0); a1 = fmaf(w.y, m.y, a1); a2 = fmaf(w.z, m.z, a2); a3 = fmaf(w.w, m.w, a3);
    }
    sed_sched_fence();
#pragma unroll
    for (int g = 0; g < MEL_GB; ++g) {
        const float4 w = s_wb[g][lane], m = mag4[gB0 + g];
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
    float* o = out + ((size_t)b * T + t) * n_mels;
    if (bandA < n_mels) o[bandA] = accA;
    if (bandB < n_mels) o[bandB] = accB;
}

// taps[(g * 64 + lane) * 4 + c]: groups g < MEL_GA = band `lane`, the others band 127 - lane; tap j = 4 g' + c - (start & 3) of the band
__global__ void mel_taps_kernel(const int* __restrict__ fb_start, const int* __restrict__ fb_len, const float* __restrict__ fb_w,
                                int fb_stride, int n_mels, float* __restrict__ taps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (MEL_GA + MEL_GB) * 64 * 4) return;
    const int c = i & 3, ln_ = (i >> 2) & 63, g = i >> 8;
    const bool isA = g < MEL_GA;
    const int band = isA ? ln_ : 127 - ln_;
    float w = 0.f;
    if (band < n_mels) {
        const int st = fb_start[band], j = 4 * (isA ? g : g - MEL_GA) + c - (st & 3);
        if (j >= 0 && j < fb_len[band]) w = fb_w[(size_t)band * fb_stride + j];
    }
    taps[i] = w;
}

SED_API int sed_mel_taps(const int* fb_start, const int* fb_len, const float* fb_w, int fb_stride, int n_mels, float* taps, void* stream) {
    if (n_mels > 128 || n_mels < 1) return SED_ERR_UNSUPPORTED;
    const int n = (MEL_GA + MEL_GB) * 64 * 4;
    SED_LAUNCH(mel_taps_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, fb_start, fb_len, fb_w, fb_stride, n_mels, taps);
    return sed_check_launch();
}

SED_API int sed_mel_fwd_wave(const float* audio, float* out, int B, int N, int T, int n_fft, int hop, int n_mels,
                             const float* window, const float* tw1024, const float* tw2048, const int* fb_start,
                             const int* fb_len, const float* fb_w, int fb_stride, const float* taps, int apply_log, void* stream) {
    if (n_fft != MEL_NFFT || n_mels > 128 || n_mels < 1 || N < n_fft / 2 + 1 || T != 1 + N / hop || !taps) return SED_ERR_UNSUPPORTED;
    if (B <= 0) return SED_OK;
    // one wave per frame and one frame per wave: a workgroup takes a run of MEL_WAVES consecutive frames of one clip, clips pinned to
    // XCDs (see mel_wave_kernel)
    const int runs_per_clip = (T + MEL_WAVES - 1) / MEL_WAVES;
    int segs_per_clip = B >= 32 ? 1 : 32 / B;       // small batches: several stretches per clip, so that all eight XCDs work
    if (segs_per_clip > runs_per_clip) segs_per_clip = runs_per_clip;
    // every XCD gets as many workgroups as it has runs (its segments x runs per segment)
    const int rps_ = (runs_per_clip + segs_per_clip - 1) / segs_per_clip;
    const long long grid_ll = 8LL * ((B * (long long)segs_per_clip + 7) / 8) * rps_;
    if (grid_ll > 0x7fffffffLL) return SED_ERR_UNSUPPORTED;
    const int grid = (int)grid_ll;
#define MELW_LAUNCH(LOG) SED_LAUNCH((mel_wave_kernel<LOG>), dim3(grid), dim3(64 * MEL_WAVES), 0, (hipStream_t)stream, audio, out, B, N, T, hop, \
                                    n_mels, window, (const float2*)tw1024, (const float2*)tw2048, fb_start, fb_len, fb_w, fb_stride,             \
                                    (const float4*)taps, runs_per_clip, segs_per_clip)
    if (apply_log) MELW_LAUNCH(true); else MELW_LAUNCH(false);
#undef MELW_LAUNCH
    return sed_check_launch();
}

SED_API int sed_mel_fwd(const float* audio, float* out, int B, int N, int T, int n_fft, int hop, int n_mels,
                           const float* window, const float* tw1024, const float* tw2048, const int* fb_start,
                           const int* fb_len, const float* fb_w, int fb_stride, int apply_log, void* stream) {
    if (n_fft != MEL_NFFT || n_mels > 128 || n_mels < 1 || N < n_fft / 2 + 1 || T != 1 + N / hop) return SED_ERR_UNSUPPORTED;
    if (B <= 0) return SED_OK;
    int S = 1;                                    // segments per clip: the smallest count that makes B * S a multiple of 8 XCDs
    while ((B * S) & 7) S <<= 1;
    const int fps = (T + S - 1) / S;
    const long long slots = ((long long)B * T + 7) / 8;
    const int grid = 8 * (int)(slots < 512 ? slots : 512);
#define MEL_LAUNCH(LOG, WPT) SED_LAUNCH((mel_kernel<LOG, WPT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, audio, out, B, N, T, hop, \
                                        n_mels, window, (const float2*)tw1024, (const float2*)tw2048, fb_start, fb_len, fb_w, fb_stride, S, fps)
    if (fb_stride <= 48 && !sed_tuning[SED_TUNE_MEL_TAPS_MEM]) {      // the recipes' filterbank (128 HTK bands up to 8 kHz: widest band 46 bins)
        if (apply_log) MEL_LAUNCH(true, 24); else MEL_LAUNCH(false, 24);
    } else {
        if (apply_log) MEL_LAUNCH(true, 0); else MEL_LAUNCH(false, 0);
    }
#undef MEL_LAUNCH
    return sed_check_launch();
}
