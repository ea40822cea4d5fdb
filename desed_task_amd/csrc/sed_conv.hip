// K6 (convolution half): 3x3 / stride 1 / pad 1 convolutions of the CNN encoder
// (desed_task/nnet/CNN.py:66-76: nn.Conv2d + the batch statistics nn.BatchNorm2d needs), forward,
// data-gradient and weight-gradient, as implicit GEMMs on the exact-f32 MFMA
// (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf chain, so parity with the fp32 reference is
// rounding-order only).
//
// Layouts: activations channels-last (B, T, F, C) fp32; packed weights Wp[tap][cin][cout] (K-major for
// the MFMA B operand).  The forward kernel also emits per-workgroup partial (sum, sum-of-squares) per
// output channel; bn_finalize reduces them in double and produces mean / invstd / scale / shift and
// the running-stat update (BatchNorm2d eps=1e-3, momentum=0.99, CNN.py:76).
//
// Workgroup = 4 (M) x WN (N) waves; block tile = 128 output pixels (TR x TF patch) x all COUT; WN = 2 for
// COUT >= 64 (8 waves = 2 per SIMD, so one wave's LDS operand reads hide under the other's 64-cycle MFMAs);
// each wave owns 32 pixels x COUT/WN (accumulators of 16 VGPRs per 32 channels).  K loop = (cin chunk of <=32)
// x 9 taps: the halo patch of the chunk sits in LDS (row stride CK+1 dwords: conflict-free A reads),
// the tap's CK x COUT weight slab is double-buffered in LDS and prefetched through registers while
// the previous tap's MFMAs run; one barrier per tap.
#include "sed_common.h"

// ---------------------------------------------------------------------------------------------
// weight packing: W (COUT, CIN, 3, 3) -> Wf[tap][ci][co] ; Wd[tap'][co][ci] = W[co][ci][2-a'][2-b']
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ W, float* __restrict__ Wf,
                                                           float* __restrict__ Wd, int COUT, int CIN) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= COUT * CIN * 9) return;
    const int b = i % 3, a = (i / 3) % 3, ci = (i / 9) % CIN, co = i / (9 * CIN);
    const float w = W[i];
    Wf[((a * 3 + b) * CIN + ci) * COUT + co] = w;
    if (Wd) Wd[(((2 - a) * 3 + (2 - b)) * COUT + co) * CIN + ci] = w;
}
SED_API int sed_conv_pack_weights(const float* W, float* Wf, float* Wd, int COUT, int CIN, void* stream) {
    const int n = COUT * CIN * 9;
    SED_LAUNCH(pack_weights_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, Wf, Wd, COUT, CIN);
    return sed_check_launch();
}
// All conv layers of one model in ONE launch (student / teacher weights change every step).
struct PackJobs {
    const float* W[8];
    float* Wf[8];
    float* Wd[8];
    int cout[8], cin[8], start[9];
    int n;
};
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(PackJobs jobs) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= jobs.start[jobs.n]) return;
    int j = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) j += (q < jobs.n && i >= jobs.start[q]) ? 1 : 0;
    const int e = i - jobs.start[j], COUT = jobs.cout[j], CIN = jobs.cin[j];
    const int b = e % 3, a = (e / 3) % 3, ci = (e / 9) % CIN, co = e / (9 * CIN);
    const float w = jobs.W[j][e];
    jobs.Wf[j][((a * 3 + b) * CIN + ci) * COUT + co] = w;
    if (jobs.Wd[j]) jobs.Wd[j][(((2 - a) * 3 + (2 - b)) * COUT + co) * CIN + ci] = w;
}
// n <= 8 layers; W/Wf/Wd: host arrays of n device pointers (Wd entries may be null); cout/cin: host int arrays.
SED_API int sed_conv_pack_multi(int n, const void* const* W, void* const* Wf, void* const* Wd, const int* cout, const int* cin,
                                   void* stream) {
    if (n < 1 || n > 8) return SED_ERR_ARG;
    PackJobs jobs;
    int tot = 0;
    for (int j = 0; j < n; ++j) {
        jobs.W[j] = (const float*)W[j]; jobs.Wf[j] = (float*)Wf[j]; jobs.Wd[j] = Wd ? (float*)Wd[j] : nullptr;
        jobs.cout[j] = cout[j]; jobs.cin[j] = cin[j]; jobs.start[j] = tot;
        tot += cout[j] * cin[j] * 9;
    }
    for (int j = n; j < 8; ++j) { jobs.W[j] = nullptr; jobs.Wf[j] = nullptr; jobs.Wd[j] = nullptr; jobs.cout[j] = 0; jobs.cin[j] = 0; jobs.start[j] = tot; }
    jobs.start[n] = tot;
    jobs.start[8] = tot;
    jobs.n = n;
    SED_LAUNCH(pack_weights_multi_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, jobs);
    return sed_check_launch();
}

// partials[part][tap][ci][co] -> dW (COUT, CIN, 3, 3): fixed-order (deterministic) sum over the workgroup partials.
// One workgroup per 64 consecutive packed outputs: 16 float4 columns x G = blockDim/16 groups of partials (the narrow
// layers have ~1000 partials of only 4.6K outputs: the partial range, not the output range, carries the parallelism).
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ parts, float* __restrict__ dW, int nparts,
                                                            int COUT, int CIN) {
    __shared__ float4 red[64][16];
    const int tid = threadIdx.x, col = tid & 15, grp = tid >> 4, G = blockDim.x >> 4;
    const int j = blockIdx.x * 64 + 4 * col;               // packed index (tap, ci, co): coalesced float4 reads
    const int n = 9 * CIN * COUT;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < n) {
#pragma unroll 4
        for (int p = grp; p < nparts; p += G) {
            const float4 v = *(const float4*)(parts + (size_t)p * n + j);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[grp][col] = acc;
    __syncthreads();
    if (grp == 0 && j < n) {
        for (int g = 1; g < G; ++g) { const float4 v = red[g][col]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        const int co = j % COUT, ci = (j / COUT) % CIN, tap = j / (COUT * CIN);     // COUT % 4 == 0: one (tap, ci) per float4
        float* d = dW + ((size_t)co * CIN + ci) * 9 + tap;
        d[0] = acc.x; d[(size_t)CIN * 9] = acc.y; d[(size_t)2 * CIN * 9] = acc.z; d[(size_t)3 * CIN * 9] = acc.w;
    }
}
// number of workgroup partials the weight-gradient launch writes (each 9*CIN*COUT floats)
static inline int wgrad_parts(int CIN, int COUT, int B, int T, int F) {
    if (CIN <= 32) {                                    // all-taps kernel: one partial per workgroup
        const int TF = F >= 32 ? 32 : F, TR = 128 / TF;
        const int ntiles = B * ((T + TR - 1) / TR) * (F / TF);
        int cap = CIN <= 16 ? 1024 : 512;             // 4 / 2 resident workgroups per CU: staging of one overlaps MFMAs of another
        if (sed_tuning[SED_TUNE_WGRAD_CAP] > 0) cap = sed_tuning[SED_TUNE_WGRAD_CAP];     // tests: several tiles per workgroup on small inputs
        return ntiles < cap ? ntiles : cap;
    }
    const int ntiles = (B * T * F + 63) / 64;           // per-tap kernel: `splits` partials
    return ntiles < 56 ? ntiles : 56;
}
// the kernel-row variant of the wide split-bf16 weight gradient runs 3 x (COUT / COH) workgroups per split: 168 / 84 splits for
// the same ~504 resident workgroups (with 56 it left half of the CUs empty: 131 vs 75 us at 64 -> 128, F = 16)
static inline bool wgrad_row_ok(int CIN, int COUT, int F) {
    return F >= 8 && 64 % F == 0 && COUT == 128 && (CIN == 64 || CIN == 128);
}
static inline int wgrad_row_parts(int CIN, int B, int T, int F) {
    const int ntiles = (B * T * F + 63) / 64, cap = CIN == 64 ? 168 : 84;
    return ntiles < cap ? ntiles : cap;
}
SED_API long long sed_conv_wgrad_scratch_floats(int B, int T, int F, int CIN, int COUT) {
    int parts = wgrad_parts(CIN, COUT, B, T, F);
    if (wgrad_row_ok(CIN, COUT, F) && wgrad_row_parts(CIN, B, T, F) > parts) parts = wgrad_row_parts(CIN, B, T, F);
    return (long long)parts * 9 * CIN * COUT;
}

// ---------------------------------------------------------------------------------------------
// generic 3x3 conv, implicit GEMM (also used for dgrad with flipped/transposed weights)
// ---------------------------------------------------------------------------------------------
#define CONV_THREADS(COUT) ((COUT) >= 64 ? 512 : 256)

// MP = output pixels per workgroup: 128, or 64 for the late 128-channel layers whose 128-pixel grids would leave
// most of the 256 CUs without a second workgroup (F <= 4: 117 / 234 tiles).  Waves: WM = MP/32 along pixels,
// WN along COUT, WM*WN = 8 for COUT >= 64 (4 otherwise).
template <int CIN, int COUT, int TF, int MP = 128>
struct ConvCfg {
    static constexpr int TR = MP / TF;
    static constexpr int PW = TF + 2, PH = TR + 2, PP = PW * PH;
    static constexpr int CK = CIN < 32 ? CIN : 32;
    static constexpr int CKP = CK + 1;
    static constexpr int NCH = CIN / CK;
    static constexpr int NT = (COUT + 31) / 32;
    static constexpr int WM = MP / 32;
    static constexpr int THREADS = CONV_THREADS(COUT);
    static constexpr int WN = THREADS / 64 / WM;        // waves along N
    static constexpr int NTW = NT / WN;                 // accumulator tiles per wave
    static_assert(NTW >= 1 && NTW * WN == NT, "wave split must tile COUT");
    static constexpr int WCH = CK * COUT;
    static constexpr int PATCH_F = (PP * CKP + 3) & ~3;
    static constexpr int SMEM = (PATCH_F + 2 * WCH) * 4;
};

template <int CIN, int COUT, int TF, bool STATS, int MP = 128>
__global__ __launch_bounds__(CONV_THREADS(COUT)) void conv3x3_kernel(const float* __restrict__ x, const float* __restrict__ Wp,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      float* __restrict__ partial, int B, int T, int F) {
    using Cfg = ConvCfg<CIN, COUT, TF, MP>;
    constexpr int TR = Cfg::TR, PW = Cfg::PW, PP = Cfg::PP, CK = Cfg::CK, CKP = Cfg::CKP, NCH = Cfg::NCH, NT = Cfg::NT,
                  WCH = Cfg::WCH, NTW = Cfg::NTW, THREADS = Cfg::THREADS, WM = Cfg::WM;
    SED_DYN_SMEM(smem);
    float* patch = (float*)smem;
    float* wbuf = patch + Cfg::PATCH_F;
    const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) % WM, wn = (tid >> 6) / WM, lo = lane & 31, hi = lane >> 5;
    const int ftiles = F / TF, ttiles = (T + TR - 1) / TR;
    const int bid = blockIdx.x;
    const int ft = bid % ftiles, tt = (bid / ftiles) % ttiles, b = bid / (ftiles * ttiles);
    const int t0 = tt * TR, f0 = ft * TF;
    const int p = 32 * w + lo;
    const int abase = ((p / TF) * PW + (p % TF)) * CKP + hi;

    f32x16 acc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[nt] = f32x16_zero();

    constexpr int WV = (WCH / 4 + THREADS - 1) / THREADS;
    float4 wreg[WV];

    for (int cc = 0; cc < NCH; ++cc) {
        // ---- stage the halo patch of this cin chunk ----
        constexpr int V = CK / 4;
        {   // all loads of the patch first, LDS stores after: one memory latency per chunk instead of one per load
            constexpr int NLD = (PP * V + THREADS - 1) / THREADS;
            float4 ld[NLD];
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int idx = tid + THREADS * u;
                const int pix = idx / V, v = idx - pix * V;
                const int i = pix / PW, j = pix - i * PW;
                const int t = t0 - 1 + i, f = f0 - 1 + j;
                ld[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < PP * V && t >= 0 && t < T && f >= 0 && f < F)
                    ld[u] = *(const float4*)(x + (((size_t)b * T + t) * F + f) * CIN + cc * CK + 4 * v);
            }
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int idx = tid + THREADS * u;
                if (idx < PP * V) {
                    const int pix = idx / V, v = idx - pix * V;
                    float* d = patch + pix * CKP + 4 * v;
                    d[0] = ld[u].x; d[1] = ld[u].y; d[2] = ld[u].z; d[3] = ld[u].w;
                }
            }
        }
        // ---- tap 0 weights straight to LDS buffer 0 ----
        {
            const float4* src = (const float4*)(Wp + ((size_t)0 * CIN + cc * CK) * COUT);
#pragma unroll
            for (int i = 0; i < WV; ++i) {
                const int idx = tid + THREADS * i;
                if (idx < WCH / 4) ((float4*)wbuf)[idx] = src[idx];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) {
                const float4* src = (const float4*)(Wp + ((size_t)(tap + 1) * CIN + cc * CK) * COUT);
#pragma unroll
                for (int i = 0; i < WV; ++i) {
                    const int idx = tid + THREADS * i;
                    float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);     // (through a temporary: a guarded `wreg[i] = *p` sends wreg to scratch)
                    if (idx < WCH / 4) wv = src[idx];
                    wreg[i] = wv;
                }
            }
            const float* wb = wbuf + (tap & 1) * WCH;
            const float* ap = patch + abase + ((tap / 3) * PW + (tap % 3)) * CKP;
#pragma unroll
            for (int k = 0; k < CK; k += 2) {
                const float av = ap[k];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const float bv = (COUT >= 32 || lo < COUT) ? wb[(k + hi) * COUT + (wn * NTW + nt) * 32 + lo] : 0.f;
                    acc[nt] = mfma32(av, bv, acc[nt]);
                }
            }
            if (tap + 1 < 9) {
                float4* dst = (float4*)(wbuf + ((tap + 1) & 1) * WCH);
#pragma unroll
                for (int i = 0; i < WV; ++i) {
                    const int idx = tid + THREADS * i;
                    if (idx < WCH / 4) dst[idx] = wreg[i];
                }
            }
            __syncthreads();
        }
    }
    // ---- epilogue: bias, store, per-channel partial statistics ----
    float* red = wbuf;   // free after the last barrier: [4 waves][2][NT*32]
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int co = (wn * NTW + nt) * 32 + lo;
        const float bv = (bias != nullptr && co < COUT) ? bias[co] : 0.f;
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pp = 32 * w + mfma32_row(r, lane);
            const int t = t0 + pp / TF, f = f0 + pp % TF;
            if (t < T && co < COUT) {
                const float v = acc[nt][r] + bv;
                y[(((size_t)b * T + t) * F + f) * COUT + co] = v;
                s += v;
                s2 += v * v;
            }
        }
        if (STATS) {
            s += __shfl_xor(s, 32);
            s2 += __shfl_xor(s2, 32);
            if (hi == 0) {
                red[(w * 2 + 0) * (NT * 32) + co] = s;
                red[(w * 2 + 1) * (NT * 32) + co] = s2;
            }
        }
    }
    if (STATS) {
        __syncthreads();
        if (tid < 2 * COUT) {
            const int which = tid / COUT, co = tid - which * COUT;
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < WM; ++ww) v += red[(ww * 2 + which) * (NT * 32) + co];
            partial[(size_t)tid * gridDim.x + bid] = v;       // [2*COUT][nblocks]: bn_finalize reads one channel's row contiguously
        }
    }
}

template <int CIN, int COUT, int TF, int MP = 128>
static int launch_conv(const float* x, const float* Wp, const float* bias, float* y, float* partial, int B, int T, int F,
                       hipStream_t s) {
    using Cfg = ConvCfg<CIN, COUT, TF, MP>;
    const int nblk = B * ((T + Cfg::TR - 1) / Cfg::TR) * (F / TF);
    if (partial) {
        SED_MAX_SMEM((conv3x3_kernel<CIN, COUT, TF, true, MP>), Cfg::SMEM);
        SED_LAUNCH((conv3x3_kernel<CIN, COUT, TF, true, MP>), dim3(nblk), dim3(Cfg::THREADS), Cfg::SMEM, s, x, Wp, bias, y, partial, B, T, F);
    } else {
        SED_MAX_SMEM((conv3x3_kernel<CIN, COUT, TF, false, MP>), Cfg::SMEM);
        SED_LAUNCH((conv3x3_kernel<CIN, COUT, TF, false, MP>), dim3(nblk), dim3(Cfg::THREADS), Cfg::SMEM, s, x, Wp, bias, y, partial, B, T, F);
    }
    return sed_check_launch();
}

static inline int conv_tf(int F) { return F >= 32 ? 32 : F; }
static inline int conv_mp(int F, int CIN, int COUT) { return (CIN == 128 && COUT == 128 && F <= 4) ? 64 : 128; }

// number of workgroups (= rows of `partial`, each 2*COUT floats) the forward launch uses
SED_API int sed_conv_fwd_blocks(int B, int T, int F, int CIN, int COUT) {
    if (CIN == 1) return B * ((T + 15) / 16);
    const int TF = conv_tf(F);
    const int TR = conv_mp(F, CIN, COUT) / TF;
    return B * ((T + TR - 1) / TR) * (F / TF);
}

// x (B,T,F,CIN), Wp packed [9][CIN][COUT], bias [COUT] or null, y (B,T,F,COUT), partial: null or
// [sed_conv_fwd_blocks][2*COUT] floats.
SED_API int sed_conv3x3(const float* x, const float* Wp, const float* bias, float* y, float* partial, int B, int T, int F,
                           int CIN, int COUT, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (B <= 0 || T <= 0) return SED_OK;
    const int TF = conv_tf(F);
    if (F % TF != 0 || (F & (F - 1)) != 0 || F < 2) return SED_ERR_UNSUPPORTED;
    if (CIN == 128 && COUT == 128 && TF == 4) return launch_conv<128, 128, 4, 64>(x, Wp, bias, y, partial, B, T, F, s);
    if (CIN == 128 && COUT == 128 && TF == 2) return launch_conv<128, 128, 2, 64>(x, Wp, bias, y, partial, B, T, F, s);
#define CONV_CASE(ci, co, tf) \
    if (CIN == ci && COUT == co && TF == tf) return launch_conv<ci, co, tf>(x, Wp, bias, y, partial, B, T, F, s);
    CONV_CASE(16, 32, 32) CONV_CASE(32, 64, 32) CONV_CASE(64, 128, 16)
    CONV_CASE(128, 128, 8)
    CONV_CASE(32, 16, 32) CONV_CASE(64, 32, 32) CONV_CASE(128, 64, 16)
    // small-shape variants used by the unit tests / other n_mels
    CONV_CASE(16, 32, 16) CONV_CASE(32, 64, 8) CONV_CASE(64, 128, 4) CONV_CASE(128, 128, 16) CONV_CASE(128, 128, 32)
    CONV_CASE(32, 16, 16) CONV_CASE(64, 32, 8) CONV_CASE(128, 64, 4) CONV_CASE(64, 128, 32) CONV_CASE(64, 128, 8)
    CONV_CASE(32, 64, 16) CONV_CASE(64, 32, 16) CONV_CASE(128, 64, 8) CONV_CASE(128, 64, 32)
#undef CONV_CASE
    return SED_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// layer 0: CIN = 1 -> COUT = 16, direct VALU (K = 9 is not a dense contraction; three 16x16x4 f32 MFMAs per 16 pixels were measured
// in round 3: 47.6 vs 41.2 us).  Fuses the SpecAugment predicate (CRNN.py:207-219) into the load.  Tile = 16 frames x F mel bins.
// ---------------------------------------------------------------------------------------------
#define C0_TR 16
// Lane layout: 4 lanes per pixel, each owning 4 of the 16 output channels -> one wave store covers 16 pixels x 64 B
// = 1 KB contiguous (the kernel is bound by writing y0, 64 B per pixel).
template <int COUT>
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                    const float* __restrict__ bias, const int* __restrict__ bounds,
                                                    float* __restrict__ y, float* __restrict__ partial, int B, int T, int F) {
    static_assert(COUT == 16, "4 lanes x 4 channels");
    __shared__ float tile[(C0_TR + 2) * (128 + 2)];
    __shared__ float red[4][2 * COUT];
    const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * C0_TR, PW = F + 2;
    const int cq = tid & 3;                                    // channel quad
    float wreg[4][9], breg[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        breg[c] = bias ? bias[4 * cq + c] : 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) wreg[c][k] = W[(4 * cq + c) * 9 + k];
    }
    int mf0 = 0, mf1 = 0, mt0 = 0, mt1 = 0;
    if (bounds) { mf0 = bounds[4 * b]; mf1 = bounds[4 * b + 1]; mt0 = bounds[4 * b + 2]; mt1 = bounds[4 * b + 3]; }
    if (!sed_stage_halo_f4<C0_TR, 256, 128>(tile, x, bounds, b, t0, T, F, PW, 0.f)) {
        // (scalar fallback for widths that are not 4 x a power of two) halo tile: unconditional loads (clamped address, predicate applied to the value), all in flight before the first store --
        // with the load inside the bounds branch every one of the ten was followed by s_waitcnt vmcnt(0) (tools/isa_exposed_loads.py)
        const int n = (C0_TR + 2) * PW;
        constexpr int NIT = ((C0_TR + 2) * (128 + 2) + 255) / 256;
        float v[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int idx = tid + 256 * u, ic = idx < n ? idx : n - 1;
            const int ii = ic / PW, j = ic - ii * PW;
            const int t = t0 - 1 + ii, f = j - 1;
            const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t), fc = f < 0 ? 0 : (f >= F ? F - 1 : f);
            const float xv = x[((size_t)b * T + tc) * F + fc];
            const bool ok = t >= 0 && t < T && f >= 0 && f < F && !((f >= mf0 && f < mf1) || (t >= mt0 && t < mt1));
            v[u] = ok ? xv : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int idx = tid + 256 * u;
            if (idx < n) tile[idx] = v[u];
        }
    }
    __syncthreads();
    // Round 4: a lane owns a VERTICAL PAIR of pixels (rows 2 rp, 2 rp + 1 of one bin) for its four channels: twelve LDS values feed
    // 36 packed FMAs (the six taps of the two middle rows serve both pixels) where one pixel per lane read nine values for 18 -- the
    // ISA of the one-pixel loop was 55 instructions of which 18 were FMAs.  Every output is still its own fmaf chain over the taps
    // 0..8 in the same order (two channels per v_pk_fma_f32), so y keeps its bits; the statistics are summed on the packed pipe
    // too, and the item -> (row pair, bin) split is carried incrementally (no runtime-F division per item).
    f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
    int rp = (tid >> 2) / F, pc = (tid >> 2) - rp * F;
    for (int p = tid >> 2; p < (C0_TR / 2) * F; p += 64) {
        const int ta = t0 + 2 * rp;
        if (ta < T) {
            float in[4][3];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int bb = 0; bb < 3; ++bb) in[a][bb] = tile[(2 * rp + a) * PW + pc + bb];
            f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f}, b01 = {0.f, 0.f}, b23 = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const f32x2 w01 = {wreg[0][k], wreg[1][k]}, w23 = {wreg[2][k], wreg[3][k]};
                const float xa = in[k / 3][k % 3], xb = in[k / 3 + 1][k % 3];
                a01 = pk_fma(f32x2{xa, xa}, w01, a01);
                a23 = pk_fma(f32x2{xa, xa}, w23, a23);
                b01 = pk_fma(f32x2{xb, xb}, w01, b01);
                b23 = pk_fma(f32x2{xb, xb}, w23, b23);
            }
            const f32x2 bias01 = {breg[0], breg[1]}, bias23 = {breg[2], breg[3]};
            a01 += bias01; a23 += bias23;
            s01 += a01; s23 += a23;
            q01 = pk_fma(a01, a01, q01); q23 = pk_fma(a23, a23, q23);
            if (y) *(float4*)(y + (((size_t)b * T + ta) * F + pc) * COUT + 4 * cq) = make_float4(a01.x, a01.y, a23.x, a23.y);
            if (ta + 1 < T) {
                b01 += bias01; b23 += bias23;
                s01 += b01; s23 += b23;
                q01 = pk_fma(b01, b01, q01); q23 = pk_fma(b23, b23, q23);
                if (y) *(float4*)(y + (((size_t)b * T + ta + 1) * F + pc) * COUT + 4 * cq) = make_float4(b01.x, b01.y, b23.x, b23.y);
            }
        }
        pc += 64;
        while (pc >= F) { pc -= F; ++rp; }
    }
    const float s[4] = {s01.x, s01.y, s23.x, s23.y}, s2[4] = {q01.x, q01.y, q23.x, q23.y};
    if (partial) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = s[c], q = s2[c];
#pragma unroll
            for (int m = 4; m <= 32; m <<= 1) { a += __shfl_xor(a, m); q += __shfl_xor(q, m); }
            if ((tid & 63) < 4) { red[tid >> 6][4 * cq + c] = a; red[tid >> 6][COUT + 4 * cq + c] = q; }
        }
        __syncthreads();
        if (tid < 2 * COUT)
            partial[(size_t)tid * (gridDim.x * gridDim.y) + (size_t)b * gridDim.x + blockIdx.x] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    }
}
// x (B,T,F) scaled log-mel; W (16,1,3,3) PyTorch layout; bounds (B,4) int32 or null.  y null = statistics pass only (the
// conv output stays in registers: first half of the fused first block, sed_block0.hip).
SED_API int sed_conv0_fwd(const float* x, const float* W, const float* bias, const int* bounds, float* y, float* partial,
                             int B, int T, int F, int COUT, void* stream) {
    if (COUT != 16 || F > 128 || F < 1) return SED_ERR_UNSUPPORTED;
    if (B <= 0 || T <= 0) return SED_OK;
    SED_LAUNCH((conv0_kernel<16>), dim3((T + C0_TR - 1) / C0_TR, B), dim3(256), 0, (hipStream_t)stream, x, W, bias, bounds, y,
               partial, B, T, F);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------
// BatchNorm statistics finalisation (one workgroup per channel; double accumulation)
// stats layout: [mean | invstd | scale = gamma*invstd | shift = beta - mean*scale], 4*C floats
// ---------------------------------------------------------------------------------------------
#define BNF_THREADS 1024
__global__ __launch_bounds__(BNF_THREADS) void bn_finalize_kernel(const float* __restrict__ partial, int nblocks, int C, float count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float momentum, float eps, float* __restrict__ stats, int training,
                                                          int update_running) {
    __shared__ double r1[BNF_THREADS / 64], r2[BNF_THREADS / 64];
    const int c = blockIdx.x, tid = threadIdx.x;
    float mean, invstd;
    // thread 0's four per-channel scalars are fetched NOW, under the reduction: read after the barrier they were one more exposed
    // memory round trip in a launch that is nothing but latency (5 us, 14 launches per step, between two chip-wide kernels each)
    float g_c = 0.f, b_c = 0.f, rm_c = 0.f, rv_c = 0.f;
    if (tid == 0) {
        g_c = gamma[c]; b_c = beta[c];
        if (!training || update_running) { rm_c = running_mean[c]; rv_c = running_var[c]; }
    }
    if (training) {
        // latency-bound launch (a few hundred to ~2000 partials per channel): 1024 threads so that every thread issues its one or
        // two loads at once instead of walking a dependent chain, then wave butterflies and one barrier
        double a = 0.0, q = 0.0;
#pragma unroll 2
        for (int i = tid; i < nblocks; i += BNF_THREADS) {
            a += (double)partial[(size_t)c * nblocks + i];
            q += (double)partial[(size_t)(C + c) * nblocks + i];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); q += __shfl_xor(q, off); }
        if ((tid & 63) == 0) { r1[tid >> 6] = a; r2[tid >> 6] = q; }
        __syncthreads();
        if (tid != 0) return;
        a = 0.0; q = 0.0;
#pragma unroll
        for (int w = 0; w < BNF_THREADS / 64; ++w) { a += r1[w]; q += r2[w]; }
        const double m = a / (double)count;
        double var = q / (double)count - m * m;
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (update_running) {
            running_mean[c] = (1.0f - momentum) * rm_c + momentum * mean;
            const float unbiased = (float)(var * ((double)count / ((double)count - 1.0)));
            running_var[c] = (1.0f - momentum) * rv_c + momentum * unbiased;
        }
    } else {
        if (tid != 0) return;
        mean = rm_c;
        invstd = 1.0f / sqrtf(rv_c + eps);
    }
    const float sc = g_c * invstd;
    stats[c] = mean;
    stats[C + c] = invstd;
    stats[2 * C + c] = sc;
    stats[3 * C + c] = b_c - mean * sc;
}
SED_API int sed_bn_finalize(const float* partial, int nblocks, int C, float count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, float* stats, int training,
                               int update_running, void* stream) {
    if (C <= 0) return SED_ERR_ARG;
    SED_LAUNCH(bn_finalize_kernel, dim3(C), dim3(BNF_THREADS), 0, (hipStream_t)stream, partial, nblocks, C, count, gamma, beta,
               running_mean, running_var, momentum, eps, stats, training, update_running);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------
// weight gradient: dWp[tap][ci][co] = sum_p x[p + tap][ci] * dy[p][co]   (K = pixels)
// grid = (splits, 9 taps).  Each workgroup walks its share of 64-pixel K-tiles; A = shifted x tile
// [64][CIN], B = dy tile [64][COUT] in LDS; waves split the COUT tiles (and K when few tiles).
// Partial results are added with fp32 atomics into dWp (zeroed by the caller).
// ---------------------------------------------------------------------------------------------
template <int CIN, int COUT>
struct WgCfg {
    static constexpr int KT = 64;                       // pixels per K tile
    static constexpr int MT = (CIN + 31) / 32, NT = (COUT + 31) / 32;
    static constexpr int WN = NT >= 4 ? 4 : NT;         // waves along N
    static constexpr int WK = 4 / WN;                   // waves along K
    static constexpr int NTW = NT / WN;                 // N tiles per wave
    static constexpr int CIP = CIN + 1, COP = COUT + 1; // padded LDS strides (reads are along the channel: contiguous)
    static constexpr int SMEM = KT * (CIN + COUT) * 4;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ dWp, int B, int T, int F) {
    using Cfg = WgCfg<CIN, COUT>;
    constexpr int KT = Cfg::KT, MT = Cfg::MT, WN = Cfg::WN, WK = Cfg::WK, NTW = Cfg::NTW;
    SED_DYN_SMEM(smem);
    float* xs = (float*)smem;            // [KT][CIN]
    float* ds = xs + KT * CIN;           // [KT][COUT]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int wn = w % WN, wk = w / WN;
    const int tap = blockIdx.y, da = tap / 3 - 1, db = tap % 3 - 1;
    const int npix = B * T * F;
    const int ntiles = (npix + KT - 1) / KT;

    f32x16 acc[MT][NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = f32x16_zero();

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int p0 = tile * KT;
        __syncthreads();
        // stage shifted x and dy for pixels p0 .. p0+KT-1 (zero outside the image / beyond npix)
        {   // loads first (x tile shifted by the tap, then the dy tile), LDS stores after
            constexpr int NX = KT * (CIN / 4) / 256, ND = KT * (COUT / 4) / 256;
            float4 lx[NX], ldy[ND];
#pragma unroll
            for (int u = 0; u < NX; ++u) {
                const int idx = tid + 256 * u, r = idx / (CIN / 4), v = idx - r * (CIN / 4);
                const int p = p0 + r;
                lx[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p < npix) {
                    const int f = p % F, t = (p / F) % T, bb = p / (F * T);
                    const int t2 = t + da, f2 = f + db;
                    if (t2 >= 0 && t2 < T && f2 >= 0 && f2 < F)
                        lx[u] = *(const float4*)(x + (((size_t)bb * T + t2) * F + f2) * CIN + 4 * v);
                }
            }
#pragma unroll
            for (int u = 0; u < ND; ++u) {
                const int idx = tid + 256 * u, r = idx / (COUT / 4), v = idx - r * (COUT / 4);
                const int p = p0 + r;
                ldy[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p < npix) ldy[u] = *(const float4*)(dy + (size_t)p * COUT + 4 * v);
            }
#pragma unroll
            for (int u = 0; u < NX; ++u) {
                const int idx = tid + 256 * u, r = idx / (CIN / 4), v = idx - r * (CIN / 4);
                *(float4*)(xs + r * CIN + 4 * v) = lx[u];
            }
#pragma unroll
            for (int u = 0; u < ND; ++u) {
                const int idx = tid + 256 * u, r = idx / (COUT / 4), v = idx - r * (COUT / 4);
                *(float4*)(ds + r * COUT + 4 * v) = ldy[u];
            }
        }
        __syncthreads();
        // this wave's K slice: rows [wk*KT/WK, (wk+1)*KT/WK)
        constexpr int KS = KT / WK;
#pragma unroll 4
        for (int k = wk * KS; k < (wk + 1) * KS; k += 2) {
            float av[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) av[m] = (CIN >= 32 || lo < CIN) ? xs[(k + hi) * CIN + m * 32 + lo] : 0.f;
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int co = (wn * NTW + n) * 32 + lo;
                const float bv = (COUT >= 32 || lo < COUT) ? ds[(k + hi) * COUT + co] : 0.f;
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m][n] = mfma32(av[m], bv, acc[m][n]);
            }
        }
    }
    // ---- this workgroup's partial: dWp[split][tap][ci][co] (deterministic; reduced by wgrad_reduce_kernel) ----
    static_assert(WK == 1, "wide layers: every wave owns its output tiles");
    float* part = dWp + ((size_t)blockIdx.x * 9 + tap) * CIN * COUT;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const int co = (wn * NTW + n) * 32 + lo;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = m * 32 + mfma32_row(r, lane);
                if (ci < CIN && co < COUT) part[(size_t)ci * COUT + co] = acc[m][n][r];
            }
        }
}

template <int CIN, int COUT>
static int launch_wgrad(const float* x, const float* dy, float* dWp, int B, int T, int F, hipStream_t s) {
    using Cfg = WgCfg<CIN, COUT>;
    const int splits = wgrad_parts(CIN, COUT, B, T, F);   // 56 x 9 taps = 504 workgroups ~ 2 per CU
    SED_MAX_SMEM((conv_wgrad_kernel<CIN, COUT>), Cfg::SMEM);
    SED_LAUNCH((conv_wgrad_kernel<CIN, COUT>), dim3(splits, 9), dim3(256), Cfg::SMEM, s, x, dy, dWp, B, T, F);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------
// weight gradient of the wide layers on the split-bf16 MFMA (same grid / partial layout as conv_wgrad_kernel).
//
// dW[tap][ci][co] = sum_p x[p + tap][ci] * dy[p][co] contracts over PIXELS, but v_mfma_f32_32x32x16_bf16 wants each lane to
// hold 8 consecutive K values of its row, i.e. both operands pixel-contiguous -- the transpose of the channels-last
// tensors.  The transposition happens in registers while staging: a thread owns a 4 pixel x 4 channel block (four
// float4 loads, coalesced along the channels), splits it into bf16 hi/lo with v_cvt_pk_bf16_f32 on PIXEL pairs and
// writes, per channel, one 8-byte row segment (4 pixels) into xT / dyT [channel][pixel] planes.  Rows are exactly the 64
// pixels of the K tile (128 B = one pass over the 32 banks) and the pixel octets of a row are XOR-swizzled with
// f(row) = (row ^ (row >> 2)) & 7.  With the gfx950 lane groups (ds_read_b128: {0-3, 12-15, 20-27}, ...; ds_write_b64: 16
// contiguous lanes) both the 16-byte fragment reads and the transposing stores are conflict-free -- the stores because two
// neighbouring lanes own the two pixel quads of one octet of the same channel quad (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
// was 0.50 with the earlier 72-element pitch and row >> 4 swizzle).  The tap shift is applied when the x block is fetched, so
// every LDS access is aligned.  The next K tile is fetched into registers under the MFMAs.
// ---------------------------------------------------------------------------------------------
template <int CIN, int COUT>
struct WgbCfg {
    static constexpr int KT = 64, RS = 64;              // pixels per K tile, LDS row stride (bf16)
    static constexpr int MT = CIN / 32, NT = COUT / 32;
    static constexpr int WM = MT >= 4 ? 2 : 1, WN = 4 / WM;
    static constexpr int MTW = MT / WM, NTW = NT / WN;
    static constexpr int NBX = (KT / 4) * (CIN / 4) / 256, NBD = (KT / 4) * (COUT / 4) / 256;   // 4x4 blocks per thread
    static constexpr int SMEM = 2 * (CIN + COUT) * RS * 2;
    static_assert(NBX >= 1 && NBD >= 1 && MTW * WM == MT && NTW * WN == NT, "tile split");
};

// 4 pixels x 4 channels (vj = pixel j) -> rows 4cq..4cq+3 of the [channel][pixel] hi / lo planes, pixels 4pq..4pq+3.
// (Plain scalars only: pointer / reference arrays over the register blocks would push them to scratch.)
__device__ __forceinline__ void wgb_store_row(float a, float b, float c, float d, unsigned short* __restrict__ hi_row,
                                              unsigned short* __restrict__ lo_row) {
    uint2 h, l;
    bf16_split2(a, b, h.x, l.x);
    bf16_split2(c, d, h.y, l.y);
    *(uint2*)hi_row = h;
    *(uint2*)lo_row = l;
}
__device__ __forceinline__ void wgb_store_block(const float4 v0, const float4 v1, const float4 v2, const float4 v3,
                                                unsigned short* __restrict__ hi_plane, unsigned short* __restrict__ lo_plane,
                                                int cq, int pq, int RS) {
    // row = 4cq + c: octet swizzle f(row) = (row ^ (row >> 2)) & 7 = (c ^ 4(cq & 1) ^ cq) & 7
    const int r0 = 4 * cq, f0 = (r0 ^ cq) & 7;
    const int o0 = r0 * RS + 4 * (pq ^ (2 * f0)), o1 = (r0 + 1) * RS + 4 * (pq ^ (2 * (f0 ^ 1)));
    const int o2 = (r0 + 2) * RS + 4 * (pq ^ (2 * (f0 ^ 2))), o3 = (r0 + 3) * RS + 4 * (pq ^ (2 * (f0 ^ 3)));
    wgb_store_row(v0.x, v1.x, v2.x, v3.x, hi_plane + o0, lo_plane + o0);
    wgb_store_row(v0.y, v1.y, v2.y, v3.y, hi_plane + o1, lo_plane + o1);
    wgb_store_row(v0.z, v1.z, v2.z, v3.z, hi_plane + o2, lo_plane + o2);
    wgb_store_row(v0.w, v1.w, v2.w, v3.w, hi_plane + o3, lo_plane + o3);
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ dWp, int B, int T, int F) {
    using Cfg = WgbCfg<CIN, COUT>;
    constexpr int KT = Cfg::KT, RS = Cfg::RS, WN = Cfg::WN, MTW = Cfg::MTW, NTW = Cfg::NTW, NBX = Cfg::NBX, NBD = Cfg::NBD;
    SED_DYN_SMEM(smem);
    unsigned short* xh = (unsigned short*)smem;      // [CIN][RS] hi, then lo
    unsigned short* xl = xh + CIN * RS;
    unsigned short* dh = xl + CIN * RS;              // [COUT][RS] hi, then lo
    unsigned short* dl = dh + COUT * RS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int wn = w % WN, wm = w / WN;
    const int tap = blockIdx.y, da = tap / 3 - 1, db = tap % 3 - 1;
    const int npix = B * T * F;
    const int ntiles = (npix + KT - 1) / KT;
    const int fsh = 31 - __builtin_clz(F);           // F is a power of two (checked by the launcher)

    f32x16 acc[MTW][NTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = f32x16_zero();

    float4 rx[NBX * 4], rd[NBD * 4];
    unsigned okx = 0, okd = 0;                            // bit 4 u + j: that pixel's x (shifted by the tap) / dy row exists
    static_assert(NBX * 4 <= 32 && NBD * 4 <= 32, "validity masks");
    // (prefetch loads are UNCONDITIONAL -- clamped addresses, a validity bit per load applied when the tile is parked in LDS: as
    //  `ok ? load : 0` the compiler put each load into its own branch and waited for it there, so the "prefetch" of the next tile
    //  was complete before the first MFMA of this one: tools/isa_exposed_loads.py)
    auto load_tile = [&](int tile) {
        const int p0 = tile * KT;
        okx = 0; okd = 0;
#pragma unroll
        for (int u = 0; u < NBX; ++u) {
            const int blk = tid + 256 * u, cq = (blk >> 1) % (CIN / 4), pq = (blk & 1) + 2 * ((blk >> 1) / (CIN / 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = p0 + 4 * pq + j, pc = p < npix ? p : 0;
                const int f = pc & (F - 1), tt = pc >> fsh, t = tt % T, bb = tt / T;
                const int t2 = t + da, f2 = f + db;
                const bool ok = p < npix && t2 >= 0 && t2 < T && f2 >= 0 && f2 < F;
                okx |= (ok ? 1u : 0u) << (4 * u + j);
                const int t2c = t2 < 0 ? 0 : (t2 >= T ? T - 1 : t2), f2c = f2 < 0 ? 0 : (f2 >= F ? F - 1 : f2);
                rx[4 * u + j] = *(const float4*)(x + (((size_t)bb * T + t2c) * F + f2c) * CIN + 4 * cq);
            }
        }
#pragma unroll
        for (int u = 0; u < NBD; ++u) {
            const int blk = tid + 256 * u, cq = (blk >> 1) % (COUT / 4), pq = (blk & 1) + 2 * ((blk >> 1) / (COUT / 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = p0 + 4 * pq + j;
                okd |= (p < npix ? 1u : 0u) << (4 * u + j);
                rd[4 * u + j] = *(const float4*)(dy + (size_t)(p < npix ? p : 0) * COUT + 4 * cq);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NBX * 4; ++i) if (!((okx >> i) & 1u)) rx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NBD * 4; ++i) if (!((okd >> i) & 1u)) rd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < NBX; ++u) {
            const int blk = tid + 256 * u;
            wgb_store_block(rx[4 * u], rx[4 * u + 1], rx[4 * u + 2], rx[4 * u + 3], xh, xl, (blk >> 1) % (CIN / 4),
                            (blk & 1) + 2 * ((blk >> 1) / (CIN / 4)), RS);
        }
#pragma unroll
        for (int u = 0; u < NBD; ++u) {
            const int blk = tid + 256 * u;
            wgb_store_block(rd[4 * u], rd[4 * u + 1], rd[4 * u + 2], rd[4 * u + 3], dh, dl, (blk >> 1) % (COUT / 4),
                            (blk & 1) + 2 * ((blk >> 1) / (COUT / 4)), RS);
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                  // previous tile's fragments are consumed
        store_tile();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
#pragma unroll
        for (int ks = 0; ks < KT / 16; ++ks) {
            const int oct = 2 * ks + hi;                  // this lane's pixel octet of the K tile
            s16x8 a_hi[MTW], a_lo[MTW];
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int row = (wm * MTW + m) * 32 + lo;
                const int off = row * RS + 8 * (oct ^ ((row ^ (row >> 2)) & 7));
                a_hi[m] = *(const s16x8*)(xh + off);
                a_lo[m] = *(const s16x8*)(xl + off);
            }
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int row = (wn * NTW + n) * 32 + lo;
                const int off = row * RS + 8 * (oct ^ ((row ^ (row >> 2)) & 7));
                const s16x8 b_hi = *(const s16x8*)(dh + off);
                const s16x8 b_lo = *(const s16x8*)(dl + off);
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    acc[m][n] = mfma32_bf16(a_lo[m], b_hi, acc[m][n]);
                    acc[m][n] = mfma32_bf16(a_hi[m], b_lo, acc[m][n]);
                    acc[m][n] = mfma32_bf16(a_hi[m], b_hi, acc[m][n]);
                }
            }
        }
    }
    // ---- this workgroup's partial: dWp[split][tap][ci][co] (reduced in a fixed order by wgrad_reduce_kernel) ----
    float* part = dWp + ((size_t)blockIdx.x * 9 + tap) * CIN * COUT;
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const int co = (wn * NTW + n) * 32 + lo;
#pragma unroll
            for (int r = 0; r < 16; ++r) part[(size_t)((wm * MTW + m) * 32 + mfma32_row(r, lane)) * COUT + co] = acc[m][n][r];
        }
}

template <int CIN, int COUT>
static int launch_wgrad_bf16(const float* x, const float* dy, float* dWp, int B, int T, int F, hipStream_t s) {
    using Cfg = WgbCfg<CIN, COUT>;
    const int splits = wgrad_parts(CIN, COUT, B, T, F);
    SED_MAX_SMEM((conv_wgrad_bf16_kernel<CIN, COUT>), Cfg::SMEM);
    SED_LAUNCH((conv_wgrad_bf16_kernel<CIN, COUT>), dim3(splits, 9), dim3(256), Cfg::SMEM, s, x, dy, dWp, B, T, F);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------
// The same weight gradient with ONE KERNEL ROW (three taps: column shifts -1, 0, +1) per workgroup (F >= 8, F | 64).
// conv_wgrad_bf16_kernel fetches, splits and transposes x and dy once per tap -- nine times per launch: 553 MB through L2 / MALL
// at 128 -> 128, F = 8, VALU issue 0.41 - 0.48 for the splitting and index arithmetic.  Here the tile of x (row-shifted by the
// kernel row, as before) and of dy is staged ONCE for three taps; the column shift is applied to the A operand in registers:
// a k-octet = 8 consecutive pixels of a channel = four dwords of bf16 pairs, and the octet shifted by one pixel is four
// v_alignbit_b32 of neighbouring dwords plus ONE extra dword from the previous / next octet (zero at a row end: F >= 8 and
// F | 64 make rows and 64-pixel tiles start on octet boundaries).  Three times the MFMA work per staged tile; the accumulators
// stay at 96 VGPRs because a workgroup owns COH = 64 of the 128 output channels when CIN = 128 (grid z = 2).  Traffic: x is read
// 3 x (COUT / COH) times, dy 3 times, instead of 9 + 9.
// ---------------------------------------------------------------------------------------------
template <int CIN, int COUT, int COH>
struct WgrCfg {
    static constexpr int KT = 64, RS = 64;
    static constexpr int MT = CIN / 32, NT = COH / 32;
    static constexpr int WM = MT >= 4 ? 2 : 1, WN = 4 / WM;
    static constexpr int MTW = MT / WM, NTW = NT / WN;
    static constexpr int NBX = (KT / 4) * (CIN / 4) / 256, NBD = (KT / 4) * (COH / 4) / 256;
    static constexpr int SMEM = 2 * (CIN + COH) * RS * 2;
    static_assert(NBX >= 1 && NBD >= 1 && MTW * WM == MT && NTW * WN == NT && MTW * NTW == 2, "tile split");
};
template <int CIN, int COUT, int COH>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_row_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  float* __restrict__ dWp, int B, int T, int F) {
    using Cfg = WgrCfg<CIN, COUT, COH>;
    constexpr int KT = Cfg::KT, RS = Cfg::RS, WN = Cfg::WN, MTW = Cfg::MTW, NTW = Cfg::NTW, NBX = Cfg::NBX, NBD = Cfg::NBD;
    SED_DYN_SMEM(smem);
    unsigned short* xh = (unsigned short*)smem;      // [CIN][RS] hi, then lo
    unsigned short* xl = xh + CIN * RS;
    unsigned short* dh = xl + CIN * RS;              // [COH][RS] hi, then lo
    unsigned short* dl = dh + COH * RS;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int tid = threadIdx.x, lo = lane & 31, hi = lane >> 5;
    const int wn = w % WN, wm = w / WN;
    const int da = (int)blockIdx.y - 1, co0 = blockIdx.z * COH;
    const int npix = B * T * F;
    const int ntiles = (npix + KT - 1) / KT;
    const int fsh = 31 - __builtin_clz(F);           // F is a power of two (checked by the launcher)

    f32x16 acc[3][MTW][NTW];
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int n = 0; n < NTW; ++n) acc[t3][m][n] = f32x16_zero();

    float4 rx[NBX * 4], rd[NBD * 4];
    unsigned okm = 0;                                     // bit u: x block u is inside the clip; bit 16 + u: dy block u exists
    // (prefetch loads are UNCONDITIONAL -- clamped addresses, a validity bit per load applied when the tile is parked in LDS: as
    //  `ok ? load : 0` the compiler put each load into its own branch and waited for it there, so the "prefetch" of the next tile
    //  was complete before the first MFMA of this one: tools/isa_exposed_loads.py)
    auto load_tile = [&](int tile) {
        const int p0 = tile * KT;
        okm = 0;
#pragma unroll
        for (int u = 0; u < NBX; ++u) {
            const int blk = tid + 256 * u, cq = (blk >> 1) % (CIN / 4), pq = (blk & 1) + 2 * ((blk >> 1) / (CIN / 4));
            // the 4 pixels of a block share one (clip, frame) row (F >= 8): one pair of divisions by the run-time T per block
            const int pb = p0 + 4 * pq, pbc = pb < npix ? pb : 0, tt = pbc >> fsh, t = tt % T, bb = tt / T, t2 = t + da, f = pbc & (F - 1);
            const bool ok = pb < npix && t2 >= 0 && t2 < T;
            okm |= (ok ? 1u : 0u) << u;
            const float* src = x + (((size_t)bb * T + (t2 < 0 ? 0 : (t2 >= T ? T - 1 : t2))) * F + f) * CIN + 4 * cq;
#pragma unroll
            for (int j = 0; j < 4; ++j) rx[4 * u + j] = *(const float4*)(src + (size_t)j * CIN);
        }
#pragma unroll
        for (int u = 0; u < NBD; ++u) {
            const int blk = tid + 256 * u, cq = (blk >> 1) % (COH / 4), pq = (blk & 1) + 2 * ((blk >> 1) / (COH / 4));
            const int pb = p0 + 4 * pq;
            okm |= (pb < npix ? 1u : 0u) << (16 + u);
            const float* src = dy + (size_t)(pb < npix ? pb : 0) * COUT + co0 + 4 * cq;
#pragma unroll
            for (int j = 0; j < 4; ++j) rd[4 * u + j] = *(const float4*)(src + (size_t)j * COUT);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < NBX; ++u)
            if (!((okm >> u) & 1u)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) rx[4 * u + j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int u = 0; u < NBD; ++u)
            if (!((okm >> (16 + u)) & 1u)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) rd[4 * u + j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int u = 0; u < NBX; ++u) {
            const int blk = tid + 256 * u;
            wgb_store_block(rx[4 * u], rx[4 * u + 1], rx[4 * u + 2], rx[4 * u + 3], xh, xl, (blk >> 1) % (CIN / 4),
                            (blk & 1) + 2 * ((blk >> 1) / (CIN / 4)), RS);
        }
#pragma unroll
        for (int u = 0; u < NBD; ++u) {
            const int blk = tid + 256 * u;
            wgb_store_block(rd[4 * u], rd[4 * u + 1], rd[4 * u + 2], rd[4 * u + 3], dh, dl, (blk >> 1) % (COH / 4),
                            (blk & 1) + 2 * ((blk >> 1) / (COH / 4)), RS);
        }
    };
    // the operand octet shifted by one pixel: to the left neighbour (x[p - 1]) or to the right one (x[p + 1])
    auto shift_m1 = [](const uint4 c, unsigned prev3) {
        uint4 r;
        r.x = sed_alignbit(c.x, prev3, 16); r.y = sed_alignbit(c.y, c.x, 16); r.z = sed_alignbit(c.z, c.y, 16); r.w = sed_alignbit(c.w, c.z, 16);
        return r;
    };
    auto shift_p1 = [](const uint4 c, unsigned next0) {
        uint4 r;
        r.x = sed_alignbit(c.y, c.x, 16); r.y = sed_alignbit(c.z, c.y, 16); r.z = sed_alignbit(c.w, c.z, 16); r.w = sed_alignbit(next0, c.w, 16);
        return r;
    };

    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        sed_opaque(tid); sed_opaque(lo); sed_opaque(hi);
        __syncthreads();                                  // previous tile's fragments are consumed
        store_tile();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
#pragma unroll
        for (int ks = 0; ks < KT / 16; ++ks) {
            const int oct = 2 * ks + hi;                  // this lane's pixel octet of the K tile
            const bool first = ((8 * oct) & (F - 1)) == 0, last = ((8 * oct + 8) & (F - 1)) == 0;     // row ends (uniform per half wave)
            s16x8 a_hi[3][MTW], a_lo[3][MTW];
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int row = (wm * MTW + m) * 32 + lo, sw = (row ^ (row >> 2)) & 7;
                const int off = row * RS + 8 * (oct ^ sw);
                const int offp = row * RS + 8 * (((oct + 7) & 7) ^ sw) + 6, offn = row * RS + 8 * (((oct + 1) & 7) ^ sw);
                const uint4 ch = *(const uint4*)(xh + off), cl = *(const uint4*)(xl + off);
                const unsigned ph = first ? 0u : *(const unsigned*)(xh + offp), pl = first ? 0u : *(const unsigned*)(xl + offp);
                const unsigned nh = last ? 0u : *(const unsigned*)(xh + offn), nl = last ? 0u : *(const unsigned*)(xl + offn);
                a_hi[0][m] = __builtin_bit_cast(s16x8, shift_m1(ch, ph)); a_lo[0][m] = __builtin_bit_cast(s16x8, shift_m1(cl, pl));
                a_hi[1][m] = __builtin_bit_cast(s16x8, ch);               a_lo[1][m] = __builtin_bit_cast(s16x8, cl);
                a_hi[2][m] = __builtin_bit_cast(s16x8, shift_p1(ch, nh)); a_lo[2][m] = __builtin_bit_cast(s16x8, shift_p1(cl, nl));
            }
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int row = (wn * NTW + n) * 32 + lo;
                const int off = row * RS + 8 * (oct ^ ((row ^ (row >> 2)) & 7));
                const s16x8 b_hi = *(const s16x8*)(dh + off);
                const s16x8 b_lo = *(const s16x8*)(dl + off);
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                    for (int m = 0; m < MTW; ++m) {
                        acc[t3][m][n] = mfma32_bf16(a_lo[t3][m], b_hi, acc[t3][m][n]);
                        acc[t3][m][n] = mfma32_bf16(a_hi[t3][m], b_lo, acc[t3][m][n]);
                        acc[t3][m][n] = mfma32_bf16(a_hi[t3][m], b_hi, acc[t3][m][n]);
                    }
            }
        }
    }
    // ---- this workgroup's partials: dWp[split][tap][ci][co], taps 3 (da + 1) .. + 2, channels co0 .. co0 + COH ----
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
        float* part = dWp + ((size_t)blockIdx.x * 9 + 3 * (da + 1) + t3) * CIN * COUT;
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int co = co0 + (wn * NTW + n) * 32 + lo;
#pragma unroll
                for (int r = 0; r < 16; ++r) part[(size_t)((wm * MTW + m) * 32 + mfma32_row(r, lane)) * COUT + co] = acc[t3][m][n][r];
            }
    }
}
template <int CIN, int COUT, int COH>
static int launch_wgrad_bf16_row(const float* x, const float* dy, float* dWp, int B, int T, int F, int splits, hipStream_t s) {
    using Cfg = WgrCfg<CIN, COUT, COH>;
    SED_MAX_SMEM((conv_wgrad_bf16_row_kernel<CIN, COUT, COH>), Cfg::SMEM);
    SED_LAUNCH((conv_wgrad_bf16_row_kernel<CIN, COUT, COH>), dim3(splits, 3, COUT / COH), dim3(256), Cfg::SMEM, s, x, dy, dWp, B, T, F);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------
// weight gradient, narrow layers (CIN <= 32): ALL 9 taps in one workgroup, so x and dy are read from HBM once
// instead of once per tap.  Tile = the forward's 128-pixel TR x TF patch: halo patch of x [PP][CIN] and the dy
// tile [128][COUT] in LDS; K = pixels of the tile, split across waves (WK) with the COUT tiles (WN); each wave
// keeps 9 x (COUT/32/WN) accumulators across all its tiles and adds them to dWp with atomics at the end.
// ---------------------------------------------------------------------------------------------
template <int CIN, int COUT, int TF>
struct WgaCfg {
    static constexpr int TR = 128 / TF, PW = TF + 2, PH = TR + 2, PP = PW * PH;
    static constexpr int NT = COUT / 32, WN = NT >= 2 ? 2 : 1, WK = 4 / WN, NTW = NT / WN;
    static constexpr int SMEM = (PP * CIN + 128 * COUT) * 4;
};
template <int CIN, int COUT, int TF>
__global__ __launch_bounds__(256) void conv_wgrad_alltaps_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 float* __restrict__ dWp, int B, int T, int F) {
    using Cfg = WgaCfg<CIN, COUT, TF>;
    constexpr int TR = Cfg::TR, PW = Cfg::PW, PP = Cfg::PP, WN = Cfg::WN, WK = Cfg::WK, NTW = Cfg::NTW;
    static_assert(CIN <= 32, "narrow layers only");
    SED_DYN_SMEM(smem);
    float* xs = (float*)smem;            // [PP][CIN]
    float* ds = xs + PP * CIN;           // [128][COUT]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int wn = w % WN, wk = w / WN;
    const int ftiles = F / TF, ttiles = (T + TR - 1) / TR, ntiles = B * ttiles * ftiles;
    // CIN == 16: the 16x16x4 MFMA (M = 16 input channels exactly; the 32x32x2 tile would run half empty), all waves split K,
    // every wave owns both 16-wide COUT tiles.  Lane (i = l&15, g = l>>4): A[ci = i][pixel 4ks + g], B[pixel 4ks + g][co = i].
    constexpr bool M16 = CIN == 16;
    constexpr int NT16 = COUT / 16, WK16 = 4;
    f32x16 acc[M16 ? 1 : 9][M16 ? 1 : NTW];
    f32x4 acc16[M16 ? 9 : 1][M16 ? NT16 : 1];
#pragma unroll
    for (int tp = 0; tp < (M16 ? 1 : 9); ++tp)
#pragma unroll
        for (int n = 0; n < (M16 ? 1 : NTW); ++n) acc[tp][n] = f32x16_zero();
#pragma unroll
    for (int tp = 0; tp < (M16 ? 9 : 1); ++tp)
#pragma unroll
        for (int n = 0; n < (M16 ? NT16 : 1); ++n) acc16[tp][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // next tile's halo patch and dy tile are fetched into registers under the current tile's MFMAs
    constexpr int NX = (PP * (CIN / 4) + 255) / 256, ND = 128 * (COUT / 4) / 256;
    float4 lx[NX], ldy[ND];
    auto load_tile = [&](int tile_) {
        const int ft = tile_ % ftiles, tt = (tile_ / ftiles) % ttiles, b = tile_ / (ftiles * ttiles);
        const int t0 = tt * TR, f0 = ft * TF;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int idx = tid + 256 * u, pix = idx / (CIN / 4), v = idx - pix * (CIN / 4);
            const int i = pix / PW, j = pix - i * PW;
            const int t = t0 - 1 + i, f = f0 - 1 + j;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < PP * (CIN / 4) && t >= 0 && t < T && f >= 0 && f < F)
                val = *(const float4*)(x + (((size_t)b * T + t) * F + f) * CIN + 4 * v);
            lx[u] = val;
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int idx = tid + 256 * u, p = idx / (COUT / 4), v = idx - p * (COUT / 4);
            const int t = t0 + p / TF, f = f0 + p % TF;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < T) val = *(const float4*)(dy + (((size_t)b * T + t) * F + f) * COUT + 4 * v);
            ldy[u] = val;
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int idx = tid + 256 * u;
            if (idx < PP * (CIN / 4)) *(float4*)(xs + (idx / (CIN / 4)) * CIN + 4 * (idx % (CIN / 4))) = lx[u];
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int idx = tid + 256 * u;
            *(float4*)(ds + (idx / (COUT / 4)) * COUT + 4 * (idx % (COUT / 4))) = ldy[u];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
        if (M16) {
            constexpr int KS16 = 128 / WK16;
            const int i16 = lane & 15, g16 = lane >> 4;
#pragma unroll 2
            for (int k = w * KS16; k < (w + 1) * KS16; k += 4) {
                const int p = k + g16;
                const int pb = (p / TF) * PW + (p % TF);
                float bv16[NT16];
#pragma unroll
                for (int n = 0; n < NT16; ++n) bv16[n] = ds[p * COUT + n * 16 + i16];
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const float av = xs[(pb + (tp / 3) * PW + (tp % 3)) * CIN + i16];
#pragma unroll
                    for (int n = 0; n < NT16; ++n) acc16[M16 ? tp : 0][M16 ? n : 0] = mfma16(av, bv16[n], acc16[M16 ? tp : 0][M16 ? n : 0]);
                }
            }
            continue;
        }
        constexpr int KS = 128 / WK;
#pragma unroll 2
        for (int k = wk * KS; k < (wk + 1) * KS; k += 2) {
            const int p = k + hi;
            const int pb = (p / TF) * PW + (p % TF);
            float bv[NTW];
#pragma unroll
            for (int n = 0; n < NTW; ++n) bv[n] = ds[p * COUT + (wn * NTW + n) * 32 + lo];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const float av = (CIN >= 32 || lo < CIN) ? xs[(pb + (tp / 3) * PW + (tp % 3)) * CIN + lo] : 0.f;
#pragma unroll
                for (int n = 0; n < NTW; ++n) acc[tp][n] = mfma32(av, bv[n], acc[tp][n]);
            }
        }
    }
    // ---- reduce the WK K-slices through LDS (tap groups of <= 5 to fit), then store this workgroup's partial ----
    float* red = (float*)smem;           // [taps in group][CIN][COUT], reuses the staging area
    float* part = dWp + (size_t)blockIdx.x * 9 * CIN * COUT;
    constexpr int TG = 5;
    if (M16) {            // same tap-group reduction, 16x16 accumulator layout: row = 4g + r (ci), col = i (co within the tile)
        const int i16 = lane & 15, g16 = lane >> 4;
#pragma unroll
        for (int g0 = 0; g0 < 9; g0 += TG) {
            for (int round = 0; round < WK16; ++round) {
                __syncthreads();
                if (w == round) {
#pragma unroll
                    for (int tp = g0; tp < g0 + TG && tp < 9; ++tp)
#pragma unroll
                        for (int n = 0; n < NT16; ++n)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float* d = red + ((tp - g0) * CIN + 4 * g16 + r) * COUT + n * 16 + i16;
                                const float v = acc16[M16 ? tp : 0][M16 ? n : 0][r];
                                *d = (round == 0) ? v : *d + v;
                            }
                }
            }
            __syncthreads();
            const int ntp = (9 - g0) < TG ? (9 - g0) : TG;
            for (int idx = tid; idx < ntp * CIN * COUT; idx += 256) part[(size_t)g0 * CIN * COUT + idx] = red[idx];
        }
        return;
    }
#pragma unroll
    for (int g0 = 0; g0 < 9; g0 += TG) {
        for (int round = 0; round < WK; ++round) {
            __syncthreads();
            if (wk == round) {
#pragma unroll
                for (int tp = g0; tp < g0 + TG && tp < 9; ++tp)
#pragma unroll
                    for (int n = 0; n < NTW; ++n) {
                        const int co = (wn * NTW + n) * 32 + lo;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ci = mfma32_row(r, lane);
                            if (ci < CIN) {
                                float* d = red + ((tp - g0) * CIN + ci) * COUT + co;
                                *d = (round == 0) ? acc[tp][n][r] : *d + acc[tp][n][r];
                            }
                        }
                    }
            }
        }
        __syncthreads();
        const int ntp = (9 - g0) < TG ? (9 - g0) : TG;
        for (int idx = tid; idx < ntp * CIN * COUT; idx += 256) part[(size_t)g0 * CIN * COUT + idx] = red[idx];
    }
}
template <int CIN, int COUT, int TF>
static int launch_wgrad_alltaps(const float* x, const float* dy, float* dWp, int B, int T, int F, hipStream_t s) {
    using Cfg = WgaCfg<CIN, COUT, TF>;
    const int grid = wgrad_parts(CIN, COUT, B, T, F);
    SED_MAX_SMEM((conv_wgrad_alltaps_kernel<CIN, COUT, TF>), Cfg::SMEM);
    SED_LAUNCH((conv_wgrad_alltaps_kernel<CIN, COUT, TF>), dim3(grid), dim3(256), Cfg::SMEM, s, x, dy, dWp, B, T, F);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------
// weight gradient, narrow layers, split-bf16 MFMA (conv_precision = "bf16x3", F >= 32): all 9 taps in one workgroup as above,
// but the contraction over the tile's 128 pixels runs on v_mfma_f32_32x32x16_bf16 (CIN = 32) / 16x16x32 (CIN = 16), three per
// product on hi / lo planes: 5.3x fewer MFMA cycles than the exact-f32 32x32x2 path, fp32-level accuracy.
// A bf16 MFMA operand is 8 consecutive k (= pixels) of one channel in one 16-byte LDS read, so both operands are staged
// TRANSPOSED, [channel][pixel], from 4-pixel x 4-channel register blocks (8-byte stores).  A tap shifts the pixel window:
// the row shift ky is a whole row pitch (aligned), the column shift kx is not, so the x patch is kept in THREE column-shifted
// copies (a thread loads 6 consecutive patch pixels once and cuts the three 4-pixel windows out of them in registers).
// Channel pitches: 25 / 17 sixteen-byte slots (32x32x16: the 16 lanes of a ds_read_b128 group are 16 distinct channels
// mod 16 -> odd pitch) and 26 / 18 slots (16x16x32: a group mixes two k-octets of complementary channel sets -> pitch = 2 mod 4).
// ---------------------------------------------------------------------------------------------
// WGN_ABL: timing-ablation mask for tools/wgrad_variants.py (0 in the product build): 1 = no MFMAs, 2 = no operand reads and no
// MFMAs, 4 = no staging (splits + LDS stores), 8 = no global loads
#ifndef WGN_ABL
#define WGN_ABL 0
#endif
template <int CIN, int COUT>
struct WgnCfg {
    static constexpr int TF = 32, TR = 4, PH = TR + 2;
    static constexpr bool M16 = CIN == 16;
    static constexpr int XS = PH * TF + (M16 ? 16 : 8), DS = TR * TF + (M16 ? 16 : 8);     // channel pitch (bf16 elements)
    static constexpr int XPLANE = CIN * XS, DPLANE = COUT * DS;
    static constexpr int SMEM_OPS = (6 * XPLANE + 2 * DPLANE) * 2, SMEM_RED = 9 * CIN * COUT * 4;
    static constexpr int SMEM = SMEM_OPS > SMEM_RED ? SMEM_OPS : SMEM_RED;
};
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_wgrad_alltaps_bf16_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                      float* __restrict__ dWp, int B, int T, int F) {
    using Cfg = WgnCfg<CIN, COUT>;
    constexpr int TF = Cfg::TF, TR = Cfg::TR, PH = Cfg::PH, XS = Cfg::XS, DS = Cfg::DS, XPLANE = Cfg::XPLANE, DPLANE = Cfg::DPLANE;
    constexpr bool M16 = Cfg::M16;
    static_assert((CIN == 16 && COUT == 32) || (CIN == 32 && COUT == 64), "the two narrow layers of the recipe");
    SED_DYN_SMEM(smem);
    unsigned short* xp = (unsigned short*)smem;          // [column shift 3][hi | lo][CIN][XS]: x[t0 - 1 + i][f0 + c + shift - 1]
    unsigned short* dp = xp + 6 * XPLANE;                // [hi | lo][COUT][DS]:                dy[t0 + r][f0 + c]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int tid = threadIdx.x, lo = lane & 31, hi = lane >> 5, i16 = lane & 15, g = lane >> 4;
    const int ftiles = F / TF, ttiles = (T + TR - 1) / TR, ntiles = B * ttiles * ftiles;

    f32x16 acc[M16 ? 1 : 9];
    f32x4 acc16[M16 ? 9 : 1][2];
#pragma unroll
    for (int tp = 0; tp < (M16 ? 1 : 9); ++tp) acc[tp] = f32x16_zero();
#pragma unroll
    for (int tp = 0; tp < (M16 ? 9 : 1); ++tp) { acc16[tp][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc16[tp][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // staging work items: x = (patch row i, column quad cq, channel quad v): 6 pixels x 4 channels in, 3 shifts x 4 channels x
    // 4 pixels out; dy = (tile row r, column quad cq, channel quad v): 4 pixels x 4 channels in and out.  Lanes run over cq
    // first: the 16 lanes of a ds_write_b64 group then cover the 64 contiguous bytes of a row segment for two channels 4 apart,
    // whose pitch is 64 bytes mod 128 (CIN = 32) -- conflict-free; with channel quads first the same stores were 4-way conflicted
    // and the staging was half of the kernel (timing ablation tools/wgrad_variants.py).  A wave's global load still covers 8 full
    // 128-byte lines (8 pixels x all channels of a quad group).
    constexpr int VX = CIN / 4, VD = COUT / 4, NXI = PH * 8 * VX, NDI = TR * 8 * VD;
    constexpr int NXU = (NXI + 255) / 256, NDU = NDI / 256;
    float4 lx[NXU][6], ld[NDU][4];
    unsigned long long okx = 0;                           // bit 6 u + k: patch pixel (u, k) is inside the clip
    unsigned okd = 0;                                     // bit u: dy row u exists
    static_assert(NXU * 6 <= 64 && NDU <= 32, "validity masks");
    // (prefetch loads are UNCONDITIONAL -- clamped addresses, a validity bit per load applied when the tile is parked in LDS: as
    //  `ok ? load : 0` the compiler put each load into its own branch and waited for it there, so the "prefetch" of the next tile
    //  was complete before the first MFMA of this one: tools/isa_exposed_loads.py)
    auto load_tile = [&](int tile_) {
        const int ft = tile_ % ftiles, tt = (tile_ / ftiles) % ttiles, b = tile_ / (ftiles * ttiles);
        const int t0 = tt * TR, f0 = ft * TF;
        okx = 0; okd = 0;
#pragma unroll
        for (int u = 0; u < NXU; ++u) {
            const int it = tid + 256 * u, cq = it % 8, v = (it / 8) % VX, i = it / (8 * VX);
            const int t = t0 - 1 + i;
            const bool rowok = it < NXI && t >= 0 && t < T && !(WGN_ABL & 8);
            const float* src = x + (((size_t)b * T + (t < 0 ? 0 : (t >= T ? T - 1 : t))) * F) * CIN + 4 * v;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int f = f0 - 1 + 4 * cq + k;
                okx |= (unsigned long long)((rowok && f >= 0 && f < F) ? 1 : 0) << (6 * u + k);
                lx[u][k] = *(const float4*)(src + (size_t)(f < 0 ? 0 : (f >= F ? F - 1 : f)) * CIN);
            }
        }
#pragma unroll
        for (int u = 0; u < NDU; ++u) {
            const int it = tid + 256 * u, cq = it % 8, v = (it / 8) % VD, r = it / (8 * VD);
            const int t = t0 + r;
            okd |= ((t < T && !(WGN_ABL & 8)) ? 1u : 0u) << u;
            const float* src = dy + (((size_t)b * T + (t < T ? t : T - 1)) * F + f0 + 4 * cq) * COUT + 4 * v;
#pragma unroll
            for (int k = 0; k < 4; ++k) ld[u][k] = *(const float4*)(src + (size_t)k * COUT);
        }
    };
    auto comp = [](const float4& q, int c) { return c == 0 ? q.x : c == 1 ? q.y : c == 2 ? q.z : q.w; };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < NXU; ++u)
#pragma unroll
            for (int k = 0; k < 6; ++k) if (!((okx >> (6 * u + k)) & 1ull)) lx[u][k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < NDU; ++u)
            if (!((okd >> u) & 1u)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) ld[u][k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int u = 0; u < NXU; ++u) {
            const int it = tid + 256 * u, cq = it % 8, v = (it / 8) % VX, i = it / (8 * VX);
            if (it < NXI) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // the five neighbouring pixel pairs of this channel, split once; shift s uses pairs (s, s + 2)
                    unsigned ph[5], pl[5];
#pragma unroll
                    for (int k = 0; k < 5; ++k) bf16_split2(comp(lx[u][k], c), comp(lx[u][k + 1], c), ph[k], pl[k]);
                    unsigned short* dst = xp + (4 * v + c) * XS + i * TF + 4 * cq;
#pragma unroll
                    for (int sft = 0; sft < 3; ++sft) {
                        uint2 hv, lv;
                        hv.x = ph[sft]; hv.y = ph[sft + 2];
                        lv.x = pl[sft]; lv.y = pl[sft + 2];
                        *(uint2*)(dst + (2 * sft) * XPLANE) = hv;
                        *(uint2*)(dst + (2 * sft + 1) * XPLANE) = lv;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NDU; ++u) {
            const int it = tid + 256 * u, cq = it % 8, v = (it / 8) % VD, r = it / (8 * VD);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint2 hv, lv;
                bf16_split2(comp(ld[u][0], c), comp(ld[u][1], c), hv.x, lv.x);
                bf16_split2(comp(ld[u][2], c), comp(ld[u][3], c), hv.y, lv.y);
                unsigned short* dst = dp + (4 * v + c) * DS + r * TF + 4 * cq;
                *(uint2*)dst = hv;
                *(uint2*)(dst + DPLANE) = lv;
            }
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        sed_opaque(tid); sed_opaque(lo); sed_opaque(hi); sed_opaque(i16); sed_opaque(g);     // per-tile addresses: recomputed, not spilled
        __syncthreads();
        if (!(WGN_ABL & 4)) store_tile();
        else if (lx[0][0].x + ld[0][0].x == 123.456f) xp[tid] = 1;      // keeps the loads alive
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);      // in flight under the MFMAs
        if (WGN_ABL & 2) continue;
        if (M16) {
            // wave w contracts tile row w (one 32-pixel k-step); both 16-wide COUT tiles.  All 22 operand reads first, then the 54
            // MFMAs (left alone the compiler emits read - wait - three dependent MFMAs per tap)
            s16x8 bh[2], bl[2], ah[9], al[9];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const unsigned short* bp = dp + (16 * nt + i16) * DS + w * TF + 8 * g;
                bh[nt] = *(const s16x8*)bp;
                bl[nt] = *(const s16x8*)(bp + DPLANE);
            }
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const unsigned short* ap = xp + (2 * (tp % 3)) * XPLANE + i16 * XS + (w + tp / 3) * TF + 8 * g;
                ah[tp] = *(const s16x8*)ap;
                al[tp] = *(const s16x8*)(ap + XPLANE);
            }
            sed_sched_fence();
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x4 a = acc16[M16 ? tp : 0][nt];
                    if (WGN_ABL & 1) { a[0] += (float)(al[tp][0] + bh[nt][1] + ah[tp][2] + bl[nt][3]); }
                    else {
                    a = mfma16_bf16(al[tp], bh[nt], a);
                    a = mfma16_bf16(ah[tp], bl[nt], a);
                    a = mfma16_bf16(ah[tp], bh[nt], a);
                    }
                    acc16[M16 ? tp : 0][nt] = a;
                }
        } else {
            // wave (wn, wk): COUT tile wn, k-steps 4 wk .. 4 wk + 3 (16 pixels each: half a tile row).  One wave per SIMD (312
            // registers), so nothing else hides the LDS latency: the 20 operand reads of k-step k + 1 are issued before the 27
            // MFMAs of k-step k (double-buffered operand registers; this kernel may use up to 512 registers anyway).
            const int wn = w & 1, wk = w >> 1;
            s16x8 bh[2], bl[2], ah[2][9], al[2][9];
            auto fetch = [&](int k4, int buf) {
                const int ks = 4 * wk + k4, r = ks >> 1, c0 = 16 * (ks & 1) + 8 * hi;
                const unsigned short* bp = dp + (32 * wn + lo) * DS + r * TF + c0;
                bh[buf] = *(const s16x8*)bp;
                bl[buf] = *(const s16x8*)(bp + DPLANE);
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const unsigned short* ap = xp + (2 * (tp % 3)) * XPLANE + lo * XS + (r + tp / 3) * TF + c0;
                    ah[buf][tp] = *(const s16x8*)ap;
                    al[buf][tp] = *(const s16x8*)(ap + XPLANE);
                }
            };
            fetch(0, 0);
            sed_sched_fence();
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int cur = k4 & 1;
                if (k4 + 1 < 4) fetch(k4 + 1, cur ^ 1);
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    f32x16 a = acc[M16 ? 0 : tp];
                    if (WGN_ABL & 1) { a[0] += (float)(al[cur][tp][0] + bh[cur][1] + ah[cur][tp][2] + bl[cur][3]); }
                    else {
                    a = mfma32_bf16(al[cur][tp], bh[cur], a);
                    a = mfma32_bf16(ah[cur][tp], bl[cur], a);
                    a = mfma32_bf16(ah[cur][tp], bh[cur], a);
                    }
                    acc[M16 ? 0 : tp] = a;
                }
                sed_sched_fence();
            }
        }
    }
    // ---- sum the waves' K slices in LDS (fixed order), store this workgroup's partial [tap][ci][co] ----
    float* red = (float*)smem;
    float* part = dWp + (size_t)blockIdx.x * 9 * CIN * COUT;
    if (M16) {
        for (int round = 0; round < 4; ++round) {
            __syncthreads();
            if (w == round) {
#pragma unroll
                for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* d = red + (tp * CIN + 4 * g + r) * COUT + 16 * nt + i16;
                            const float v = acc16[M16 ? tp : 0][nt][r];
                            *d = round == 0 ? v : *d + v;
                        }
            }
        }
    } else {
        const int wn = w & 1, wk = w >> 1;
        for (int round = 0; round < 2; ++round) {
            __syncthreads();
            if (wk == round) {
#pragma unroll
                for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float* d = red + (tp * CIN + mfma32_row(r, lane)) * COUT + 32 * wn + lo;
                        const float v = acc[M16 ? 0 : tp][r];
                        *d = round == 0 ? v : *d + v;
                    }
            }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 9 * CIN * COUT; idx += 256) part[idx] = red[idx];
}
template <int CIN, int COUT>
static int launch_wgrad_alltaps_bf16(const float* x, const float* dy, float* dWp, int B, int T, int F, int grid, hipStream_t s) {
    using Cfg = WgnCfg<CIN, COUT>;
    SED_MAX_SMEM((conv_wgrad_alltaps_bf16_kernel<CIN, COUT>), Cfg::SMEM);
    SED_LAUNCH((conv_wgrad_alltaps_bf16_kernel<CIN, COUT>), dim3(grid), dim3(256), Cfg::SMEM, s, x, dy, dWp, B, T, F);
    return sed_check_launch();
}

static int conv_wgrad_impl(const float* x, const float* dy, float* dWp, float* dW, int B, int T, int F, int CIN, int COUT,
                           bool split_bf16, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int rc = SED_ERR_UNSUPPORTED;
    const int TF = conv_tf(F);
    if (F % TF != 0 || (F & (F - 1)) != 0) return SED_ERR_UNSUPPORTED;
    // narrow layers at the production widths: the split-bf16 all-taps kernel (sed_set_tuning(SED_TUNE_WGRAD_NARROW, 1): exact f32)
    int nparts = wgrad_parts(CIN, COUT, B, T, F);
    if (split_bf16 && TF == 32 && sed_tuning[SED_TUNE_WGRAD_NARROW] != 1 && ((CIN == 16 && COUT == 32) || (CIN == 32 && COUT == 64))) {
        // one (CIN = 32: 111 KB of LDS) / two resident workgroups per CU; every workgroup beyond that only adds a partial to write
        // and reduce (18 / 74 KB each: 1024 / 512 partials were 22 / 29 us of fixed cost per launch)
        const int capb = sed_tuning[SED_TUNE_WGRAD_CAP] > 0 ? sed_tuning[SED_TUNE_WGRAD_CAP] : (CIN == 16 ? 512 : 256);
        if (nparts > capb) nparts = capb;
        if (CIN == 16) rc = launch_wgrad_alltaps_bf16<16, 32>(x, dy, dWp, B, T, F, nparts, s);
        else rc = launch_wgrad_alltaps_bf16<32, 64>(x, dy, dWp, B, T, F, nparts, s);
    }
#define WGA_CASE(ci, co, tf) if (rc != SED_OK && CIN == ci && COUT == co && TF == tf) rc = launch_wgrad_alltaps<ci, co, tf>(x, dy, dWp, B, T, F, s);
    WGA_CASE(16, 32, 32) WGA_CASE(16, 32, 16) WGA_CASE(16, 32, 8) WGA_CASE(32, 64, 32) WGA_CASE(32, 64, 16) WGA_CASE(32, 64, 8)
    WGA_CASE(32, 64, 4)
#undef WGA_CASE
    // wide layers, split-bf16: one kernel row (three taps) per workgroup when a row is whole octets (sed_set_tuning key 9 = 1: one tap)
    if (rc != SED_OK && split_bf16 && wgrad_row_ok(CIN, COUT, F) && sed_tuning[SED_TUNE_WGRAD_WIDE] != 1) {
        nparts = wgrad_row_parts(CIN, B, T, F);
        if (CIN == 64) rc = launch_wgrad_bf16_row<64, 128, 128>(x, dy, dWp, B, T, F, nparts, s);
        else rc = launch_wgrad_bf16_row<128, 128, 64>(x, dy, dWp, B, T, F, nparts, s);
    }
#define WG_CASE(ci, co)                                                                                        \
    if (rc != SED_OK && CIN == ci && COUT == co)                                                               \
        rc = split_bf16 ? launch_wgrad_bf16<ci, co>(x, dy, dWp, B, T, F, s) : launch_wgrad<ci, co>(x, dy, dWp, B, T, F, s);
    WG_CASE(64, 128) WG_CASE(128, 128)
#undef WG_CASE
    if (rc != SED_OK) return rc;
    const int n = COUT * CIN * 9;
    SED_LAUNCH(wgrad_reduce_kernel, dim3((n + 63) / 64), dim3(nparts >= 256 ? 1024 : 256), 0, s, (const float*)dWp, dW, nparts, COUT, CIN);
    return sed_check_launch();
}
// x (B,T,F,CIN), dy (B,T,F,COUT) -> dW (COUT,CIN,3,3).  dWp: scratch of sed_conv_wgrad_scratch_floats() floats.
SED_API int sed_conv_wgrad(const float* x, const float* dy, float* dWp, float* dW, int B, int T, int F, int CIN, int COUT,
                              void* stream) {
    return conv_wgrad_impl(x, dy, dWp, dW, B, T, F, CIN, COUT, false, stream);
}
// Same contract; every layer with F >= 32 (narrow ones) or CIN >= 64 contracts on the split-bf16 MFMA (fp32-level accuracy,
// ~8e-6 relative); narrow layers on tiles narrower than 32 columns use the exact-f32 all-taps kernel.
SED_API int sed_conv_wgrad_bf16x3(const float* x, const float* dy, float* dWp, float* dW, int B, int T, int F, int CIN,
                                     int COUT, void* stream) {
    return conv_wgrad_impl(x, dy, dWp, dW, B, T, F, CIN, COUT, true, stream);
}

// ---------------------------------------------------------------------------------------------
// layer-0 weight gradient: dW[co][tap] = sum_p x[p + tap] * dy[p][co], CIN = 1, COUT = 16.
// K = 9 taps x 16 channels is no dense contraction: direct VALU.  Lanes = (pixel, channel quad) as in the forward,
// 36 accumulators per lane; same 16 x F LDS tile of x (SpecAugment predicate fused).  The training-mode
// BatchNorm backward (dz -> dy, see bn_bwd_apply_kernel) is fused into the load: nobody else consumes dy of the
// first block, so the 3 x 246 MB in-place pass is replaced by reading y alongside dz here.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv0_wgrad_kernel(const float* __restrict__ x, const int* __restrict__ bounds,
                                                          const float* __restrict__ dyz, const float* __restrict__ y,
                                                          const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                          float* __restrict__ dW, int B, int T, int F, int tiles_t, int fuse_bn,
                                                          int training, float inv_count) {
    constexpr int C = 16;
    __shared__ float tile[(C0_TR + 2) * (128 + 2)];
    __shared__ float red[4][144];
    const int tid = threadIdx.x, PW = F + 2, cq = tid & 3;
    float acc[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[c][k] = 0.f;
    float mean[4], istd[4], m1[4], m2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = 4 * cq + c;
        mean[c] = fuse_bn ? stats[ch] : 0.f;
        istd[c] = fuse_bn ? stats[C + ch] : 1.f;
        m1[c] = (fuse_bn && training) ? gamma[ch] * dbeta[ch] * inv_count : 0.f;
        m2[c] = (fuse_bn && training) ? gamma[ch] * dgamma[ch] * inv_count : 0.f;
    }
    const int ntiles = B * tiles_t;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int b = tl / tiles_t, t0 = (tl - b * tiles_t) * C0_TR;
        int mf0 = 0, mf1 = 0, mt0 = 0, mt1 = 0;
        if (bounds) { mf0 = bounds[4 * b]; mf1 = bounds[4 * b + 1]; mt0 = bounds[4 * b + 2]; mt1 = bounds[4 * b + 3]; }
        __syncthreads();
        for (int idx = tid; idx < (C0_TR + 2) * PW; idx += 256) {
            const int i = idx / PW, j = idx - i * PW;
            const int t = t0 - 1 + i, f = j - 1;
            float v = 0.f;
            if (t >= 0 && t < T && f >= 0 && f < F) {
                v = x[((size_t)b * T + t) * F + f];
                if ((f >= mf0 && f < mf1) || (t >= mt0 && t < mt1)) v = 0.f;
            }
            tile[idx] = v;
        }
        __syncthreads();
        for (int p = tid >> 2; p < C0_TR * F; p += 64) {
            const int pr = p / F, pc = p - pr * F, t = t0 + pr;
            if (t < T) {
                const size_t off = (((size_t)b * T + t) * F + pc) * C + 4 * cq;
                const float4 gz = *(const float4*)(dyz + off);
                float g[4] = {gz.x, gz.y, gz.z, gz.w};
                if (fuse_bn) {
                    const float4 yv = *(const float4*)(y + off);
                    const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) g[c] = istd[c] * (g[c] - m1[c] - (yy[c] - mean[c]) * istd[c] * m2[c]);
                }
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int bb = 0; bb < 3; ++bb) {
                        const float xv = tile[(pr + a) * PW + pc + bb];
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c][a * 3 + bb] = fmaf(xv, g[c], acc[c][a * 3 + bb]);
                    }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float v = acc[c][k];
#pragma unroll
            for (int m = 4; m <= 32; m <<= 1) v += __shfl_xor(v, m);
            if ((tid & 63) < 4) red[tid >> 6][(4 * cq + c) * 9 + k] = v;
        }
    __syncthreads();
    if (tid < 144) atomicAdd(dW + tid, red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
}
// dW (16,1,3,3) PyTorch layout, accumulated with atomics (zeroed here).
// fuse_bn = 0: `dyz` is dy.  fuse_bn = 1: `dyz` is dz = dL/d(xhat) and the BatchNorm backward is applied on the fly
// (y, stats, gamma, dgamma, dbeta as for sed_bn_bwd_apply); dbias (16) then receives the conv-bias gradient.
SED_API int sed_conv0_wgrad(const float* x, const int* bounds, const float* dyz, const float* y, const float* stats,
                               const float* gamma, const float* dgamma, const float* dbeta, float* dW, float* dbias, int B, int T,
                               int F, int COUT, int fuse_bn, int training, void* stream) {
    if (COUT != 16 || F > 128 || F < 1) return SED_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    sed_zero4(s, dW, 16 * 9, fuse_bn ? dbias : nullptr, fuse_bn ? 16 : 0, nullptr, 0, nullptr, 0);
    if (B <= 0 || T <= 0) return SED_OK;
    if (fuse_bn && !training) return SED_ERR_UNSUPPORTED;      // eval-mode BN backward keeps the unfused path (bias gradient != 0)
    const int tiles_t = (T + C0_TR - 1) / C0_TR;
    int grid = B * tiles_t;
    if (grid > 1024) grid = 1024;
    SED_LAUNCH(conv0_wgrad_kernel, dim3(grid), dim3(256), 0, s, x, bounds, dyz, y, stats, gamma, dgamma, dbeta, dW, B, T, F, tiles_t,
               fuse_bn, training, 1.0f / ((float)B * (float)T * (float)F));
    return sed_check_launch();
}
