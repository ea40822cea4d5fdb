// the frozen BEATs feature extractor (recipes/dcase2023_task4_baseline/local/beats/{BEATs.py:109-204,
// backbone.py:23-160,214-296,446-700}) as an inference path on the MI355X.  fp32 storage everywhere, fp32-level accuracy:
//   K-B1 kaldi_fbank_kernel   torchaudio.compliance.kaldi.fbank(25 ms / 10 ms, povey window, 512-point FFT, 128 Kaldi-mel bins, log)
//                             + the (x - mean) / (2 std) of BEATs.preprocess                                  (BEATs.py:109-133)
//   K-B2 patchify16_kernel    the 16 x 16 / stride 16 patch gather; the patch embedding itself, every Linear of the encoder and
//                             the FFN (with its GELU in the epilogue) are the split-bf16 MFMA GEMM of sed_gemm_bf16.hip
//   K-B3 layernorm_kernel     y = LayerNorm(alpha * residual + x): the post-LN / deep-norm residual of every encoder sub-layer
//   K-B4 posconv_kernel       x + GELU(grouped Conv1d(k = 128, 16 groups)(x)): the convolutional position embedding
//   K-B5 attention_kernel     multi-head attention with the bucketed relative position bias scaled per (head, query) by the
//                             GRU-style gate computed from the query (backbone.py:662-682); online softmax, no T x T tensor in HBM
// The Linear layers (86 % of the extractor's FLOPs), the attention and the position convolution all run on the split-bf16 MFMA
// (vector-pipe versions of the last two are kept as sed_posconv and behind sed_set_tuning).
#include "sed_common.h"
#include <type_traits>

// ---------------------------------------------------------------------------------------------
// K-B1: one workgroup (256 threads) per frame.  audio (B, N) in [-1, 1) -> out (B, M, n_mels), M = 1 + (N - 400) / 160.
// window[400] (povey), tw[256] complex exp(-2 pi i k / 512), sparse Kaldi-mel bank (start, len, weights[n_mels][stride]).
// ---------------------------------------------------------------------------------------------
#define FB_LEN 400
#define FB_SHIFT 160
#define FB_NFFT 512
__global__ __launch_bounds__(256) void kaldi_fbank_kernel(const float* __restrict__ audio, float* __restrict__ out, int N, int M,
                                                          int n_mels, const float* __restrict__ window, const float* __restrict__ tw,
                                                          const int* __restrict__ fb_start, const int* __restrict__ fb_len,
                                                          const float* __restrict__ fb_w, int fb_stride, float preemph, float scale,
                                                          float norm_mean, float norm_inv) {
    __shared__ float re[FB_NFFT], im[FB_NFFT], pw[FB_NFFT / 2 + 1], red[4];
    const int tid = threadIdx.x, f = blockIdx.x, b = blockIdx.y;
    const float* src = audio + (size_t)b * N + (size_t)f * FB_SHIFT;
    // frame + DC removal
    float v0 = tid < FB_LEN ? src[tid] * scale : 0.f, v1 = tid + 256 < FB_LEN ? src[tid + 256] * scale : 0.f;
    float s = wave_sum(v0 + v1);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)FB_LEN;
    // pre-emphasis needs the left neighbour: park the DC-free frame in `re` first (natural order)
    re[tid] = tid < FB_LEN ? v0 - mean : 0.f;
    re[tid + 256] = tid + 256 < FB_LEN ? v1 - mean : 0.f;
    __syncthreads();
    float y[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 256 * u;
        y[u] = i < FB_LEN ? (re[i] - preemph * re[i > 0 ? i - 1 : 0]) * window[i] : 0.f;
    }
    __syncthreads();
    // bit-reversed load for the radix-2 decimation-in-time FFT (9 stages of 256 butterflies)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 256 * u;
        const int r = (int)(__brev((unsigned)i) >> 23);
        re[r] = y[u];
        im[r] = 0.f;
    }
    __syncthreads();
    for (int st = 0; st < 9; ++st) {
        const int half = 1 << st, j = tid & (half - 1), i0 = ((tid >> st) << (st + 1)) + j, i1 = i0 + half;
        const int k = j << (8 - st);                              // twiddle exp(-2 pi i j / (2 half)) = tw[j * 256 / half]
        const float wr = tw[2 * k], wi = tw[2 * k + 1];
        const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
        const float tr = br * wr - bi * wi, ti = br * wi + bi * wr;
        re[i0] = ar + tr; im[i0] = ai + ti;
        re[i1] = ar - tr; im[i1] = ai - ti;
        __syncthreads();
    }
    pw[tid] = re[tid] * re[tid] + im[tid] * im[tid];
    if (tid == 0) pw[256] = re[256] * re[256] + im[256] * im[256];
    __syncthreads();
    if (tid < n_mels) {
        const int st = fb_start[tid], ln = fb_len[tid];
        const float* w = fb_w + (size_t)tid * fb_stride;
        float acc = 0.f;
        for (int i = 0; i < ln; ++i) acc = fmaf(w[i], pw[st + i], acc);
        const float v = logf(fmaxf(acc, 1.1920928955078125e-07f));
        out[((size_t)b * M + f) * n_mels + tid] = (v - norm_mean) * norm_inv;
    }
}
SED_API int sed_kaldi_fbank(const float* audio, float* out, int B, int N, int n_mels, const float* window, const float* tw,
                               const int* fb_start, const int* fb_len, const float* fb_w, int fb_stride, float norm_mean,
                               float norm_inv, void* stream) {
    if (n_mels < 1 || n_mels > 256) return SED_ERR_UNSUPPORTED;
    if (B <= 0 || N < FB_LEN) return SED_OK;
    const int M = 1 + (N - FB_LEN) / FB_SHIFT;
    SED_LAUNCH(kaldi_fbank_kernel, dim3(M, B), dim3(256), 0, (hipStream_t)stream, audio, out, N, M, n_mels, window, tw, fb_start,
               fb_len, fb_w, fb_stride, 0.97f, 32768.0f, norm_mean, norm_inv);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------
// K-B2: fbank (B, M, F) -> patches (B * Tp * Fp, P * P), Tp = M / P, Fp = F / P (floor: trailing frames are dropped exactly as
// Conv2d(stride = kernel = P) does), row = (b * Tp + tp) * Fp + fp (BEATs.py:154-156: tokens run frequency-fastest), column = i * P + j.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ fb, float* __restrict__ out, int M, int F, int P,
                                                       int Tp, int Fp, size_t n) {
    const int PP = P * P;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
        const size_t row = idx / PP;
        const int col = (int)(idx - row * PP), i = col / P, j = col - i * P;
        const int fp = (int)(row % Fp);
        const size_t r2 = row / Fp;
        const int tp = (int)(r2 % Tp), b = (int)(r2 / Tp);
        out[idx] = fb[((size_t)b * M + (size_t)tp * P + i) * F + fp * P + j];
    }
}
SED_API int sed_patchify(const float* fbank, float* patches, int B, int M, int F, int P, void* stream) {
    if (P < 1 || F % P != 0) return SED_ERR_ARG;
    const int Tp = M / P, Fp = F / P;
    const size_t n = (size_t)B * Tp * Fp * P * P;
    if (n == 0) return SED_OK;
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    SED_LAUNCH(patchify_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, fbank, patches, M, F, P, Tp, Fp, n);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------
// K-B3: y[m] = LayerNorm(alpha * res[m] + x[m]) * gamma + beta over D (one wave per row; D % 64 == 0, D <= 1024), eps 1e-5,
// biased variance (torch.nn.LayerNorm).  res may be null (plain LayerNorm); y may alias x.
// ---------------------------------------------------------------------------------------------
template <int NPL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ res, float alpha,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ y, int Mrows, float eps) {
    constexpr int D = NPL * 64;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Mrows) return;
    // all loads of the row are issued before the first use: with `res ? fmaf(alpha, res[i], x[i]) : x[i]` inside the loop every
    // element's loads sat in their own branch with s_waitcnt vmcnt(0) behind them (tools/isa_exposed_loads.py)
    float v[NPL];
    const size_t i0 = (size_t)row * D + lane;
#pragma unroll
    for (int u = 0; u < NPL; ++u) v[u] = x[i0 + 64 * u];
    if (res) {                                          // (uniform)
        float r[NPL];
#pragma unroll
        for (int u = 0; u < NPL; ++u) r[u] = res[i0 + 64 * u];
#pragma unroll
        for (int u = 0; u < NPL; ++u) v[u] = fmaf(alpha, r[u], v[u]);
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NPL; ++u) s += v[u];
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < NPL; ++u) { const float d = v[u] - mean; q = fmaf(d, d, q); }
    const float inv = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int u = 0; u < NPL; ++u) {
        const int c = lane + 64 * u;
        y[(size_t)row * D + c] = (v[u] - mean) * inv * gamma[c] + beta[c];
    }
}
SED_API int sed_layernorm(const float* x, const float* res, float alpha, const float* gamma, const float* beta, float* y, int M,
                             int D, float eps, void* stream) {
    if (M <= 0) return SED_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((M + 3) / 4);
#define LN_CASE(d) \
    if (D == d) { SED_LAUNCH((layernorm_kernel<d / 64>), grid, dim3(256), 0, s, x, res, alpha, gamma, beta, y, M, eps); return sed_check_launch(); }
    LN_CASE(128) LN_CASE(256) LN_CASE(512) LN_CASE(768) LN_CASE(1024)
#undef LN_CASE
    return SED_ERR_UNSUPPORTED;
}

// Round 6: the same LayerNorm that ALSO writes y as the K-tiled bf16 hi / lo image the next Linear's LDS-DMA kernel reads
// (sed_gemm_bf16.hip, sed_linear_tiles_bf16x3: block (row / 256, c / 16) = [hi | lo][256][16], octet o of row r at slot o ^ ((r >> 3) & 1)).
// A lane owns 4 consecutive channels (float4 in, float4 out, one 8-byte piece per plane); D % 256 == 0.  The sums run in a different
// order than layernorm_kernel's (4 channels per lane instead of every 64th), so y agrees with it to rounding, not bit for bit.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_tiles_kernel(const float* __restrict__ x, const float* __restrict__ x2, const float* __restrict__ res, float alpha,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ y, unsigned short* __restrict__ yt, int Mrows, float eps) {
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Mrows) return;
    float4 v[NV];
    const size_t i0 = (size_t)row * D + 4 * lane;
#pragma unroll
    for (int u = 0; u < NV; ++u) v[u] = *(const float4*)(x + i0 + 256 * u);
    if (x2) {                                           // (uniform) the second K half's partial sum of a split Linear
        float4 r[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) r[u] = *(const float4*)(x2 + i0 + 256 * u);
#pragma unroll
        for (int u = 0; u < NV; ++u) { v[u].x = sed_sadd(v[u].x, r[u].x); v[u].y = sed_sadd(v[u].y, r[u].y);
                                       v[u].z = sed_sadd(v[u].z, r[u].z); v[u].w = sed_sadd(v[u].w, r[u].w); }
    }
    if (res) {                                          // (uniform)
        float4 r[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) r[u] = *(const float4*)(res + i0 + 256 * u);
#pragma unroll
        for (int u = 0; u < NV; ++u) { v[u].x = fmaf(alpha, r[u].x, v[u].x); v[u].y = fmaf(alpha, r[u].y, v[u].y);
                                       v[u].z = fmaf(alpha, r[u].z, v[u].z); v[u].w = fmaf(alpha, r[u].w, v[u].w); }
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) s += sed_sadd(sed_sadd(v[u].x, v[u].y), sed_sadd(v[u].z, v[u].w));   // (scalar adds: the compiler's packed pair sum is the
                                                                                                       // op_sel-on-src1 form of sed_common.h's hazard note)
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const float a = v[u].x - mean, b = v[u].y - mean, c = v[u].z - mean, d = v[u].w - mean;
        q = fmaf(a, a, q); q = fmaf(b, b, q); q = fmaf(c, c, q); q = fmaf(d, d, q);
    }
    const float inv = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    const int panel = row >> 8, r8 = row & 255;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = 4 * lane + 256 * u;
        const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
        const float4 o = make_float4((v[u].x - mean) * inv * g.x + b.x, (v[u].y - mean) * inv * g.y + b.y, (v[u].z - mean) * inv * g.z + b.z,
                                     (v[u].w - mean) * inv * g.w + b.w);
        *(float4*)(y + (size_t)row * D + c) = o;
        unsigned h0, l0, h1, l1;
        bf16_split2(o.x, o.y, h0, l0);
        bf16_split2(o.z, o.w, h1, l1);
        unsigned short* d = yt + ((size_t)panel * (D / 16) + (c >> 4)) * 8192 + r8 * 16 + ((((c >> 3) & 1) ^ ((r8 >> 3) & 1)) << 3) + (c & 7);
        *(uint2*)d = make_uint2(h0, h1);
        *(uint2*)(d + 4096) = make_uint2(l0, l1);
    }
}
SED_API int sed_layernorm_tiles(const float* x, const float* x2, const float* res, float alpha, const float* gamma, const float* beta, float* y,
                                unsigned short* yt, int M, int D, float eps, void* stream) {
    if (!x || !gamma || !beta || !y || !yt || M < 0) return SED_ERR_ARG;
    if (M == 0) return SED_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((M + 3) / 4);
#define LN_CASE(d) \
    if (D == d) { SED_LAUNCH((layernorm_tiles_kernel<d / 256>), grid, dim3(256), 0, s, x, x2, res, alpha, gamma, beta, y, yt, M, eps); return sed_check_launch(); }
    LN_CASE(256) LN_CASE(512) LN_CASE(768) LN_CASE(1024)
#undef LN_CASE
    return SED_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// K-B4: y = x + GELU(bias + grouped Conv1d(x)), kernel K (even: the reference drops the last output, SamePad), padding K / 2,
// CG channels per group (backbone.py:30-43,118-120).  x, y (B, T, D); wt (D / CG groups, K, CG co, CG ci): the weight-normalised
// filter, transposed on the host once (frozen weights) so that a chunk of taps is one contiguous read.
// Workgroup = 64 tokens x one group: thread (tq = tid % 16, c = tid / 16) owns tokens 4 tq .. 4 tq + 3 x output channels
// 3 c .. 3 c + 2.  Odd LDS pitches: the 16 token quads of a wave hit 16 distinct banks, the co triples likewise.
// ---------------------------------------------------------------------------------------------
#define PC_CG 48
#define PC_TT 64
#define PC_KC 4
#define PC_XP 49
__global__ __launch_bounds__(256) void posconv_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                      const float* __restrict__ bias, float* __restrict__ y, int T, int D, int K) {
    SED_DYN_SMEM(smem);
    float* xs = (float*)smem;                                     // [PC_TT + K - 1][PC_XP]
    float* ws = xs + (PC_TT + K - 1) * PC_XP;                     // [PC_KC][PC_CG co][PC_XP ci]
    const int tid = threadIdx.x, tq = tid & 15, c = tid >> 4;
    const int t0 = blockIdx.x * PC_TT, g = blockIdx.y, b = blockIdx.z;
    const int nrows = PC_TT + K - 1, half = K / 2;
    for (int i = tid; i < nrows * PC_CG; i += 256) {
        const int r = i / PC_CG, ci = i - r * PC_CG, t = t0 - half + r;
        xs[r * PC_XP + ci] = (t >= 0 && t < T) ? x[((size_t)b * T + t) * D + g * PC_CG + ci] : 0.f;
    }
    float acc[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[j][q] = 0.f;
    const float* wg = wt + (size_t)g * K * PC_CG * PC_CG;
    for (int k0 = 0; k0 < K; k0 += PC_KC) {
        __syncthreads();
        for (int i = tid; i < PC_KC * PC_CG * PC_CG; i += 256) {
            const int kk = i / (PC_CG * PC_CG), rem = i - kk * PC_CG * PC_CG, co = rem / PC_CG, ci = rem - co * PC_CG;
            ws[(kk * PC_CG + co) * PC_XP + ci] = (k0 + kk < K) ? wg[(size_t)(k0 + kk) * PC_CG * PC_CG + rem] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < PC_KC; ++kk) {
            const float* xr = xs + (4 * tq + k0 + kk) * PC_XP;
            const float* wr = ws + (kk * PC_CG + 3 * c) * PC_XP;
#pragma unroll 4
            for (int ci = 0; ci < PC_CG; ++ci) {
                const float w0 = wr[ci], w1 = wr[PC_XP + ci], w2 = wr[2 * PC_XP + ci];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xv = xr[j * PC_XP + ci];
                    acc[j][0] = fmaf(xv, w0, acc[j][0]);
                    acc[j][1] = fmaf(xv, w1, acc[j][1]);
                    acc[j][2] = fmaf(xv, w2, acc[j][2]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = t0 + 4 * tq + j;
        if (t < T) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int ch = g * PC_CG + 3 * c + q;
                const float v = acc[j][q] + bias[ch];
                const size_t o = ((size_t)b * T + t) * D + ch;
                y[o] = x[o] + 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
            }
        }
    }
}
// ---------------------------------------------------------------------------------------------
// K-B4 on the matrix cores: the grouped position convolution as a GEMM per (clip, group) -- M = tokens, N = 48 output channels,
// K = 128 taps x 48 input channels -- on v_mfma_f32_16x16x32_bf16 with split operands (three MFMAs per product).
// A operand: 8 consecutive input channels of token t + tap - K/2 = one 16-byte read from the [token + halo][48] hi / lo planes of
// the tile (pitch 48 bf16 = 6 slots = 2 mod 4: conflict-free); B operand: 8 consecutive input channels of (tap, output channel)
// from the frozen weights, split into bf16 hi / lo planes ONCE on the host side (beats.py::_pack), streamed through LDS four
// taps at a time by plain 16-byte copies.  Workgroup = 64 tokens x one group, wave = 16 tokens x the 48 output channels.
// ---------------------------------------------------------------------------------------------
#define PCM_TT 64
#define PCM_KC 4
__global__ __launch_bounds__(256) void posconv_mfma_kernel(const float* __restrict__ x, const unsigned short* __restrict__ wsplit,
                                                           const float* __restrict__ bias, float* __restrict__ y, int T, int D, int K,
                                                           int G) {
    SED_DYN_SMEM(smem);
    const int nrows = PCM_TT + K - 1, XPL = nrows * PC_CG, WPL = PCM_KC * PC_CG * PC_CG;
    unsigned short* xh = (unsigned short*)smem;           // [nrows][48] hi | lo
    unsigned short* wh = xh + 2 * XPL;                    // [PCM_KC][48 co][48 ci] hi | lo
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const int t0 = blockIdx.x * PCM_TT, grp = blockIdx.y, b = blockIdx.z, half = K / 2;
    for (int i = tid; i < nrows * (PC_CG / 4); i += 256) {
        const int r = i / (PC_CG / 4), c4 = i - r * (PC_CG / 4), t = t0 - half + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < T) v = *(const float4*)(x + ((size_t)b * T + t) * D + grp * PC_CG + 4 * c4);
        uint2 hv, lv;
        bf16_split2(v.x, v.y, hv.x, lv.x);
        bf16_split2(v.z, v.w, hv.y, lv.y);
        *(uint2*)(xh + r * PC_CG + 4 * c4) = hv;
        *(uint2*)(xh + XPL + r * PC_CG + 4 * c4) = lv;
    }
    f32x4 acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const size_t plane = (size_t)G * K * PC_CG * PC_CG;                         // elements per hi / lo plane of the weights
    // The next chunk's weights travel HBM -> registers under this chunk's MFMAs (round 5; they used to be copied global -> LDS between
    // the two barriers, one exposed memory latency per 54 MFMAs: wait_any 0.68, profiles/r05i_pmc_beats_wait.md)
    // (nine named registers: as an array the compiler kept them in scratch memory -- loads waited for one by one)
    static_assert(2 * (PCM_KC * PC_CG * PC_CG / 8) == 9 * 256, "nine 16-byte pieces per thread and chunk");
#define PCM_EACH(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#define PCM_DECL(q)                                                                                                          \
    uint4 w##q;                                                                                                              \
    const int wp##q = (tid + 256 * q) / (WPL / 8), we##q = (tid + 256 * q) - wp##q * (WPL / 8);                              \
    const unsigned short* const ws##q = wsplit + (size_t)grp * K * PC_CG * PC_CG + wp##q * plane + 8 * (size_t)we##q;        \
    unsigned short* const wd##q = wh + wp##q * WPL + 8 * we##q;
    PCM_EACH(PCM_DECL)
#define PCM_LOAD(q) w##q = *(const uint4*)(ws##q + koff);
#define PCM_PARK(q) *(uint4*)wd##q = w##q;
    {
        const size_t koff = 0;
        PCM_EACH(PCM_LOAD)
    }
    for (int k0 = 0; k0 < K; k0 += PCM_KC) {
        __syncthreads();                                                         // previous chunk consumed (first pass: x staged)
        PCM_EACH(PCM_PARK)
        __syncthreads();
        if (k0 + PCM_KC < K) {
            const size_t koff = (size_t)(k0 + PCM_KC) * PC_CG * PC_CG;
            PCM_EACH(PCM_LOAD)
        }
#pragma unroll
        for (int ks = 0; ks < PCM_KC * PC_CG / 32; ++ks) {                       // 6 k-steps of 32 = 4 octets of (tap, 8 channels)
            const int o = 4 * ks + g, tapl = o / (PC_CG / 8), c8 = o - tapl * (PC_CG / 8);
            const unsigned short* ap = xh + (16 * w + i16 + k0 + tapl) * PC_CG + 8 * c8;
            const s16x8 ah = *(const s16x8*)ap, al = *(const s16x8*)(ap + XPL);
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) {
                const unsigned short* bp = wh + (tapl * PC_CG + 16 * nb + i16) * PC_CG + 8 * c8;
                const s16x8 bh = *(const s16x8*)bp, bl = *(const s16x8*)(bp + WPL);
                acc[nb] = mfma16_bf16(al, bh, acc[nb]);
                acc[nb] = mfma16_bf16(ah, bl, acc[nb]);
                acc[nb] = mfma16_bf16(ah, bh, acc[nb]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + 16 * w + 4 * g + r;
        if (t < T) {
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) {
                const int ch = grp * PC_CG + 16 * nb + i16;
                const float v = acc[nb][r] + bias[ch];
                const size_t oidx = ((size_t)b * T + t) * D + ch;
                y[oidx] = x[oidx] + 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
            }
        }
    }
}
// wsplit: (2 planes hi | lo, groups, K, 48 co, 48 ci) bf16 bit patterns of the weight-normalised filter
SED_API int sed_posconv_bf16x3(const float* x, const unsigned short* wsplit, const float* bias, float* y, int B, int T, int D, int K,
                                  int groups, void* stream) {
    if (groups < 1 || D % groups != 0 || D / groups != PC_CG || K < 2 || (K & 1) || K % PCM_KC != 0 || D % 4 != 0) return SED_ERR_UNSUPPORTED;
    if (B <= 0 || T <= 0) return SED_OK;
    const int smem = (2 * (PCM_TT + K - 1) * PC_CG + 2 * PCM_KC * PC_CG * PC_CG) * 2;
    if (smem > 150 * 1024) return SED_ERR_UNSUPPORTED;
    SED_MAX_SMEM(posconv_mfma_kernel, smem);
    SED_LAUNCH(posconv_mfma_kernel, dim3((T + PCM_TT - 1) / PCM_TT, groups, B), dim3(256), smem, (hipStream_t)stream, x, wsplit, bias, y,
               T, D, K, groups);
    return sed_check_launch();
}

SED_API int sed_posconv(const float* x, const float* wt, const float* bias, float* y, int B, int T, int D, int K, int groups,
                           void* stream) {
    if (groups < 1 || D % groups != 0 || D / groups != PC_CG || K < 2 || (K & 1) || K % PC_KC != 0) return SED_ERR_UNSUPPORTED;
    if (B <= 0 || T <= 0) return SED_OK;
    const int smem = ((PC_TT + K - 1) * PC_XP + PC_KC * PC_CG * PC_XP) * 4;
    SED_MAX_SMEM(posconv_kernel, smem);
    SED_LAUNCH(posconv_kernel, dim3((T + PC_TT - 1) / PC_TT, groups, B), dim3(256), smem, (hipStream_t)stream, x, wt, bias, y, T, D, K);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------
// K-B5: attention.  qkv (B * T, 3 D): row = b * T + t, [q | k | v], head h at columns h * 64.  out (B * T, D).
// scores(t, s) = q_t . k_s / 8 + gate(b, h, t) * relb[h][s - t + T - 1]   (the reference's q / 32, (s - max) * 32 is this up to the
// softmax's shift invariance, backbone.py:529-531,640-643); gate = ga * (gb * grep_a[h] - 1) + 2 with
// (ga, gb) = sigmoid(sum of 4 | sum of 4 of grep_linear(q_t))  (:662-682).  relb (H, 2 T - 1): the bucket embedding gathered per
// relative offset on the host (it depends on s - t only, :390-444); null = no position bias; grep_w null = ungated bias.
// Workgroup = 64 queries of one (b, h); 4 lanes per query, lane p owns keys 16 p .. 16 p + 15 of every 64-key tile for the scores
// and output dims 16 p .. 16 p + 15 for P V; online softmax; K, V and P tiles in LDS (row pitch 65).
// ---------------------------------------------------------------------------------------------
#define AT_HD 64
#define AT_TQ 64
#define AT_P 65
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, const float* __restrict__ relb,
                                                        const float* __restrict__ grep_w, const float* __restrict__ grep_b,
                                                        const float* __restrict__ grep_a, float* __restrict__ out, int T, int H,
                                                        float scaling) {
    SED_DYN_SMEM(smem);
    float* ks = (float*)smem;                  // [64][65]
    float* vs = ks + AT_TQ * AT_P;             // [64][65]
    float* ps = vs + AT_TQ * AT_P;             // [64][65]
    float* rb = ps + AT_TQ * AT_P;             // [2 T - 1]
    const int tid = threadIdx.x, r = tid >> 2, p = tid & 3;
    const int q0 = blockIdx.x * AT_TQ, h = blockIdx.y, b = blockIdx.z, D = H * AT_HD, LD = 3 * D;
    const int t = q0 + r;
    const bool live = t < T;
    if (relb)
        for (int i = tid; i < 2 * T - 1; i += 256) rb[i] = relb[(size_t)h * (2 * T - 1) + i];
    float q[AT_HD];
    {
        const float* qr = qkv + ((size_t)b * T + (live ? t : 0)) * LD + h * AT_HD;
#pragma unroll
        for (int d = 0; d < AT_HD; d += 4) {
            const float4 v = *(const float4*)(qr + d);
            q[d] = v.x; q[d + 1] = v.y; q[d + 2] = v.z; q[d + 3] = v.w;
        }
    }
    float gate = relb ? 1.0f : 0.0f;
    if (relb && grep_w) {
        float ga = 0.f, gb = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float a = grep_b[o];
#pragma unroll
            for (int d = 0; d < AT_HD; ++d) a = fmaf(grep_w[o * AT_HD + d], q[d], a);
            if (o < 4) ga += a; else gb += a;
        }
        ga = sed_sigmoid(ga); gb = sed_sigmoid(gb);
        gate = ga * (gb * grep_a[h] - 1.0f) + 2.0f;
    }
    float m = -INFINITY, l = 0.f, o[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) o[d] = 0.f;
    for (int s0 = 0; s0 < T; s0 += AT_TQ) {
        __syncthreads();                                           // previous tile consumed (and rb staged)
        for (int i = tid; i < AT_TQ * (AT_HD / 4); i += 256) {
            const int kr = i / (AT_HD / 4), d4 = i - kr * (AT_HD / 4), s = s0 + kr;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (s < T) {
                const float* base = qkv + ((size_t)b * T + s) * LD + h * AT_HD + 4 * d4;
                kv = *(const float4*)(base + D);
                vv = *(const float4*)(base + 2 * D);
            }
            float* kd = ks + kr * AT_P + 4 * d4;
            float* vd = vs + kr * AT_P + 4 * d4;
            kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
            vd[0] = vv.x; vd[1] = vv.y; vd[2] = vv.z; vd[3] = vv.w;
        }
        __syncthreads();
        float sc[16], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int kk = 16 * p + j, s = s0 + kk;
            const float* kr = ks + kk * AT_P;
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < AT_HD; ++d) a = fmaf(q[d], kr[d], a);
            a *= scaling;
            if (relb && live && s < T) a = fmaf(gate, rb[s - t + T - 1], a);
            sc[j] = s < T ? a : -INFINITY;
            mx = fmaxf(mx, sc[j]);
        }
        mx = fmaxf(mx, sed_quad_xor1(mx));
        mx = fmaxf(mx, sed_quad_xor2(mx));
        const float mn = fmaxf(m, mx);
        const float corr = expf(m - mn);                           // m = -inf on the first tile: exp(-inf) = 0
        float ls = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float e = expf(sc[j] - mn);
            ps[r * AT_P + 16 * p + j] = e;
            ls += e;
        }
        ls = sed_quad_sum(ls);
        l = l * corr + ls;
        m = mn;
        __syncthreads();                                           // the row's four lanes wrote its P values
#pragma unroll
        for (int d = 0; d < 16; ++d) o[d] *= corr;
        for (int kk = 0; kk < AT_TQ; ++kk) {
            const float pv = ps[r * AT_P + kk];
            const float* vr = vs + kk * AT_P + 16 * p;
#pragma unroll
            for (int d = 0; d < 16; ++d) o[d] = fmaf(pv, vr[d], o[d]);
        }
    }
    if (live) {
        const float inv = 1.0f / l;
        float* dst = out + ((size_t)b * T + t) * D + h * AT_HD + 16 * p;
#pragma unroll
        for (int d = 0; d < 16; d += 4) *(float4*)(dst + d) = make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
    }
}
// ---------------------------------------------------------------------------------------------
// K-B5 on the matrix cores (default): the same attention, Q K^T and P V on v_mfma_f32_16x16x32_bf16 with split operands (three
// MFMAs per product: fp32-level accuracy, the extractor's parity bound is 2e-4).  Round 5: the TRANSPOSED products, so that the
// probabilities never leave the registers (rounds 2-4 computed S = Q K^T, whose accumulator layout had to go through wave-private
// LDS planes -- 32 two-byte stores per lane and tile -- to become the A operand of P V; with the 4-rows-per-lane softmax and IEEE
// expf the vector pipe did ~550 instructions per 48 MFMAs: MFMA-busy 0.10, profiles/r05g_pmc_beats_mfma_busy.md).
//   Workgroup = 128 queries of one (b, h); wave w = queries 32 w .. 32 w + 31 as TWO 16-query blocks j that share every K / V
//   fragment read (LDS operand traffic per MFMA halves); key tiles of 64.
//   S^T[key][query] = K Q^T:  A = K rows from [key][dim] hi / lo planes, B = Q (lane (i16, g) = query i16, dims 8 g + e of a
//            32-dim half; scaled by log2(e) / sqrt(d), as bf16 hi / lo fragments in registers for the whole kernel).
//            D[key = 16 kb + 4 g + r][query = i16]: a lane holds 16 scores OF ONE QUERY per block.
//   softmax: per query = per lane: max / sum over the lane's 16 values, then over the 4 lanes of the column (xor 16, xor 32); ONE
//            running maximum, sum and correction per lane and block; base-2 exponentials (v_exp_f32); the gated relative-position
//            bias (pre-multiplied by log2 e when staged) is one LDS base + 16 immediate offsets per lane.
//   O^T[dim][query] = V^T P^T: B = P^T comes STRAIGHT from the score accumulators -- the contraction index of an MFMA may be any
//            permutation of the keys as long as both operands use it: slot (g, e) of K-half ks := key 16 (2 ks + (e >> 2)) + 4 g +
//            (e & 3), which is exactly what lane (query, g) holds; A = V^T from [dim][slot] hi / lo planes staged in that order.
//            D[dim = 16 db + 4 g + r][query = i16]: the result leaves as 16-byte stores.
// Plane pitch 80 bf16 = 10 sixteen-byte slots (= 2 mod 4: conflict-free ds_read_b128 for the 16x16x32 lane groups).
// ---------------------------------------------------------------------------------------------
#define ATM_P 80
#define ATM_QG 128
__device__ __forceinline__ float atm_exp2(float x) {
#ifdef SED_EMU
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);
#endif
}
template <bool BIAS>
__global__ __launch_bounds__(256, 2) void attention_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ relb,
                                                             const float* __restrict__ grep_w, const float* __restrict__ grep_b,
                                                             const float* __restrict__ grep_a, float* __restrict__ out, int T, int H,
                                                             int n_pairs, float scaling) {
    SED_DYN_SMEM(smem);
    unsigned short* kh = (unsigned short*)smem;           // K   [64 keys][ATM_P] hi | lo
    unsigned short* kl = kh + 64 * ATM_P;
    unsigned short* vh = kl + 64 * ATM_P;                 // V^T [64 dims][ATM_P] hi | lo, keys in MFMA-slot order
    unsigned short* vl = vh + 64 * ATM_P;
    float* rb = (float*)(vl + 64 * ATM_P);                // [2 T - 1 (+ 64 zeros)] bias row of this head, times log2 e
    const float LOG2E = 1.44269504088896341f;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i16 = lane & 15, g = lane >> 4;
    // XCD-aware walk: workgroup n runs on XCD n & 7; the query tiles of one (b, h) are consecutive workgroups OF ONE XCD, so that head's
    // K / V rows (254 KB) are fetched into one L2 once instead of into four
    const int nq = (T + ATM_QG - 1) / ATM_QG, mloc = blockIdx.x >> 3, pair = (blockIdx.x & 7) + 8 * (mloc / nq);
    if (pair >= n_pairs) return;
    const int q0 = (mloc % nq) * ATM_QG, h = pair % H, b = pair / H, D = H * AT_HD, LD = 3 * D;
    if (BIAS)
        for (int i = tid; i < 2 * T - 1 + 64; i += 256) rb[i] = i < 2 * T - 1 ? relb[(size_t)h * (2 * T - 1) + i] * LOG2E : 0.f;
    // ---- Q fragments and gates: lane (i16, g) = query q0 + 32 w + 16 j + i16, dims 32 ks + 8 g + e ----
    s16x8 qh[2][2], ql[2][2];
    float gate[2], m2[2], l2[2];
    int tq[2], rbase[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        tq[j] = q0 + 32 * w + 16 * j + i16;
        const int tc = tq[j] < T ? tq[j] : T - 1;                      // dead queries compute on a clamped row and are not stored
        rbase[j] = T - 1 - tc + 4 * g;
        const float* qr = qkv + ((size_t)b * T + tc) * LD + h * AT_HD;
        float qv[16];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float4 a = *(const float4*)(qr + 32 * ks + 8 * g), c = *(const float4*)(qr + 32 * ks + 8 * g + 4);
            qv[8 * ks] = a.x; qv[8 * ks + 1] = a.y; qv[8 * ks + 2] = a.z; qv[8 * ks + 3] = a.w;
            qv[8 * ks + 4] = c.x; qv[8 * ks + 5] = c.y; qv[8 * ks + 6] = c.z; qv[8 * ks + 7] = c.w;
        }
        float gt = BIAS ? 1.0f : 0.0f;
        if (BIAS && grep_w) {
            float ga = 0.f, gb = 0.f;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                float a = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) a = sed_sfma(grep_w[o * AT_HD + 32 * ks + 8 * g + e], qv[8 * ks + e], a);
                // (sed_sfma: the compiler paired two outputs o and broadcast qv[odd] with v_pk_fma_f32 ... op_sel:[0,1,0] -- the form
                //  sed_common.h's "gfx950 hazard" note forbids, in a kernel whose other waves issue bf16 MFMAs; 128 FMAs per workgroup)
                a += __shfl_xor(a, 16);
                a += __shfl_xor(a, 32);
                a += grep_b[o];
                if (o < 4) ga += a; else gb += a;
            }
            ga = sed_sigmoid(ga); gb = sed_sigmoid(gb);
            gt = ga * (gb * grep_a[h] - 1.0f) + 2.0f;
        }
        gate[j] = gt;
        m2[j] = -INFINITY; l2[j] = 0.f;
        const float qs = scaling * LOG2E;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            unsigned hv[4], lv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) bf16_split2(qv[8 * ks + 2 * e] * qs, qv[8 * ks + 2 * e + 1] * qs, hv[e], lv[e]);
            uint4 hq, lq;
            hq.x = hv[0]; hq.y = hv[1]; hq.z = hv[2]; hq.w = hv[3];
            lq.x = lv[0]; lq.y = lv[1]; lq.z = lv[2]; lq.w = lv[3];
            qh[j][ks] = __builtin_bit_cast(s16x8, hq);
            ql[j][ks] = __builtin_bit_cast(s16x8, lq);
        }
    }
    f32x4 o[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int db = 0; db < 4; ++db) o[j][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kq = tid & 15, dq = tid >> 4;                            // staging block: keys 4 kq .. +3, dims 4 dq .. +3
    // keys 4 kq .. 4 kq + 3 = (kb = kq >> 2, g = kq & 3, r = 0..3) sit at MFMA slots 32 (kb >> 1) + 8 g + 4 (kb & 1) + r of V^T
    const int vpos = 32 * (kq >> 3) + 8 * (kq & 3) + 4 * ((kq >> 2) & 1);
    // The K / V rows of the NEXT key tile are fetched into registers under this tile's MFMAs and softmax (unconditional loads on
    // clamped rows; keys past T are zeroed when the tile is parked in LDS).
    float4 kr[4], vr[4];
    // (K needs no transposition, so its rows are fetched and parked row-contiguously: lanes = 16 dim quads of key dq + 16 j.  With
    // V's 4 x 4 blocks the 8-byte LDS stores of 16 lanes fell on two bank groups: lds_conflict 0.61 of the LDS cycles)
    auto load_kv = [&](int s0_) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = s0_ + 4 * kq + j, sk = s0_ + dq + 16 * j;
            vr[j] = *(const float4*)(qkv + ((size_t)b * T + (s < T ? s : T - 1)) * LD + h * AT_HD + 4 * dq + 2 * D);
            kr[j] = *(const float4*)(qkv + ((size_t)b * T + (sk < T ? sk : T - 1)) * LD + h * AT_HD + 4 * kq + D);
        }
    };
    load_kv(0);
    for (int s0 = 0; s0 < T; s0 += 64) {
        __syncthreads();                                               // previous tile consumed (first pass: rb staged)
        {
            if (s0 + 64 > T) {                                         // (uniform: the ragged last tile only)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (s0 + 4 * kq + j >= T) vr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (s0 + dq + 16 * j >= T) kr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {                              // K rows: dims 4 kq .. + 3 of key dq + 16 j
                uint2 hv, lv;
                bf16_split2(kr[j].x, kr[j].y, hv.x, lv.x);
                bf16_split2(kr[j].z, kr[j].w, hv.y, lv.y);
                *(uint2*)(kh + (dq + 16 * j) * ATM_P + 4 * kq) = hv;
                *(uint2*)(kl + (dq + 16 * j) * ATM_P + 4 * kq) = lv;
            }
            auto vt = [&](float a, float c, float e, float f, int d) {  // V^T rows: 4 keys of dim d
                uint2 hv, lv;
                bf16_split2(a, c, hv.x, lv.x);
                bf16_split2(e, f, hv.y, lv.y);
                *(uint2*)(vh + d * ATM_P + vpos) = hv;
                *(uint2*)(vl + d * ATM_P + vpos) = lv;
            };
            vt(vr[0].x, vr[1].x, vr[2].x, vr[3].x, 4 * dq);
            vt(vr[0].y, vr[1].y, vr[2].y, vr[3].y, 4 * dq + 1);
            vt(vr[0].z, vr[1].z, vr[2].z, vr[3].z, 4 * dq + 2);
            vt(vr[0].w, vr[1].w, vr[2].w, vr[3].w, 4 * dq + 3);
        }
        __syncthreads();
        if (s0 + 64 < T) load_kv(s0 + 64);
        // ---- scores, transposed: four 16-key blocks x two query blocks ----
        f32x4 sc[2][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned short* ap = kh + (16 * kb + i16) * ATM_P + 32 * ks + 8 * g;
                const s16x8 ah = *(const s16x8*)ap, al = *(const s16x8*)(ap + 64 * ATM_P);
                a0 = mfma16_bf16(al, qh[0][ks], a0);
                a1 = mfma16_bf16(al, qh[1][ks], a1);
                a0 = mfma16_bf16(ah, ql[0][ks], a0);
                a1 = mfma16_bf16(ah, ql[1][ks], a1);
                a0 = mfma16_bf16(ah, qh[0][ks], a0);
                a1 = mfma16_bf16(ah, qh[1][ks], a1);
            }
            sc[0][kb] = a0;
            sc[1][kb] = a1;
        }
        // ---- bias, mask, online softmax in base 2: everything of a query is in its lane column ----
        s16x8 ph[2][2], pl[2][2];
        // (BIAS and the ragged last tile are compile-time / loop-versioned: as per-element conditions they became 32 basic blocks,
        // each one LDS read waited for on its own)
        auto soft = [&](auto ragged_c) {
            constexpr bool RAGGED = decltype(ragged_c)::value;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float bias[16];
                if (BIAS) {
                    const float* rbj = rb + rbase[j] + s0;
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) bias[4 * kb + r] = rbj[16 * kb + r];
                }
                float mx = -INFINITY;
                const float gj = gate[j];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = sc[j][kb][r];
                        if (BIAS) v = fmaf(gj, bias[4 * kb + r], v);
                        if (RAGGED) v = (s0 + 16 * kb + 4 * g + r) < T ? v : -INFINITY;
                        sc[j][kb][r] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float mn = fmaxf(m2[j], mx);
                const float corr = atm_exp2(m2[j] - mn);               // m = -inf on the first tile: exp2(-inf) = 0
                float ls = 0.f;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = atm_exp2(sc[j][kb][r] - mn);
                        sc[j][kb][r] = e;
                        ls += e;
                    }
                ls += __shfl_xor(ls, 16);
                ls += __shfl_xor(ls, 32);
                l2[j] = l2[j] * corr + ls;
                m2[j] = mn;
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[j][db][r] *= corr;
                // P^T fragments: slot e of K-half ks = sc[2 ks + (e >> 2)][e & 3]
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    uint4 hq, lq;
                    bf16_split2(sc[j][2 * ks][0], sc[j][2 * ks][1], hq.x, lq.x);
                    bf16_split2(sc[j][2 * ks][2], sc[j][2 * ks][3], hq.y, lq.y);
                    bf16_split2(sc[j][2 * ks + 1][0], sc[j][2 * ks + 1][1], hq.z, lq.z);
                    bf16_split2(sc[j][2 * ks + 1][2], sc[j][2 * ks + 1][3], hq.w, lq.w);
                    ph[j][ks] = __builtin_bit_cast(s16x8, hq);
                    pl[j][ks] = __builtin_bit_cast(s16x8, lq);
                }
            }
        };
        if (s0 + 64 > T) soft(std::true_type{}); else soft(std::false_type{});
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const unsigned short* ap = vh + (16 * db + i16) * ATM_P + 32 * ks + 8 * g;
                const s16x8 ah = *(const s16x8*)ap, al = *(const s16x8*)(ap + 64 * ATM_P);
                o[0][db] = mfma16_bf16(al, ph[0][ks], o[0][db]);
                o[1][db] = mfma16_bf16(al, ph[1][ks], o[1][db]);
                o[0][db] = mfma16_bf16(ah, pl[0][ks], o[0][db]);
                o[1][db] = mfma16_bf16(ah, pl[1][ks], o[1][db]);
                o[0][db] = mfma16_bf16(ah, ph[0][ks], o[0][db]);
                o[1][db] = mfma16_bf16(ah, ph[1][ks], o[1][db]);
            }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (tq[j] < T) {
            const float inv = 1.0f / l2[j];
            float* dst = out + ((size_t)b * T + tq[j]) * D + h * AT_HD + 4 * g;
#pragma unroll
            for (int db = 0; db < 4; ++db)
                *(float4*)(dst + 16 * db) = make_float4(o[j][db][0] * inv, o[j][db][1] * inv, o[j][db][2] * inv, o[j][db][3] * inv);
        }
}
SED_API int sed_attention_relpos(const float* qkv, const float* relb, const float* grep_w, const float* grep_b,
                                    const float* grep_a, float* out, int B, int T, int H, int head_dim, void* stream) {
    if (head_dim != AT_HD || T > 4096) return SED_ERR_UNSUPPORTED;
    if (B <= 0 || T <= 0) return SED_OK;
    if (sed_tuning[SED_TUNE_ATTN_VALU] == 0) {          // default: the matrix-core kernel
        const int smem_m = 4 * 64 * ATM_P * 2 + (64 + 2 * T) * 4;
        const int nwg = ((T + ATM_QG - 1) / ATM_QG) * ((H * B + 7) / 8) * 8;
        if (relb) {
            SED_MAX_SMEM(attention_mfma_kernel<true>, smem_m);
            SED_LAUNCH(attention_mfma_kernel<true>, dim3(nwg), dim3(256), smem_m, (hipStream_t)stream, qkv, relb,
                       grep_w, grep_b, grep_a, out, T, H, H * B, 1.0f / sqrtf((float)head_dim));
        } else {
            SED_MAX_SMEM(attention_mfma_kernel<false>, smem_m);
            SED_LAUNCH(attention_mfma_kernel<false>, dim3(nwg), dim3(256), smem_m, (hipStream_t)stream, qkv, relb,
                       grep_w, grep_b, grep_a, out, T, H, H * B, 1.0f / sqrtf((float)head_dim));
        }
        return sed_check_launch();
    }
    const int smem = (3 * AT_TQ * AT_P + 2 * T) * 4;
    SED_MAX_SMEM(attention_kernel, smem);
    SED_LAUNCH(attention_kernel, dim3((T + AT_TQ - 1) / AT_TQ, H, B), dim3(256), smem, (hipStream_t)stream, qkv, relb, grep_w, grep_b,
               grep_a, out, T, H, 1.0f / sqrtf((float)head_dim));
    return sed_check_launch();
}
