// K6 (activation half): BatchNorm-apply + GLU + Dropout + AvgPool of one CNN block, fused, forward and
// backward.  Reference: desed_task/nnet/CNN.py:11-16 (GLU = Linear_CxC(x) * sigmoid(x), both branches on
// the BN output), :76 (BatchNorm2d eps 1e-3), :90-91 (Dropout), :96-98 (AvgPool2d, floor mode).
//
// Forward : y (B,T,F,C) pre-BN conv output -> out (B,T/PT,F/PF,C).  One read of y, one (4x or 2x
//           smaller) write; the CxC GLU linear runs on the exact-f32 MFMA with the weight resident in LDS.
// Backward: g_out -> dz = dL/d(xhat) (B,T,F,C) plus the reductions dgamma, dbeta (BatchNorm), dWg, dbg
//           (GLU linear): three MFMA GEMMs per tile sharing one LDS image of xhat; nothing but y was saved
//           by the forward (xn, lin, sigmoid and the dropout mask are recomputed).
// bn_bwd_apply then turns dz into dy = dL/d(conv output) in place (training-mode BN backward).
//
// M tile rows are ordered by pooling window (row = window*WIN + q), so that the MFMA accumulator groups
// of 4 consecutive rows held by one lane are exactly one 2x2 window (or two 1x2 windows): pooling and
// un-pooling are lane-local.
#include "sed_common.h"

// GLU_ABL: timing-ablation mask for tools/glu_variants.py (0 in the product build): 1 = no MFMA, 2 = no epilogue
// math, 4 = no global tile loads, 8 = no global stores; glu32_bwd_kernel: 16 = no identity-operand MFMAs, 32 = no GEMM3.
#ifndef GLU_ABL
#define GLU_ABL 0
#endif

template <int C, int PT, int PF>
struct GluGeom {
    static constexpr int WIN = PT * PF;
    static constexpr int CP = C + 1;
    static constexpr int NT = C >= 32 ? C / 32 : 1;
};

// map row m of a tile starting at window o0 to the input pixel; returns false when the window is out of range
// Fo = 1 << fsh (F is a power of two by the conv kernels' contract), so the window -> (to, fo) split is shift/mask.
template <int PT, int PF>
__device__ __forceinline__ bool row_pixel(int m, int o0, int NWC, int fsh, int& o, int& t, int& f) {
    constexpr int WIN = PT * PF;
    const int q = m % WIN;
    o = o0 + m / WIN;
    const int to = o >> fsh, fo = o & ((1 << fsh) - 1);
    t = to * PT + q / PF;
    f = fo * PF + q % PF;
    return o < NWC;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int C, int PT, int PF>
__global__ __launch_bounds__(256) void glu_fwd_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                      const float* __restrict__ Wg, const float* __restrict__ bg,
                                                      float* __restrict__ out, int B, int T, int F, uint32_t seed,
                                                      uint32_t thr24, float dscale, const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;            // per-step entropy in device memory (hipGraph replays)
    using G = GluGeom<C, PT, PF>;
    constexpr int WIN = G::WIN, CP = G::CP, NT = G::NT, ROWS = 128, NW = ROWS / WIN;
    SED_DYN_SMEM(smem);
    float* wg = (float*)smem;           // [C][CP]
    float* xs = wg + C * CP;            // [ROWS][CP]  BN output xn
    float* sc = xs + ROWS * CP;         // scale[C], shift[C], bg[C]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int To = T / PT, Fo = F / PF, NWC = To * Fo;
    const int fsh = 31 - __builtin_clz(Fo);
    const int tiles_per_clip = (NWC + NW - 1) / NW, ntiles = B * tiles_per_clip;

    for (int i = tid; i < C * C; i += 256) wg[(i / C) * CP + (i % C)] = Wg[i];
    if (tid < C) { sc[tid] = stats[2 * C + tid]; sc[C + tid] = stats[3 * C + tid]; sc[2 * C + tid] = bg[tid]; }

    // the next tile's rows are fetched into registers under the current tile's MFMAs / epilogue (rows past the clip: zeros)
    constexpr int NLD = ROWS * (C / 4) / 256;
    float4 ld[NLD];
    auto load_tile = [&](int tile_) {
        const int b_ = tile_ / tiles_per_clip, o0_ = (tile_ - b_ * tiles_per_clip) * NW;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = tid + 256 * u, m = idx / (C / 4), v = idx - m * (C / 4);
            int o, t, f;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_pixel<PT, PF>(m, o0_, NWC, fsh, o, t, f)) val = *(const float4*)(y + (((size_t)b_ * T + t) * F + f) * C + 4 * v);
            ld[u] = val;
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_clip, o0 = (tile - b * tiles_per_clip) * NW;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = tid + 256 * u, m = idx / (C / 4), v = idx - m * (C / 4);
            int o, t, f;
            float4 val = ld[u];
            if (row_pixel<PT, PF>(m, o0, NWC, fsh, o, t, f)) {
                const float* s = sc + 4 * v;
                val.x = fmaf(val.x, s[0], s[C + 0]); val.y = fmaf(val.y, s[1], s[C + 1]);
                val.z = fmaf(val.z, s[2], s[C + 2]); val.w = fmaf(val.w, s[3], s[C + 3]);
            }
            float* d = xs + m * CP + 4 * v;
            d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x16_zero();
        const float* ap = xs + (32 * w + lo) * CP + hi;
#pragma unroll 8
        for (int k = 0; k < C; k += 2) {
            const float av = ap[k];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float bv = (C >= 32 || lo < C) ? wg[(nt * 32 + lo) * CP + k + hi] : 0.f;
                acc[nt] = mfma32(av, bv, acc[nt]);
            }
        }
        // ---- epilogue: gate, dropout, pool (row -> pixel arithmetic once per row, shared by all channel tiles) ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mbase = 32 * w + 8 * j + 4 * hi;
            uint32_t ebase[4];
            bool okr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int o, t, f;
                okr[q] = row_pixel<PT, PF>(mbase + q, o0, NWC, fsh, o, t, f);
                ebase[q] = (uint32_t)((((size_t)b * T + t) * F + f) * C);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = nt * 32 + lo;
                if (n < C) {
                    const float bias = sc[2 * C + n];
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float xn = xs[(mbase + q) * CP + n];
                        float r = (acc[nt][4 * j + q] + bias) * sed_fast_sigmoid(xn);
                        r = (okr[q] && sed_keep(ebase[q] + n, seed, thr24)) ? r * dscale : 0.f;
                        v[q] = r;
                    }
                    if (WIN == 4) {
                        const int o = o0 + mbase / 4;
                        if (o < NWC) out[((size_t)b * NWC + o) * C + n] = 0.25f * ((v[0] + v[1]) + (v[2] + v[3]));
                    } else {
                        const int o = o0 + mbase / 2;
                        if (o < NWC) out[((size_t)b * NWC + o) * C + n] = 0.5f * (v[0] + v[1]);
                        if (o + 1 < NWC) out[((size_t)b * NWC + o + 1) * C + n] = 0.5f * (v[2] + v[3]);
                    }
                }
            }
        }
    }
}

template <int C, int PT, int PF>
static int launch_glu_fwd(const float* y, const float* stats, const float* Wg, const float* bg, float* out, int B, int T, int F,
                          uint32_t seed, uint32_t thr24, float dscale, const unsigned* seed_dev, hipStream_t s) {
    using G = GluGeom<C, PT, PF>;
    constexpr int SMEM = (C * G::CP + 128 * G::CP + 3 * C) * 4;
    const int NWC = (T / PT) * (F / PF);
    const int ntiles = B * ((NWC + 128 / G::WIN - 1) / (128 / G::WIN));
    const int per_cu = SMEM > 80 * 1024 ? 1 : (SMEM > 40 * 1024 ? 2 : 4);
    int grid = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
    if (grid < 1) return SED_OK;
    SED_MAX_SMEM((glu_fwd_kernel<C, PT, PF>), SMEM);
    SED_LAUNCH((glu_fwd_kernel<C, PT, PF>), dim3(grid), dim3(256), SMEM, s, y, stats, Wg, bg, out, B, T, F, seed, thr24, dscale, seed_dev);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------
// C = 16 (first block, 2x2 pooling): no LDS, one wave per 16-pixel tile (4 pooling windows along F) on the
// 16x16x4 f32 MFMA.  Lane (i = l&15, g = l>>4) loads ONE float4 = pixel i, channels 4g..4g+3, and that same
// register quad is the MFMA operand with the contraction index ordered (k-step q, lane group g) -> channel 4g+q.
// The GEMM is computed transposed (lin^T = Wg . xn^T) so that the result lands on the lane that already holds
// xn for the same (pixel, channels): the gate is lane-local, the 2x2 pooling is two DPP adds inside a lane quad.
// HBM-bound: reads y once (64 B/pixel), writes 16 B/pixel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void glu16_fwd_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                        const float* __restrict__ Wg, const float* __restrict__ bg,
                                                        float* __restrict__ out, int B, int T, int F, uint32_t seed,
                                                        uint32_t thr24, float dscale, const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;            // per-step entropy in device memory (hipGraph replays)
    constexpr int C = 16;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    float wa[4], sc[4], sh[4], bgr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        wa[q] = Wg[i * C + 4 * g + q];
        sc[q] = stats[2 * C + 4 * g + q];
        sh[q] = stats[3 * C + 4 * g + q];
        bgr[q] = bg[4 * g + q];
    }
    const int To = T / 2, Fo = F / 2, tpr = Fo / 4, ntiles = B * To * tpr;
    const int nwaves = gridDim.x * 4;
    // Work unit of a wave = one ROW of pooling windows (b, to): the index arithmetic (two runtime divisions) is paid once per
    // row of tpr tiles, not per tile -- these kernels are VALU-bound (a register prefetch of the next tile made them slower).
    const int w = i >> 2, q = i & 3;
    const int nrows = B * To;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += nwaves)
    for (int tr = 0; tr < tpr; ++tr) {
        const int b = row / To, to = row - b * To;
        const size_t pix = ((size_t)b * T + 2 * to + (q >> 1)) * F + 2 * (4 * tr + w) + (q & 1);
        const float4 yv = *(const float4*)(y + pix * C + 4 * g);
        float xn[4] = {fmaf(yv.x, sc[0], sh[0]), fmaf(yv.y, sc[1], sh[1]), fmaf(yv.z, sc[2], sh[2]), fmaf(yv.w, sc[3], sh[3])};
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = mfma16(wa[k], xn[k], acc);      // D[n = 4g+r][pixel i]
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float o = (acc[r] + bgr[r]) * sed_fast_sigmoid(xn[r]);
            const uint32_t e = (uint32_t)(pix * C + 4 * g + r);
            o = sed_keep(e, seed, thr24) ? o * dscale : 0.f;
            v[r] = 0.25f * sed_quad_sum(o);
        }
        if (q == 0)
            *(float4*)(out + (((size_t)b * To + to) * Fo + 4 * tr + w) * C + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// backward of the above: 24 MFMA 16x16x4 per 16-pixel tile (GEMM1^T, GEMM2 with e folded in through an identity
// operand, two identity "transposes" into accumulator layout, and dWg accumulation straight from accumulator
// registers: the accumulator layout of X is the A-operand layout of X^T with k = pixel).
__global__ __launch_bounds__(256) void glu16_bwd_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ Wg, const float* __restrict__ bg,
                                                        const float* __restrict__ gout, float* __restrict__ dz,
                                                        float* __restrict__ part, int B, int T, int F,
                                                        uint32_t seed, uint32_t thr24, float dscale, const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;            // per-step entropy in device memory (hipGraph replays)
    constexpr int C = 16, NP = C * C + 3 * C;
    __shared__ float red[4][NP];                // per-wave sums -> one partial per workgroup (see glu32_bwd_kernel)
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    float wa1[4], wb2[4], idb[4], mu[4], istd[4], gam4[4], bet4[4], bgr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = 4 * g + q;
        wa1[q] = Wg[i * C + c];
        wb2[q] = Wg[c * C + i];
        idb[q] = (c == i) ? 1.0f : 0.0f;
        mu[q] = stats[c]; istd[q] = stats[C + c];
        gam4[q] = gamma[c]; bet4[q] = beta[c];
        bgr[q] = bg[c];
    }
    const float gam_i = gamma[i], bet_i = beta[i];
    f32x4 P = {0.f, 0.f, 0.f, 0.f};
    float a_dgam = 0.f, a_dbet = 0.f, a_dbg = 0.f;
    const int To = T / 2, Fo = F / 2, tpr = Fo / 4, ntiles = B * To * tpr;
    const int nwaves = gridDim.x * 4;
    // one row of pooling windows (b, to) per wave iteration, tiles of the row in the inner loop (see glu16_fwd_kernel)
    const int w = i >> 2, q = i & 3;
    const int nrows = B * To;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += nwaves)
    for (int tr = 0; tr < tpr; ++tr) {
        const int b = row / To, to = row - b * To;
        const size_t pix = ((size_t)b * T + 2 * to + (q >> 1)) * F + 2 * (4 * tr + w) + (q & 1);
        const float4 yv = *(const float4*)(y + pix * C + 4 * g);
        const float4 go = *(const float4*)(gout + (((size_t)b * To + to) * Fo + 4 * tr + w) * C + 4 * g);
        float xh[4] = {(yv.x - mu[0]) * istd[0], (yv.y - mu[1]) * istd[1], (yv.z - mu[2]) * istd[2], (yv.w - mu[3]) * istd[3]};
        float xn[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xn[k] = fmaf(xh[k], gam4[k], bet4[k]);
        f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) acc1 = mfma16(wa1[k], xn[k], acc1);   // lin^T: D[n = 4g+r][pixel i]
        const float gv[4] = {go.x, go.y, go.z, go.w};
        float dlin[4], e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lin = acc1[r] + bgr[r];
            const float sg = sed_fast_sigmoid(xn[r]);
            const uint32_t ei = (uint32_t)(pix * C + 4 * g + r);
            const float gr = sed_keep(ei, seed, thr24) ? gv[r] * 0.25f * dscale : 0.f;
            dlin[r] = gr * sg;
            e[r] = gr * lin * sg * (1.0f - sg);
        }
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f}, accx = {0.f, 0.f, 0.f, 0.f}, accd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc2 = mfma16(dlin[k], wb2[k], acc2);       // dxn[pixel 4g+r][c = i] = dlin . Wg
            acc2 = mfma16(e[k], idb[k], acc2);          //                         + e
            accx = mfma16(xh[k], idb[k], accx);         // xhat  in accumulator layout
            accd = mfma16(dlin[k], idb[k], accd);       // dlin  in accumulator layout  [pixel 4g+r][n' = i]
        }
        float xnD[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tp = 2 * to + (r >> 1), fp = 2 * (4 * tr + g) + (r & 1);       // pixel of accumulator row 4g+r
            const float dxn = acc2[r];
            a_dgam += dxn * accx[r];
            a_dbet += dxn;
            a_dbg += accd[r];
            dz[(((size_t)b * T + tp) * F + fp) * C + i] = dxn * gam_i;
            xnD[r] = fmaf(accx[r], gam_i, bet_i);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) P = mfma16(accd[k], xnD[k], P);     // P[n' = 4g+r][c = i] += sum_pixels dlin[p][n'] xn[p][c]
    }
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv][(4 * g + r) * C + i] = P[r];
    a_dgam += __shfl_xor(a_dgam, 16); a_dgam += __shfl_xor(a_dgam, 32);
    a_dbet += __shfl_xor(a_dbet, 16); a_dbet += __shfl_xor(a_dbet, 32);
    a_dbg += __shfl_xor(a_dbg, 16); a_dbg += __shfl_xor(a_dbg, 32);
    if (g == 0) { red[wv][C * C + i] = a_dbg; red[wv][C * C + C + i] = a_dgam; red[wv][C * C + 2 * C + i] = a_dbet; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NP; idx += 256)
        part[(size_t)blockIdx.x * NP + idx] = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
}

// ---------------------------------------------------------------------------------------------
// C = 32 (second block, 2x2 pooling): the glu16 scheme on the 32x32x2 f32 MFMA.  One wave per 32-pixel tile = 8 pooling
// windows along F; lane (lo = 4 w + q, hi) is pixel q of window w and loads the 16 channels S_hi = {8j + 4hi + e} (four
// float4, the two lanes of a pixel interleave 16-byte chunks of its 128-byte row).  The contraction index of MFMA call ks
// is ordered (ks, hi) -> channel S_hi[ks], so register ks of a lane IS its B operand, and the transposed product
// lin^T = Wg . xn^T lands on the lane that holds xn for the same (pixel, channels): S_hi[r] is exactly the accumulator's
// row map (r&3) + 8(r>>2) + 4hi.  Gate, dropout and the 2x2 pooling (two xor-shuffles) are lane-local; no LDS, no barrier.
// The backward mirrors glu16_bwd_kernel: GEMM2 with e folded in through an identity operand, two identity "transposes" into
// accumulator layout (lane = channel) -- through wave-private LDS since round 3, 48 instead of 96 MFMAs per tile --, dWg accumulated
// straight from accumulator registers.
// Measured: forward 52 -> 44 us; the backward runs at the speed of the LDS-tiled generic kernel (214 vs 217 us; neither a
// register prefetch of the next tile nor two waves per SIMD moved it).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int glu32_ch(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// (a waves-per-SIMD hint of 3 makes the allocator settle on 128 registers -- four waves per SIMD instead of three at 148 + 16 -- but it
//  also serialises the four row loads of a tile, each waited for behind its issue (tests/test_isa_audit.py): same box 3.243 vs 3.228 ms
//  per step, inside the noise; not kept.  A register prefetch of the next tile measured the same as well.)
__global__ __launch_bounds__(256) void glu32_fwd_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                        const float* __restrict__ Wg, const float* __restrict__ bg,
                                                        float* __restrict__ out, int B, int T, int F, uint32_t seed,
                                                        uint32_t thr24, float dscale, const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;
    constexpr int C = 32;
    const int lane = threadIdx.x & 63, lo = lane & 31, hi = lane >> 5, w = lo >> 2, q = lo & 3;
    float wa[16], sc[16], sh[16], bgr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = glu32_ch(r, hi);
        wa[r] = Wg[lo * C + c];
        sc[r] = stats[2 * C + c]; sh[r] = stats[3 * C + c];
        bgr[r] = bg[c];
    }
    const int To = T / 2, Fo = F / 2, tpr = Fo / 8, nrows = B * To;
    const int nwaves = gridDim.x * 4;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += nwaves)
    for (int tr = 0; tr < tpr; ++tr) {
        const int b = row / To, to = row - b * To;
        const size_t pix = ((size_t)b * T + 2 * to) * F + 16 * tr + ((q >> 1) ? F : 0) + 2 * w + (q & 1);
        float xn[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 yv = *(const float4*)(y + pix * C + 8 * j + 4 * hi);
            xn[4 * j] = fmaf(yv.x, sc[4 * j], sh[4 * j]); xn[4 * j + 1] = fmaf(yv.y, sc[4 * j + 1], sh[4 * j + 1]);
            xn[4 * j + 2] = fmaf(yv.z, sc[4 * j + 2], sh[4 * j + 2]); xn[4 * j + 3] = fmaf(yv.w, sc[4 * j + 3], sh[4 * j + 3]);
        }
        f32x16 acc = f32x16_zero();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = mfma32(wa[ks], xn[ks], acc);       // D[n = S_hi[r]][pixel lo]
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float o = (acc[r] + bgr[r]) * sed_fast_sigmoid(xn[r]);
            const uint32_t e = (uint32_t)(pix * C + glu32_ch(r, hi));
            o = sed_keep(e, seed, thr24) ? o * dscale : 0.f;
            v[r] = 0.25f * sed_quad_sum(o);
        }
        if (q == 0) {
            float* dst = out + (((size_t)b * To + to) * Fo + 8 * tr + w) * C + 4 * hi;
#pragma unroll
            for (int j = 0; j < 4; ++j) *(float4*)(dst + 8 * j) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
        }
    }
}

__global__ __launch_bounds__(256, 2) void glu32_bwd_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ Wg, const float* __restrict__ bg,
                                                        const float* __restrict__ gout, float* __restrict__ dz,
                                                        float* __restrict__ part, int B, int T, int F,
                                                        uint32_t seed, uint32_t thr24, float dscale, const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;
    constexpr int C = 32, NP = C * C + 3 * C;
    // Three wave-private 32 x 32 transposition buffers per wave (e, xhat, dlin: operand layout -> accumulator layout, round 3; until
    // then 48 of the tile's 96 MFMAs were identity-operand "transposes" on the exact-f32 pipe, which was 0.48 busy).  Row pitch 36
    // floats: 16-byte rows for the float4 reads, lanes of a store run along a row (conflict-free).  The per-wave partial record
    // `red` of the kernel's tail aliases the buffers.
    constexpr int TP = 36, TB = 32 * TP;
    __shared__ __attribute__((aligned(16))) float tbuf[4][3 * TB];
    float (*red)[NP] = (float (*)[NP])&tbuf[0][0];              // [4][NP] floats inside tbuf (NP = 1120 <= 3 TB = 3456)
    static_assert(NP <= 3 * TB, "the partial record must fit its wave's transposition buffers");
    __shared__ float cst[5 * C];                  // mean | invstd | gamma | beta | bg: read per use, 80 VGPRs would not fit
    const int lane = threadIdx.x & 63;
    float* tE = &tbuf[threadIdx.x >> 6][0];
    float* tX = tE + TB;
    float* tD = tE + 2 * TB;
    int lo = lane & 31, hi = lane >> 5, w = lo >> 2, q = lo & 3;
    if (threadIdx.x < C) {
        cst[threadIdx.x] = stats[threadIdx.x]; cst[C + threadIdx.x] = stats[C + threadIdx.x];
        cst[2 * C + threadIdx.x] = gamma[threadIdx.x]; cst[3 * C + threadIdx.x] = beta[threadIdx.x];
        cst[4 * C + threadIdx.x] = bg[threadIdx.x];
    }
    __syncthreads();
    float wa1[16], wb2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = glu32_ch(r, hi);
        wa1[r] = Wg[lo * C + c];                  // A of GEMM1^T: Wg[n = lo][c]
        wb2[r] = Wg[c * C + lo];                  // B of GEMM2:   Wg[n' = c][c_out = lo]
    }
    const float gam_lo = cst[2 * C + lo], bet_lo = cst[3 * C + lo];
    f32x16 P = f32x16_zero();
    float a_dgam = 0.f, a_dbet = 0.f, a_dbg = 0.f;
    const int To = T / 2, Fo = F / 2, tpr = Fo / 8, nrows = B * To;
    const int nwaves = gridDim.x * 4;
    // Software pipeline over the wave's tiles: the conv output AND the upstream gradient of the NEXT tile are fetched while this one
    // is computed.  Before, both groups of loads were issued where they were consumed -- two exposed memory latencies per tile with
    // two waves per SIMD to cover them (and the scheduler hoists the `* 0.25 dscale` of a just-issued load right behind it, so
    // issuing this tile's loads at the top of its own iteration does not help: tools/isa_exposed_loads.py).
    auto y_src = [&](int row_, int tr_) -> const float* {
        const int b_ = row_ / To, to_ = row_ - b_ * To;
        const size_t p_ = ((size_t)b_ * T + 2 * to_) * F + 16 * tr_ + ((q >> 1) ? F : 0) + 2 * w + (q & 1);
        return y + p_ * C + 4 * hi;
    };
    auto g_src = [&](int row_, int tr_) -> const float* { return gout + ((size_t)row_ * Fo + 8 * tr_ + w) * C + 4 * hi; };   // row = b To + to
    float4 yn[4], gn[4];
    {
        const int row_ = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (row_ < nrows) {
            const float* src = y_src(row_, 0);
            const float* gs = g_src(row_, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { yn[j] = *(const float4*)(src + 8 * j); gn[j] = *(const float4*)(gs + 8 * j); }
        }
    }
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += nwaves)
    for (int tr = 0; tr < tpr; ++tr) {
        sed_opaque(lo); sed_opaque(hi); sed_opaque(w); sed_opaque(q);
        const int b = row / To, to = row - b * To;
        const size_t pix0 = ((size_t)b * T + 2 * to) * F + 16 * tr;              // first pixel of the tile's upper row
        const size_t pix = pix0 + ((q >> 1) ? F : 0) + 2 * w + (q & 1);
        float4 gq[4], yc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { yc[j] = yn[j]; gq[j] = gn[j]; }
        {
            int row2 = row, tr2 = tr + 1;
            if (tr2 == tpr) { tr2 = 0; row2 = row + nwaves; }
            if (row2 >= nrows) { row2 = row; tr2 = tr; }                          // last tile: a harmless re-read
            const float* src = y_src(row2, tr2);
            const float* gs = g_src(row2, tr2);
#pragma unroll
            for (int j = 0; j < 4; ++j) { yn[j] = *(const float4*)(src + 8 * j); gn[j] = *(const float4*)(gs + 8 * j); }
        }
        float xh[16], xn[16], dlin[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 yv = yc[j];
            const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 8 * j + 4 * hi + e;
                xh[4 * j + e] = (yy[e] - cst[c]) * cst[C + c];
                xn[4 * j + e] = fmaf(xh[4 * j + e], cst[2 * C + c], cst[3 * C + c]);
            }
        }
        f32x16 acc1 = f32x16_zero();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc1 = mfma32(wa1[ks], xn[ks], acc1);     // lin^T: D[n = S_hi[r]][pixel lo]
        f32x16 acc2 = f32x16_zero();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 gv4 = gq[j];
            const float gv[4] = {gv4.x, gv4.y, gv4.z, gv4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * j + e, c = 8 * j + 4 * hi + e;
                const float lin = acc1[r] + cst[4 * C + c];
                const float sg = (GLU_ABL & 2) ? xn[r] : sed_fast_sigmoid(xn[r]);
                const float gr = ((GLU_ABL & 2) || sed_keep((uint32_t)(pix * C + c), seed, thr24)) ? gv[e] * 0.25f * dscale : 0.f;
                dlin[r] = gr * sg;
                const float ev = gr * lin * sg * (1.0f - sg);
                // operand layout (lane = pixel lo, register = channel c) -> LDS [channel][pixel]
                if (!(GLU_ABL & 16)) { tE[c * TP + lo] = ev; tX[c * TP + lo] = xh[r]; tD[c * TP + lo] = dlin[r]; }
                else acc2[r] += ev;
            }
        }
        f32x16 accx, accd;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc2 = mfma32(dlin[ks], wb2[ks], acc2);   // dxn[pixel S_hi[r]][c = lo] = dlin . Wg
        if (!(GLU_ABL & 16)) {
            sed_wave_sync();
            // accumulator layout: lane = channel lo, register r = tile pixel S_hi[r] = 8 (r >> 2) + 4 hi + (r & 3): four float4 per matrix
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 ev4 = *(const float4*)(tE + lo * TP + 8 * k + 4 * hi);
                const float4 xv4 = *(const float4*)(tX + lo * TP + 8 * k + 4 * hi);
                const float4 dv4 = *(const float4*)(tD + lo * TP + 8 * k + 4 * hi);
                acc2[4 * k] += ev4.x; acc2[4 * k + 1] += ev4.y; acc2[4 * k + 2] += ev4.z; acc2[4 * k + 3] += ev4.w;
                accx[4 * k] = xv4.x; accx[4 * k + 1] = xv4.y; accx[4 * k + 2] = xv4.z; accx[4 * k + 3] = xv4.w;
                accd[4 * k] = dv4.x; accd[4 * k + 1] = dv4.y; accd[4 * k + 2] = dv4.z; accd[4 * k + 3] = dv4.w;
            }
            sed_wave_sync();                                                      // reads done before the next tile's stores
        } else {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) { accx[ks] = xh[ks]; accd[ks] = dlin[ks]; }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // accumulator row r is tile pixel S_hi[r] = 4 wp + qp with qp = r & 3 (compile time) and wp = 2 (r >> 2) + hi
            const size_t pp = pix0 + (((r & 3) >> 1) ? F : 0) + 2 * (2 * (r >> 2) + hi) + (r & 1);
            const float dxn = acc2[r];
            a_dgam = fmaf(dxn, accx[r], a_dgam);
            a_dbet += dxn;
            a_dbg += accd[r];
            if (!(GLU_ABL & 8)) dz[pp * C + lo] = dxn * gam_lo;
            accx[r] = fmaf(accx[r], gam_lo, bet_lo);                              // -> xn in accumulator layout
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) { if (!(GLU_ABL & 32)) P = mfma32(accd[ks], accx[ks], P); else P[ks] += accd[ks] * accx[ks]; }   // P[n' = S_hi[r]][c = lo] += sum_pixels dlin xn
        // the frame that floor-mode time pooling drops (T odd: frame T - 1 of every clip) gets no gradient: the wave that owns the
        // clip's last window row zeroes the 16 pixels x 32 channels below its tile (it was a launch of its own on the backward chain)
        if ((T & 1) && to == To - 1) {
            float4* zr = (float4*)(dz + (((size_t)b * T + (T - 1)) * F + 16 * tr) * C);
            zr[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
            zr[64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // one partial per workgroup (plain stores), summed in a fixed order by glu_bwd_reduce_kernel: 4096 waves x 1024 device-scope
    // float atomics on the same 1 K addresses cost tens of microseconds
    __syncthreads();                              // `red` aliases the transposition buffers of all four waves
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wv][glu32_ch(r, hi) * C + lo] = P[r];
    a_dgam += __shfl_xor(a_dgam, 32); a_dbet += __shfl_xor(a_dbet, 32); a_dbg += __shfl_xor(a_dbg, 32);
    if (hi == 0) { red[wv][C * C + lo] = a_dbg; red[wv][C * C + C + lo] = a_dgam; red[wv][C * C + 2 * C + lo] = a_dbet; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NP; idx += 256)
        part[(size_t)blockIdx.x * NP + idx] = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
}

// ---------------------------------------------------------------------------------------------
// C = 64 / 128 with (1,2) pooling: "weight-stationary in registers".  8 waves; wave (wm, wn) owns N tile wn
// (32 output channels) and keeps its whole B operand -- Wg[n][k] for its 32 n, all k -- in C/2 VGPRs, so LDS
// holds only the activation tile (66 KB at C = 128 -> two workgroups per CU instead of one) and the MFMA loop
// issues one LDS read per MFMA.  The next tile is prefetched into registers under the current tile's MFMAs.
// ---------------------------------------------------------------------------------------------

// acc += sum_k A[k] * bf(k) over K2 k-steps with the A operand read from LDS at ap[2*ks] in software-pipelined groups
// of G (one group in flight under the previous group's MFMAs; the fences keep the unrolled chain from hoisting more).
template <int K2, int G, class BF>
__device__ __forceinline__ void mfma_chain(const float* ap, BF&& bf, f32x16& acc) {
    float a[G], an[G];
#pragma unroll
    for (int i = 0; i < G; ++i) a[i] = ap[2 * i];
#pragma unroll
    for (int ch = 0; ch < K2 / G; ++ch) {
        if (ch + 1 < K2 / G) {
#pragma unroll
            for (int i = 0; i < G; ++i) an[i] = ap[2 * (G * (ch + 1) + i)];
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if (GLU_ABL & 1) acc[0] += a[i] * bf(G * ch + i);
            else acc = mfma32(a[i], bf(G * ch + i), acc);
        }
#pragma unroll
        for (int i = 0; i < G; ++i) a[i] = an[i];
        sed_sched_fence();
    }
}

template <int C>
__global__ __launch_bounds__(512, 4) void glu_wide_fwd_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                           const float* __restrict__ Wg, const float* __restrict__ bg,
                                                           float* __restrict__ out, int B, int T, int F, uint32_t seed,
                                                           uint32_t thr24, float dscale, const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;            // per-step entropy in device memory (hipGraph replays)
    // With (1,2) pooling the pooled pair (f, f+1) is two consecutive pixels, so a tile is simply ROWS consecutive
    // rows of the flattened (B*T*F, C) activation: row r <-> pixel r, window r/2, dropout counter r*C + n.
    constexpr int CP = C + 1, NT = C / 32;
    constexpr int WN = NT, WM = 8 / WN, MS = 1, ROWS = 32 * WM * MS;   // C=128: 4 x 2 waves, 64 rows; C=64: 2 x 4, 128
    constexpr int NLD = ROWS * (C / 4) / 512;
    static_assert(512 % (C / 4) == 0, "each thread keeps one channel quad");
    SED_DYN_SMEM(smem);
    float* xs = (float*)smem;                                          // [ROWS][CP] BN output xn
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int wn = w % WN, wm = w / WN;
    const int R = B * T * F, ntiles = (R + ROWS - 1) / ROWS;
    const int n = wn * 32 + lo;
    // B fragments: Wg goes through LDS so the global reads stay coalesced (a per-lane row gather would cost
    // ~32 cache lines per load instruction, x C/2 instructions x 16 waves on one CU's address unit)
    float breg[C / 2];
    constexpr int WROWS = ROWS < C ? ROWS : C;
#pragma unroll
    for (int p = 0; p < C / WROWS; ++p) {
        __syncthreads();
        for (int i = tid; i < WROWS * (C / 4); i += 512) {
            const int row = i / (C / 4), q = i - row * (C / 4);
            const float4 val = *(const float4*)(Wg + (size_t)(p * WROWS + row) * C + 4 * q);
            float* d = xs + row * CP + 4 * q;
            d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
        }
        __syncthreads();
        if (wn * 32 / WROWS == p) {
#pragma unroll
            for (int ks = 0; ks < C / 2; ++ks) breg[ks] = xs[(n - p * WROWS) * CP + 2 * ks + hi];
        }
    }
    const float bias_n = bg[n];
    const int v = tid % (C / 4), r0 = tid / (C / 4);                   // this thread's channel quad / first row when staging
    float sc4[4], sh4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { sc4[i] = stats[2 * C + 4 * v + i]; sh4[i] = stats[3 * C + 4 * v + i]; }

    float4 ld[NLD];
    auto load_tile = [&](int tile) {
        const int row0 = tile * ROWS;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int m = r0 + (512 / (C / 4)) * u;
            ld[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + m < R && !(GLU_ABL & 4)) {
                float4 val = *(const float4*)(y + (size_t)(row0 + m) * C + 4 * v);
                val.x = fmaf(val.x, sc4[0], sh4[0]); val.y = fmaf(val.y, sc4[1], sh4[1]);
                val.z = fmaf(val.z, sc4[2], sh4[2]); val.w = fmaf(val.w, sc4[3], sh4[3]);
                ld[u] = val;
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            float* d = xs + (r0 + (512 / (C / 4)) * u) * CP + 4 * v;
            d[0] = ld[u].x; d[1] = ld[u].y; d[2] = ld[u].z; d[3] = ld[u].w;
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * ROWS;
        __syncthreads();                                               // previous tile fully consumed
        store_tile();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);    // in flight under the MFMAs below
        f32x16 acc[MS];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) acc[ms] = f32x16_zero();
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            mfma_chain<C / 2, (C > 64 ? 4 : 8)>(xs + ((wm * MS + ms) * 32 + lo) * CP + hi, [&](int ks) { return breg[ks]; }, acc[ms]);
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int mbase = (wm * MS + ms) * 32 + 8 * j + 4 * hi;
                float vv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (GLU_ABL & 2) { vv[q] = acc[ms][4 * j + q]; continue; }
                    const uint32_t e = (uint32_t)(row0 + mbase + q) * (uint32_t)C + (uint32_t)n;
                    const float xn = xs[(mbase + q) * CP + n];
                    float r = (acc[ms][4 * j + q] + bias_n) * sed_fast_sigmoid(xn);
                    vv[q] = sed_keep(e, seed, thr24) ? r * dscale : 0.f;
                }
                if (GLU_ABL & 8) { if (vv[0] + vv[1] + vv[2] + vv[3] == 123.456f) out[0] = 1.f; continue; }
                const int o = (row0 + mbase) / 2;                      // rows past R hold zeros and are never stored
                if (2 * o < R) out[(size_t)o * C + n] = 0.5f * (vv[0] + vv[1]);
                if (2 * o + 2 < R) out[(size_t)(o + 1) * C + n] = 0.5f * (vv[2] + vv[3]);
            }
        }
    }
}
// ---------------------------------------------------------------------------------------------
// Split-bf16 variant of the weight-stationary forward (conv_precision = "bf16x3"): the C x C linear of the gate runs as
// three v_mfma_f32_32x32x16_bf16 per product on operands split x = hi + lo (see sed_common.h), 3/16 of the f32-MFMA
// cost, fp32-level accuracy.  The activation tile sits in LDS as bf16 hi / lo planes [row][C + 8] (16-byte fragment
// reads, conflict-free per quarter wave), the wave's Wg fragments (hi and lo) stay in C/2 VGPRs as before, and the
// epilogue rebuilds xn = hi + lo (exact to 2^-17 relative) for the sigmoid gate.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_pair_sum(unsigned short h, unsigned short l) { return bf16_to_f32(h) + bf16_to_f32(l); }

template <int C>
__global__ __launch_bounds__(512, 4) void glu_wide_fwd_b_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                                const float* __restrict__ Wg, const float* __restrict__ bg,
                                                                float* __restrict__ out, int B, int T, int F, uint32_t seed,
                                                                uint32_t thr24, float dscale, const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;
    constexpr int RS = C + 8, NT = C / 32, WN = NT, WM = 8 / WN, ROWS = 32 * WM, KS = C / 16;
    constexpr int NLD = ROWS * (C / 4) / 512;
    static_assert(512 % (C / 4) == 0 && ROWS >= 64, "staging layout");
    SED_DYN_SMEM(smem);
    unsigned short* xh = (unsigned short*)smem;                        // [ROWS][RS] hi plane of xn
    unsigned short* xl = xh + ROWS * RS;                               // lo plane
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int lo = lane & 31, hi = lane >> 5;
    const int wn = w % WN, wm = w / WN;
    const int R = B * T * F, ntiles = (R + ROWS - 1) / ROWS;
    int n = wn * 32 + lo;
    int v = tid % (C / 4), r0 = tid / (C / 4);                         // this thread's channel quad / first row when staging
    constexpr int RSTEP = 512 / (C / 4);
    // ---- B fragments: Wg rows through LDS (coalesced global reads), split into bf16 planes, 64 rows per pass ----
    s16x8 bh[KS], bl[KS];
#pragma unroll
    for (int p = 0; p < C / 64; ++p) {
        __syncthreads();
        for (int i = tid; i < 64 * (C / 4); i += 512) {
            const int row = i / (C / 4), q = i - row * (C / 4);
            const float4 val = *(const float4*)(Wg + (size_t)(p * 64 + row) * C + 4 * q);
            uint2 hv, lv;
            bf16_split2(val.x, val.y, hv.x, lv.x);
            bf16_split2(val.z, val.w, hv.y, lv.y);
            *(uint2*)(xh + row * RS + 4 * q) = hv;
            *(uint2*)(xl + row * RS + 4 * q) = lv;
        }
        __syncthreads();
        if (wn * 32 / 64 == p) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bh[ks] = *(const s16x8*)(xh + (n - p * 64) * RS + 16 * ks + 8 * hi);
                bl[ks] = *(const s16x8*)(xl + (n - p * 64) * RS + 16 * ks + 8 * hi);
            }
        }
    }
    const float bias_n = bg[n];
    float sc4[4], sh4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { sc4[i] = stats[2 * C + 4 * v + i]; sh4[i] = stats[3 * C + 4 * v + i]; }

    // Register prefetch of the next tile.  The loads are unconditional (rows past R read row R - 1) and the raw values stay in
    // flight: BatchNorm and the zeroing of the rows past R happen when the tile is parked in LDS.  (With `if (row < R) { load; fma }`
    // the compiler waited for every load inside its branch -- s_waitcnt vmcnt(0) right behind each of them, so nothing was ever in
    // flight under the MFMAs: tools/isa_exposed_loads.py.)
    float4 ld[NLD];
    auto load_tile = [&](int tile) {
        const int row0 = tile * ROWS;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int row = row0 + r0 + RSTEP * u;
            ld[u] = *(const float4*)(y + (size_t)(row < R ? row : R - 1) * C + 4 * v);
        }
    };
    auto store_tile = [&](int row0) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int m = r0 + RSTEP * u;
            const bool in = row0 + m < R;
            ld[u].x = in ? fmaf(ld[u].x, sc4[0], sh4[0]) : 0.f; ld[u].y = in ? fmaf(ld[u].y, sc4[1], sh4[1]) : 0.f;
            ld[u].z = in ? fmaf(ld[u].z, sc4[2], sh4[2]) : 0.f; ld[u].w = in ? fmaf(ld[u].w, sc4[3], sh4[3]) : 0.f;
            uint2 hv, lv;
            bf16_split2(ld[u].x, ld[u].y, hv.x, lv.x);
            bf16_split2(ld[u].z, ld[u].w, hv.y, lv.y);
            *(uint2*)(xh + m * RS + 4 * v) = hv;
            *(uint2*)(xl + m * RS + 4 * v) = lv;
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * ROWS;
        sed_opaque(lo); sed_opaque(hi); sed_opaque(n); sed_opaque(v); sed_opaque(r0);   // per-tile addresses: recomputed, not spilled
        __syncthreads();                                               // previous tile fully consumed
        store_tile(row0);
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);    // in flight under the MFMAs below
        // (two accumulator chains -- cross terms | hi x hi -- were measured: 43.8 vs 34.5 us at F = 16, the 128-VGPR budget spills more)
        f32x16 acc = f32x16_zero();
        const unsigned short* ap = xh + (wm * 32 + lo) * RS + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const s16x8 a_hi = *(const s16x8*)(ap + 16 * ks);
            const s16x8 a_lo = *(const s16x8*)(ap + ROWS * RS + 16 * ks);
            acc = mfma32_bf16(a_lo, bh[ks], acc);
            acc = mfma32_bf16(a_hi, bl[ks], acc);
            acc = mfma32_bf16(a_hi, bh[ks], acc);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mbase = wm * 32 + 8 * j + 4 * hi;
            float vv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t e = (uint32_t)(row0 + mbase + q) * (uint32_t)C + (uint32_t)n;
                const float xn = bf16_pair_sum(xh[(mbase + q) * RS + n], xl[(mbase + q) * RS + n]);
                const float r = (acc[4 * j + q] + bias_n) * sed_fast_sigmoid(xn);
                vv[q] = sed_keep(e, seed, thr24) ? r * dscale : 0.f;
            }
            const int o = (row0 + mbase) / 2;                          // rows past R hold zeros and are never stored
            if (2 * o < R) out[(size_t)o * C + n] = 0.5f * (vv[0] + vv[1]);
            if (2 * o + 2 < R) out[(size_t)(o + 1) * C + n] = 0.5f * (vv[2] + vv[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// C = 128 forward on v_mfma_f32_16x16x32_bf16 (same tiling as glu128_bwd_c_kernel below): a wave owns 16 output columns x all 64
// rows of the tile, Wg fragments = 32 VGPRs (64 with the 32x32 tiling, which does not fit the 128-VGPR budget of two resident
// workgroups without scratch), four independent accumulator chains, unpadded swizzled planes, per-tile address recomputation.
//
// DB (round 4; sed_set_tuning glu_fwd128 = 3, NOT the default -- measured neutral, see the end of this comment): the activation
// planes are DOUBLE-BUFFERED in LDS (2 x 32 KB).  The single-buffered kernel serialised every
// tile into [park the tile | barrier | MFMAs + gate / dropout / pooling epilogue | barrier]: its own timing ablation said loads 7 us,
// parking 7, epilogue 9, MFMAs + stores 12 of the 35 us at F = 16, none of them overlapping inside a workgroup (r03: 2.0 - 2.6 TB/s,
// SQ_WAIT_ANY 0.45 - 0.52).  Here tile t + 1 is parked into the other buffer BEFORE the MFMAs of tile t -- its ds_writes drain while the
// matrix pipe works -- and one barrier per tile is left (everybody done with buffer t, buffer t + 1 complete).  Same arithmetic, same
// operation order per element: bit-identical output.  Measured (same box, alternating, tools/ab_tuning.sh): step 3.232 / 3.212 ms with it,
// 3.223 / 3.225 without; per launch inside the replayed step (rocprofv3) 22.99 vs 22.29 us on average over the four shapes: the barrier
// and the parking were NOT what the kernel waits for -- with two resident workgroups per CU one's parking already overlaps the other's
// MFMAs, and the 25 issue slots per element of the gate (exp, rcp, dropout hash) put the kernel's VALU floor at ~ 6 TB/s-equivalent
// (DESIGN.md section 11).  Kept selectable so that the number can be reproduced.
// ---------------------------------------------------------------------------------------------
template <bool DB>
__global__ __launch_bounds__(512, 4) void glu128_fwd_c_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                              const float* __restrict__ Wg, const float* __restrict__ bg,
                                                              float* __restrict__ out, int B, int T, int F, uint32_t seed,
                                                              uint32_t thr24, float dscale, const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;
    constexpr int C = 128, ROWS = 64, RS = C, WS = C + 8, KS = C / 32, RB = ROWS / 16;
    constexpr int BUF = 2 * ROWS * RS;               // one buffer = hi plane + lo plane (ushorts)
    SED_DYN_SMEM(smem);
    unsigned short* xbase = (unsigned short*)smem;   // xn [ROWS][RS] hi | lo, octet o of row m at slot o ^ (m & 15); DB: two such buffers
    unsigned short* wh = xbase;                      // Wg staging (before the tile loop): plain [64][WS] hi | lo
    unsigned short* wl = xbase + 64 * WS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int i16 = lane & 15, g = lane >> 4;
    int n = 16 * w + i16;
    const int R = B * T * F, ntiles = (R + ROWS - 1) / ROWS;
    auto rm_off = [](int m, int k) { return m * RS + ((((k >> 3) ^ (m & 15))) << 3) + (k & 7); };
    int cq = tid % (C / 4), rq = tid / (C / 4);      // this thread's 4x4 staging block: channels 4cq.., rows 4rq..

    float4 ld0, ld1, ld2, ld3;
    auto load_tile = [&](int tile) {
        const int row = tile * ROWS + 4 * rq;
        if ((tile + 1) * ROWS <= R && !(GLU_ABL & 4)) {
            const float* src = y + (size_t)row * C + 4 * cq;
            ld0 = *(const float4*)src; ld1 = *(const float4*)(src + C); ld2 = *(const float4*)(src + 2 * C); ld3 = *(const float4*)(src + 3 * C);
        } else {
            auto lr = [&](int r) -> float4 {
                float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < R && !(GLU_ABL & 4)) val = *(const float4*)(y + (size_t)r * C + 4 * cq);
                return val;
            };
            ld0 = lr(row); ld1 = lr(row + 1); ld2 = lr(row + 2); ld3 = lr(row + 3);
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);              // in flight under the fragment set-up

    // ---- Wg fragments: B[k = c][j = n] = Wg[n][c]; lane (i16, g) holds c = 32 ks + 8 g + e of row n ----
    s16x8 bh[KS], bl[KS];
#pragma unroll
    for (int p = 0; p < C / 64; ++p) {
        __syncthreads();
        for (int i = tid; i < 64 * (C / 4); i += 512) {
            const int row = i / (C / 4), q = i - row * (C / 4);
            const float4 val = *(const float4*)(Wg + (size_t)(p * 64 + row) * C + 4 * q);
            uint2 hv, lv;
            bf16_split2(val.x, val.y, hv.x, lv.x);
            bf16_split2(val.z, val.w, hv.y, lv.y);
            *(uint2*)(wh + row * WS + 4 * q) = hv;
            *(uint2*)(wl + row * WS + 4 * q) = lv;
        }
        __syncthreads();
        if (w / 4 == p) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bh[ks] = *(const s16x8*)(wh + (n - p * 64) * WS + 32 * ks + 8 * g);
                bl[ks] = *(const s16x8*)(wl + (n - p * 64) * WS + 32 * ks + 8 * g);
            }
        }
    }
    const float bias_n = bg[n];
    auto store_rm = [&](unsigned short* xh, float4 val, int m, bool live, const float4& sc, const float4& sh) {
        const float k = live ? 1.0f : 0.0f;                             // rows past the end: xn = 0
        val.x = fmaf(val.x, sc.x, sh.x) * k; val.y = fmaf(val.y, sc.y, sh.y) * k;
        val.z = fmaf(val.z, sc.z, sh.z) * k; val.w = fmaf(val.w, sc.w, sh.w) * k;
        uint2 hv, lv;
        bf16_split2(val.x, val.y, hv.x, lv.x);
        bf16_split2(val.z, val.w, hv.y, lv.y);
        *(uint2*)(xh + rm_off(m, 4 * cq)) = hv;
        *(uint2*)(xh + ROWS * RS + rm_off(m, 4 * cq)) = lv;
    };
    auto park = [&](unsigned short* xh, int row0) {     // BatchNorm + bf16 split of the tile held in ld0..3 -> planes of buffer xh
        const float4 sc = *(const float4*)(stats + 2 * C + 4 * cq), sh = *(const float4*)(stats + 3 * C + 4 * cq);
        const int row = row0 + 4 * rq;
        store_rm(xh, ld0, 4 * rq, row < R, sc, sh); store_rm(xh, ld1, 4 * rq + 1, row + 1 < R, sc, sh);
        store_rm(xh, ld2, 4 * rq + 2, row + 2 < R, sc, sh); store_rm(xh, ld3, 4 * rq + 3, row + 3 < R, sc, sh);
    };
    int buf = 0;
    if (DB) {
        __syncthreads();                                                // Wg staging consumed: the planes may be written
        if (tile < ntiles) {
            park(xbase, tile * ROWS);
            if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
        }
        __syncthreads();
    }
    for (; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * ROWS;
        sed_opaque(i16); sed_opaque(g); sed_opaque(n); sed_opaque(cq); sed_opaque(rq);   // addresses recomputed per tile, not spilled
        unsigned short* xh = xbase + (DB ? buf * BUF : 0);
        unsigned short* xl = xh + ROWS * RS;
        if (DB) {
            // tile t + 1 (raw values loaded one iteration ago) goes into the OTHER buffer now; nobody reads that buffer before the
            // barrier at the end of this iteration, and its last readers passed the barrier at the end of the previous one
            const int tn = tile + (int)gridDim.x;
            if (tn < ntiles) {
                park(xbase + (buf ^ 1) * BUF, tn * ROWS);
                if (tn + (int)gridDim.x < ntiles) load_tile(tn + gridDim.x);   // in flight under the MFMAs and the epilogue below
            }
        } else {
            __syncthreads();                                            // previous tile fully consumed (first pass: Wg staging)
            park(xh, row0);
            __syncthreads();
            if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);     // in flight under the MFMAs below
        }
        f32x4 acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(GLU_ABL & 1)) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s16x8 ah[RB], al[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const unsigned short* ap = xh + rm_off(16 * rb + i16, 32 * ks + 8 * g);
                    ah[rb] = *(const s16x8*)ap;
                    al[rb] = *(const s16x8*)(ap + ROWS * RS);
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma16_bf16(al[rb], bh[ks], acc[rb]);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma16_bf16(ah[rb], bl[ks], acc[rb]);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma16_bf16(ah[rb], bh[ks], acc[rb]);
                sed_sched_fence();
            }
        }
        // ---- epilogue: gate, dropout, (1,2) pooling over the lane's row pairs ----
        const bool full = row0 + ROWS <= R;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            float vv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * rb + 4 * g + r;
                const uint32_t e = (uint32_t)(row0 + m) * (uint32_t)C + (uint32_t)n;
                const float xn = bf16_pair_sum(xh[rm_off(m, n)], xl[rm_off(m, n)]);
                const float v = (acc[rb][r] + bias_n) * ((GLU_ABL & 2) ? xn : sed_fast_sigmoid(xn));
                vv[r] = ((GLU_ABL & 2) || sed_keep(e, seed, thr24)) ? v * dscale : 0.f;
            }
            const int o = (row0 + 16 * rb + 4 * g) / 2;                  // R is even: a pooling pair is inside or outside together
            if (!(GLU_ABL & 8)) {
                if (full || 2 * o < R) out[(size_t)o * C + n] = 0.5f * (vv[0] + vv[1]);
                if (full || 2 * o + 2 < R) out[(size_t)(o + 1) * C + n] = 0.5f * (vv[2] + vv[3]);
            }
        }
        if (DB) {
            __syncthreads();                                            // buffer `buf` is free again, buffer `buf ^ 1` is complete
            buf ^= 1;
        }
    }
}

// persistent-grid cap; sed_set_tuning(SED_TUNE_GLU_GRID_CAP, n) (tests) forces several tiles per workgroup on small problems
// zero the gradient rows of the frames that floor-mode time pooling drops: dz (B, T, F*C), rows [t0, T) of every clip
__global__ __launch_bounds__(256) void glu_zero_tail_kernel(float* __restrict__ dz, long long n4, int T, int t0, int tail4,
                                                            long long clip4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const long long b = i / tail4, r = i - b * tail4;
        ((float4*)dz)[b * clip4 + (long long)t0 * (clip4 / T) + r] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
static inline int glu_grid_cap(int dflt) {
    const int e = sed_tuning[SED_TUNE_GLU_GRID_CAP];
    return e > 0 ? e : dflt;
}
template <int C, bool SPLIT>
static int launch_glu_wide_fwd(const float* y, const float* stats, const float* Wg, const float* bg, float* out, int B, int T, int F,
                               uint32_t seed, uint32_t thr24, float dscale, const unsigned* seed_dev, hipStream_t s) {
    constexpr int ROWS = 32 * (8 / (C / 32));
    constexpr int SMEM = SPLIT ? 2 * ROWS * (C + 8) * 2 : ROWS * (C + 1) * 4;
    const int ntiles = (B * T * F + ROWS - 1) / ROWS;
    const int cap = glu_grid_cap(512);                                  // two 8-wave workgroups per CU
    int grid = ntiles < cap ? ntiles : cap;
    if (grid < 1) return SED_OK;
    if (SPLIT && C == 128 && sed_tuning[SED_TUNE_GLU_FWD128] != 1) {      // 1 = the 32x32x16 tiling, 3 = double-buffered planes (A/B runs)
        if (sed_tuning[SED_TUNE_GLU_FWD128] != 3) {
            constexpr int SMEM_C = 2 * 64 * (128 + 8) * 2;                // Wg staging [64][C + 8] hi | lo; the tile planes need 32 KB
            SED_MAX_SMEM(glu128_fwd_c_kernel<false>, SMEM_C);
            SED_LAUNCH(glu128_fwd_c_kernel<false>, dim3(grid), dim3(512), SMEM_C, s, y, stats, Wg, bg, out, B, T, F, seed, thr24, dscale, seed_dev);
        } else {
            constexpr int SMEM_D = 2 * 2 * 64 * 128 * 2;                  // two buffers of hi | lo planes (the Wg staging fits inside)
            SED_MAX_SMEM(glu128_fwd_c_kernel<true>, SMEM_D);
            SED_LAUNCH(glu128_fwd_c_kernel<true>, dim3(grid), dim3(512), SMEM_D, s, y, stats, Wg, bg, out, B, T, F, seed, thr24, dscale, seed_dev);
        }
        return sed_check_launch();
    }
    if (SPLIT) {
        SED_MAX_SMEM((glu_wide_fwd_b_kernel<C>), SMEM);
        SED_LAUNCH((glu_wide_fwd_b_kernel<C>), dim3(grid), dim3(512), SMEM, s, y, stats, Wg, bg, out, B, T, F, seed, thr24, dscale, seed_dev);
    } else {
        SED_MAX_SMEM((glu_wide_fwd_kernel<C>), SMEM);
        SED_LAUNCH((glu_wide_fwd_kernel<C>), dim3(grid), dim3(512), SMEM, s, y, stats, Wg, bg, out, B, T, F, seed, thr24, dscale, seed_dev);
    }
    return sed_check_launch();
}

// y (B,T,F,C); stats 4*C (mean, invstd, scale, shift); Wg (C,C) [out][in]; out (B,T/PT,F/PF,C).
// dropout: keep element e when hash(e, seed) >> 8 >= thr24 (thr24 = round(p * 2^24)); dscale = 1/(1-p).
SED_API int sed_glu_fwd(const float* y, const float* stats, const float* Wg, const float* bg, float* out, int B, int T, int F,
                           int C, int PT, int PF, unsigned seed, unsigned thr24, float dscale,
                           const unsigned* seed_dev, int split_bf16, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (F % PF != 0) return SED_ERR_UNSUPPORTED;
    if (C == 16 && PT == 2 && PF == 2 && F % 8 == 0) {
        const int ntiles = B * (T / 2) * (F / 8);
        if (ntiles <= 0) return SED_OK;
        int grid = (ntiles + 3) / 4;
        if (grid > 2048) grid = 2048;
        SED_LAUNCH(glu16_fwd_kernel, dim3(grid), dim3(256), 0, s, y, stats, Wg, bg, out, B, T, F, seed, thr24, dscale, seed_dev);
        return sed_check_launch();
    }
    if (C == 32 && PT == 2 && PF == 2 && F % 16 == 0) {
        const int nrows = B * (T / 2);
        if (nrows <= 0) return SED_OK;
        int grid = (nrows + 3) / 4;
        if (grid > 2048) grid = 2048;
        SED_LAUNCH(glu32_fwd_kernel, dim3(grid), dim3(256), 0, s, y, stats, Wg, bg, out, B, T, F, seed, thr24, dscale, seed_dev);
        return sed_check_launch();
    }
    if (PT == 1 && PF == 2) {
        if (C == 128 && split_bf16) return launch_glu_wide_fwd<128, true>(y, stats, Wg, bg, out, B, T, F, seed, thr24, dscale, seed_dev, s);
        if (C == 64 && split_bf16) return launch_glu_wide_fwd<64, true>(y, stats, Wg, bg, out, B, T, F, seed, thr24, dscale, seed_dev, s);
        if (C == 128) return launch_glu_wide_fwd<128, false>(y, stats, Wg, bg, out, B, T, F, seed, thr24, dscale, seed_dev, s);
        if (C == 64) return launch_glu_wide_fwd<64, false>(y, stats, Wg, bg, out, B, T, F, seed, thr24, dscale, seed_dev, s);
    }
#define GLU_CASE(c, pt, pf) \
    if (C == c && PT == pt && PF == pf) return launch_glu_fwd<c, pt, pf>(y, stats, Wg, bg, out, B, T, F, seed, thr24, dscale, seed_dev, s);
    GLU_CASE(16, 2, 2) GLU_CASE(32, 2, 2)
#undef GLU_CASE
    return SED_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
template <int C, int PT, int PF>
__global__ __launch_bounds__(256) void glu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ Wg, const float* __restrict__ bg,
                                                      const float* __restrict__ gout, float* __restrict__ dz,
                                                      float* __restrict__ dWg, float* __restrict__ dbg,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int T, int F,
                                                      uint32_t seed, uint32_t thr24, float dscale, const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;            // per-step entropy in device memory (hipGraph replays)
    using G = GluGeom<C, PT, PF>;
    constexpr int WIN = G::WIN, CP = G::CP, NT = G::NT;
    constexpr int ROWS = C >= 64 ? 64 : 128, WM = ROWS / 32, WN = 4 / WM, NTW = NT / WN, NW = ROWS / WIN;
    constexpr int NTILES3 = NT * NT, TPW = NTILES3 >= 4 ? NTILES3 / 4 : 1;
    SED_DYN_SMEM(smem);
    float* wg = (float*)smem;           // [C][CP]
    float* xh = wg + C * CP;            // [ROWS][CP]  xhat
    float* dl = xh + ROWS * CP;         // [ROWS][CP]  d lin
    float* sc = dl + ROWS * CP;         // mean[C], invstd[C], gamma[C], beta[C], bg[C]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int wm = w % WM, wn = w / WM;
    const int To = T / PT, Fo = F / PF, NWC = To * Fo;
    const int fsh = 31 - __builtin_clz(Fo);
    const int tiles_per_clip = (NWC + NW - 1) / NW, ntiles = B * tiles_per_clip;

    for (int i = tid; i < C * C; i += 256) wg[(i / C) * CP + (i % C)] = Wg[i];
    if (tid < C) {
        sc[tid] = stats[tid]; sc[C + tid] = stats[C + tid];
        sc[2 * C + tid] = gamma[tid]; sc[3 * C + tid] = beta[tid]; sc[4 * C + tid] = bg[tid];
    }
    const float* gam = sc + 2 * C;
    const float* bet = sc + 3 * C;

    f32x16 P[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) P[i] = f32x16_zero();
    float a_dbg[NTW], a_dgam[NTW], a_dbet[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) { a_dbg[i] = 0.f; a_dgam[i] = 0.f; a_dbet[i] = 0.f; }

    // the next tile's rows are fetched into registers under the current tile's three GEMMs (see glu_fwd_kernel)
    constexpr int NLD = ROWS * (C / 4) / 256;
    float4 ld[NLD];
    auto load_tile = [&](int tile_) {
        const int b_ = tile_ / tiles_per_clip, o0_ = (tile_ - b_ * tiles_per_clip) * NW;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = tid + 256 * u, m = idx / (C / 4), v = idx - m * (C / 4);
            int o, t, f;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_pixel<PT, PF>(m, o0_, NWC, fsh, o, t, f) && !(GLU_ABL & 4)) val = *(const float4*)(y + (((size_t)b_ * T + t) * F + f) * C + 4 * v);
            ld[u] = val;
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_clip, o0 = (tile - b * tiles_per_clip) * NW;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = tid + 256 * u, m = idx / (C / 4), v = idx - m * (C / 4);
            int o, t, f;
            float4 val = ld[u];
            if (row_pixel<PT, PF>(m, o0, NWC, fsh, o, t, f)) {
                const float* mu = sc + 4 * v;
                val.x = (val.x - mu[0]) * mu[C + 0]; val.y = (val.y - mu[1]) * mu[C + 1];
                val.z = (val.z - mu[2]) * mu[C + 2]; val.w = (val.w - mu[3]) * mu[C + 3];
            }
            float* d = xh + m * CP + 4 * v;
            d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
        // ---- GEMM1: lin = xn . Wg^T ----
        f32x16 acc[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[nt] = f32x16_zero();
        {
            const float* ap = xh + (32 * wm + lo) * CP + hi;
#pragma unroll 8
            for (int k = 0; k < C; k += 2) {
                const float av = fmaf(ap[k], gam[k + hi], bet[k + hi]);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int n = (wn * NTW + nt) * 32 + lo;
                    const float bv = (C >= 32 || lo < C) ? wg[n * CP + k + hi] : 0.f;
                    acc[nt] = (GLU_ABL & 1) ? acc[nt] + av * bv : mfma32(av, bv, acc[nt]);
                }
            }
        }
        // ---- epilogue 1: dlin -> LDS, e -> acc (seed of GEMM2); row -> pixel arithmetic once per row ----
        {
            bool nokv[NTW];
            float biasv[NTW], gnv[NTW], bnv[NTW];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int n = (wn * NTW + nt) * 32 + lo;
                nokv[nt] = n < C;
                biasv[nt] = nokv[nt] ? sc[4 * C + n] : 0.f;
                gnv[nt] = nokv[nt] ? gam[n] : 0.f;
                bnv[nt] = nokv[nt] ? bet[n] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 32 * wm + mfma32_row(r, lane);
                int o, t, f;
                const bool rok = row_pixel<PT, PF>(m, o0, NWC, fsh, o, t, f);
                const uint32_t ebase = (uint32_t)((((size_t)b * T + t) * F + f) * C);
                const size_t gbase = ((size_t)b * NWC + o) * C;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int n = (wn * NTW + nt) * 32 + lo;
                    float dlin = 0.f, e = 0.f;
                    if (GLU_ABL & 2) { dlin = acc[nt][r]; e = dlin; }
                    else if (rok && nokv[nt]) {
                        const float xn = fmaf(xh[m * CP + n], gnv[nt], bnv[nt]);
                        const float sg = sed_fast_sigmoid(xn);
                        const float lin = acc[nt][r] + biasv[nt];
                        float g = gout[gbase + n] * (1.0f / WIN);
                        g = sed_keep(ebase + n, seed, thr24) ? g * dscale : 0.f;
                        dlin = g * sg;
                        e = g * lin * sg * (1.0f - sg);
                    }
                    if (nokv[nt]) dl[m * CP + n] = dlin;
                    acc[nt][r] = e;
                    a_dbg[nt] += dlin;
                }
            }
        }
        __syncthreads();
        // ---- GEMM2: dxn = dlin . Wg + e ----
        {
            const float* ap = dl + (32 * wm + lo) * CP + hi;
#pragma unroll 8
            for (int k = 0; k < C; k += 2) {
                const float av = ap[k];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int c = (wn * NTW + nt) * 32 + lo;
                    const float bv = (C >= 32 || lo < C) ? wg[(k + hi) * CP + c] : 0.f;
                    acc[nt] = (GLU_ABL & 1) ? acc[nt] + av * bv : mfma32(av, bv, acc[nt]);
                }
            }
        }
        // ---- epilogue 2: dz = dxn * gamma, BN reductions ----
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 32 * wm + mfma32_row(r, lane);
            int o, t, f;
            if (row_pixel<PT, PF>(m, o0, NWC, fsh, o, t, f)) {
                float* dzr = dz + (((size_t)b * T + t) * F + f) * C;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int c = (wn * NTW + nt) * 32 + lo;
                    if (c < C) {
                        const float dxn = acc[nt][r];
                        a_dgam[nt] += dxn * xh[m * CP + c];
                        a_dbet[nt] += dxn;
                        if (!(GLU_ABL & 8)) dzr[c] = dxn * gam[c];
                    }
                }
            }
        }
        // ---- GEMM3: P[n'][c] += sum_rows dlin[row][n'] * xn[row][c] ----
        if (NTILES3 >= 4) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int id = w * TPW + i, mt = id / NT, ct = id % NT;
                const int c = ct * 32 + lo;
                const float gc = gam[c], bc = bet[c];
#pragma unroll 8
                for (int k = 0; k < ROWS; k += 2) {
                    const float av = dl[(k + hi) * CP + mt * 32 + lo];
                    const float bv = fmaf(xh[(k + hi) * CP + c], gc, bc);
                    P[i] = (GLU_ABL & 1) ? P[i] + av * bv : mfma32(av, bv, P[i]);
                }
            }
        } else {   // single 32x32 output tile: the 4 waves split K (rows)
            const bool cok = lo < C;
            const float gc = cok ? gam[lo] : 0.f, bc = cok ? bet[lo] : 0.f;
#pragma unroll 8
            for (int k = 32 * w; k < 32 * w + 32; k += 2) {
                const float av = cok ? dl[(k + hi) * CP + lo] : 0.f;
                const float bv = cok ? fmaf(xh[(k + hi) * CP + lo], gc, bc) : 0.f;
                P[0] = (GLU_ABL & 1) ? P[0] + av * bv : mfma32(av, bv, P[0]);
            }
        }
    }
    // ---- flush the per-workgroup reductions ----
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int id = NTILES3 >= 4 ? w * TPW + i : 0;
        const int mt = id / NT, ct = id % NT, c = ct * 32 + lo;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = mt * 32 + mfma32_row(r, lane);
            if (n < C && c < C) atomicAdd(dWg + (size_t)n * C + c, P[i][r]);
        }
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int c = (wn * NTW + nt) * 32 + lo;
        float v0 = a_dbg[nt], v1 = a_dgam[nt], v2 = a_dbet[nt];
        v0 += __shfl_xor(v0, 32); v1 += __shfl_xor(v1, 32); v2 += __shfl_xor(v2, 32);
        if (hi == 0 && c < C) { atomicAdd(dbg + c, v0); atomicAdd(dgamma + c, v1); atomicAdd(dbeta + c, v2); }
    }
}

template <int C, int PT, int PF>
static int launch_glu_bwd(const float* y, const float* stats, const float* gamma, const float* beta, const float* Wg,
                          const float* bg, const float* gout, float* dz, float* dWg, float* dbg, float* dgamma, float* dbeta,
                          int B, int T, int F, uint32_t seed, uint32_t thr24, float dscale, const unsigned* seed_dev, hipStream_t s) {
    using G = GluGeom<C, PT, PF>;
    constexpr int ROWS = C >= 64 ? 64 : 128;
    constexpr int SMEM = (C * G::CP + 2 * ROWS * G::CP + 5 * C) * 4;
    const int NWC = (T / PT) * (F / PF);
    const int NW = ROWS / G::WIN;
    const int ntiles = B * ((NWC + NW - 1) / NW);
    const int per_cu = SMEM > 80 * 1024 ? 1 : (SMEM > 40 * 1024 ? 2 : 4);
    int grid = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
    if (grid < 1) return SED_OK;
    SED_MAX_SMEM((glu_bwd_kernel<C, PT, PF>), SMEM);
    SED_LAUNCH((glu_bwd_kernel<C, PT, PF>), dim3(grid), dim3(256), SMEM, s, y, stats, gamma, beta, Wg, bg, gout, dz, dWg, dbg,
               dgamma, dbeta, B, T, F, seed, thr24, dscale, seed_dev);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------
// C = 64 / 128, (1,2) pooling: weight-stationary backward.  8 waves; wave (wm, wn) owns the 32x32 output tile
// (row block wm, channel block wn) of GEMM1 and GEMM2 and keeps BOTH orientations of Wg in registers
// (b1[k] = gamma_k Wg[n][k] -- the BN affine is folded into the weights, beta's contribution into the bias --
// and b2[k] = Wg[k][c]), so LDS holds only xhat and dlin tiles and every MFMA needs one LDS read.  The next
// tile's y rows and this tile's gout values are fetched into registers before GEMM1.
// ---------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(512) void glu_wide_bwd_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ Wg, const float* __restrict__ bg,
                                                           const float* __restrict__ gout, float* __restrict__ dz,
                                                           float* __restrict__ part, int B, int T,
                                                           int F, uint32_t seed, uint32_t thr24, float dscale, const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;            // per-step entropy in device memory (hipGraph replays)
    constexpr int CP = C + 1, NT = C / 32, WN = NT, WM = 8 / WN, ROWS = 32 * WM;   // flat row tiles, see glu_wide_fwd_kernel
    constexpr int NLD = ROWS * (C / 4) / 512, RSTEP = 512 / (C / 4);
    constexpr int NT3 = NT * NT, TPW = NT3 >= 8 ? NT3 / 8 : 1, KSPLIT = NT3 >= 8 ? 1 : 8 / NT3, KROWS = ROWS / KSPLIT;
    SED_DYN_SMEM(smem);
    float* xh = (float*)smem;           // [ROWS][CP] xhat
    float* dl = xh + ROWS * CP;         // [ROWS][CP] d lin
    float* wg = dl + ROWS * CP;         // [C][CP] Wg, only when B2_LDS
    constexpr bool B2_LDS = C > 64;     // 2 x C/2 weight registers do not fit next to the accumulators at C = 128
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int wn = w % WN, wm = w / WN, n = wn * 32 + lo;
    const int R = B * T * F, ntiles = (R + ROWS - 1) / ROWS;

    // Wg through LDS (coalesced global reads; see glu_wide_fwd_kernel): resident in `wg` at C = 128, staged in the
    // not-yet-used xhat tile at C = 64
    float b1[C / 2], b2[B2_LDS ? 1 : C / 2];
    float* wl = B2_LDS ? wg : xh;
    static_assert(B2_LDS || ROWS >= C, "xhat tile must hold Wg while the fragments are read");
    for (int i = tid; i < C * (C / 4); i += 512) {
        const int row = i / (C / 4), q = i - row * (C / 4);
        const float4 val = *(const float4*)(Wg + (size_t)row * C + 4 * q);
        float* d = wl + row * CP + 4 * q;
        d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
    }
    __syncthreads();
    float biasp = bg[n];
#pragma unroll
    for (int ks = 0; ks < C / 2; ++ks) {
        const float w0 = wl[n * CP + 2 * ks], w1 = wl[n * CP + 2 * ks + 1];
        biasp = fmaf(beta[2 * ks], w0, biasp);
        biasp = fmaf(beta[2 * ks + 1], w1, biasp);
        b1[ks] = gamma[2 * ks + hi] * (hi ? w1 : w0);
        if (!B2_LDS) b2[ks] = wl[(2 * ks + hi) * CP + n];
        if ((ks & 7) == 7) sed_sched_fence();
    }
    const float gn = gamma[n], bn = beta[n];
    const int v = tid % (C / 4), r0 = tid / (C / 4);
    float mu4[4], is4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { mu4[i] = stats[4 * v + i]; is4[i] = stats[C + 4 * v + i]; }

    f32x16 P[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) P[i] = f32x16_zero();
    float a_dbg = 0.f, a_dgam = 0.f, a_dbet = 0.f;

    float4 ld[NLD];
    auto load_tile = [&](int tile) {
        const int row0 = tile * ROWS;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int m = r0 + RSTEP * u;
            ld[u] = make_float4(mu4[0], mu4[1], mu4[2], mu4[3]);                     // -> xhat 0 for rows past the end
            if (row0 + m < R && !(GLU_ABL & 4)) ld[u] = *(const float4*)(y + (size_t)(row0 + m) * C + 4 * v);
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * ROWS;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            float* d = xh + (r0 + RSTEP * u) * CP + 4 * v;
            d[0] = (ld[u].x - mu4[0]) * is4[0]; d[1] = (ld[u].y - mu4[1]) * is4[1];
            d[2] = (ld[u].z - mu4[2]) * is4[2]; d[3] = (ld[u].w - mu4[3]) * is4[3];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
        // this lane's 16 accumulator rows pair up into 8 pooling windows: fetch their gout values now
        float g8[8];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rr = row0 + 32 * wm + 8 * j + 4 * hi + 2 * h;
                g8[2 * j + h] = (rr < R && !(GLU_ABL & 4)) ? gout[(size_t)(rr / 2) * C + n] * (0.5f * dscale) : 0.f;
            }
        // ---- GEMM1: lin = xn . Wg^T  (= xhat . (gamma Wg)^T + beta-folded bias) ----
        f32x16 acc = f32x16_zero();
        {
            mfma_chain<C / 2, 8>(xh + (32 * wm + lo) * CP + hi, [&](int ks) { return b1[ks]; }, acc);
        }
        // ---- epilogue 1: dlin -> LDS, e -> acc (seed of GEMM2) ----
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 32 * wm + mfma32_row(r, lane);
            const uint32_t e_idx = (uint32_t)(row0 + m) * (uint32_t)C + (uint32_t)n;
            float dlin = 0.f, e = 0.f;
            if (GLU_ABL & 2) { dlin = acc[r]; e = g8[r & 7]; }
            else if (row0 + m < R) {
                const float xn = fmaf(xh[m * CP + n], gn, bn);
                const float sg = sed_fast_sigmoid(xn);
                const float lin = acc[r] + biasp;
                const float g = sed_keep(e_idx, seed, thr24) ? g8[(r >> 2) * 2 + ((r & 3) >> 1)] : 0.f;
                dlin = g * sg;
                e = g * lin * sg * (1.0f - sg);
            }
            dl[m * CP + n] = dlin;
            acc[r] = e;
            a_dbg += dlin;
        }
        __syncthreads();
        // ---- GEMM2: dxn = dlin . Wg + e ----
        {
            const float* ap = dl + (32 * wm + lo) * CP + hi;
            if (B2_LDS) {
                const float* bp = wg + hi * CP + n;
#pragma unroll 8
                for (int ks = 0; ks < C / 2; ++ks) {
                    if (GLU_ABL & 1) acc[0] += ap[2 * ks] * bp[2 * ks * CP];
                    else acc = mfma32(ap[2 * ks], bp[2 * ks * CP], acc);
                }
            } else {
                mfma_chain<C / 2, 8>(ap, [&](int ks) { return b2[B2_LDS ? 0 : ks]; }, acc);
            }
        }
        // ---- epilogue 2: dz = dxn * gamma, BN reductions ----
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 32 * wm + mfma32_row(r, lane);
            if (row0 + m < R) {
                const float dxn = acc[r];
                a_dgam = fmaf(dxn, xh[m * CP + n], a_dgam);
                a_dbet += dxn;
                if (!(GLU_ABL & 8)) dz[(size_t)(row0 + m) * C + n] = dxn * gn;
            }
        }
        // ---- GEMM3: P[n'][c] += sum_rows dlin[row][n'] * xn[row][c] ----
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int id = NT3 >= 8 ? w * TPW + i : w % NT3, mt = id / NT, ct = id % NT;
            const int k0 = NT3 >= 8 ? 0 : (w / NT3) * KROWS;
            const int c = ct * 32 + lo;
            const float gc = gamma[c], bc = beta[c];
#pragma unroll 8
            for (int k = k0; k < k0 + KROWS; k += 2) {
                const float av = dl[(k + hi) * CP + mt * 32 + lo];
                const float bv = fmaf(xh[(k + hi) * CP + c], gc, bc);
                if (GLU_ABL & 1) P[i][0] += av * bv;
                else P[i] = mfma32(av, bv, P[i]);
            }
        }
    }
    // ---- per-workgroup partial sums -> scratch (plain stores; glu_bwd_reduce_kernel adds them in a fixed order:
    // device-scope float atomics from 256 workgroups onto the same 16K addresses cost more than the kernel body) ----
    float* mine = part + (size_t)blockIdx.x * (KSPLIT * C * C + WM * 3 * C);
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int id = NT3 >= 8 ? w * TPW + i : w % NT3, mt = id / NT, ct = id % NT, c = ct * 32 + lo;
        float* dst = mine + (NT3 >= 8 ? 0 : w / NT3) * C * C;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(mt * 32 + mfma32_row(r, lane)) * C + c] = P[i][r];
    }
    a_dbg += __shfl_xor(a_dbg, 32); a_dgam += __shfl_xor(a_dgam, 32); a_dbet += __shfl_xor(a_dbet, 32);
    if (hi == 0) {
        float* dst = mine + KSPLIT * C * C + wm * 3 * C;
        dst[n] = a_dbg; dst[C + n] = a_dgam; dst[2 * C + n] = a_dbet;
    }
}
// ---------------------------------------------------------------------------------------------
// Split-bf16 variant of the weight-stationary backward: all three GEMMs of a tile on v_mfma_f32_32x32x16_bf16 (three per
// product, fp32-level accuracy).  GEMM1 (lin = xhat (gamma Wg)^T) and GEMM2 (dxn = dlin Wg + e) contract over channels,
// so their A operands are row-major [row][C + 8] hi/lo planes; GEMM3 (dWg' = dlin^T xhat) contracts over ROWS, so it reads
// [channel][ROWS + 8] planes with the pixel octets XOR-swizzled by bits 4..6 of the row (as in conv_wgrad_bf16_kernel).
// Both layouts of xhat are written from one 4 row x 4 channel register block per thread; both layouts of dlin straight
// from the accumulator registers of epilogue 1 (a lane holds 4 consecutive rows of its column: one 8-byte transposed store).
// Both orientations of Wg live in registers as bf16 fragments.  GEMM3 uses xhat instead of xn = gamma xhat + beta:
// dWg[n][c] = gamma_c dWg'[n][c] + beta_c dbg[n] is applied by glu_bwd_reduce_kernel (fix = 1).
// ---------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(512) void glu_wide_bwd_b_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ Wg, const float* __restrict__ bg,
                                                             const float* __restrict__ gout, float* __restrict__ dz,
                                                             float* __restrict__ part, int B, int T, int F, uint32_t seed,
                                                             uint32_t thr24, float dscale, const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;
    constexpr int NT = C / 32, WN = NT, WM = 8 / WN, ROWS = 32 * WM, RS = C + 8, RT = ROWS + 8, KS = C / 16;
    constexpr int NT3 = NT * NT, TPW = NT3 >= 8 ? NT3 / 8 : 1, KSPLIT = NT3 >= 8 ? 1 : 8 / NT3, KS3 = ROWS / 16 / KSPLIT;
    static_assert(ROWS * C / 16 == 512, "one 4x4 block per thread");
    SED_DYN_SMEM(smem);
    unsigned short* xh = (unsigned short*)smem;      // xhat  [ROWS][RS] hi | lo
    unsigned short* xl = xh + ROWS * RS;
    unsigned short* xth = xl + ROWS * RS;            // xhat^T [C][RT] hi | lo (swizzled octets)
    unsigned short* xtl = xth + C * RT;
    unsigned short* dh = xtl + C * RT;               // dlin  [ROWS][RS] hi | lo
    unsigned short* dl = dh + ROWS * RS;
    unsigned short* dth = dl + ROWS * RS;            // dlin^T [C][RT] hi | lo (swizzled octets)
    unsigned short* dtl = dth + C * RT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int lo = lane & 31, hi = lane >> 5;
    const int wn = w % WN, wm = w / WN;
    int n = wn * 32 + lo;
    const int R = B * T * F, ntiles = (R + ROWS - 1) / ROWS;

    // ---- Wg fragments: b2 = columns of Wg (GEMM2), b1 = rows of gamma (.) Wg (GEMM1); beta folded into the bias ----
    s16x8 b1h[KS], b1l[KS], b2h[KS], b2l[KS];
    float biasp = bg[n];
#pragma unroll
    for (int p = 0; p < C / 64; ++p) {               // 64 rows of Wg per pass through the (still unused) xhat planes
#pragma unroll
        for (int fold = 0; fold < 2; ++fold) {
            __syncthreads();
            for (int i = tid; i < 64 * (C / 4); i += 512) {
                const int row = i / (C / 4), q = i - row * (C / 4);
                float4 val = *(const float4*)(Wg + (size_t)(p * 64 + row) * C + 4 * q);
                if (fold) { val.x *= gamma[4 * q]; val.y *= gamma[4 * q + 1]; val.z *= gamma[4 * q + 2]; val.w *= gamma[4 * q + 3]; }
                uint2 hv, lv;
                bf16_split2(val.x, val.y, hv.x, lv.x);
                bf16_split2(val.z, val.w, hv.y, lv.y);
                *(uint2*)(xh + row * RS + 4 * q) = hv;
                *(uint2*)(xl + row * RS + 4 * q) = lv;
            }
            __syncthreads();
            if (!fold) {
                // column n of rows 64p..64p+63: k-steps 4p..4p+3 of this lane's GEMM2 fragment; and beta . Wg[n][:] for rows here
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    s16x8 fh, fl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        fh[e] = (short)xh[(16 * kk + 8 * hi + e) * RS + n];
                        fl[e] = (short)xl[(16 * kk + 8 * hi + e) * RS + n];
                    }
                    b2h[4 * p + kk] = fh; b2l[4 * p + kk] = fl;
                }
                if (wn * 32 / 64 == p) {
                    const int rown = n - p * 64;
                    for (int k = 0; k < C; ++k) biasp = fmaf(beta[k], bf16_pair_sum(xh[rown * RS + k], xl[rown * RS + k]), biasp);
                }
            } else if (wn * 32 / 64 == p) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    b1h[ks] = *(const s16x8*)(xh + (n - p * 64) * RS + 16 * ks + 8 * hi);
                    b1l[ks] = *(const s16x8*)(xl + (n - p * 64) * RS + 16 * ks + 8 * hi);
                }
            }
        }
    }
    const float gn = gamma[n], bn = beta[n];
    int cq = tid % (C / 4), rq = tid / (C / 4);            // this thread's 4x4 staging block: channels 4cq.., rows 4rq..
    float mu4[4], is4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { mu4[i] = stats[4 * cq + i]; is4[i] = stats[C + 4 * cq + i]; }

    f32x16 P[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) P[i] = f32x16_zero();
    float a_dbg = 0.f, a_dgam = 0.f, a_dbet = 0.f;

    float4 ld0, ld1, ld2, ld3;
    // Register prefetch of the next tile: unconditional loads (rows past R read row R - 1), RAW values in flight under the GEMMs;
    // they are normalised (and the rows past R zeroed) when the tile is parked in LDS.  With `if (row < R) load; (val - mu) * is`
    // the compiler waited for each of the four loads right behind it (tools/isa_exposed_loads.py).
    auto load_row = [&](int row) -> float4 { return *(const float4*)(y + (size_t)(row < R ? row : R - 1) * C + 4 * cq); };
    auto load_tile = [&](int tile) {
        const int row = tile * ROWS + 4 * rq;
        ld0 = load_row(row); ld1 = load_row(row + 1); ld2 = load_row(row + 2); ld3 = load_row(row + 3);
    };
    auto normalise = [&](float4& val, int row) {
        const float keep = row < R ? 1.0f : 0.0f;                       // xhat 0 for rows past the end
        val.x = (val.x - mu4[0]) * is4[0] * keep; val.y = (val.y - mu4[1]) * is4[1] * keep;
        val.z = (val.z - mu4[2]) * is4[2] * keep; val.w = (val.w - mu4[3]) * is4[3] * keep;
    };
    auto store_rm = [&](const float4 val, int m) {                      // one row of the block -> row-major planes
        uint2 hv, lv;
        bf16_split2(val.x, val.y, hv.x, lv.x);
        bf16_split2(val.z, val.w, hv.y, lv.y);
        *(uint2*)(xh + m * RS + 4 * cq) = hv;
        *(uint2*)(xl + m * RS + 4 * cq) = lv;
    };
    auto store_tr = [&](float a, float b, float c, float d, int ch) {   // one channel of the block -> transposed planes
        uint2 hv, lv;
        bf16_split2(a, b, hv.x, lv.x);
        bf16_split2(c, d, hv.y, lv.y);
        const int off = ch * RT + 4 * (rq ^ (2 * ((ch >> 4) & 7)));
        *(uint2*)(xth + off) = hv;
        *(uint2*)(xtl + off) = lv;
    };
    // xhat of this lane's row quad 8 wm + 2 j + hi (rows 32 wm + 8 j + 4 hi .. + 3) of its column n, out of the TRANSPOSED planes:
    // one 8-byte read per plane instead of four 2-byte reads from the row-major planes (round 4; same bf16 pairs, same sums)
    auto xhat4 = [&](int j, float* out4) {
        const int rqd = 8 * wm + 2 * j + hi, off = n * RT + 4 * (rqd ^ (2 * ((n >> 4) & 7)));
        const uint2 hv = *(const uint2*)(xth + off), lv = *(const uint2*)(xtl + off);
        out4[0] = bf16_pair_sum((unsigned short)(hv.x & 0xFFFFu), (unsigned short)(lv.x & 0xFFFFu));
        out4[1] = bf16_pair_sum((unsigned short)(hv.x >> 16), (unsigned short)(lv.x >> 16));
        out4[2] = bf16_pair_sum((unsigned short)(hv.y & 0xFFFFu), (unsigned short)(lv.y & 0xFFFFu));
        out4[3] = bf16_pair_sum((unsigned short)(hv.y >> 16), (unsigned short)(lv.y >> 16));
    };
    const float gsc = 0.5f * dscale;
    float g8n[8];
    auto load_g8 = [&](int tile_) {
        const int rbase = tile_ * ROWS + 32 * wm + 4 * hi;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rr = rbase + 8 * j + 2 * h;
                g8n[2 * j + h] = gout[(size_t)((rr < R ? rr : R - 1) / 2) * C + n];
            }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) { load_tile(tile); load_g8(tile); }
    for (; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * ROWS;
        sed_opaque(lo); sed_opaque(hi); sed_opaque(n); sed_opaque(cq); sed_opaque(rq);   // per-tile addresses: recomputed, not spilled
        normalise(ld0, row0 + 4 * rq); normalise(ld1, row0 + 4 * rq + 1); normalise(ld2, row0 + 4 * rq + 2); normalise(ld3, row0 + 4 * rq + 3);
        __syncthreads();
        store_rm(ld0, 4 * rq); store_rm(ld1, 4 * rq + 1); store_rm(ld2, 4 * rq + 2); store_rm(ld3, 4 * rq + 3);
        store_tr(ld0.x, ld1.x, ld2.x, ld3.x, 4 * cq);
        store_tr(ld0.y, ld1.y, ld2.y, ld3.y, 4 * cq + 1);
        store_tr(ld0.z, ld1.z, ld2.z, ld3.z, 4 * cq + 2);
        store_tr(ld0.w, ld1.w, ld2.w, ld3.w, 4 * cq + 3);
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
        // the eight upstream-gradient values of this lane's pooling windows: this tile's were fetched one iteration ago, the next
        // tile's go out now (unconditional, clamped rows; scale and row predicate are applied at the use).  As `rr < R ? gout[..] * c : 0`
        // each of the eight loads sat in its own branch with s_waitcnt vmcnt(0) behind it -- which also drained the y prefetch above.
        float g8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) g8[u] = g8n[u];
        load_g8(tile + (int)gridDim.x < ntiles ? tile + gridDim.x : tile);
        // ---- GEMM1 ----
        f32x16 acc = f32x16_zero();
        {
            const unsigned short* ap = xh + (32 * wm + lo) * RS + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const s16x8 a_hi = *(const s16x8*)(ap + 16 * ks);
                const s16x8 a_lo = *(const s16x8*)(ap + ROWS * RS + 16 * ks);
                acc = mfma32_bf16(a_lo, b1h[ks], acc);
                acc = mfma32_bf16(a_hi, b1l[ks], acc);
                acc = mfma32_bf16(a_hi, b1h[ks], acc);
            }
        }
        // ---- epilogue 1: dlin -> both LDS layouts, e -> acc (seed of GEMM2) ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float dv[4], xq[4];
            xhat4(j, xq);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * j + q, m = 32 * wm + 8 * j + 4 * hi + q;
                const uint32_t e_idx = (uint32_t)(row0 + m) * (uint32_t)C + (uint32_t)n;
                float dlin = 0.f, e = 0.f;
                if (row0 + m < R) {
                    const float xn = fmaf(xq[q], gn, bn);
                    const float sg = sed_fast_sigmoid(xn);
                    const float lin = acc[r] + biasp;
                    const float g = sed_keep(e_idx, seed, thr24) ? g8[2 * j + (q >> 1)] * gsc : 0.f;
                    dlin = g * sg;
                    e = g * lin * sg * (1.0f - sg);
                }
                unsigned short hh, ll;
                bf16_split(dlin, hh, ll);
                dh[m * RS + n] = hh;
                dl[m * RS + n] = ll;
                dv[q] = dlin;
                acc[r] = e;
                a_dbg += dlin;
            }
            uint2 hv, lv;
            bf16_split2(dv[0], dv[1], hv.x, lv.x);
            bf16_split2(dv[2], dv[3], hv.y, lv.y);
            const int rqd = 8 * wm + 2 * j + hi;                        // row quad of rows 32wm + 8j + 4hi .. +3
            const int off = n * RT + 4 * (rqd ^ (2 * ((n >> 4) & 7)));
            *(uint2*)(dth + off) = hv;
            *(uint2*)(dtl + off) = lv;
        }
        __syncthreads();
        // ---- GEMM2: dxn = dlin . Wg + e ----
        {
            const unsigned short* ap = dh + (32 * wm + lo) * RS + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const s16x8 a_hi = *(const s16x8*)(ap + 16 * ks);
                const s16x8 a_lo = *(const s16x8*)(ap + ROWS * RS + 16 * ks);
                acc = mfma32_bf16(a_lo, b2h[ks], acc);
                acc = mfma32_bf16(a_hi, b2l[ks], acc);
                acc = mfma32_bf16(a_hi, b2h[ks], acc);
            }
        }
        // ---- epilogue 2: dz = dxn * gamma, BN reductions ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float xq[4];
            xhat4(j, xq);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * j + q, m = 32 * wm + 8 * j + 4 * hi + q;      // == 32 wm + mfma32_row(r, lane)
                if (row0 + m < R) {
                    const float dxn = acc[r];
                    a_dgam = fmaf(dxn, xq[q], a_dgam);
                    a_dbet += dxn;
                    dz[(size_t)(row0 + m) * C + n] = dxn * gn;
                }
            }
        }
        // ---- GEMM3: P'[n'][c] += sum_rows dlin[row][n'] * xhat[row][c] ----
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int id = NT3 >= 8 ? w * TPW + i : w % NT3, mt = id / NT, ct = id % NT;
            const int ks0 = NT3 >= 8 ? 0 : (w / NT3) * KS3;
            const int rowa = mt * 32 + lo, rowb = ct * 32 + lo;
#pragma unroll
            for (int ks = 0; ks < KS3; ++ks) {
                const int oct = 2 * (ks0 + ks) + hi;
                const int offa = rowa * RT + 8 * (oct ^ ((rowa >> 4) & 7)), offb = rowb * RT + 8 * (oct ^ ((rowb >> 4) & 7));
                const s16x8 a_hi = *(const s16x8*)(dth + offa), a_lo = *(const s16x8*)(dtl + offa);
                const s16x8 b_hi = *(const s16x8*)(xth + offb), b_lo = *(const s16x8*)(xtl + offb);
                P[i] = mfma32_bf16(a_lo, b_hi, P[i]);
                P[i] = mfma32_bf16(a_hi, b_lo, P[i]);
                P[i] = mfma32_bf16(a_hi, b_hi, P[i]);
            }
        }
    }
    float* mine = part + (size_t)blockIdx.x * (KSPLIT * C * C + WM * 3 * C);
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int id = NT3 >= 8 ? w * TPW + i : w % NT3, mt = id / NT, ct = id % NT, c = ct * 32 + lo;
        float* dst = mine + (NT3 >= 8 ? 0 : w / NT3) * C * C;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(mt * 32 + mfma32_row(r, lane)) * C + c] = P[i][r];
    }
    a_dbg += __shfl_xor(a_dbg, 32); a_dgam += __shfl_xor(a_dgam, 32); a_dbet += __shfl_xor(a_dbet, 32);
    if (hi == 0) {
        float* dst = mine + KSPLIT * C * C + wm * 3 * C;
        dst[n] = a_dbg; dst[C + n] = a_dgam; dst[2 * C + n] = a_dbet;
    }
}
// ---------------------------------------------------------------------------------------------
// C = 128 on v_mfma_f32_16x16x32_bf16: the same three split-bf16 GEMMs per 64-row tile and the same LDS planes as
// glu_wide_bwd_b_kernel, but a wave owns 16 output columns x all 64 rows.  Both orientations of Wg are then 4 k-steps x
// (hi, lo) x 2 GEMMs = 64 VGPRs of fragments per wave instead of 128 (the 32x32 tiling of the kernel above spills 83 dwords
// per lane at C = 128), four independent accumulator chains (one per 16-row block) keep the MFMA pipe busy, and GEMM3 is
// 8 blocks of 16x16 per wave (n' block = wave, all 8 channel blocks) = 32 accumulator VGPRs.  ~170 VGPRs, no scratch.
// The price is LDS read volume: every wave reads the whole A tile (16 KB per plane pair and GEMM), so the planes are laid
// out for conflict-free ds_read_b128 (see rm_off / tr_off below).
// One [C][C] slab + one [3][C] vector slab per workgroup partial (KS = 1, WMS = 1 for glu_bwd_reduce_kernel, fix = 1).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void glu128_bwd_c_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ Wg, const float* __restrict__ bg,
                                                           const float* __restrict__ gout, float* __restrict__ dz,
                                                           float* __restrict__ part, int B, int T, int F, uint32_t seed,
                                                           uint32_t thr24, float dscale, const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;
    constexpr int C = 128, ROWS = 64, RS = C, RT = ROWS, WS = C + 8, KS = C / 32, RB = ROWS / 16, NB = C / 16, KS3 = ROWS / 32;
    SED_DYN_SMEM(smem);
    unsigned short* xh = (unsigned short*)smem;      // xhat  [ROWS][RS] hi | lo
    unsigned short* xl = xh + ROWS * RS;
    unsigned short* xth = xl + ROWS * RS;            // xhat^T [C][RT] hi | lo (swizzled octets)
    unsigned short* xtl = xth + C * RT;
    unsigned short* dh = xtl + C * RT;               // dlin  [ROWS][RS] hi | lo
    unsigned short* dl = dh + ROWS * RS;
    unsigned short* dth = dl + ROWS * RS;            // dlin^T [C][RT] hi | lo (swizzled octets)
    unsigned short* dtl = dth + C * RT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int i16 = lane & 15, g = lane >> 4;
    int n = 16 * w + i16;                            // this lane's column: GEMM1 output n, GEMM2 output channel c
    const int R = B * T * F, ntiles = (R + ROWS - 1) / ROWS;
    // Unpadded planes, 16-byte octets XOR-swizzled so that every ds_read_b128 lane group ({g even: rows 0-3,12-15} + {g odd:
    // rows 4-11} and its complement) covers the 16 slots of a 256-byte bank row exactly once:
    //   row-major  [row][128]: octet o of row m      at slot o ^ (m & 15)
    //   transposed [ch][64]:   octet o of channel ch at slot o ^ ((ch >> 1) & 7)   (two channels per bank row)
    auto rm_off = [](int m, int k) { return m * RS + ((((k >> 3) ^ (m & 15))) << 3) + (k & 7); };
    auto tr_off = [](int ch, int row) { return ch * RT + ((((row >> 3) ^ ((ch >> 1) & 7))) << 3) + (row & 7); };
    unsigned short* wh = xh;                         // Wg staging (before the tile loop): plain [64][WS] hi | lo over the planes
    unsigned short* wl = xh + 64 * WS;

    // ---- Wg fragments (16x16x32: lane (i16, g) holds k = 32 ks + 8 g + e of its column) ----
    s16x8 b1h[KS], b1l[KS], b2h[KS], b2l[KS];
    float biasp = bg[n];
#pragma unroll
    for (int p = 0; p < C / 64; ++p) {               // 64 rows of Wg per pass through the (still unused) xhat planes
#pragma unroll
        for (int fold = 0; fold < 2; ++fold) {
            __syncthreads();
            for (int i = tid; i < 64 * (C / 4); i += 512) {
                const int row = i / (C / 4), q = i - row * (C / 4);
                float4 val = *(const float4*)(Wg + (size_t)(p * 64 + row) * C + 4 * q);
                if (fold) { val.x *= gamma[4 * q]; val.y *= gamma[4 * q + 1]; val.z *= gamma[4 * q + 2]; val.w *= gamma[4 * q + 3]; }
                uint2 hv, lv;
                bf16_split2(val.x, val.y, hv.x, lv.x);
                bf16_split2(val.z, val.w, hv.y, lv.y);
                *(uint2*)(wh + row * WS + 4 * q) = hv;
                *(uint2*)(wl + row * WS + 4 * q) = lv;
            }
            __syncthreads();
            if (!fold) {
                // column n of rows 64p..64p+63 = k-steps 2p, 2p+1 of the GEMM2 fragment; beta . Wg[n][:] where row n is staged
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    s16x8 fh, fl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        fh[e] = (short)wh[(32 * kk + 8 * g + e) * WS + n];
                        fl[e] = (short)wl[(32 * kk + 8 * g + e) * WS + n];
                    }
                    b2h[2 * p + kk] = fh; b2l[2 * p + kk] = fl;
                }
                if (w / 4 == p) {                                       // the four g lanes of column n take 32 k each
                    const int rown = n - p * 64;
                    float part_b = 0.f;
                    for (int k = 32 * g; k < 32 * g + 32; ++k)
                        part_b = fmaf(beta[k], bf16_pair_sum(wh[rown * WS + k], wl[rown * WS + k]), part_b);
                    part_b += __shfl_xor(part_b, 16);
                    part_b += __shfl_xor(part_b, 32);
                    biasp += part_b;
                }
            } else if (w / 4 == p) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    b1h[ks] = *(const s16x8*)(wh + (n - p * 64) * WS + 32 * ks + 8 * g);
                    b1l[ks] = *(const s16x8*)(wl + (n - p * 64) * WS + 32 * ks + 8 * g);
                }
            }
        }
    }
    const float gn = gamma[n], bn = beta[n];
    int cq = tid % (C / 4), rq = tid / (C / 4);            // this thread's 4x4 staging block: channels 4cq.., rows 4rq..

    f32x4 P[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) P[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a_dbg = 0.f, a_dgam = 0.f, a_dbet = 0.f;

    float4 ld0, ld1, ld2, ld3;
    // raw rows of the next tile (prefetched under GEMM3; normalised when they are staged: mean / invstd are re-read from the
    // cache then instead of living in 8 VGPRs through the GEMMs)
    // (full tiles take straight-line code: per-row / per-element guards compile to one branch each and fence the scheduler)
    auto load_row = [&](int row) -> float4 {
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < R && !(GLU_ABL & 4)) val = *(const float4*)(y + (size_t)row * C + 4 * cq);
        return val;
    };
    auto load_tile = [&](int tile) {
        const int row = tile * ROWS + 4 * rq;
        if ((tile + 1) * ROWS <= R && !(GLU_ABL & 4)) {
            const float* src = y + (size_t)row * C + 4 * cq;
            ld0 = *(const float4*)src; ld1 = *(const float4*)(src + C); ld2 = *(const float4*)(src + 2 * C); ld3 = *(const float4*)(src + 3 * C);
        } else {
            ld0 = load_row(row); ld1 = load_row(row + 1); ld2 = load_row(row + 2); ld3 = load_row(row + 3);
        }
    };
    auto normalise = [&](int tile) {
        const float4 mu = *(const float4*)(stats + 4 * cq), is = *(const float4*)(stats + C + 4 * cq);
        const int row = tile * ROWS + 4 * rq;
        auto nrm = [&](float4& v, int r) {
            const float keep = r < R ? 1.0f : 0.0f;                     // rows past the end: xhat = 0
            v.x = (v.x - mu.x) * is.x * keep; v.y = (v.y - mu.y) * is.y * keep;
            v.z = (v.z - mu.z) * is.z * keep; v.w = (v.w - mu.w) * is.w * keep;
        };
        nrm(ld0, row); nrm(ld1, row + 1); nrm(ld2, row + 2); nrm(ld3, row + 3);      // rows past the end stay 0
    };
    auto store_rm = [&](const float4 val, int m) {                      // one row of the block -> row-major planes
        uint2 hv, lv;
        bf16_split2(val.x, val.y, hv.x, lv.x);
        bf16_split2(val.z, val.w, hv.y, lv.y);
        *(uint2*)(xh + rm_off(m, 4 * cq)) = hv;
        *(uint2*)(xl + rm_off(m, 4 * cq)) = lv;
    };
    auto store_tr = [&](float a, float b, float c, float d, int ch) {   // one channel of the block -> transposed planes
        uint2 hv, lv;
        bf16_split2(a, b, hv.x, lv.x);
        bf16_split2(c, d, hv.y, lv.y);
        const int off = tr_off(ch, 4 * rq);
        *(uint2*)(xth + off) = hv;
        *(uint2*)(xtl + off) = lv;
    };
    // xhat of this lane's four rows 16 rb + 4 g .. + 3 of its column n, out of the TRANSPOSED planes: the four rows are one 8-byte
    // half-octet of channel n there (round 4: 2 ds_read_b64 instead of 8 ds_read_u16 from the row-major planes per block and
    // epilogue -- 48 of the ~ 220 LDS instructions a lane issues per tile; same bf16 pairs, same sums)
    auto xhat4 = [&](int rb, float* out4) {
        const int off = tr_off(n, 16 * rb + 4 * g);
        const uint2 hv = *(const uint2*)(xth + off), lv = *(const uint2*)(xtl + off);
        out4[0] = bf16_pair_sum((unsigned short)(hv.x & 0xFFFFu), (unsigned short)(lv.x & 0xFFFFu));
        out4[1] = bf16_pair_sum((unsigned short)(hv.x >> 16), (unsigned short)(lv.x >> 16));
        out4[2] = bf16_pair_sum((unsigned short)(hv.y & 0xFFFFu), (unsigned short)(lv.y & 0xFFFFu));
        out4[3] = bf16_pair_sum((unsigned short)(hv.y >> 16), (unsigned short)(lv.y >> 16));
    };
    // acc[rb] += A[16rb.., :] . B for the four 16-row blocks, A = hi | lo planes (lo plane ROWS * RS further), B = this
    // wave's fragments.  Per k-step: the eight A reads, then the twelve MFMAs pass by pass across the four blocks, so that
    // consecutive MFMAs never share an accumulator (left alone the compiler emits read - wait - three dependent MFMAs).
    // Double-buffering the reads across k-steps was measured and bought nothing (85.8 vs 87.0 us at F = 16).
    auto gemm_rows = [&](const unsigned short* ph, const s16x8* bh, const s16x8* bl, f32x4* acc) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            s16x8 ah[RB], al[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const unsigned short* ap = ph + rm_off(16 * rb + i16, 32 * ks + 8 * g);
                ah[rb] = *(const s16x8*)ap;
                al[rb] = *(const s16x8*)(ap + ROWS * RS);
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma16_bf16(al[rb], bh[ks], acc[rb]);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma16_bf16(ah[rb], bl[ks], acc[rb]);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma16_bf16(ah[rb], bh[ks], acc[rb]);
            sed_sched_fence();
        }
    };
    int tile = blockIdx.x;
    float g8n[2 * RB];
    auto load_g8 = [&](int tile_) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rr = tile_ * ROWS + 16 * rb + 4 * g + 2 * h;
                g8n[2 * rb + h] = (GLU_ABL & 4) ? 1.0f : gout[(size_t)((rr < R ? rr : 0) / 2) * C + n];      // clamped address: no branch
            }
    };
    if (tile < ntiles) { load_tile(tile); load_g8(tile); }
    for (; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * ROWS;
        sed_opaque(i16); sed_opaque(g); sed_opaque(n); sed_opaque(cq); sed_opaque(rq);   // addresses are recomputed per tile, not spilled
        normalise(tile);
        __syncthreads();
        store_rm(ld0, 4 * rq); store_rm(ld1, 4 * rq + 1); store_rm(ld2, 4 * rq + 2); store_rm(ld3, 4 * rq + 3);
        store_tr(ld0.x, ld1.x, ld2.x, ld3.x, 4 * cq);
        store_tr(ld0.y, ld1.y, ld2.y, ld3.y, 4 * cq + 1);
        store_tr(ld0.z, ld1.z, ld2.z, ld3.z, 4 * cq + 2);
        store_tr(ld0.w, ld1.w, ld2.w, ld3.w, 4 * cq + 3);
        __syncthreads();
        // the pooled upstream gradient of this lane's 16 elements: this tile's values were fetched one tile ago (raw; scale and row
        // predicate are applied here), the next tile's loads go out now and fly under all three GEMMs.  Loaded and scaled in place
        // they were waited for right behind their issue, in front of GEMM1 (tools/isa_exposed_loads.py).
        float g8[2 * RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rr = row0 + 16 * rb + 4 * g + 2 * h;          // R is even: a pooling pair is inside or outside together
                g8[2 * rb + h] = rr < R ? g8n[2 * rb + h] * (0.5f * dscale) : 0.f;
            }
        load_g8(tile + (int)gridDim.x < ntiles ? tile + gridDim.x : tile);
        // ---- GEMM1: lin = xhat . (gamma Wg)^T, four independent 16-row chains ----
        f32x4 acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(GLU_ABL & 1)) gemm_rows(xh, b1h, b1l, acc);
        // ---- epilogue 1: dlin -> both LDS layouts (one split feeds both), e -> acc (seed of GEMM2) ----
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            unsigned short hh[4], ll[4];
            float xq[4];
            xhat4(rb, xq);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * rb + 4 * g + r;
                const uint32_t e_idx = (uint32_t)(row0 + m) * (uint32_t)C + (uint32_t)n;
                // rows past the end: g8 = 0 there, so dlin = e = 0 without a guard (their xhat is 0: everything stays finite)
                const float xn = fmaf(xq[r], gn, bn);
                const float sg = (GLU_ABL & 2) ? xn : sed_fast_sigmoid(xn);
                const float lin = acc[rb][r] + biasp;
                const float gq = ((GLU_ABL & 2) || sed_keep(e_idx, seed, thr24)) ? g8[2 * rb + (r >> 1)] : 0.f;
                const float dlin = gq * sg;
                const float e = gq * lin * sg * (1.0f - sg);
                bf16_split(dlin, hh[r], ll[r]);
                dh[rm_off(m, n)] = hh[r];
                dl[rm_off(m, n)] = ll[r];
                acc[rb][r] = e;
                a_dbg += dlin;
            }
            uint2 hv, lv;
            hv.x = (unsigned)hh[0] | ((unsigned)hh[1] << 16); hv.y = (unsigned)hh[2] | ((unsigned)hh[3] << 16);
            lv.x = (unsigned)ll[0] | ((unsigned)ll[1] << 16); lv.y = (unsigned)ll[2] | ((unsigned)ll[3] << 16);
            const int off = tr_off(n, 16 * rb + 4 * g);                 // rows 16rb + 4g .. +3 of channel n: half an octet
            *(uint2*)(dth + off) = hv;
            *(uint2*)(dtl + off) = lv;
        }
        __syncthreads();
        // ---- GEMM2: dxn = dlin . Wg + e ----
        if (!(GLU_ABL & 1)) gemm_rows(dh, b2h, b2l, acc);
        // ---- epilogue 2: dz = dxn * gamma, BN reductions (dxn = 0 on rows past the end) ----
        const bool full = row0 + ROWS <= R;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            float xq[4];
            xhat4(rb, xq);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * rb + 4 * g + r;
                const float dxn = acc[rb][r];
                a_dgam = fmaf(dxn, xq[r], a_dgam);
                a_dbet += dxn;
                if (!(GLU_ABL & 8) && (full || row0 + m < R)) dz[(size_t)(row0 + m) * C + n] = dxn * gn;
            }
        }
        // ---- GEMM3: P'[n'][c] += sum_rows dlin[row][n'] * xhat[row][c]   (n' block = wave, every channel block) ----
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
        if (!(GLU_ABL & 1)) {
            // four steps of (k-step, half of the channel blocks): 8 (+2) reads, then 12 MFMAs pass by pass over 4 accumulators
            const int rowa = 16 * w + i16;
            s16x8 ah3, al3;
#pragma unroll
            for (int st = 0; st < 2 * KS3; ++st) {
                const int ks = st >> 1, c0 = 4 * (st & 1);
                if ((st & 1) == 0) {
                    const int offa = tr_off(rowa, 32 * ks + 8 * g);
                    ah3 = *(const s16x8*)(dth + offa);
                    al3 = *(const s16x8*)(dtl + offa);
                }
                s16x8 bh3[4], bl3[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int offb = tr_off(16 * (c0 + q) + i16, 32 * ks + 8 * g);
                    bh3[q] = *(const s16x8*)(xth + offb);
                    bl3[q] = *(const s16x8*)(xtl + offb);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) P[c0 + q] = mfma16_bf16(al3, bh3[q], P[c0 + q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) P[c0 + q] = mfma16_bf16(ah3, bl3[q], P[c0 + q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) P[c0 + q] = mfma16_bf16(ah3, bh3[q], P[c0 + q]);
                sed_sched_fence();
            }
        }
    }
    float* mine = part + (size_t)blockIdx.x * (C * C + 3 * C);
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[(16 * w + 4 * g + r) * C + 16 * cb + i16] = P[cb][r];
    a_dbg += __shfl_xor(a_dbg, 16); a_dgam += __shfl_xor(a_dgam, 16); a_dbet += __shfl_xor(a_dbet, 16);
    a_dbg += __shfl_xor(a_dbg, 32); a_dgam += __shfl_xor(a_dgam, 32); a_dbet += __shfl_xor(a_dbet, 32);
    if (g == 0) {
        float* dst = mine + C * C;
        dst[n] = a_dbg; dst[C + n] = a_dgam; dst[2 * C + n] = a_dbet;
    }
}
// sums the nblk per-workgroup partials of glu_wide_bwd_kernel: [KS][C][C] dWg slabs, then [WMS][3][C] (dbg, dgamma, dbeta).
// One workgroup per 64 consecutive outputs: 16 float4 columns x 64 groups of partials, fixed summation order.  1024 threads:
// the launch is a latency chain of dependent loads (256-1024 partials per output), so the walk per thread must be short.
#define GBR_THREADS 1024
#define GBR_GROUPS (GBR_THREADS / 16)
__global__ __launch_bounds__(GBR_THREADS) void glu_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dWg,
                                                             float* __restrict__ dbg, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int nblk, int C, int KS, int WMS,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, int fix) {
    // fix != 0 (split-bf16 kernel): the dWg slabs hold dWg' = dlin^T xhat; dWg[n][c] = gamma_c dWg'[n][c] + beta_c dbg[n]
    __shared__ float4 red[GBR_GROUPS][16];
    __shared__ float sdb[GBR_THREADS / 64];
    const int tid = threadIdx.x, col = tid & 15, grp = tid >> 4, e = blockIdx.x * 64 + 4 * col;
    const int CC = C * C, PART = KS * CC + WMS * 3 * C;
    // outputs [0, CC) are dWg with KS slabs per partial; [CC, CC + 3C) the three vectors with WMS slabs (C % 4 == 0 keeps
    // a float4 inside one of them)
    const bool isw = e < CC;
    const int nsl = isw ? KS : WMS, sstride = isw ? CC : 3 * C;
    const size_t off = isw ? (size_t)e : (size_t)KS * CC + (e - CC);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < CC + 3 * C) {
        const int total = nblk * nsl;                                  // (partial, slab) pairs, split over the groups
#pragma unroll 4
        for (int i = grp; i < total; i += GBR_GROUPS) {
            const int b = i / nsl, k = i - b * nsl;
            const float4 v = *(const float4*)(part + (size_t)b * PART + off + (size_t)k * sstride);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    // Everything this launch reads is requested BEFORE its one barrier (round 4): dbg[n] of the block's row and the gamma / beta quads
    // used to be fetched after it, behind a ten-barrier LDS tree -- two more exposed round trips in a launch that is nothing but
    // latency (6 - 9 us, six of them on the backward chain between a GLU backward and its data-gradient convolution).
    const bool need_dbn = fix && blockIdx.x * 64 < CC;     // dbg[n] for this block's row n = (blockIdx.x * 64) / C   (C >= 64)
    float sacc = 0.f;
    if (need_dbn) {
        const int nrow = (blockIdx.x * 64) / C;
#pragma unroll 2
        for (int i = tid; i < nblk * WMS; i += GBR_THREADS) {
            const int b = i / WMS, k = i - b * WMS;
            sacc += part[(size_t)b * PART + (size_t)KS * CC + k * 3 * C + nrow];
        }
    }
    float4 gq = make_float4(0.f, 0.f, 0.f, 0.f), bq = gq;
    if (fix && isw && grp == 0) {                        // (four scalar loads: a parameter is only guaranteed 4-byte alignment)
        const int c = e % C;
        gq = make_float4(gamma[c], gamma[c + 1], gamma[c + 2], gamma[c + 3]);
        bq = make_float4(beta[c], beta[c + 1], beta[c + 2], beta[c + 3]);
    }
    red[grp][col] = acc;
    if (need_dbn) {
        sacc = wave_sum(sacc);
        if ((tid & 63) == 0) sdb[tid >> 6] = sacc;
    }
    __syncthreads();
    float dbn = 0.f;
    if (need_dbn) {
#pragma unroll
        for (int w = 0; w < GBR_THREADS / 64; ++w) dbn += sdb[w];
    }
    if (grp == 0 && e < CC + 3 * C) {
#pragma unroll 8
        for (int g = 1; g < GBR_GROUPS; ++g) { const float4 v = red[g][col]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        if (fix && isw) {
            acc.x = fmaf(gq.x, acc.x, bq.x * dbn); acc.y = fmaf(gq.y, acc.y, bq.y * dbn);
            acc.z = fmaf(gq.z, acc.z, bq.z * dbn); acc.w = fmaf(gq.w, acc.w, bq.w * dbn);
        }
        float* dst;
        if (isw) dst = dWg + e;
        else {
            const int j2 = e - CC, which = j2 / C;
            dst = (which == 0 ? dbg : which == 1 ? dgamma : dbeta) + (j2 - which * C);
        }
        *(float4*)dst = acc;
    }
}
template <int C, bool SPLIT>
static int launch_glu_wide_bwd(const float* y, const float* stats, const float* gamma, const float* beta, const float* Wg,
                               const float* bg, const float* gout, float* dz, float* dWg, float* dbg, float* dgamma, float* dbeta,
                               float* scratch, int B, int T, int F, uint32_t seed, uint32_t thr24, float dscale, const unsigned* seed_dev, hipStream_t s) {
    constexpr int ROWS = 32 * (8 / (C / 32)), WMS = 8 / (C / 32), KS = (C / 32) * (C / 32) >= 8 ? 1 : 8 / ((C / 32) * (C / 32));
    if (!scratch) return SED_ERR_ARG;
    constexpr int SMEM = SPLIT ? 8 * (ROWS * (C + 8) + C * (ROWS + 8)) : (2 * ROWS + (C > 64 ? C : 0)) * (C + 1) * 4;
    const int ntiles = (B * T * F + ROWS - 1) / ROWS;
    const int cap = glu_grid_cap(256);                                  // register-bound: one workgroup per CU
    int grid = ntiles < cap ? ntiles : cap;
    if (grid < 1) { sed_zero4(s, dWg, C * C, dbg, C, dgamma, C, dbeta, C); return SED_OK; }
    if (SPLIT && C == 128 && sed_tuning[SED_TUNE_GLU_BWD128_SPLIT] != 1) {
        // 16x16x32 tiling: one [C][C] + one [3][C] slab per partial
        constexpr int SMEM_C = 8 * 64 * 128 * 2;                          // eight unpadded bf16 planes of one 64 x 128 tile
        SED_MAX_SMEM(glu128_bwd_c_kernel, SMEM_C);
        SED_LAUNCH(glu128_bwd_c_kernel, dim3(grid), dim3(512), SMEM_C, s, y, stats, gamma, beta, Wg, bg, gout, dz, scratch, B, T, F,
                   seed, thr24, dscale, seed_dev);
        SED_LAUNCH(glu_bwd_reduce_kernel, dim3((C * C + 3 * C + 63) / 64), dim3(GBR_THREADS), 0, s, scratch, dWg, dbg, dgamma, dbeta, grid, C, 1, 1,
                   gamma, beta, 1);
        return sed_check_launch();
    }
    if (SPLIT) {
        SED_MAX_SMEM((glu_wide_bwd_b_kernel<C>), SMEM);
        SED_LAUNCH((glu_wide_bwd_b_kernel<C>), dim3(grid), dim3(512), SMEM, s, y, stats, gamma, beta, Wg, bg, gout, dz, scratch, B, T, F,
                   seed, thr24, dscale, seed_dev);
    } else {
        SED_MAX_SMEM((glu_wide_bwd_kernel<C>), SMEM);
        SED_LAUNCH((glu_wide_bwd_kernel<C>), dim3(grid), dim3(512), SMEM, s, y, stats, gamma, beta, Wg, bg, gout, dz, scratch, B, T, F,
                   seed, thr24, dscale, seed_dev);
    }
    SED_LAUNCH(glu_bwd_reduce_kernel, dim3((C * C + 3 * C + 63) / 64), dim3(GBR_THREADS), 0, s, scratch, dWg, dbg, dgamma, dbeta, grid, C, KS, WMS,
               gamma, beta, SPLIT ? 1 : 0);
    return sed_check_launch();
}

// Floats of scratch sed_glu_bwd needs: the 64/128-channel (1,2)-pooled blocks keep one partial (dWg, dbg, dgamma,
// dbeta) per workgroup (at most 256) and reduce them in a fixed order; the narrow blocks accumulate with atomics (0).
SED_API long long sed_glu_bwd_scratch_floats(int B, int T, int F, int C, int PT, int PF) {
    (void)B; (void)T; (void)F;
    if (PT == 1 && PF == 2 && (C == 64 || C == 128)) {
        const int nt3 = (C / 32) * (C / 32), ks = nt3 >= 8 ? 1 : 8 / nt3, wms = 8 / (C / 32);
        return 256LL * (ks * C * C + wms * 3 * C);
    }
    if (PT == 2 && PF == 2 && (C == 16 || C == 32)) return 1024LL * (C * C + 3 * C);       // one partial per workgroup
    return 0;
}

// gout (B,T/PT,F/PF,C) -> dz (B,T,F,C) = dL/d xhat; dWg (C,C), dbg, dgamma, dbeta (C) are overwritten (zeroed here and
// accumulated with fp32 atomics on the narrow blocks; reduced from `scratch`, see above, on the wide ones).
// When T % PT != 0 the dropped frames of dz are zeroed too.
SED_API int sed_glu_bwd(const float* y, const float* stats, const float* gamma, const float* beta, const float* Wg,
                           const float* bg, const float* gout, float* dz, float* dWg, float* dbg, float* dgamma, float* dbeta,
                           float* scratch, int B, int T, int F, int C, int PT, int PF, unsigned seed, unsigned thr24, float dscale,
                           const unsigned* seed_dev, int split_bf16, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (F % PF != 0) return SED_ERR_UNSUPPORTED;
    if (PT == 1 && PF == 2) {
        // C = 128, split-bf16: the 16x16x32 kernel (glu128_bwd_c_kernel; 87 us at B = 48, F = 16 against 174 us of the exact-f32
        // kernel and 204 us of the 32x32x16 tiling, whose two fragment sets of 128 VGPRs spill).
        // sed_set_tuning(SED_TUNE_GLU_BWD128_SPLIT, v) for A/B runs: 1 = the 32x32x16 split kernel, 3 = exact f32.
        if (C == 128 && split_bf16 && sed_tuning[SED_TUNE_GLU_BWD128_SPLIT] != 3)
            return launch_glu_wide_bwd<128, true>(y, stats, gamma, beta, Wg, bg, gout, dz, dWg, dbg, dgamma, dbeta, scratch, B, T, F, seed, thr24, dscale, seed_dev, s);
        if (C == 64 && split_bf16) return launch_glu_wide_bwd<64, true>(y, stats, gamma, beta, Wg, bg, gout, dz, dWg, dbg, dgamma, dbeta, scratch, B, T, F, seed, thr24, dscale, seed_dev, s);
        if (C == 128) return launch_glu_wide_bwd<128, false>(y, stats, gamma, beta, Wg, bg, gout, dz, dWg, dbg, dgamma, dbeta, scratch, B, T, F, seed, thr24, dscale, seed_dev, s);
        if (C == 64) return launch_glu_wide_bwd<64, false>(y, stats, gamma, beta, Wg, bg, gout, dz, dWg, dbg, dgamma, dbeta, scratch, B, T, F, seed, thr24, dscale, seed_dev, s);
    }
    const bool lds_free32 = C == 32 && PT == 2 && PF == 2 && F % 16 == 0 && T >= 2;      // glu32_bwd_kernel zeroes the dropped frame itself
    if (T % PT != 0 && !lds_free32) {
        // frames the floor-mode pooling drops get no gradient: zero just those rows (zeroing all of dz was a 123 MB memset
        // per step for block 1)
        const int tail = T % PT, rowf4 = F * C / 4;
        const long long n4 = (long long)B * tail * rowf4;
        int grid = (int)((n4 + 255) / 256);
        if (grid > 2048) grid = 2048;
        SED_LAUNCH(glu_zero_tail_kernel, dim3(grid), dim3(256), 0, s, dz, n4, T, T - tail, tail * rowf4, (long long)T * rowf4);
    }
    if ((C == 16 && PT == 2 && PF == 2 && F % 8 == 0) || (C == 32 && PT == 2 && PF == 2 && F % 16 == 0)) {
        // LDS-free MFMA kernels: one row of pooling windows per wave iteration, one partial per workgroup + fixed-order reduce
        const int nrows = B * (T / 2);
        int grid = (nrows + 3) / 4;
        const int cap = C == 32 ? glu_grid_cap(512) : 1024;   // glu32_bwd: 228 registers -> two 4-wave workgroups per CU are resident
        if (grid > cap) grid = cap;
        if (grid < 1) { sed_zero4(s, dWg, C * C, dbg, C, dgamma, C, dbeta, C); return SED_OK; }
        if (!scratch) return SED_ERR_ARG;
        if (C == 16)
            SED_LAUNCH(glu16_bwd_kernel, dim3(grid), dim3(256), 0, s, y, stats, gamma, beta, Wg, bg, gout, dz, scratch, B, T, F, seed, thr24,
                       dscale, seed_dev);
        else
            SED_LAUNCH(glu32_bwd_kernel, dim3(grid), dim3(256), 0, s, y, stats, gamma, beta, Wg, bg, gout, dz, scratch, B, T, F, seed, thr24,
                       dscale, seed_dev);
        SED_LAUNCH(glu_bwd_reduce_kernel, dim3((C * C + 3 * C + 63) / 64), dim3(GBR_THREADS), 0, s, scratch, dWg, dbg, dgamma, dbeta, grid, C, 1, 1,
                   gamma, beta, 0);
        return sed_check_launch();
    }
    sed_zero4(s, dWg, C * C, dbg, C, dgamma, C, dbeta, C);
#define GLU_CASE(c, pt, pf)                                                                                              \
    if (C == c && PT == pt && PF == pf)                                                                                  \
        return launch_glu_bwd<c, pt, pf>(y, stats, gamma, beta, Wg, bg, gout, dz, dWg, dbg, dgamma, dbeta, B, T, F, seed, thr24, \
                                         dscale, seed_dev, s);
    GLU_CASE(16, 2, 2) GLU_CASE(32, 2, 2)
#undef GLU_CASE
    return SED_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm backward apply, in place: dz (= dL/d xhat) -> dy = dL/d(conv output)
//   training: dy = invstd * (dz - mean(dz) - xhat * mean(dz * xhat));  eval: dy = invstd * dz
// with sum(dz) = gamma*dbeta and sum(dz*xhat) = gamma*dgamma.  Also emits the conv-bias gradient
// (= sum dy: analytically 0 in training mode, invstd*gamma*dbeta in eval mode).
// ---------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ y, float* __restrict__ dz,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                           float* __restrict__ dbias, size_t npix, float inv_count, int training) {
    constexpr int V = C / 4, RPI = 256 / V;
    const int tid = threadIdx.x, v = tid % V, r0 = tid / V;
    float mean[4], istd[4], m1[4], m2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 4 * v + i;
        mean[i] = stats[c];
        istd[i] = stats[C + c];
        m1[i] = training ? gamma[c] * dbeta[c] * inv_count : 0.f;
        m2[i] = training ? gamma[c] * dgamma[c] * inv_count : 0.f;
    }
    if (blockIdx.x == 0 && tid < C) dbias[tid] = training ? 0.f : stats[C + tid] * gamma[tid] * dbeta[tid];
    for (size_t p = (size_t)blockIdx.x * RPI + r0; p < npix; p += (size_t)gridDim.x * RPI) {
        const float4 yv = *(const float4*)(y + p * C + 4 * v);
        float4 g = *(const float4*)(dz + p * C + 4 * v);
        g.x = istd[0] * (g.x - m1[0] - (yv.x - mean[0]) * istd[0] * m2[0]);
        g.y = istd[1] * (g.y - m1[1] - (yv.y - mean[1]) * istd[1] * m2[1]);
        g.z = istd[2] * (g.z - m1[2] - (yv.z - mean[2]) * istd[2] * m2[2]);
        g.w = istd[3] * (g.w - m1[3] - (yv.w - mean[3]) * istd[3] * m2[3]);
        *(float4*)(dz + p * C + 4 * v) = g;
    }
}
SED_API int sed_bn_bwd_apply(const float* y, float* dz, const float* stats, const float* gamma, const float* dgamma,
                                const float* dbeta, float* dbias, long long npix, int C, int training, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (npix <= 0) return SED_OK;
    const float inv = 1.0f / (float)npix;
    int grid = (int)((npix * (C / 4) + 255) / 256);
    if (grid > 4096) grid = 4096;
#define BN_CASE(c) \
    if (C == c) { SED_LAUNCH((bn_bwd_apply_kernel<c>), dim3(grid), dim3(256), 0, s, y, dz, stats, gamma, dgamma, dbeta, dbias, (size_t)npix, inv, training); return sed_check_launch(); }
    BN_CASE(16) BN_CASE(32) BN_CASE(64) BN_CASE(128)
#undef BN_CASE
    return SED_ERR_UNSUPPORTED;
}
