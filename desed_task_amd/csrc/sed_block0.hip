// K6, first block, WITHOUT its pre-BatchNorm tensor in HBM.
//
// Block 0 of the CNN (desed_task/nnet/CNN.py:66-98 with n_in_channel = 1, 16 filters, pooling (2,2)) turns a 15 MB input
// (B,T,F) into a 246 MB conv output y (B,T,F,16) that the unfused path wrote once and read three times (BN+GLU forward, GLU
// backward, weight gradient) and whose gradient dz it wrote and read once more: ~2.2 GB of HBM traffic per step for a K = 9
// convolution that costs 36 FMAs per lane to recompute.  Here y only ever exists in registers:
//
//   forward  (training): sed_conv0_fwd(y = NULL)  conv0 in registers -> per-workgroup (sum, sum^2) for BatchNorm   [reads x]
//                        sed_bn_finalize          batch statistics, running-stat update (momentum 0.99, eps 1e-3)
//                        sed_block0_fwd           conv0 + BN + GLU + Dropout + AvgPool(2,2) -> pooled output        [reads x, writes out]
//   forward  (eval)    : sed_bn_finalize (running stats) + sed_block0_fwd
//   backward (training): sed_block0_bwd           recompute conv0/BN/GLU, GLU + BN backward, ALL six parameter gradients
//                                                 (conv weight/bias, BN gamma/beta, GLU weight/bias)               [reads x, gout]
//
// The conv weight gradient needs dy = invstd (dz - mean(dz) - xhat mean(dz xhat)), i.e. two batch-wide means that are only
// complete when the pass is over.  It is linear in dy, so the pass accumulates the three raw correlations
//      S1[c][tap] = sum_p x_tap(p) dz_c(p),   S2[c][tap] = sum_p x_tap(p) xhat_c(p),   Sx[tap] = sum_p x_tap(p)
// and a tiny second kernel forms  dW[c][tap] = invstd_c (S1 - m1_c Sx - m2_c S2),  m1 = gamma dbeta / N, m2 = gamma dgamma / N
// in double.  S1 and m1 Sx cancel down to N cov(x_tap, dz_c): to keep that difference out of fp32 rounding each workgroup
// accumulates against x - k (k = a local estimate of the input mean, any constant is algebraically exact) and the reduction
// adds k sum(dz_c), k sum(xhat_c), k n back in double.
//
// Lane layout = the glu16 kernels' (sed_glu.hip): one wave per 16-pixel tile = 4 pooling windows along F on the 16x16x4 f32
// MFMA; operand lane (i = pixel = 4 w + q, g = channel quad) computes its own four conv outputs (36 FMAs from a 16 x F LDS
// tile of the input, SpecAugment predicate applied while staging), which are directly the MFMA operand.
#include "sed_common.h"

#define B0_TR 16            // input rows per workgroup tile (8 pooled rows, two per wave)
#define B0_MAXF 128
// LDS row pitch of the halo tile.  A wave's tap read covers 8 consecutive columns of TWO consecutive tile rows (16 pixels = 4 pooling
// windows), and ds_read_b32 banks are (address / 4) mod 32 per group of 32 lanes: with the natural pitch F + 2 (== 2 mod 32 for every
// F the recipes use) six of the eight columns of the two rows shared a bank -- every tap read was a 2-way conflict
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.49 on block0_fwd_kernel, profiles/r03f_pmc_wait.md).  F + 8 is == 8 mod 32: disjoint.
#define B0_PITCH(F) ((F) + 8)

// stage the (B0_TR + 2) x (F + 2) halo tile of clip b, rows t0 - 1 .. t0 + B0_TR, minus `center` (zero padding and
// SpecAugment-masked bins hold 0 - center)
template <int BATCH>
__device__ __forceinline__ void b0_stage(float* tile, const float* __restrict__ x, const int* __restrict__ bounds, int b, int t0, int T,
                                         int F, float center) {
    // (forward only: the persistent backward kernel sits at 254 registers and spills 17 of them with the float4 path)
    if constexpr (BATCH >= 8) {
        if (sed_stage_halo_f4<B0_TR, 256, B0_MAXF>(tile, x, bounds, b, t0, T, F, B0_PITCH(F), center)) return;
    }
    const int PW = F + 2, PT = B0_PITCH(F), n = (B0_TR + 2) * PW;
    int mf0 = 0, mf1 = 0, mt0 = 0, mt1 = 0;
    if (bounds) { mf0 = bounds[4 * b]; mf1 = bounds[4 * b + 1]; mt0 = bounds[4 * b + 2]; mt1 = bounds[4 * b + 3]; }
    // every load is unconditional (clamped address, the predicate is applied to the value) and all of a thread's loads are issued
    // before the first store: with the load inside the bounds branch the compiler waited for each one right behind it -- ten
    // exposed memory latencies per tile (tools/isa_exposed_loads.py)
    // (BATCH loads in flight per thread: the persistent backward kernels have few registers to spare)
    // (round 4: a division-free (row, bin) thread map for this scalar path was built for the backward -- 274 -> ~ 35 instructions per
    //  batch -- and measured 191.8 vs 188.4 us per launch: at 254 registers the kernel spilled 4 - 7 of them for it; kept as it was)
    for (int base = threadIdx.x; base < n; base += 256 * BATCH) {
        float v[BATCH];
        bool ok[BATCH];
        int pad[BATCH];                                         // row * (pitch - width): where the element lands in the padded tile
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int idx = base + 256 * u, ic = idx < n ? idx : n - 1;
            const int i = ic / PW, j = ic - i * PW;
            const int t = t0 - 1 + i, f = j - 1;
            const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t), fc = f < 0 ? 0 : (f >= F ? F - 1 : f);
            v[u] = x[((size_t)b * T + tc) * F + fc];
            pad[u] = i * (PT - PW);
            ok[u] = t >= 0 && t < T && f >= 0 && f < F && !((f >= mf0 && f < mf1) || (t >= mt0 && t < mt1));
        }
        // all BATCH loads are issued before the first value is touched: without the fence the scheduler pairs every load with its
        // use again, and without the pins the optimiser sinks each load back under its bounds predicate (a branch + a wait per load)
        sed_sched_fence();
#pragma unroll
        for (int u = 0; u < BATCH; ++u) sed_pin(v[u]);
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int idx = base + 256 * u;
            if (idx < n) tile[idx + pad[u]] = (ok[u] ? v[u] : 0.f) - center;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward: x (B,T,F) -> out (B,T/2,F/2,16).  grid = (ceil(T/16), B), 256 threads.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void block0_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, const int* __restrict__ bounds,
                                                         const float* __restrict__ stats, const float* __restrict__ Wg,
                                                         const float* __restrict__ bg, float* __restrict__ out, int B, int T, int F,
                                                         uint32_t seed, uint32_t thr24, float dscale,
                                                         const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;            // per-step entropy in device memory (hipGraph replays)
    constexpr int C = 16;
    __shared__ float tile[(B0_TR + 2) * B0_PITCH(B0_MAXF)];
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const int b = blockIdx.y, t0 = blockIdx.x * B0_TR, PW = B0_PITCH(F);
    float wreg[4][9], breg[4], wa[4], sc[4], sh[4], bgr[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        breg[c] = bias ? bias[4 * g + c] : 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) wreg[c][k] = W[(4 * g + c) * 9 + k];
        wa[c] = Wg[i * C + 4 * g + c];
        sc[c] = stats[2 * C + 4 * g + c];
        sh[c] = stats[3 * C + 4 * g + c];
        bgr[c] = bg[4 * g + c];
    }
    b0_stage<10>(tile, x, bounds, b, t0, T, F, 0.f);
    __syncthreads();
    const int To = T / 2, Fo = F / 2, tpr = Fo / 4;
    const int w = i >> 2, q = i & 3;
    for (int pr = wv; pr < B0_TR / 2; pr += 4) {
        const int to = (t0 >> 1) + pr;
        if (to >= To) break;
        const int lr = 2 * pr + (q >> 1);                       // tile-local row of this lane's pixel
        for (int tr = 0; tr < tpr; ++tr) {
            const int col = 2 * (4 * tr + w) + (q & 1);
            // (the K = 9 convolution as three 16x16x4 f32 MFMAs -- taps 0-3 | 4-7 | 8, operands straight from the tile, 3 LDS reads
            //  instead of 9 -- was built and measured in round 3: 88.8 vs 85.9 us here, 47.6 vs 41.2 us in the statistics pass: a
            //  dependent chain of three 40-cycle matrix instructions in front of the four of the gate is longer than 18 packed FMAs)
            float in[9];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int bb = 0; bb < 3; ++bb) in[a * 3 + bb] = tile[(lr + a) * PW + col + bb];
            float xn[4];
#pragma unroll
            for (int c = 0; c < 4; c += 2) {                    // two channels per v_pk_fma_f32 (each half is the same fmaf chain)
                f32x2 acc2 = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 9; ++k) acc2 = pk_fma(f32x2{in[k], in[k]}, f32x2{wreg[c][k], wreg[c + 1][k]}, acc2);
                const float a0 = acc2.x + breg[c], a1 = acc2.y + breg[c + 1];   // == conv0_kernel's y, bit for bit (same operation order)
                xn[c] = fmaf(a0, sc[c], sh[c]);
                xn[c + 1] = fmaf(a1, sc[c + 1], sh[c + 1]);
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = mfma16(wa[k], xn[k], acc);      // D[n = 4g+r][pixel i]
            const size_t pix = ((size_t)b * T + t0 + lr) * F + col;
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
}

SED_API int sed_block0_fwd(const float* x, const float* W, const float* bias, const int* bounds, const float* stats,
                              const float* Wg, const float* bg, float* out, int B, int T, int F, unsigned seed, unsigned thr24,
                              float dscale, const unsigned* seed_dev, void* stream) {
    if (F > B0_MAXF || F < 8 || F % 8 != 0) return SED_ERR_UNSUPPORTED;
    if (B <= 0 || T < 2) return SED_OK;
    SED_LAUNCH(block0_fwd_kernel, dim3((T + B0_TR - 1) / B0_TR, B), dim3(256), 0, (hipStream_t)stream, x, W, bias, bounds, stats, Wg,
               bg, out, B, T, F, seed, thr24, dscale, seed_dev);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------
// backward (training-mode BatchNorm).  Persistent workgroups over the (clip, 16-row tile) pairs; per workgroup ONE
// partial record, reduced in a fixed order (double) by block0_bwd_reduce_kernel.
// ---------------------------------------------------------------------------------------------
// partial record (floats): P = dWg (256) | dbg (16) | dgamma (16) | dbeta (16) | S1 (16 x 9) | S2 (16 x 9) | sum xhat (16) |
//                          Sx (9) | k | pixel count | pad
#define B0_O_DBG 256
#define B0_O_DGAM 272
#define B0_O_DBET 288
#define B0_O_S1 304
#define B0_O_S2 448
#define B0_O_XH 592
#define B0_O_SX 608
#define B0_O_K 617
#define B0_O_CNT 618
#define B0_NP 624
#define B0_NSUM 617         // entries [0, 617) are sums over the workgroup's pixels

#ifndef B0_BWD_BATCH
#define B0_BWD_BATCH 4
#endif
#define B0_TS 20            // row pitch (floats) of the wave-private 16 x 16 transposition buffers: 16-byte rows, and the
                            // strided reads of a lane group (rows 4g + kk, column i) fall into 16 distinct banks per group
__global__ __launch_bounds__(256, 2) void block0_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, const int* __restrict__ bounds,
                                                         const float* __restrict__ stats, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ Wg,
                                                         const float* __restrict__ bg, const float* __restrict__ gout,
                                                         float* __restrict__ part, int B, int T, int F, int tiles_t, uint32_t seed,
                                                         uint32_t thr24, float dscale, const unsigned* __restrict__ seed_dev,
                                                         int center) {
    if (seed_dev) seed += *seed_dev;
    constexpr int C = 16;
    __shared__ float tile[(B0_TR + 2) * B0_PITCH(B0_MAXF)];
    __shared__ float red[4][B0_NP];
    __shared__ __attribute__((aligned(16))) float tbuf[4][2][16 * B0_TS];
    __shared__ float kred[4];
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const int PW = B0_PITCH(F), ntiles = B * tiles_t;
    const int To = T / 2, Fo = F / 2, tpr = Fo / 4;
    // ---- centring constant: mean of a sample of this workgroup's first tile (any value is exact; a good one keeps S1 small) ----
    float k;
    {
        const int tl = blockIdx.x, b = tl / tiles_t, t0 = (tl - b * tiles_t) * B0_TR;
        float s = 0.f;
        int n = 0;
#pragma unroll
        for (int u = 0; u < B0_TR * B0_MAXF / 512; ++u) {                     // (unconditional, independent loads: see b0_stage)
            const int idx = threadIdx.x + 512 * u, ic = idx < B0_TR * F ? idx : 0;
            const int r = ic / F, f = ic - r * F, t = t0 + r;
            const float xv = x[((size_t)b * T + (t < T ? t : T - 1)) * F + f];
            if (idx < B0_TR * F && t < T) { s += xv; ++n; }
        }
        float nf = (float)n;
        s = wave_sum(s); nf = wave_sum(nf);
        if (lane == 0) { kred[wv] = s; red[wv][0] = nf; }
        __syncthreads();
        // (sed_sadd: left to itself the compiler paired the two sums and finished them with v_pk_add_f32 ... op_sel:[0,1] -- the form
        //  sed_common.h's "gfx950 hazard" note forbids; this kernel runs beside block 1's split-bf16 weight gradient)
        const float tot = sed_sadd(sed_sadd(kred[0], kred[1]), sed_sadd(kred[2], kred[3]));
        const float cnt = sed_sadd(sed_sadd(red[0][0], red[1][0]), sed_sadd(red[2][0], red[3][0]));
        k = (center && cnt > 0.f) ? tot / cnt : 0.f;
        __syncthreads();
    }
    // Everything up to the weight gradient stays in the OPERAND layout -- lane = (pixel i = 4 w + q, channel quad g), registers =
    // the quad's four channels: conv0, BN, GEMM1 (lin^T = Wg xn^T), the gate epilogue, GEMM2 computed transposed (dxn^T = Wg^T dlin^T,
    // so that it lands on the lane that holds xhat and e for the same pixel and channels), the BN reductions and the raw
    // correlations S1 / S2 against the nine taps this lane already loaded for its convolution.  Only GEMM3 (dWg += dlin^T xn)
    // contracts over pixels and needs channel-indexed lanes: dlin and xn go through a wave-private 16 x 16 LDS transposition.
    float wreg[4][9], breg[4], wa1[4], wb2[4], mu[4], istd[4], gam4[4], bet4[4], bgr[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = 4 * g + c;
        float ws = 0.f;
#pragma unroll
        for (int kk = 0; kk < 9; ++kk) { wreg[c][kk] = W[ch * 9 + kk]; ws += wreg[c][kk]; }
        breg[c] = (bias ? bias[ch] : 0.f) + k * ws;             // conv over (x - k) + k sum(w) = conv over x
        wa1[c] = Wg[i * C + ch];                                // GEMM1:  A[n = i][c = 4g + kk]
        wb2[c] = Wg[ch * C + i];                                // GEMM2^T: A[c = i][n = 4g + kk]
        mu[c] = stats[ch]; istd[c] = stats[C + ch];
        gam4[c] = gamma[ch]; bet4[c] = beta[ch];
        bgr[c] = bg[ch];
    }
    // the per-lane constants are complete before the tile loop (sed_pin: otherwise their first use in every iteration carries a
    // conservative vmcnt wait that drains the upstream-gradient prefetch issued a few instructions earlier)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int kk = 0; kk < 9; ++kk) sed_pin(wreg[c][kk]);
        sed_pin(breg[c]); sed_pin(wa1[c]); sed_pin(wb2[c]); sed_pin(mu[c]); sed_pin(istd[c]); sed_pin(gam4[c]); sed_pin(bet4[c]); sed_pin(bgr[c]);
    }
    f32x4 P = {0.f, 0.f, 0.f, 0.f};
    float a_dgam[4], a_dbet[4], a_dbg[4], a_xh[4], a_sx[9], a_cnt = 0.f;
    f32x2 S12[4][9];                                    // {S1, S2} pairs: one v_pk_fma_f32 per (channel, tap) and pixel
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        a_dgam[c] = 0.f; a_dbet[c] = 0.f; a_dbg[c] = 0.f; a_xh[c] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 9; ++kk) S12[c][kk] = f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int kk = 0; kk < 9; ++kk) a_sx[kk] = 0.f;
    const int w = i >> 2, q = i & 3;
    float* t1 = tbuf[wv][0];
    float* t2 = tbuf[wv][1];
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int b = tl / tiles_t, t0 = (tl - b * tiles_t) * B0_TR;
        __syncthreads();                                          // previous tile fully consumed
        b0_stage<B0_BWD_BATCH>(tile, x, bounds, b, t0, T, F, k);
        __syncthreads();
        for (int pr = wv; pr < B0_TR / 2; pr += 4) {
            const int to = (t0 >> 1) + pr;
            // T odd: floor-mode pooling drops frame T - 1.  It gets no gradient from above (dz = 0) but its dy is not zero -- the
            // two batch means of the BatchNorm backward reach every pixel -- so it still enters Sx, S2 and sum(xhat).  It is
            // walked as the top row of a window whose bottom row (frame T, outside the clip) is masked out.
            const bool tail = (to == To) && (T & 1);
            if (to >= To && !tail) break;
            const int lr = 2 * pr + (q >> 1);
            const float vm = (tail && (q >> 1)) ? 0.f : 1.f;      // this lane's pixel exists
            const float gsc = tail ? 0.f : 0.25f * dscale;
            const float* grow = gout + (((size_t)b * To + (tail ? To - 1 : to)) * Fo + w) * C + 4 * g;
            float4 go_n = *(const float4*)grow;
            for (int tr = 0; tr < tpr; ++tr) {
                const float4 go = go_n;
                go_n = *(const float4*)(grow + (size_t)(4 * C) * (tr + 1 < tpr ? tr + 1 : tr));
                const int col = 2 * (4 * tr + w) + (q & 1);
                float in[9];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int bb = 0; bb < 3; ++bb) in[a * 3 + bb] = tile[(lr + a) * PW + col + bb];
                float xh[4], xn[4];
#pragma unroll
                for (int c = 0; c < 4; c += 2) {                // two channels per v_pk_fma_f32
                    f32x2 acc2 = {0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < 9; ++kk) acc2 = pk_fma(f32x2{in[kk], in[kk]}, f32x2{wreg[c][kk], wreg[c + 1][kk]}, acc2);
                    const float a0 = acc2.x + breg[c], a1 = acc2.y + breg[c + 1];
                    xh[c] = (a0 - mu[c]) * istd[c] * vm;
                    xh[c + 1] = (a1 - mu[c + 1]) * istd[c + 1] * vm;
                    xn[c] = fmaf(xh[c], gam4[c], bet4[c]);
                    xn[c + 1] = fmaf(xh[c + 1], gam4[c + 1], bet4[c + 1]);
                }
                f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc1 = mfma16(wa1[kk], xn[kk], acc1);   // lin^T: D[n = 4g+r][pixel i]
                const float gv[4] = {go.x, go.y, go.z, go.w};
                const size_t pix = ((size_t)b * T + t0 + lr) * F + col;
                float dlin[4], e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float lin = acc1[r] + bgr[r];
                    const float sg = sed_fast_sigmoid(xn[r]);
                    const uint32_t ei = (uint32_t)(pix * C + 4 * g + r);
                    const float gr = sed_keep(ei, seed, thr24) ? gv[r] * gsc : 0.f;
                    dlin[r] = gr * sg;
                    e[r] = gr * lin * sg * (1.0f - sg);
                }
                // wave-private transposition of dlin and xn for GEMM3 (written now, read after GEMM2)
                sed_wave_sync();                                  // the previous iteration's reads are done
                *(float4*)(t1 + i * B0_TS + 4 * g) = make_float4(dlin[0], dlin[1], dlin[2], dlin[3]);
                *(float4*)(t2 + i * B0_TS + 4 * g) = make_float4(xn[0], xn[1], xn[2], xn[3]);
                f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc2 = mfma16(wb2[kk], dlin[kk], acc2);  // dxn^T: D[c = 4g+r][pixel i]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dxn = acc2[r] + e[r];
                    a_dgam[r] = fmaf(dxn, xh[r], a_dgam[r]);
                    a_dbet[r] += dxn;
                    a_dbg[r] += dlin[r];
                    a_xh[r] += xh[r];
                    const float dzr = dxn * gam4[r];              // dL/d xhat of (pixel i, channel 4g + r)
#pragma unroll
                    for (int kk = 0; kk < 9; ++kk) {
                        S12[r][kk] = pk_fma(f32x2{in[kk], in[kk]}, f32x2{dzr, xh[r]}, S12[r][kk]);
                    }
                }
#pragma unroll
                for (int kk = 0; kk < 9; ++kk) a_sx[kk] = fmaf(in[kk], vm, a_sx[kk]);
                a_cnt += vm;
                sed_wave_sync();
                float dT[4], xT[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) { dT[kk] = t1[(4 * g + kk) * B0_TS + i]; xT[kk] = t2[(4 * g + kk) * B0_TS + i]; }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) P = mfma16(dT[kk], xT[kk], P);   // P[n' = 4g+r][c = i] += dlin[p][n'] xn[p][c]
            }
        }
    }
    // ---- per-workgroup partial record: sum the 16 pixel lanes of every channel quad, then the four waves ----
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv][(4 * g + r) * C + i] = P[r];
#pragma unroll
    for (int m = 1; m <= 8; m <<= 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            a_dgam[c] += __shfl_xor(a_dgam[c], m); a_dbet[c] += __shfl_xor(a_dbet[c], m);
            a_dbg[c] += __shfl_xor(a_dbg[c], m); a_xh[c] += __shfl_xor(a_xh[c], m);
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) { S12[c][kk].x += __shfl_xor(S12[c][kk].x, m); S12[c][kk].y += __shfl_xor(S12[c][kk].y, m); }
        }
#pragma unroll
        for (int kk = 0; kk < 9; ++kk) a_sx[kk] += __shfl_xor(a_sx[kk], m);
        a_cnt += __shfl_xor(a_cnt, m);
    }
    if (i == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ch = 4 * g + c;
            red[wv][B0_O_DBG + ch] = a_dbg[c]; red[wv][B0_O_DGAM + ch] = a_dgam[c]; red[wv][B0_O_DBET + ch] = a_dbet[c];
            red[wv][B0_O_XH + ch] = a_xh[c];
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) { red[wv][B0_O_S1 + ch * 9 + kk] = S12[c][kk].x; red[wv][B0_O_S2 + ch * 9 + kk] = S12[c][kk].y; }
        }
        if (g == 0) {                                             // every channel quad saw the same pixels: count them once
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) red[wv][B0_O_SX + kk] = a_sx[kk];
            red[wv][B0_O_CNT] = a_cnt; red[wv][B0_O_K] = 0.f;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < B0_NP; idx += 256) {
        float v = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
        if (idx == B0_O_K) v = k;
        if (idx > B0_O_CNT) v = 0.f;
        part[(size_t)blockIdx.x * B0_NP + idx] = v;
    }
}

// sums[o] (double) = sum over the partial records, un-centred: S1 += k gamma_c dbeta_c, S2 += k sum(xhat_c), Sx += k n.
// One workgroup per 32 outputs, 32 walkers per output (the launch is a latency chain of dependent loads: keep the walks short);
// fixed summation order.
__global__ __launch_bounds__(1024) void block0_bwd_reduce_kernel(const float* __restrict__ part, int nparts,
                                                                 const float* __restrict__ gamma, double* __restrict__ sums) {
    __shared__ double red[32][33];
    const int tid = threadIdx.x, col = tid & 31, grp = tid >> 5, o = blockIdx.x * 32 + col;
    double acc = 0.0;
    if (o < B0_NP) {
        int o2 = -1;                                            // the companion column multiplied by k
        float scale = 1.0f;
        if (o >= B0_O_S1 && o < B0_O_S2) { const int c = (o - B0_O_S1) / 9; o2 = B0_O_DBET + c; scale = gamma[c]; }
        else if (o >= B0_O_S2 && o < B0_O_XH) { const int c = (o - B0_O_S2) / 9; o2 = B0_O_XH + c; }
        else if (o >= B0_O_SX && o < B0_O_K) o2 = B0_O_CNT;
        if (o2 >= 0) {
#pragma unroll 4
            for (int p = grp; p < nparts; p += 32) {
                const float* rec = part + (size_t)p * B0_NP;
                acc += (double)rec[o] + (double)rec[B0_O_K] * (double)scale * (double)rec[o2];
            }
        } else {
#pragma unroll 4
            for (int p = grp; p < nparts; p += 32) acc += (double)part[(size_t)p * B0_NP + o];
        }
    }
    red[grp][col] = acc;
    __syncthreads();
    if (grp == 0 && o < B0_NP) {
#pragma unroll
        for (int gq = 1; gq < 32; ++gq) acc += red[gq][col];
        sums[o] = acc;
    }
}

// the six parameter gradients from the reduced sums
__global__ __launch_bounds__(256) void block0_bwd_final_kernel(const double* __restrict__ sums, const float* __restrict__ stats,
                                                               const float* __restrict__ gamma, float* __restrict__ dW,
                                                               float* __restrict__ dbias, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ dWg,
                                                               float* __restrict__ dbg, double count) {
    constexpr int C = 16;
    const int tid = threadIdx.x;
    dWg[tid] = (float)sums[tid];                                  // 256 threads = the 16 x 16 GLU weight gradient
    if (tid < C) {
        dbg[tid] = (float)sums[B0_O_DBG + tid];
        dgamma[tid] = (float)sums[B0_O_DGAM + tid];
        dbeta[tid] = (float)sums[B0_O_DBET + tid];
        if (dbias) dbias[tid] = 0.f;                              // analytically zero under training-mode BatchNorm
    }
    if (tid < C * 9) {
        const int c = tid / 9, tap = tid - 9 * c;
        const double g = (double)gamma[c], is = (double)stats[C + c];
        const double m1 = g * sums[B0_O_DBET + c] / count, m2 = g * sums[B0_O_DGAM + c] / count;
        dW[tid] = (float)(is * (sums[B0_O_S1 + tid] - m1 * sums[B0_O_SX + tap] - m2 * sums[B0_O_S2 + tid]));
    }
}

static inline int block0_bwd_grid(int B, int T) {
    const int ntiles = B * ((T + B0_TR - 1) / B0_TR);
    const int forced = sed_tuning[SED_TUNE_GLU_GRID_CAP];                 // tests: several tiles per workgroup at toy sizes
    const int cap = forced > 0 ? forced : 512;                            // two workgroups per CU are resident (register-bound)
    return ntiles < cap ? ntiles : cap;
}
// floats of scratch: one partial record per workgroup + the reduced sums (doubles)
SED_API long long sed_block0_bwd_scratch_floats(int B, int T, int F) {
    (void)F;
    return (long long)block0_bwd_grid(B, T) * B0_NP + 2 * B0_NP + 2;
}

SED_API int sed_block0_bwd(const float* x, const float* W, const float* bias, const int* bounds, const float* stats,
                              const float* gamma, const float* beta, const float* Wg, const float* bg, const float* gout,
                              float* dW, float* dbias, float* dgamma, float* dbeta, float* dWg, float* dbg, float* scratch, int B,
                              int T, int F, unsigned seed, unsigned thr24, float dscale, const unsigned* seed_dev, void* stream) {
    if (F > B0_MAXF || F < 8 || F % 8 != 0) return SED_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int grid = block0_bwd_grid(B, T);
    if (grid < 1 || T < 2) {
        sed_zero4(s, dWg, 256, dW, 144, dgamma, 16, dbeta, 16);
        sed_zero4(s, dbg, 16, dbias, dbias ? 16 : 0, nullptr, 0, nullptr, 0);
        return sed_check_launch();
    }
    if (!scratch) return SED_ERR_ARG;
    const int tiles_t = (T + B0_TR - 1) / B0_TR;
    float* part = scratch;
    size_t off = (size_t)grid * B0_NP;
    off += off & 1;                                               // 8-byte alignment of the doubles
    double* sums = (double*)(scratch + off);
    SED_LAUNCH(block0_bwd_kernel, dim3(grid), dim3(256), 0, s, x, W, bias, bounds, stats, gamma, beta, Wg, bg, gout, part, B, T, F,
               tiles_t, seed, thr24, dscale, seed_dev, sed_tuning[SED_TUNE_B0_NOCENTER] ? 0 : 1);
    SED_LAUNCH(block0_bwd_reduce_kernel, dim3((B0_NP + 31) / 32), dim3(1024), 0, s, (const float*)part, grid, gamma, sums);
    SED_LAUNCH(block0_bwd_final_kernel, dim3(1), dim3(256), 0, s, (const double*)sums, stats, gamma, dW, dbias, dgamma, dbeta, dWg, dbg,
               (double)B * (double)T * (double)F);
    return sed_check_launch();
}
