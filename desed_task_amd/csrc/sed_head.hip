// K8 + K9: attention-pooling head and the mean-teacher losses, fp32.
//  head : desed_task/nnet/CRNN.py:152-178 -- Dropout (CRNN.py:304) -> strong = sigmoid(dense(x)),
//         sof = clamp(softmax_over_CLASSES(dense_softmax(x)), 1e-7, 1), weak = sum_t(strong*sof) / sum_t(sof).
//  loss : recipes/dcase2023_task4_baseline/local/sed_trainer.py:309-342 -- BCE (strong on the first n_strong
//         clips, weak on the next n_weak), the two teacher BCEs (logging), MSE student-vs-teacher on all clips;
//         emits the scalars and the gradient seeds d(total)/d(strong_s), d(total)/d(weak_s).
// Layout: x (B,T,D), D = 2 * n_RNN_cell = 256 (2023 recipe) or 384 (2024 recipe); strong/sof (B,T,NC) (the Python side returns the (B,NC,T) transposed view);
// labels stay in the reference layout (B,NC,T).
#include "sed_common.h"


// Forward: one workgroup per clip (the attention pooling sums over the clip's frames), FOUR lanes per frame: lane
// quarter q owns the input features k with (k / 4) % 4 == q, so the four lanes of a frame read 64 contiguous bytes of
// x and adjacent LDS banks of the weights; the 2*NC partial logits are combined with two xor-shuffles.
#define HEAD_THREADS 1024
template <int NC, int D>
__global__ __launch_bounds__(HEAD_THREADS) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W1,
                                                                const float* __restrict__ b1, const float* __restrict__ W2,
                                                                const float* __restrict__ b2, float* __restrict__ strong,
                                                                float* __restrict__ psoft, float* __restrict__ weak,
                                                                float* __restrict__ den, int T, uint32_t seed, uint32_t thr24,
                                                                float dscale, const unsigned* __restrict__ seed_dev,
                                                                const unsigned char* __restrict__ cvalid,
                                                                const unsigned char* __restrict__ pad) {
    if (seed_dev) seed += *seed_dev;            // per-step entropy in device memory (hipGraph replays)
    static_assert(NC <= 32, "the class mask of a frame is one 32-bit set");
    constexpr int FPP = HEAD_THREADS / 4;                      // frames per pass
    SED_DYN_SMEM(smem_w);                                      // (27 classes x 384 features x 2 matrices = 83 KB: above the static limit)
    float* w1 = (float*)smem_w;                                // [NC][D]
    float* w2 = w1 + NC * D;                                   // [NC][D]
    __shared__ float red[HEAD_THREADS / 64][2 * NC];
    const int tid = threadIdx.x, b = blockIdx.x, q = tid & 3;
    for (int i = tid; i < NC * D; i += HEAD_THREADS) { w1[i] = W1[i]; w2[i] = W2[i]; }
    __syncthreads();
    float num[NC], dn[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { num[c] = 0.f; dn[c] = 0.f; }
    for (int t0 = 0; t0 < T; t0 += FPP) {                      // uniform trip count: the shuffles below need whole quads
        const int t = t0 + (tid >> 2);
        const bool live = t < T;
        const float* xr = x + ((size_t)b * T + (live ? t : 0)) * D;
        float l1[NC], l2[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { l1[c] = 0.f; l2[c] = 0.f; }
        if (live) {
#pragma unroll 4
            for (int i = 0; i < D / 16; ++i) {
                const int k = 16 * i + 4 * q;
                float4 v = *(const float4*)(xr + k);
                const uint32_t e = (uint32_t)(((size_t)b * T + t) * D + k);
                v.x = sed_keep(e, seed, thr24) ? v.x * dscale : 0.f;
                v.y = sed_keep(e + 1, seed, thr24) ? v.y * dscale : 0.f;
                v.z = sed_keep(e + 2, seed, thr24) ? v.z * dscale : 0.f;
                v.w = sed_keep(e + 3, seed, thr24) ? v.w * dscale : 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float4 a = *(const float4*)(w1 + c * D + k);
                    const float4 g = *(const float4*)(w2 + c * D + k);
                    l1[c] = fmaf(v.x, a.x, fmaf(v.y, a.y, fmaf(v.z, a.z, fmaf(v.w, a.w, l1[c]))));
                    l2[c] = fmaf(v.x, g.x, fmaf(v.y, g.y, fmaf(v.z, g.z, fmaf(v.w, g.w, l2[c]))));
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            l1[c] = sed_quad_sum(l1[c]);
            l2[c] = sed_quad_sum(l2[c]);
        }
        if (live && q == 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) { l1[c] += b1[c]; l2[c] += b2[c]; }
            // CRNN.py:160-166: padded frames and the classes a clip's data set does not annotate cannot be attended to
            // (masked_fill(-1e30) before the class softmax; a fully masked frame therefore attends uniformly)
            // (the masks are gathered into one bit set first and applied with selects: the branchy form -- a conditional store
            // per class behind `padded || (cvalid && !cvalid[...])` -- lost the fill in hipcc 7.2's code for gfx950)
            unsigned inval = 0u;                                    // classes outside the clip's data set
            if (cvalid) {
#pragma unroll
                for (int c = 0; c < NC; ++c) inval |= (cvalid[b * NC + c] ? 0u : 1u) << c;
            }
            const unsigned gone = (pad && pad[(size_t)b * T + t]) ? 0xFFFFFFFFu : inval;
#pragma unroll
            for (int c = 0; c < NC; ++c) l2[c] = ((gone >> c) & 1u) ? -1e30f : l2[c];
            float mx = l2[0];
#pragma unroll
            for (int c = 1; c < NC; ++c) mx = fmaxf(mx, l2[c]);
            float se = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) { l2[c] = expf(l2[c] - mx); se += l2[c]; }
            const float inv = 1.0f / se;
            float* so = strong + ((size_t)b * T + t) * NC;
            float* po = psoft + ((size_t)b * T + t) * NC;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float s = sed_sigmoid(l1[c]);
                const float p = l2[c] * inv;
                const float a = fminf(fmaxf(p, 1e-7f), 1.0f);
                so[c] = ((inval >> c) & 1u) ? 0.f : s;                    // CRNN.py:173-175 (after the pooling below)
                po[c] = p;
                num[c] += s * a;
                dn[c] += a;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float a = wave_sum(num[c]), g = wave_sum(dn[c]);
        if ((tid & 63) == 0) { red[tid >> 6][c] = a; red[tid >> 6][NC + c] = g; }
    }
    __syncthreads();
    if (tid < NC) {
        float n = 0.f, d = 0.f;
#pragma unroll
        for (int w = 0; w < HEAD_THREADS / 64; ++w) { n += red[w][tid]; d += red[w][NC + tid]; }
        weak[b * NC + tid] = (cvalid && !cvalid[b * NC + tid]) ? 0.f : n / d;
        den[b * NC + tid] = d;
    }
}

// backward: d_strong (B,T,NC), d_weak (B,NC) -> dx (B,T,D), dW1,dW2 (NC,D), db1,db2 (NC).  Every workgroup writes ONE partial record
// (dW1 | dW2 | db1 | db2) and head_bwd_reduce_kernel sums the records in a fixed order (round 3: the 144 workgroups x 5 120 float
// atomics on the same 5 120 addresses were most of the launch's 62 us, and made it irreproducible).
// Grid = (clip, frame slice of HEAD_TS frames): nothing in the backward couples frames, so the slices fill the chip.
// Waves 0-3: HEAD_BL lanes per frame (logit gradients by all of them, then this lane's share of the dx row);
// waves 4-7: thread k owns input feature k over the slice's frames (weight gradients).  The two halves only share the logit
// gradients (dl, behind one barrier) and ran one after the other on the same four waves until round 4: a slice is one wave per SIMD
// of latency-bound work either way, so the second set of waves is free.
#define HEAD_TS 32                // frames per slice: 256 frame threads = EIGHT lanes per frame (round 4: 64 frames x 4 lanes -- a slice is a
                                  // latency chain per frame, so twice the lanes on half the features each and 240 instead of 144 workgroups)
#define HEAD_BL 8                 // lanes per frame in the backward
// (27 classes: 456 registers per thread -- one set of four waves does both halves in turn, as before)
#define HEAD_BWD_THREADS(NC) ((NC) <= 10 ? 512 : 256)
template <int NC, int D>
__global__ __launch_bounds__(HEAD_BWD_THREADS(NC)) void head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W1,
                                                       const float* __restrict__ W2, const float* __restrict__ strong,
                                                       const float* __restrict__ psoft, const float* __restrict__ weak,
                                                       const float* __restrict__ den, const float* __restrict__ d_strong,
                                                       const float* __restrict__ d_weak, float* __restrict__ dx,
                                                       float* __restrict__ part, int T, uint32_t seed, uint32_t thr24, float dscale, const unsigned* __restrict__ seed_dev,
                                                       const unsigned char* __restrict__ cvalid, const unsigned char* __restrict__ pad) {
    if (seed_dev) seed += *seed_dev;            // per-step entropy in device memory (hipGraph replays)
    constexpr int TS = HEAD_TS, NP = 2 * NC * D + 2 * NC;
    float* mine = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NP;      // this workgroup's record: dW1 | dW2 | db1 | db2
    SED_DYN_SMEM(smem);
    float* w1 = (float*)smem;            // NC*D
    float* w2 = w1 + NC * D;             // NC*D
    float* dl = w2 + NC * D;             // TS * 2*NC: (d logit1 | d logit2) per frame of the slice
    const int tid = threadIdx.x, b = blockIdx.y, tbeg = blockIdx.x * TS, tn = min(TS, T - tbeg), q = tid & (HEAD_BL - 1);
    static_assert(HEAD_TS * HEAD_BL == 256 && D % (4 * HEAD_BL) == 0, "256 frame threads: HEAD_BL lanes on interleaved feature quads per frame");
    constexpr int NTH = HEAD_BWD_THREADS(NC);
    constexpr bool SPLIT = NTH == 512;
    const bool rows = !SPLIT || tid < 256;                        // waves 0-3: frames; waves 4-7: input features
    const int ft = SPLIT ? tid - 256 : tid;                       // feature thread index
    for (int i = tid; i < NC * D; i += NTH) { w1[i] = W1[i]; w2[i] = W2[i]; }
    for (int i = tid; i < TS * 2 * NC; i += NTH) dl[i] = 0.f;
    __syncthreads();
    // ---- per-frame logit gradients (TS == 256 / HEAD_BL frames in one pass) ----
    float g1[NC], g2[NC];
    const int tl = (tid & 255) / HEAD_BL, t = tbeg + tl;
    const size_t bt = (size_t)b * T + t;
    // the feature threads fetch their first eight frames of x while the frame threads work out the logit gradients
    constexpr int KPT = D / 256 + (D % 256 ? 1 : 0);             // input features per feature thread (k, k + 256)
    float v[KPT][8];
    auto first_round = [&]() {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int k = ft + 256 * j;
#pragma unroll
            for (int u = 0; u < 8; ++u) v[j][u] = k < D ? x[((size_t)b * T + tbeg + (u < tn ? u : tn - 1)) * D + k] : 0.f;
        }
    };
    if (SPLIT && !rows) first_round();
    if (rows && tl < tn) {
        {
            float dot = 0.f;
            const bool padded = pad && pad[bt];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                // masked outputs (strong and weak of an invalid class are constants 0) pass no gradient; the stored `strong`
                // of such a class is the masked 0, its sigmoid is not needed: every term below carries a zero factor
                const bool ok = !(cvalid && !cvalid[b * NC + c]);
                const float s = strong[bt * NC + c], p = psoft[bt * NC + c];
                const float a = fminf(fmaxf(p, 1e-7f), 1.0f);
                const float dw = ok ? d_weak[b * NC + c] : 0.f, dd = den[b * NC + c], wk = weak[b * NC + c];
                const float ds = (ok ? d_strong[bt * NC + c] : 0.f) + dw * a / dd;
                g1[c] = ds * s * (1.0f - s);
                const float da = dw * (s - wk) / dd;
                const float dp = (p >= 1e-7f && p <= 1.0f) ? da : 0.f;
                g2[c] = dp;
                dot += p * dp;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                g2[c] = padded ? 0.f : psoft[bt * NC + c] * (g2[c] - dot);     // filled logits are constants
                if (q == 0) { dl[tl * 2 * NC + c] = g1[c]; dl[tl * 2 * NC + NC + c] = g2[c]; }
            }
        }
    }
    __syncthreads();
    if (rows) {
        if (tl < tn) {
            // ---- dx rows ----
            float* dr = dx + bt * D;
#pragma unroll 2
            for (int i = 0; i < D / (4 * HEAD_BL); ++i) {
                const int k = 4 * HEAD_BL * i + 4 * q;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float4 a = *(const float4*)(w1 + c * D + k);
                    const float4 g = *(const float4*)(w2 + c * D + k);
                    acc.x = fmaf(g1[c], a.x, fmaf(g2[c], g.x, acc.x));
                    acc.y = fmaf(g1[c], a.y, fmaf(g2[c], g.y, acc.y));
                    acc.z = fmaf(g1[c], a.z, fmaf(g2[c], g.z, acc.z));
                    acc.w = fmaf(g1[c], a.w, fmaf(g2[c], g.w, acc.w));
                }
                const uint32_t e = (uint32_t)(bt * D + k);
                acc.x = sed_keep(e, seed, thr24) ? acc.x * dscale : 0.f;
                acc.y = sed_keep(e + 1, seed, thr24) ? acc.y * dscale : 0.f;
                acc.z = sed_keep(e + 2, seed, thr24) ? acc.z * dscale : 0.f;
                acc.w = sed_keep(e + 3, seed, thr24) ? acc.w * dscale : 0.f;
                *(float4*)(dr + k) = acc;
            }
        }
        if (tid < 2 * NC) {
            float sm = 0.f;
            for (int i = 0; i < tn; ++i) sm += dl[i * 2 * NC + tid];
            mine[2 * NC * D + tid] = sm;
        }
        if (SPLIT) return;
    }
    // ---- weight gradients: thread k owns input feature k (and k + 256 when D = 384) ----
    if (!SPLIT) first_round();
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const int k = ft + 256 * j;
        if (k >= D) break;
        float a1[NC], a2[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { a1[c] = 0.f; a2[c] = 0.f; }
        // eight frames per round, the NEXT round's loads in flight under this round's FMAs (one load per iteration made this loop 64
        // dependent HBM round trips; eight loads and then their FMAs still exposed one round trip per round)
        for (int tl0 = 0; tl0 < tn; tl0 += 8) {
            float vn[8];                                        // (v[j] holds this round: its first one was fetched before the barrier)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int nt = tl0 + 8 + u < tn ? tl0 + 8 + u : tn - 1;
                vn[u] = tl0 + 8 < tn ? x[((size_t)b * T + tbeg + nt) * D + k] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int tl2 = tl0 + u;
                const size_t bt2 = (size_t)b * T + tbeg + tl2;
                const float vv = (tl2 < tn && sed_keep((uint32_t)(bt2 * D + k), seed, thr24)) ? v[j][u] * dscale : 0.f;
                const int tc = tl2 < tn ? tl2 : 0;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    a1[c] = fmaf(dl[tc * 2 * NC + c], vv, a1[c]);
                    a2[c] = fmaf(dl[tc * 2 * NC + NC + c], vv, a2[c]);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) v[j][u] = vn[u];
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) { mine[c * D + k] = a1[c]; mine[NC * D + c * D + k] = a2[c]; }
    }
}
// out[j] = sum over the nrec records of part[r][j], in a fixed order; j < NP = 2 NC D + 2 NC, laid out dW1 | dW2 | db1 | db2.
// 64 columns per workgroup x 4 record groups (group g: records g, g + 4, ... with four independent partial sums), combined in LDS.
__global__ __launch_bounds__(256) void head_bwd_reduce_kernel(const float* __restrict__ part, int nrec, int NP, int ncd,
                                                              float* __restrict__ dW1, float* __restrict__ dW2,
                                                              float* __restrict__ db1, float* __restrict__ db2, int NC) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6, j = blockIdx.x * 64 + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < NP) {
        int r = grp;
        for (; r + 12 < nrec; r += 16) {
            s0 += part[(size_t)r * NP + j]; s1 += part[(size_t)(r + 4) * NP + j];
            s2 += part[(size_t)(r + 8) * NP + j]; s3 += part[(size_t)(r + 12) * NP + j];
        }
        for (; r < nrec; r += 4) s0 += part[(size_t)r * NP + j];
    }
    red[grp][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && j < NP) {
        const float v = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        if (j < ncd) dW1[j] = v;
        else if (j < 2 * ncd) dW2[j - ncd] = v;
        else if (j < 2 * ncd + NC) db1[j - 2 * ncd] = v;
        else db2[j - 2 * ncd - NC] = v;
    }
}
SED_API long long sed_head_bwd_scratch_floats(int B, int T, int D, int NC) {
    return (long long)B * ((T + HEAD_TS - 1) / HEAD_TS) * (2 * NC * D + 2 * NC);
}

SED_API int sed_head_fwd(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, float* strong,
                            float* psoft, float* weak, float* den, int B, int T, int D, int NC, unsigned seed, unsigned thr24,
                            float dscale, const unsigned* seed_dev, const unsigned char* classes_valid, const unsigned char* pad_mask,
                            void* stream) {
    if (D != 256 && D != 384) return SED_ERR_UNSUPPORTED;
    if (B <= 0 || T <= 0) return SED_OK;
    hipStream_t s = (hipStream_t)stream;
    const int smem_f = 2 * NC * D * 4;
#define HEAD_CASE(nc, d) \
    if (NC == nc && D == d) { SED_MAX_SMEM((head_fwd_kernel<nc, d>), smem_f); SED_LAUNCH((head_fwd_kernel<nc, d>), dim3(B), dim3(HEAD_THREADS), smem_f, s, x, W1, b1, W2, b2, strong, psoft, weak, den, T, seed, thr24, dscale, seed_dev, classes_valid, pad_mask); return sed_check_launch(); }
    HEAD_CASE(10, 256) HEAD_CASE(27, 256) HEAD_CASE(10, 384) HEAD_CASE(27, 384)
#undef HEAD_CASE
    return SED_ERR_UNSUPPORTED;
}

SED_API int sed_head_bwd(const float* x, const float* W1, const float* W2, const float* strong, const float* psoft,
                            const float* weak, const float* den, const float* d_strong, const float* d_weak, float* dx, float* dW1,
                            float* dW2, float* db1, float* db2, int B, int T, int D, int NC, unsigned seed, unsigned thr24,
                            float dscale, const unsigned* seed_dev, const unsigned char* classes_valid, const unsigned char* pad_mask,
                            float* scratch, void* stream) {
    if (D != 256 && D != 384) return SED_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const bool reduce = dW1 || dW2 || db1 || db2;          // all four null: records only, summed later by sed_head_bwd_reduce
    if (reduce && !(dW1 && dW2 && db1 && db2)) return SED_ERR_ARG;
    if (B <= 0 || T <= 0) { if (reduce) sed_zero4(s, dW1, NC * D, dW2, NC * D, db1, NC, db2, NC); return SED_OK; }
    if (!scratch) return SED_ERR_ARG;
    const int smem = (2 * NC * D + HEAD_TS * 2 * NC) * 4;
    const int gx = (T + HEAD_TS - 1) / HEAD_TS, NP = 2 * NC * D + 2 * NC;
    int rc = SED_ERR_UNSUPPORTED;
#define HEAD_CASE(nc, d) \
    if (NC == nc && D == d) { SED_MAX_SMEM((head_bwd_kernel<nc, d>), smem); SED_LAUNCH((head_bwd_kernel<nc, d>), dim3(gx, B), dim3(HEAD_BWD_THREADS(nc)), smem, s, x, W1, W2, strong, psoft, weak, den, d_strong, d_weak, dx, scratch, T, seed, thr24, dscale, seed_dev, classes_valid, pad_mask); rc = sed_check_launch(); }
    HEAD_CASE(10, 256) HEAD_CASE(27, 256) HEAD_CASE(10, 384) HEAD_CASE(27, 384)
#undef HEAD_CASE
    if (rc != SED_OK || !reduce) return rc;
    SED_LAUNCH(head_bwd_reduce_kernel, dim3((NP + 63) / 64), dim3(256), 0, s, (const float*)scratch, gx * B, NP, NC * D, dW1, dW2, db1, db2, NC);
    return sed_check_launch();
}
// The second half of sed_head_bwd on its own: the weight / bias gradients from the records a sed_head_bwd call with null gradient
// pointers left in `scratch` (same B, T, D, NC).  Nothing on the backward chain reads them.
SED_API int sed_head_bwd_reduce(const float* scratch, float* dW1, float* dW2, float* db1, float* db2, int B, int T, int D, int NC,
                                   void* stream) {
    if (D != 256 && D != 384) return SED_ERR_UNSUPPORTED;
    if (!(dW1 && dW2 && db1 && db2)) return SED_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (B <= 0 || T <= 0) { sed_zero4(s, dW1, NC * D, dW2, NC * D, db1, NC, db2, NC); return SED_OK; }
    if (!scratch) return SED_ERR_ARG;
    const int gx = (T + HEAD_TS - 1) / HEAD_TS, NP = 2 * NC * D + 2 * NC;
    SED_LAUNCH(head_bwd_reduce_kernel, dim3((NP + 63) / 64), dim3(256), 0, s, scratch, gx * B, NP, NC * D, dW1, dW2, db1, db2, NC);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------
// mean-teacher losses + gradient seeds.  One workgroup per clip.
// scalars[0..5] = BCE strong (student), BCE weak (student), BCE strong (teacher), BCE weak (teacher),
//                 MSE strong, MSE weak.  g_strong (B,T,NC), g_weak (B,NC) = d(total)/d(student outputs) with
// total = BCE_s + BCE_w + weight * (MSE_s + MSE_w).  torch.nn.BCELoss semantics: log clamped at -100,
// gradient (s - y) / max(s (1-s), 1e-12).  labels (B,NC,T) reference layout; labels_weak (n_weak, NC).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bce_term(float s, float y) {
    return -(y * fmaxf(logf(s), -100.0f) + (1.0f - y) * fmaxf(logf(1.0f - s), -100.0f));
}
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ strong_s, const float* __restrict__ weak_s,
                                                   const float* __restrict__ strong_t, const float* __restrict__ weak_t,
                                                   const float* __restrict__ labels, const float* __restrict__ labels_weak,
                                                   float* __restrict__ scalars, float* __restrict__ g_strong,
                                                   float* __restrict__ g_weak, int B, int T, int NC, int n_strong, int n_weak,
                                                   float weight, const float* __restrict__ weight_dev, int selfsup_bce,
                                                   int selfsup_from, const unsigned char* __restrict__ valid,
                                                   float* __restrict__ work, int finish) {
    if (weight_dev) weight = *weight_dev;       // consistency weight in device memory (hipGraph replays)
    // one workgroup per clip writes its eight pre-scaled per-clip sums to work[b][8]; the workgroup that draws the last ticket
    // (work[8 B], an integer counter it resets to 0) adds them up in clip order: no zero-fill launch, no float atomics,
    // the same bits every run
    __shared__ float red[4][6];
    __shared__ int is_last;
    const int tid = threadIdx.x, b = blockIdx.x;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // selfsup_from: the consistency terms average over clips [selfsup_from, B) only (2024 recipe: everything but MAESTRO,
    // dcase2024 sed_trainer_pretrained.py:337,399-406); valid (B,NC): labels of classes a clip's data set does not annotate
    // count as 0 (:352-356)
    const int ns_el = n_strong * T * NC, all_el = (B - selfsup_from) * T * NC, clip_el = T * NC;
    const float inv_bs = ns_el > 0 ? 1.0f / (float)ns_el : 0.f, inv_all = all_el > 0 ? 1.0f / (float)all_el : 0.f;
    const bool selfsup = b >= selfsup_from;
    for (int j = tid; j < clip_el; j += 256) {
        const int i = b * clip_el + j, c = j % NC, t = j / NC;
        const float s = strong_s[i], q = strong_t[i];
        float g = 0.f;
        if (!selfsup) {
        } else if (selfsup_bce) {      // self_sup_loss: bce (sed_trainer.py:99-100): BCELoss(student, teacher), teacher as the target
            g = weight * (s - q) / fmaxf(s * (1.0f - s), 1e-12f) * inv_all;
            acc[4] += bce_term(s, q);
        } else {
            const float d = s - q;
            g = weight * 2.0f * d * inv_all;
            acc[4] += d * d;
        }
        if (b < n_strong) {
            float y = labels[((size_t)b * NC + c) * T + t];
            if (valid && !valid[b * NC + c]) y = 0.f;
            acc[0] += bce_term(s, y);
            acc[2] += bce_term(q, y);
            g += (s - y) / fmaxf(s * (1.0f - s), 1e-12f) * inv_bs;
        }
        g_strong[i] = g;
    }
    const int nw_el = n_weak * NC, allw = (B - selfsup_from) * NC;
    const float inv_bw = nw_el > 0 ? 1.0f / (float)nw_el : 0.f, inv_allw = allw > 0 ? 1.0f / (float)allw : 0.f;
    if (tid < NC) {
        const int c = tid, i = b * NC + c;
        const float s = weak_s[i], q = weak_t[i];
        float g = 0.f;
        if (!selfsup) {
        } else if (selfsup_bce) {
            g = weight * (s - q) / fmaxf(s * (1.0f - s), 1e-12f) * inv_allw;
            acc[5] += bce_term(s, q);
        } else {
            const float d = s - q;
            g = weight * 2.0f * d * inv_allw;
            acc[5] += d * d;
        }
        if (b >= n_strong && b < n_strong + n_weak) {
            float y = labels_weak[(b - n_strong) * NC + c];
            if (valid && !valid[b * NC + c]) y = 0.f;
            acc[1] += bce_term(s, y);
            acc[3] += bce_term(q, y);
            g += (s - y) / fmaxf(s * (1.0f - s), 1e-12f) * inv_bw;
        }
        g_weak[i] = g;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float v = wave_sum(acc[k]);
        if ((tid & 63) == 0) red[tid >> 6][k] = v;
    }
    __syncthreads();
    if (tid < 8) {
        // slots 0..5: the six means; 6: weight * (self_strong + self_weak) (logged as tot_self_loss); 7: the total loss
        float part[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float sc = k == 0 || k == 2 ? inv_bs : (k == 1 || k == 3 ? inv_bw : (k == 4 ? inv_all : inv_allw));
            part[k] = ((red[0][k] + red[1][k]) + (red[2][k] + red[3][k])) * sc;
        }
        const float self_tot = weight * (part[4] + part[5]);
        const float v = tid < 6 ? part[tid] : (tid == 6 ? self_tot : part[0] + part[1] + self_tot);
        work[8 * b + tid] = v;
    }
    // finish == 0 (sed_mt_loss_records): the per-clip records are all this launch leaves; loss_finish_kernel adds them later.  The
    // gradient seeds above are complete per clip -- nothing on the backward chain needs the eight sums, and the agent-scope fence in
    // front of the ticket makes every workgroup write its XCD's dirty L2 lines back (what such a hand-over costs).
    if (!finish) return;
    __threadfence();
    __syncthreads();
    if (tid == 0) is_last = atomicAdd((unsigned*)(work + 8 * (size_t)B), 1u) == (unsigned)(B - 1);
    __syncthreads();
    if (is_last) {
        __threadfence();
        // Round 4: the 8 B per-clip records come in with ALL 256 threads at once and are then added in clip order out of LDS.  Eight
        // threads walking them in global memory was a chain of B dependent L2 round trips -- most of this kernel's 20 us, on the
        // critical path between the forward and the backward pass.  Same order of additions: same bits.
        __shared__ float fin[8 * 256];
        const bool staged = B <= 256;
        if (staged) {
            for (int idx = tid; idx < 8 * B; idx += 256) fin[idx] = ((volatile float*)work)[idx];
            __syncthreads();
        }
        if (tid < 8) {
            float sum = 0.f;
            if (staged) for (int i = 0; i < B; ++i) sum += fin[8 * i + tid];
            else for (int i = 0; i < B; ++i) sum += ((volatile float*)work)[8 * i + tid];
            scalars[tid] = sum;
            if (tid == 7) scalars[8] = sum;     // the total once more: the differentiable 0-d output is a view of this slot
        }
        if (tid == 0) *(unsigned*)(work + 8 * (size_t)B) = 0u;
    }
}
SED_API int sed_mt_loss(const float* strong_s, const float* weak_s, const float* strong_t, const float* weak_t,
                           const float* labels, const float* labels_weak, float* scalars, float* g_strong, float* g_weak, int B,
                           int T, int NC, int n_strong, int n_weak, float weight, const float* weight_dev, int selfsup_bce,
                           int selfsup_from, const unsigned char* valid, float* work, void* stream) {
    if (B <= 0 || T <= 0 || NC <= 0 || NC > 256 || n_strong + n_weak > B || selfsup_from < 0 || selfsup_from > B) return SED_ERR_ARG;
    if (work == nullptr) return SED_ERR_ARG;
    SED_LAUNCH(loss_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, strong_s, weak_s, strong_t, weak_t, labels, labels_weak,
               scalars, g_strong, g_weak, B, T, NC, n_strong, n_weak, weight, weight_dev, selfsup_bce, selfsup_from, valid, work, 1);
    return sed_check_launch();
}
// The eight sums of sed_mt_loss from the per-clip records of a sed_mt_loss_records launch: one workgroup, clip order (the same order of
// additions, the same bits as the one-call form).
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* __restrict__ work, float* __restrict__ scalars, int B) {
    __shared__ float fin[8 * 256];
    const int tid = threadIdx.x;
    const bool staged = B <= 256;
    if (staged) {
        for (int idx = tid; idx < 8 * B; idx += 256) fin[idx] = work[idx];
        __syncthreads();
    }
    if (tid < 8) {
        float sum = 0.f;
        if (staged) for (int i = 0; i < B; ++i) sum += fin[8 * i + tid];
        else for (int i = 0; i < B; ++i) sum += work[8 * i + tid];
        scalars[tid] = sum;
        if (tid == 7) scalars[8] = sum;
    }
}
// sed_mt_loss in two halves: this one writes the gradient seeds (all the backward pass needs) and the per-clip records into `work`
// (8 B floats; the ticket word is not touched) -- no fence, no ticket; sed_mt_loss_finish adds the records up, on any stream ordered
// after it.  Same arguments as sed_mt_loss (`scalars` is not written here).
SED_API int sed_mt_loss_records(const float* strong_s, const float* weak_s, const float* strong_t, const float* weak_t,
                                   const float* labels, const float* labels_weak, float* scalars, float* g_strong, float* g_weak, int B,
                                   int T, int NC, int n_strong, int n_weak, float weight, const float* weight_dev, int selfsup_bce,
                                   int selfsup_from, const unsigned char* valid, float* work, void* stream) {
    if (B <= 0 || T <= 0 || NC <= 0 || NC > 256 || n_strong + n_weak > B || selfsup_from < 0 || selfsup_from > B) return SED_ERR_ARG;
    if (work == nullptr) return SED_ERR_ARG;
    SED_LAUNCH(loss_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, strong_s, weak_s, strong_t, weak_t, labels, labels_weak,
               scalars, g_strong, g_weak, B, T, NC, n_strong, n_weak, weight, weight_dev, selfsup_bce, selfsup_from, valid, work, 0);
    return sed_check_launch();
}
SED_API int sed_mt_loss_finish(const float* work, float* scalars, int B, void* stream) {
    if (B <= 0 || work == nullptr || scalars == nullptr) return SED_ERR_ARG;
    SED_LAUNCH(loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, work, scalars, B);
    return sed_check_launch();
}
