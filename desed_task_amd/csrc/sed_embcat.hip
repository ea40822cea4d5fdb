// K14: embedding fusion in front of the recurrent stage, `aggregation_type: "pool1d"` of
// desed_task/nnet/CRNN.py:283-296 (the 2023 "pretrained" / BEATs configuration):
//     reshape_emb = adaptive_avg_pool1d(embeddings (B, E, Te), T).transpose(1, 2)            (B, T, E)
//     z           = dropout(cat((x (B, T, C), reshape_emb), -1))                             (B, T, C + E)
//     x           = cat_tf(z)                                                                 Linear(C + E -> C)
// sed_embcat_fwd builds z in ONE pass over the embeddings (pooling, transposition, concatenation and the dropout mask fused:
// the (B, T, E) pooled tensor and the undropped concatenation never exist in HBM); the Linear and its weight / input
// gradients are the split-bf16 GEMMs of K7 (sed_gemm_bf16x3) on z, which is also what the backward needs saved.
// sed_embcat_bwd applies the same mask to the x-columns of dz (the embeddings are frozen features: no gradient).
//
// HBM-bound: algorithmic bytes per clip = read E*Te*4 + C*T*4, write (C + E)*T*4  (768 x 496, 128 x 156: 2.16 MB).
// Layout: embeddings (B, E, Te) time-contiguous exactly as the reference stores them; a workgroup stages the input frames of
// EMB_TCH output frames x EMB_TILE channel rows in LDS with coalesced reads along time (odd row stride: the strided pooling
// reads are conflict-free) and writes 256-byte channel runs of z; ~33 KB of LDS, four workgroups per CU.
#include "sed_common.h"

#define EMB_TILE 64        // embedding channels per workgroup: 256-byte channel runs of z per output frame
#define EMB_TCH 40         // output frames per workgroup (bounds the LDS stage: 64 x ~130 floats for 496 -> 156)
#define EMB_THREADS 256

// pooling window of output frame t: [floor(t * Te / T), ceil((t + 1) * Te / T))  (torch adaptive pooling); T * Te < 2^31
__device__ __forceinline__ int emb_win_lo(int t, int Te, int T) { return (int)((unsigned)(t * Te) / (unsigned)T); }
__device__ __forceinline__ int emb_win_hi(int t, int Te, int T) { return (int)((unsigned)((t + 1) * Te + T - 1) / (unsigned)T); }

__global__ __launch_bounds__(EMB_THREADS) void embcat_fwd_kernel(const float* __restrict__ x, const float* __restrict__ emb,
                                                                  float* __restrict__ z, int T, int Te, int C, int E, int RS,
                                                                  uint32_t seed, uint32_t thr24, float dscale,
                                                                  const unsigned* __restrict__ seed_dev,
                                                                  const int* __restrict__ tmask, int mode) {
    if (seed_dev) seed += *seed_dev;            // per-step entropy in device memory (hipGraph replays)
    SED_DYN_SMEM(smem_raw);
    float* rows = (float*)smem_raw;             // [EMB_TILE][RS]: the input frames this chunk of output frames pools over
    __shared__ int win[EMB_TCH + 1][2];         // window bounds per output frame, relative to the staged span
    const int b = blockIdx.z, tile = blockIdx.x, tid = threadIdx.x;
    const int t0 = blockIdx.y * EMB_TCH, t1 = min(T, t0 + EMB_TCH);
    const int ntile = (E + EMB_TILE - 1) / EMB_TILE, W = C + E;
    // dropstep_recurrent (CRNN.py:288-294): one time span of the CNN features and, drawn independently, one of the embeddings
    // is zeroed per clip before the concatenation is dropped out.  tmask (B,4) = [x0, x1, e0, e1) in output frames.
    int mx0 = 0, mx1 = 0, me0 = 0, me1 = 0;
    if (tmask) { mx0 = tmask[4 * b]; mx1 = tmask[4 * b + 1]; me0 = tmask[4 * b + 2]; me1 = tmask[4 * b + 3]; }
    if (tile == ntile) {                        // the x columns of these frames: rows of x are contiguous, rows of z W apart
        const size_t m0 = (size_t)b * T + t0;
        const float* xs = x + m0 * C;
        const int n = (t1 - t0) * C, dt = EMB_THREADS / C, dc = EMB_THREADS % C;
        int t = tid / C, c = tid % C;
        for (int i = tid; i < n; i += EMB_THREADS) {
            const size_t o = (m0 + t) * W + c;
            const bool gone = (t0 + t) >= mx0 && (t0 + t) < mx1;
            z[o] = (!gone && sed_keep((uint32_t)o, seed, thr24)) ? xs[i] * dscale : 0.f;
            t += dt; c += dc;
            if (c >= C) { c -= C; ++t; }
        }
        return;
    }
    const int e0 = tile * EMB_TILE, ne = min(EMB_TILE, E - e0);
    const int s_lo = emb_win_lo(t0, Te, T), len = emb_win_hi(t1 - 1, Te, T) - s_lo;
    if (tid < t1 - t0) {
        if (mode == 1) {        // aggregation_type "interpolate" (CRNN.py:271-279): nearest-exact, source = floor((t + 0.5) Te / T)
            int src = (int)(((long long)(2 * (t0 + tid) + 1) * Te) / (2 * T));
            src = min(src, Te - 1);
            win[tid][0] = src - s_lo;
            win[tid][1] = src - s_lo + 1;
        } else {
            win[tid][0] = emb_win_lo(t0 + tid, Te, T) - s_lo;
            win[tid][1] = emb_win_hi(t0 + tid, Te, T) - s_lo;
        }
    }
    const float* src = emb + ((size_t)b * E + e0) * Te + s_lo;
    // half a wave per channel row: 32 consecutive frames (128 B) per row and pass, eight rows per workgroup pass
    for (int r = tid >> 5; r < ne; r += EMB_THREADS / 32) {
        const float* row = src + (size_t)r * Te;
        for (int s = tid & 31; s < len; s += 32) rows[r * RS + s] = row[s];
    }
    __syncthreads();
    const int e = tid & (EMB_TILE - 1);
    if (e >= ne) return;
    for (int t = t0 + tid / EMB_TILE; t < t1; t += EMB_THREADS / EMB_TILE) {
        const int w0 = win[t - t0][0], w1 = win[t - t0][1];
        float acc = 0.f;
        for (int s = w0; s < w1; ++s) acc += rows[e * RS + s];  // RS is odd: the 64 channel rows hit distinct banks
        const float v = acc / (float)(w1 - w0);
        const size_t o = ((size_t)b * T + t) * W + C + e0 + e;
        z[o] = (!(t >= me0 && t < me1) && sed_keep((uint32_t)o, seed, thr24)) ? v * dscale : 0.f;
    }
}

// x (B,T,C), emb (B,E,Te) -> z (B,T,C+E).  thr24 = 0 disables the dropout (dscale is then 1).
SED_API int sed_embcat_fwd(const float* x, const float* emb, float* z, int B, int T, int Te, int C, int E, unsigned seed,
                              unsigned thr24, float dscale, const unsigned* seed_dev, const int* tmask, int mode, void* stream) {
    if (B <= 0 || T <= 0) return SED_OK;
    if (Te < 1 || C < 1 || E < 1 || mode < 0 || mode > 1) return SED_ERR_ARG;
    // longest input span of one chunk of EMB_TCH output frames (+2: the floor / ceil at either end), odd row stride
    int RS = (int)(((long long)EMB_TCH * Te + T - 1) / T) + 2;
    RS |= 1;
    const size_t smem = (size_t)EMB_TILE * RS * sizeof(float);
    if (smem > 150 * 1024 || (size_t)B * T * (C + E) >= (1ull << 32) || (long long)(T + 1) * Te + T >= (1ll << 31) || C > (1 << 20))
        return SED_ERR_UNSUPPORTED;
    const int ntile = (E + EMB_TILE - 1) / EMB_TILE, nchunk = (T + EMB_TCH - 1) / EMB_TCH;
    SED_MAX_SMEM(embcat_fwd_kernel, smem);
    SED_LAUNCH(embcat_fwd_kernel, dim3(ntile + 1, nchunk, B), dim3(EMB_THREADS), smem, (hipStream_t)stream, x, emb, z, T, Te, C, E,
               RS, seed, thr24, dscale, seed_dev, tmask, mode);
    return sed_check_launch();
}

__global__ __launch_bounds__(256) void embcat_bwd_kernel(const float* __restrict__ dzx, float* __restrict__ dx, size_t n, int C,
                                                          int W, uint32_t seed, uint32_t thr24, float dscale,
                                                          const unsigned* __restrict__ seed_dev, const int* __restrict__ tmask, int T) {
    if (seed_dev) seed += *seed_dev;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t m = i / C;
        const int c = (int)(i - m * C);
        bool gone = false;
        if (tmask) {
            const int b = (int)(m / T), t = (int)(m - (size_t)b * T);
            gone = t >= tmask[4 * b] && t < tmask[4 * b + 1];
        }
        dx[i] = (!gone && sed_keep((uint32_t)(m * W + c), seed, thr24)) ? dzx[i] * dscale : 0.f;
    }
}

// dzx (M, C) = the first C columns of dz = dy . W_cat_tf -> dx (M, C) = dzx masked with the forward's dropout mask.
SED_API int sed_embcat_bwd(const float* dzx, float* dx, int M, int C, int E, unsigned seed, unsigned thr24, float dscale,
                              const unsigned* seed_dev, const int* tmask, int T, void* stream) {
    if (M <= 0) return SED_OK;
    if (C < 1 || E < 1 || (tmask && (T < 1 || M % T != 0))) return SED_ERR_ARG;
    const size_t n = (size_t)M * C;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    SED_LAUNCH(embcat_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dzx, dx, n, C, C + E, seed, thr24, dscale, seed_dev, tmask, T);
    return sed_check_launch();
}
