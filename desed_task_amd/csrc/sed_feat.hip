// K2-K5: spectrogram-domain elementwise ops, all on (B, L) row-major clips (L = T * n_mels).
//  * sed_logscale_fwd  : take_log (sed_trainer.py:253-264) + TorchScaler instance/minmax
//                        (desed_task/utils/scaler.py:114-120): two launches, deterministic
//                        (per-chunk partial min/max, no atomics).
//  * sed_mixup         : desed_task/data_augm.py:31-51 on a group of n clips (soft/hard labels too).
//  * sed_specaug       : the two torchaudio axis masks of desed_task/nnet/CRNN.py:207-219, given
//                        per-clip [f0,f1,t0,t1) bounds.
// All are HBM-bound streaming kernels: float4 per lane, grid-stride.
#include "sed_common.h"
#include <string.h>

#define FEAT_CHUNKS 32

__device__ __forceinline__ float logdb(float v) {
    v = 20.0f * log10f(fmaxf(v, 1e-5f));
    return fminf(fmaxf(v, -50.0f), 80.0f);
}

// pass 1: y = LOG ? logdb(x) : x ; partial[b][chunk] = (min, max) of y over the chunk
template <bool LOG>
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             float* __restrict__ partial, int L) {
    __shared__ float smin[4], smax[4];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int per = (L + FEAT_CHUNKS - 1) / FEAT_CHUNKS;
    const int lo = chunk * per, hi = min(L, lo + per);
    const float* xi = x + (size_t)b * L;
    float* yo = y + (size_t)b * L;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        float v = xi[i];
        if (LOG) v = logdb(v);
        if (LOG || yo != xi) yo[i] = v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[((size_t)b * FEAT_CHUNKS + chunk) * 2 + 0] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        partial[((size_t)b * FEAT_CHUNKS + chunk) * 2 + 1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
}

// pass 2: out = (y - mn) / (mx - mn + eps) * 2 - 1 (same op order as the reference); also emits (mn, mx)
__global__ __launch_bounds__(256) void minmax_apply_kernel(const float* __restrict__ y, float* __restrict__ out,
                                                           const float* __restrict__ partial, float* __restrict__ minmax,
                                                           int L, float eps) {
    __shared__ float s_mn, s_mx;
    const int b = blockIdx.y;
    if (threadIdx.x < 64) {
        float mn = INFINITY, mx = -INFINITY;
        if (threadIdx.x < FEAT_CHUNKS) {
            mn = partial[((size_t)b * FEAT_CHUNKS + threadIdx.x) * 2 + 0];
            mx = partial[((size_t)b * FEAT_CHUNKS + threadIdx.x) * 2 + 1];
        }
        mn = wave_min(mn);
        mx = wave_max(mx);
        if (threadIdx.x == 0) { s_mn = mn; s_mx = mx; }
    }
    __syncthreads();
    const float mn = s_mn, den = s_mx - s_mn + eps;
    if (blockIdx.x == 0 && threadIdx.x == 0 && minmax) { minmax[2 * b] = s_mn; minmax[2 * b + 1] = s_mx; }
    const float* yi = y + (size_t)b * L;
    float* oo = out + (size_t)b * L;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) oo[i] = (yi[i] - mn) / den * 2.0f - 1.0f;
}

// x: (B, L) linear mel (apply_log=1) or any tensor (apply_log=0).  logbuf: (B, L) scratch for the log values
// (may alias out).  partial: B*32*2 floats.  minmax: optional (B,2) output.  out may alias x when apply_log=0.
SED_API int sed_logscale_fwd(const float* x, float* logbuf, float* out, float* partial, float* minmax, int B, int L,
                                int apply_log, float eps, void* stream) {
    if (B <= 0 || L <= 0) return B < 0 || L < 0 ? SED_ERR_ARG : SED_OK;
    hipStream_t s = (hipStream_t)stream;
    if (apply_log)
        SED_LAUNCH((minmax_partial_kernel<true>), dim3(FEAT_CHUNKS, B), dim3(256), 0, s, x, logbuf, partial, L);
    else
        SED_LAUNCH((minmax_partial_kernel<false>), dim3(FEAT_CHUNKS, B), dim3(256), 0, s, x, (float*)x, partial, L);
    const float* src = apply_log ? logbuf : x;
    int gx = (L + 256 * 8 - 1) / (256 * 8);
    SED_LAUNCH(minmax_apply_kernel, dim3(gx, B), dim3(256), 0, s, src, out, partial, minmax, L, eps);
    return sed_check_launch();
}

// log only (take_log as a standalone op)
__global__ __launch_bounds__(256) void logdb_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = logdb(x[i]);
}
SED_API int sed_take_log(const float* x, float* y, long long n, void* stream) {
    if (n <= 0) return SED_OK;
    int grid = (int)min((long long)4096, (n + 255) / 256);
    SED_LAUNCH(logdb_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, (size_t)n);
    return sed_check_launch();
}

// ---- mixup ------------------------------------------------------------------------------------
// data[i] = c*src[i] + omc*src[perm[i]]   (src = snapshot of the group; mode 1/2 = soft/hard label clamp)
__global__ __launch_bounds__(256) void mixup_kernel(float* __restrict__ data, const float* __restrict__ src,
                                                    const int* __restrict__ perm, float c, float omc, int L, int mode,
                                                    const float* __restrict__ c_dev) {
    if (c_dev) {                                // coefficient in device memory (hipGraph replays)
        c = c_dev[0]; omc = c_dev[1];           // {c, 1-c} exactly as the host would have passed them by value
        if (c > 1.5f) return;                   // sentinel 2 = "no mixup this step" (a Beta(0.2,0.2) draw can round to exactly 1.0f,
    }                                           // and hard labels are still clamp(t + t[perm]) then)
    const int i = blockIdx.y;
    const int j = perm[i];
    const float* a = src + (size_t)i * L;
    const float* p = src + (size_t)j * L;
    float* o = data + (size_t)i * L;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < L; e += gridDim.x * 256) {
        float v;
        if (mode == 2) v = a[e] + p[e];
        else v = c * a[e] + omc * p[e];
        if (mode != 0) v = fminf(fmaxf(v, 0.0f), 1.0f);
        o[e] = v;
    }
}
// data: (n, L) in place; tmp: (n, L) scratch; perm: n int32 on device.  mode 0 = features, 1 = soft labels, 2 = hard labels.
SED_API int sed_mixup(float* data, float* tmp, const int* perm, float c, float one_minus_c, int n, int L, int mode,
                         const float* c_dev, void* stream) {
    if (n <= 0 || L <= 0) return SED_OK;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(tmp, data, (size_t)n * L * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return SED_ERR_LAUNCH;
    int gx = min(64, (L + 255) / 256);
    SED_LAUNCH(mixup_kernel, dim3(gx, n), dim3(256), 0, s, data, (const float*)tmp, perm, c, one_minus_c, L, mode, c_dev);
    return sed_check_launch();
}

// Batched form: up to MIX_JOBS groups in one launch, no scratch copy.  blockIdx.y = job; a thread owns position e of the clip block
// for ALL clips of the group (n <= MIX_NMAX): every clip's value and its partner's are in registers before the first store, and no
// other thread touches position e, so mixing in place is safe.  The partner loads hit the lines the thread has just read.
#define MIX_JOBS 8
#define MIX_NMAX 32
struct MixJobs {
    float* data[MIX_JOBS];
    const int* perm[MIX_JOBS];
    const float* c_dev[MIX_JOBS];
    float c[MIX_JOBS], omc[MIX_JOBS];
    int n[MIX_JOBS], L[MIX_JOBS], mode[MIX_JOBS];
};
__global__ __launch_bounds__(256) void mixup_multi_kernel(MixJobs J) {
    const int job = blockIdx.y;
    float c = J.c[job], omc = J.omc[job];
    const float* cd = J.c_dev[job];
    if (cd) {
        c = cd[0]; omc = cd[1];
        if (c > 1.5f) return;                   // sentinel: no mixup this step
    }
    const int n = J.n[job], L = J.L[job], mode = J.mode[job];
    float* data = J.data[job];
    const int* perm = J.perm[job];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < L; e += gridDim.x * 256) {
        float a[MIX_NMAX], p[MIX_NMAX];
#pragma unroll
        for (int i = 0; i < MIX_NMAX; ++i)
            if (i < n) a[i] = data[(size_t)i * L + e];
#pragma unroll
        for (int i = 0; i < MIX_NMAX; ++i)
            if (i < n) p[i] = data[(size_t)perm[i] * L + e];
#pragma unroll
        for (int i = 0; i < MIX_NMAX; ++i)
            if (i < n) {
                float v = mode == 2 ? a[i] + p[i] : c * a[i] + omc * p[i];
                if (mode != 0) v = fminf(fmaxf(v, 0.0f), 1.0f);
                data[(size_t)i * L + e] = v;
            }
    }
}
SED_API int sed_mixup_multi(const long long* jobs, int njobs, void* stream) {
    if (njobs <= 0) return SED_OK;
    if (njobs > MIX_JOBS || jobs == nullptr) return SED_ERR_ARG;
    MixJobs J;
    int maxL = 0;
    for (int j = 0; j < MIX_JOBS; ++j) {
        const long long* r = jobs + 8 * (j < njobs ? j : 0);
        J.data[j] = (float*)(uintptr_t)r[0];
        J.perm[j] = (const int*)(uintptr_t)r[1];
        J.c_dev[j] = (const float*)(uintptr_t)r[2];
        const unsigned cb = (unsigned)r[3], ob = (unsigned)r[4];
        memcpy(&J.c[j], &cb, 4);
        memcpy(&J.omc[j], &ob, 4);
        J.n[j] = j < njobs ? (int)r[5] : 0;
        J.L[j] = j < njobs ? (int)r[6] : 0;
        J.mode[j] = (int)r[7];
        if (j < njobs) {
            if (J.n[j] < 0 || J.n[j] > MIX_NMAX || J.L[j] < 0) return SED_ERR_UNSUPPORTED;
            if (J.n[j] > 0 && (J.data[j] == nullptr || J.perm[j] == nullptr)) return SED_ERR_ARG;
            if (J.n[j] > 0 && J.L[j] > maxL) maxL = J.L[j];
        }
    }
    if (maxL == 0) return SED_OK;
    int gx = (maxL + 255) / 256;
    if (gx > 1024) gx = 1024;
    SED_LAUNCH(mixup_multi_kernel, dim3(gx, njobs), dim3(256), 0, (hipStream_t)stream, J);
    return sed_check_launch();
}

// ---- SpecAugment ------------------------------------------------------------------------------
// x, y: (B, T, F); bounds: (B, 4) int32 = [f0, f1, t0, t1): y = 0 inside either band else x
__global__ __launch_bounds__(256) void specaug_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      const int* __restrict__ bounds, int T, int Fq) {
    const int b = blockIdx.y;
    const int f0 = bounds[4 * b], f1 = bounds[4 * b + 1], t0 = bounds[4 * b + 2], t1 = bounds[4 * b + 3];
    const int L = T * Fq;
    const float* xi = x + (size_t)b * L;
    float* yo = y + (size_t)b * L;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < L; e += gridDim.x * 256) {
        const int t = e / Fq, f = e - t * Fq;
        const bool masked = (f >= f0 && f < f1) || (t >= t0 && t < t1);
        yo[e] = masked ? 0.0f : xi[e];
    }
}
SED_API int sed_specaug(const float* x, float* y, const int* bounds, int B, int T, int Fq, void* stream) {
    if (B <= 0) return SED_OK;
    int gx = min(64, (T * Fq + 255) / 256);
    SED_LAUNCH(specaug_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x, y, bounds, T, Fq);
    return sed_check_launch();
}

// ---- SpecAugment mask draws (CRNN.apply_specaugment, desed_task/nnet/CRNN.py:207-219 = torchaudio mask_along_axis[_iid]) ----
// u_f / u_t: (2, n) uniform draws per axis (row 0 -> mask length, row 1 -> start) or null when that axis is off; n = B
// (per-clip masks) or 1 (one mask for the batch).  bounds (B, 4) int32 = [f0, f1, t0, t1).  Same float32 arithmetic as the
// reference: value = u0 * mask_param, min_value = u1 * (axis_len - value), start = trunc(min_value), end = start + trunc(value).
// One launch instead of ~25 single-element torch kernels per model call.
__global__ __launch_bounds__(256) void specaug_bounds_kernel(const float* __restrict__ u_f, const float* __restrict__ u_t,
                                                             int* __restrict__ bounds, int B, int n, int f_param, int n_freq,
                                                             int t_param, int n_time) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const int i = n == 1 ? 0 : b;
    int f0 = 0, f1 = 0, t0 = 0, t1 = 0;
    if (u_f != nullptr) {
        const float value = u_f[i] * (float)f_param;
        const float min_value = u_f[n + i] * ((float)n_freq - value);
        f0 = (int)min_value;
        f1 = f0 + (int)value;
    }
    if (u_t != nullptr) {
        const float value = u_t[i] * (float)t_param;
        const float min_value = u_t[n + i] * ((float)n_time - value);
        t0 = (int)min_value;
        t1 = t0 + (int)value;
    }
    bounds[4 * b] = f0; bounds[4 * b + 1] = f1; bounds[4 * b + 2] = t0; bounds[4 * b + 3] = t1;
}
// The same masks from a counter-based generator instead of uniform tensors: u_k(clip i) = top 24 bits of sed_hash(4 i + k, seed)
// / 2^24, k = 0 / 1 frequency-mask length / start, 2 / 3 time-mask length / start -- the draw costs no launch of its own (two
// torch.rand launches per model call plus, under a hipGraph, the generator's two bookkeeping fills per replay otherwise).
__global__ __launch_bounds__(256) void specaug_bounds_seeded_kernel(int* __restrict__ bounds, int B, int n, int f_param, int n_freq,
                                                                    int t_param, int n_time, uint32_t seed,
                                                                    const unsigned* __restrict__ seed_dev) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    if (seed_dev) seed += *seed_dev;
    sed_specaug_draw(bounds, b, n, f_param, n_freq, t_param, n_time, seed);
}
SED_API int sed_specaug_bounds_seeded(int* bounds, int B, int n, int f_param, int n_freq, int t_param, int n_time, unsigned seed,
                                         const unsigned* seed_dev, void* stream) {
    if (B <= 0) return SED_OK;
    if (n != 1 && n != B) return SED_ERR_ARG;
    SED_LAUNCH(specaug_bounds_seeded_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, bounds, B, n, f_param, n_freq,
               t_param, n_time, (uint32_t)seed, seed_dev);
    return sed_check_launch();
}

SED_API int sed_specaug_bounds(const float* u_f, const float* u_t, int* bounds, int B, int n, int f_param, int n_freq,
                                  int t_param, int n_time, void* stream) {
    if (B <= 0) return SED_OK;
    if (n != 1 && n != B) return SED_ERR_ARG;
    SED_LAUNCH(specaug_bounds_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, u_f, u_t, bounds, B, n, f_param,
               n_freq, t_param, n_time);
    return sed_check_launch();
}

// ---- weak labels of the weakly annotated clips: (sum_t labels[b, c, :] > 0) as float (sed_trainer.py:292) ----
// labels (n, NC, T) -> out (n, NC); one wave per (clip, class) row (a thread per row would walk 156 dependent loads)
__global__ __launch_bounds__(256) void weak_labels_kernel(const float* __restrict__ labels, float* __restrict__ out, int rows, int T) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* p = labels + (size_t)r * T;
    float acc = 0.f;
    for (int t = lane; t < T; t += 64) acc += p[t];
    acc = wave_sum(acc);
    if (lane == 0) out[r] = acc > 0.f ? 1.0f : 0.0f;
}
SED_API int sed_weak_labels(const float* labels, float* out, int n, int NC, int T, void* stream) {
    if (n <= 0 || NC <= 0) return SED_OK;
    if (T <= 0) return SED_ERR_ARG;
    SED_LAUNCH(weak_labels_kernel, dim3((n * NC + 3) / 4), dim3(256), 0, (hipStream_t)stream, labels, out, n * NC, T);
    return sed_check_launch();
}

// ---- dropstep_recurrent without embeddings (desed_task/nnet/CRNN.py:296-301) -------------------------------------------------
// y = dropout(time_mask(x)) on (B,T,C): frames [t0, t1) of clip b are zeroed (bounds (B,2) int32 or null), then the usual
// counter-hash dropout over the element index.  The operator is diagonal: the backward is the same launch on the gradient.
__global__ __launch_bounds__(256) void dropstep_kernel(const float* __restrict__ x, float* __restrict__ y, const int* __restrict__ tb,
                                                       int T, int C, size_t n, uint32_t seed, uint32_t thr24, float dscale,
                                                       const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t m = i / C;
        const int b = (int)(m / T), t = (int)(m - (size_t)b * T);
        const bool gone = tb && t >= tb[2 * b] && t < tb[2 * b + 1];
        y[i] = (!gone && sed_keep((uint32_t)i, seed, thr24)) ? x[i] * dscale : 0.f;
    }
}
SED_API int sed_dropstep(const float* x, float* y, const int* bounds, int B, int T, int C, unsigned seed, unsigned thr24,
                            float dscale, const unsigned* seed_dev, void* stream) {
    if (B <= 0 || T <= 0 || C <= 0) return SED_OK;
    const size_t n = (size_t)B * T * C;
    if (n >= (1ull << 32)) return SED_ERR_UNSUPPORTED;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    SED_LAUNCH(dropstep_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, bounds, T, C, n, seed, thr24, dscale, seed_dev);
    return sed_check_launch();
}
