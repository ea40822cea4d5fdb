// K13: inference post-processing of the frame-level posteriors on the device, replacing the per-clip
// `.cpu().numpy()` loop of recipes/dcase2023_task4_baseline/local/utils.py:16-73 (batched_decode_preds):
//   * sed_median_filter    : scipy.ndimage.median_filter(scores (T, NC), size=(win, 1)) for every clip of the batch --
//                            median over time with scipy's default 'reflect' boundary (d c b a | a b c d | d c b a),
//                            window centred with origin 0 (covers t - win/2 .. t + (win-1)/2); utils.py:55;
//   * sed_threshold_events : `scores > threshold` (utils.py:62) followed by the contiguous-region search that
//                            ManyHotEncoder.decode_strong (desed_task/utils/encoder.py:189-211) delegates to
//                            dcase_util's DecisionEncoder.find_contiguous_regions: [onset_frame, offset_frame) pairs per
//                            (threshold, clip, class), in frame units; the host only converts frames to seconds.
// Layout: scores (B, T, NC) frame-major, the native layout of the head kernel (the reference's (B, NC, T) tensor is a
// transposed view of it).  Integer / selection work: results are bit-exact.
#include "sed_common.h"

#define POST_MAX_WIN 15

// index of the reflected sample: period 2T, ... 2 1 0 | 0 1 2 ... T-1 | T-1 T-2 ...
__device__ __forceinline__ int reflect_index(int i, int T) {
    const int p = 2 * T;
    i %= p;
    if (i < 0) i += p;
    return i < T ? i : p - 1 - i;
}

__global__ __launch_bounds__(256) void median_filter_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int T,
                                                            int NC, int win) {
    const size_t n = (size_t)B * T * NC;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % NC), t = (int)((e / NC) % T);
        const size_t b = e / ((size_t)NC * T);
        const float* col = x + b * T * NC + c;
        float v[POST_MAX_WIN];
#pragma unroll
        for (int k = 0; k < POST_MAX_WIN; ++k)
            if (k < win) v[k] = col[(size_t)reflect_index(t - win / 2 + k, T) * NC];
        // insertion sort of <= 15 values (fully unrolled: the array stays in registers); NaNs are not expected in posteriors
#pragma unroll
        for (int a = 1; a < POST_MAX_WIN; ++a) {
            if (a < win) {
#pragma unroll
                for (int b2 = POST_MAX_WIN - 1; b2 >= 1; --b2) {
                    if (b2 <= a) {
                        const float lo = fminf(v[b2 - 1], v[b2]), hi = fmaxf(v[b2 - 1], v[b2]);
                        v[b2 - 1] = lo; v[b2] = hi;
                    }
                }
            }
        }
        float m = v[0];
#pragma unroll
        for (int k = 0; k < POST_MAX_WIN; ++k)
            if (k == win / 2) m = v[k];
        y[e] = m;
    }
}

// scores (B,T,NC) -> out (B,T,NC); win odd or even, 1 <= win <= 15 (scipy picks element win/2 of the sorted window).
SED_API int sed_median_filter(const float* scores, float* out, int B, int T, int NC, int win, void* stream) {
    if (win < 1 || win > POST_MAX_WIN) return SED_ERR_UNSUPPORTED;
    if (B <= 0 || T <= 0 || NC <= 0) return SED_OK;
    const size_t n = (size_t)B * T * NC;
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    SED_LAUNCH(median_filter_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, scores, out, B, T, NC, win);
    return sed_check_launch();
}

// One thread per (threshold, clip, class): walks the frames once and appends [onset, offset) pairs.
// counts (n_thr, B, NC) int32; events (n_thr, B, NC, max_events, 2) int32.  true_len (B) int32 or null: frames past it are
// ignored (utils.py:48-50, padded clips).  A (clip, class) column cannot hold more than (T + 1) / 2 regions.
__global__ __launch_bounds__(256) void threshold_events_kernel(const float* __restrict__ scores, const float* __restrict__ thr,
                                                               const int* __restrict__ true_len, int* __restrict__ counts,
                                                               int* __restrict__ events, int B, int T, int NC, int n_thr,
                                                               int max_events) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_thr * B * NC) return;
    const int c = idx % NC, b = (idx / NC) % B, k = idx / (NC * B);
    const float th = thr[k];
    const int len = true_len ? min(max(true_len[b], 0), T) : T;
    const float* col = scores + (size_t)b * T * NC + c;
    int* ev = events + (size_t)idx * max_events * 2;
    int n = 0, onset = -1;
    for (int t = 0; t < len; ++t) {
        const bool on = col[(size_t)t * NC] > th;
        if (on && onset < 0) onset = t;
        if (!on && onset >= 0) {
            if (n < max_events) { ev[2 * n] = onset; ev[2 * n + 1] = t; }
            ++n; onset = -1;
        }
    }
    if (onset >= 0) {
        if (n < max_events) { ev[2 * n] = onset; ev[2 * n + 1] = len; }
        ++n;
    }
    counts[idx] = n;
}

SED_API int sed_threshold_events(const float* scores, const float* thresholds, const int* true_len, int* counts, int* events,
                                    int B, int T, int NC, int n_thr, int max_events, void* stream) {
    if (n_thr <= 0 || max_events < (T + 1) / 2) return SED_ERR_ARG;
    if (B <= 0 || T <= 0 || NC <= 0) return SED_OK;
    const int n = n_thr * B * NC;
    SED_LAUNCH(threshold_events_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, scores, thresholds, true_len, counts,
               events, B, T, NC, n_thr, max_events);
    return sed_check_launch();
}
