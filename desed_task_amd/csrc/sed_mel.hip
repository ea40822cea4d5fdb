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

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// WPT > 0: the thread's filterbank taps (every second tap of its band, zero-padded to WPT) live in registers for all its frames.
// With the weights read from memory inside the frame loop the mel stage was a chain of dependent loads -- start, length, then one
// weight per tap, each waited for before the next (tools/isa_exposed_loads.py): ~22 cache round trips per frame on the widest band.
template <bool LOG, int WPT>
__global__ __launch_bounds__(256, 4) void mel_kernel(const float* __restrict__ audio, float* __restrict__ out,
                                                  int B, int N, int T, int hop, int n_mels,
                                                  const float* __restrict__ window, const float2* __restrict__ tw1024,
                                                  const float2* __restrict__ tw2048, const int* __restrict__ fb_start,
                                                  const int* __restrict__ fb_len, const float* __restrict__ fb_w, int fb_stride) {
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
    const int n_frames = B * T;
    // the next frame's samples are fetched into registers while the current frame is transformed (a workgroup would
    // otherwise expose one L2/HBM latency per frame in front of ~10 short barrier-separated phases)
    float2 smp[4];
    auto load_frame = [&](int frame_) {
        const int b_ = frame_ / T, t_ = frame_ - b_ * T;
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
    int frame = blockIdx.x;
    if (frame < n_frames) load_frame(frame);
    for (; frame < n_frames; frame += gridDim.x) {
        const int b = frame / T, t = frame - b * T;
        // ---- reflect-padded, windowed samples packed as z[n] = x[2n] + i x[2n+1] ----
#pragma unroll
        for (int q = 0; q < 4; ++q) buf0[mel_sw(tid + 256 * q)] = make_float2(smp[q].x * win[q].x, smp[q].y * win[q].y);
        if (frame + (int)gridDim.x < n_frames) load_frame(frame + gridDim.x);
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
                v1 = cmul(v1, tw[0]);
                v2 = cmul(v2, tw[Ns]);
                v3 = cmul(v3, tw[2 * Ns]);
            }
            // DFT-4 (forward, e^{-i pi/2} = -i)
            const float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y);
            const float2 a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
            const float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y);
            const float2 a3 = make_float2(v1.x - v3.x, v1.y - v3.y);   // (v1 - v3)
            const int d = ((tid - k) << 2) + k;
            dst[mel_sw(d)] = make_float2(a0.x + a2.x, a0.y + a2.y);
            dst[mel_sw(d + Ns)] = make_float2(a1.x + a3.y, a1.y - a3.x);      // a1 - i a3
            dst[mel_sw(d + 2 * Ns)] = make_float2(a0.x - a2.x, a0.y - a2.y);
            dst[mel_sw(d + 3 * Ns)] = make_float2(a1.x - a3.y, a1.y + a3.x);  // a1 + i a3
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
                const float2 xe = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                const float dr = zk.x - zm.x, di = zk.y + zm.y;            // zk - conj(zm)
                const float2 xo = make_float2(0.5f * di, -0.5f * dr);       // -i/2 * (zk - conj(zm))
                const float2 wx = cmul(tw2[q], xo);
                const float re = xe.x + wx.x, im = xe.y + wx.y;
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
    }
}

SED_API int sed_mel_fwd(const float* audio, float* out, int B, int N, int T, int n_fft, int hop, int n_mels,
                           const float* window, const float* tw1024, const float* tw2048, const int* fb_start,
                           const int* fb_len, const float* fb_w, int fb_stride, int apply_log, void* stream) {
    if (n_fft != MEL_NFFT || n_mels > 128 || n_mels < 1 || N < n_fft / 2 + 1 || T != 1 + N / hop) return SED_ERR_UNSUPPORTED;
    if (B <= 0) return SED_OK;
    int grid = B * T < 4096 ? B * T : 4096;
#define MEL_LAUNCH(LOG, WPT) SED_LAUNCH((mel_kernel<LOG, WPT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, audio, out, B, N, T, hop, \
                                        n_mels, window, (const float2*)tw1024, (const float2*)tw2048, fb_start, fb_len, fb_w, fb_stride)
    if (fb_stride <= 48 && !sed_tuning[SED_TUNE_MEL_TAPS_MEM]) {      // the recipes' filterbank (128 HTK bands up to 8 kHz: widest band 46 bins)
        if (apply_log) MEL_LAUNCH(true, 24); else MEL_LAUNCH(false, 24);
    } else {
        if (apply_log) MEL_LAUNCH(true, 0); else MEL_LAUNCH(false, 0);
    }
#undef MEL_LAUNCH
    return sed_check_launch();
}
