/* C ABI of libsed_hip.so -- the MI355X (gfx950) kernels behind the DESED_task mel + CRNN
 * mean-teacher training step.
 *
 * The reference (DCASE-REPO/DESED_task) is pure Python with no FFI of its own; the boundary a
 * maintainer binds is therefore this header (ctypes stub: INTEGRATION.md).  Each entry point names
 * the reference call site it replaces (paths relative to the reference repository).
 *
 * Conventions: every function returns 0 (SED_OK) or a negative error code, never throws, never
 * allocates or frees -- the caller (PyTorch) owns all buffers, including workspaces -- and is
 * asynchronous on `stream` (a hipStream_t passed as void*).  Pointers are device pointers.
 * Re-entrant across streams; no global mutable state and no reads of the process environment (the only process-wide
 * setting is the explicit tuning override of sed_set_tuning, used by tests and sweep tools).
 * Activation layout is channels-last: (B, T, F, C) fp32, T = time frames, F = mel bins.
 */
#ifndef SED_HIP_H
#define SED_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* K1: torchaudio MelSpectrogram(n_fft=2048, hop=256, hamming(periodic=False), power=1, center, reflect)
 * built at recipes/dcase2023_task4_baseline/local/sed_trainer.py:80-91, called :282.
 * audio (B,N) -> out (B,T,n_mels), T = 1 + N/hop.  Tables are host-built once: window[2048],
 * tw1024[1024] complex exp(-2 pi i n/1024), tw2048[1024] complex exp(-2 pi i k/2048), sparse filterbank
 * (start bin, length, weights[n_mels][fb_stride]).  apply_log=1 fuses take_log (:253-264). */
int sed_mel_fwd(const float* audio, float* out, int B, int N, int T, int n_fft, int hop, int n_mels,
                const float* window, const float* tw1024, const float* tw2048, const int* fb_start,
                const int* fb_len, const float* fb_w, int fb_stride, int apply_log, void* stream);

/* K1, round 5: the same transform with one WAVE per frame (1024-point packed FFT as 16 x 16 x 4 in registers, two wave-private LDS
 * exchanges, no workgroup barrier in the frame loop; runs of consecutive frames of a clip stay on one XCD).  `taps`: the filterbank in
 * the kernel's LDS layout, (4 + 12) * 64 * 4 floats written once by sed_mel_taps from the same sparse filterbank arrays.  Same
 * reference call site as sed_mel_fwd (local/sed_trainer.py:80-91, :282), which remains as the generic one-frame-per-workgroup form. */
int sed_mel_taps(const int* fb_start, const int* fb_len, const float* fb_w, int fb_stride, int n_mels, float* taps, void* stream);
int sed_mel_fwd_wave(const float* audio, float* out, int B, int N, int T, int n_fft, int hop, int n_mels,
                     const float* window, const float* tw1024, const float* tw2048, const int* fb_start,
                     const int* fb_len, const float* fb_w, int fb_stride, const float* taps, int apply_log, void* stream);

/* K3+K4: SEDTask4.take_log (sed_trainer.py:253-264) + TorchScaler("instance","minmax") forward
 * (desed_task/utils/scaler.py:114-120) on (B, L) clips.  partial: B*64 floats scratch; minmax: optional (B,2). */
int sed_logscale_fwd(const float* x, float* logbuf, float* out, float* partial, float* minmax, int B, int L,
                     int apply_log, float eps, void* stream);

/* K3 alone: amp_to_db(mels).clamp(-50, 80) with amin = 1e-5 (sed_trainer.py:262-264). */
int sed_take_log(const float* x, float* y, long long n, void* stream);

/* ---- step-varying arguments in device memory (hipGraph replay) ---------------------------------------------------
 * A captured training step freezes every by-value launch argument.  The entry points whose arguments change from
 * step to step therefore take one extra, NULLABLE device pointer that overrides the by-value argument when set:
 *   seed_dev   (sed_glu_*, sed_head_*) : *seed_dev is added to `seed` (wrapping) -- the dropout entropy of this step;
 *   c_dev      (sed_mixup)             : {c, 1-c} (the sentinel c = 2 makes the launch a no-op);
 *   weight_dev (sed_mt_loss)           : consistency-loss weight;
 *   alpha_dev  (sed_ema_update)        : {alpha, 1-alpha};
 *   hyper_dev  (sed_adam_step)         : {step_size, inv_bc2_sqrt}.
 * Null pointers give the plain eager behaviour.  Host side: desed_task_amd/graph.py. */

/* K2: desed_task/data_augm.py:31-51 mixup on a group of n clips of L floats, in place (tmp = scratch copy).
 * mode 0 = features, 1 = soft labels (clamp 0..1), 2 = hard labels. */
int sed_mixup(float* data, float* tmp, const int* perm, float c, float one_minus_c, int n, int L, int mode,
              const float* c_dev, void* stream);

/* K2, batched: up to 8 mixup jobs (e.g. weak features + weak labels + strong features + strong labels of one training step,
 * sed_trainer.py:294-301) in ONE launch and without scratch copies: a thread owns one position of a clip block and all n <= 32
 * clips of the group -- it reads every clip's value and its partner's before it writes any.  `jobs` is a HOST array of njobs
 * records of 8 x int64, consumed during the call: {data (device address), perm (device address of n int32), c_dev (device address
 * of {c, 1-c}, or 0), bit pattern of float c, bit pattern of float 1-c, n, L, mode}.  Jobs of one call must not overlap in memory. */
int sed_mixup_multi(const long long* jobs, int njobs, void* stream);

/* K5: the two axis masks of CRNN.apply_specaugment (desed_task/nnet/CRNN.py:207-219) on (B,T,F);
 * bounds (B,4) int32 = [f0,f1,t0,t1). */
int sed_specaug(const float* x, float* y, const int* bounds, int B, int T, int Fq, void* stream);

/* The mask draws of CRNN.apply_specaugment (desed_task/nnet/CRNN.py:207-219, torchaudio mask_along_axis[_iid]) from uniform
 * numbers: u_f / u_t (2, n) per axis (row 0 -> length, row 1 -> start; null = axis off), n = B (per-clip masks) or 1;
 * bounds (B,4) int32 = [f0, f1, t0, t1).  float32 arithmetic identical to the reference's tensor ops. */
int sed_specaug_bounds(const float* u_f, const float* u_t, int* bounds, int B, int n, int f_param, int n_freq, int t_param,
                       int n_time, void* stream);

/* The same draws from a counter-based generator keyed by `seed` (+ *seed_dev when non-null: hipGraph replays) instead of uniform
 * tensors: u_k(clip i) = (sed_hash(4 i + k, seed) >> 8) / 2^24, k = 0..3 = frequency length, frequency start, time length, time
 * start; f_param / t_param < 1 switch an axis off.  Same float32 mask arithmetic. */
int sed_specaug_bounds_seeded(int* bounds, int B, int n, int f_param, int n_freq, int t_param, int n_time, unsigned seed,
                              const unsigned* seed_dev, void* stream);

/* labels_weak = (sum over frames of labels (n,NC,T) > 0) as float (n,NC) (sed_trainer.py:292). */
int sed_weak_labels(const float* labels, float* out, int n, int NC, int T, void* stream);

/* ---- K6: CNN block (desed_task/nnet/CNN.py:66-98), channels-last (B,T,F,C) ---------------------------------- */

/* nn.Conv2d weight (COUT,CIN,3,3) -> packed Wf[9][CIN][COUT] (forward) and Wd[9][COUT][CIN] (data gradient:
 * flipped taps, transposed channels); Wd may be null. */
int sed_conv_pack_weights(const float* W, float* Wf, float* Wd, int COUT, int CIN, void* stream);

/* The same for up to 8 layers in one launch.  W/Wf/Wd: HOST arrays of n device pointers (Wd or its entries may be null). */
int sed_conv_pack_multi(int n, const void* const* W, void* const* Wf, void* const* Wd, const int* cout, const int* cin,
                        void* stream);

/* Number of workgroups a forward conv launch writes partial statistics for: `partial` is [2*COUT][nblocks] floats
 * (row c = per-workgroup sums of channel c, row COUT + c = sums of squares), the layout sed_bn_finalize reads. */
int sed_conv_fwd_blocks(int B, int T, int F, int CIN, int COUT);

/* Conv2d(k=3,s=1,p=1) (CNN.py:69-72) as implicit GEMM on f32 MFMA.  x (B,T,F,CIN), Wp packed, bias or null,
 * y (B,T,F,COUT); partial (or null) receives per-workgroup (sum, sumsq) per channel for BatchNorm.
 * The same entry computes the data gradient when given Wd and the output gradient as x. */
int sed_conv3x3(const float* x, const float* Wp, const float* bias, float* y, float* partial, int B, int T, int F,
                int CIN, int COUT, void* stream);

/* Split-bf16 ("bf16x3") variant of the two entries above: weights pre-split into bf16 hi/lo planes (buffers of the same
 * byte size as the fp32 packs), the convolution on v_mfma_f32_32x32x16_bf16 with three MFMAs per product
 * (hi*hi + hi*lo + lo*hi): fp32-level accuracy (~8e-6 relative) at 3/16 of the f32 MFMA cost.  Same tensors/contract. */
int sed_conv_pack_multi_bf16(int n, const void* const* W, void* const* Wf, void* const* Wd, const int* cout,
                             const int* cin, void* stream);
int sed_conv3x3_bf16x3(const float* x, const void* Wp, const float* bias, float* y, float* partial, int B, int T, int F,
                       int CIN, int COUT, void* stream);
/* The prologue of one CNN forward (desed_task/nnet/CRNN.py:207-232: apply_specaugment, then self.cnn) in ONE launch: the packs of
 * sed_conv_pack_multi_bf16 (n = 0: none) + the SpecAugment bands of sed_specaug_bounds_seeded (bounds = null: none) + a private
 * copy of copy_n floats copy_src -> copy_dst (null: none; both 16-byte aligned) for a caller whose input buffer is rewritten
 * before the backward pass reads it. */
int sed_cnn_prologue_bf16(int n, const void* const* W, void* const* Wf, void* const* Wd, const int* cout, const int* cin,
                          int* bounds, int B, int nb, int f_param, int n_freq, int t_param, int n_time, unsigned seed,
                          const unsigned* seed_dev, const float* copy_src, float* copy_dst, long long copy_n, void* stream);

/* The data gradient of a TRAINING-mode block with its BatchNorm backward (desed_task/nnet/CNN.py:76) folded into the operand
 * staging: dz (B,T,F,CIN) = dL/d(xhat) from sed_glu_bwd, ybn = the block's pre-BN conv output, stats = mean | invstd,
 * dgamma / dbeta = the finished BatchNorm parameter gradients; the kernel forms dy = invstd (dz - gamma dbeta / n - xhat gamma
 * dgamma / n) -- sed_bn_bwd_apply's expression -- on the fly, convolves it with the data-gradient pack Wd into dx (B,T,F,COUT) and
 * writes dy to dy_out (B,T,F,CIN; must not alias dz) for sed_conv_wgrad_bf16x3; dbias[CIN] (nullable) = 0. */
int sed_conv3x3_bf16x3_bnbwd(const float* dz, const float* ybn, const float* stats, const float* gamma, const float* dgamma,
                             const float* dbeta, const void* Wd, float* dx, float* dy_out, float* dbias, int B, int T, int F,
                             int CIN, int COUT, void* stream);
int sed_conv_fwd_blocks_bf16(int B, int T, int F, int CIN, int COUT);   /* rows of `partial` for the bf16x3 forward */

/* Layer 0 (CIN=1): direct conv with the SpecAugment predicate (CRNN.py:207-219) fused into the load.
 * x (B,T,F) scaled log-mel; W (16,1,3,3) PyTorch layout; bounds (B,4) int32 [f0,f1,t0,t1) or null.  y may be null: only the
 * BatchNorm partial statistics are produced (first pass of the fused first block below). */
int sed_conv0_fwd(const float* x, const float* W, const float* bias, const int* bounds, float* y, float* partial,
                  int B, int T, int F, int COUT, void* stream);

/* ---- First block without its pre-BatchNorm tensor in HBM (CNN.py:66-98 for n_in_channel = 1, 16 filters, pooling (2,2)) ----
 * The 246 MB conv output of block 0 (B = 48) is never written: it is recomputed from the 15 MB input wherever it is needed.
 *   training forward : sed_conv0_fwd(y = NULL, partial) -> sed_bn_finalize -> sed_block0_fwd
 *   eval forward     : sed_bn_finalize(training = 0)    -> sed_block0_fwd
 *   training backward: sed_block0_bwd (all six parameter gradients of the block in one pass over x and gout)
 * sed_block0_fwd: conv0 (SpecAugment predicate fused, as sed_conv0_fwd) + BN-apply + GLU + Dropout + AvgPool(2,2):
 * x (B,T,F), W (16,1,3,3), stats from sed_bn_finalize -> out (B,T/2,F/2,16).  F % 8 == 0, F <= 128.  Dropout as sed_glu_fwd
 * (element index of the (B,T,F,16) tensor: identical masks to the unfused path). */
int sed_block0_fwd(const float* x, const float* W, const float* bias, const int* bounds, const float* stats, const float* Wg,
                   const float* bg, float* out, int B, int T, int F, unsigned seed, unsigned thr24, float dscale,
                   const unsigned* seed_dev, void* stream);
/* Floats of scratch sed_block0_bwd needs (one partial record per workgroup + the reduced sums). */
long long sed_block0_bwd_scratch_floats(int B, int T, int F);
/* Backward of the whole first block under training-mode BatchNorm: gout (B,T/2,F/2,16) -> dW (16,1,3,3), dbias (16; analytically
 * zero), dgamma, dbeta (16), dWg (16,16), dbg (16), all overwritten.  The weight gradient is assembled from raw correlations
 * accumulated in the same pass (see sed_block0.hip); nothing of size B*T*F*16 is read or written. */
int sed_block0_bwd(const float* x, const float* W, const float* bias, const int* bounds, const float* stats, const float* gamma,
                   const float* beta, const float* Wg, const float* bg, const float* gout, float* dW, float* dbias, float* dgamma,
                   float* dbeta, float* dWg, float* dbg, float* scratch, int B, int T, int F, unsigned seed, unsigned thr24,
                   float dscale, const unsigned* seed_dev, void* stream);

/* BatchNorm2d(eps=1e-3, momentum=0.99) statistics (CNN.py:76): reduce the partials (training) or read the
 * running stats (eval); stats = [mean | invstd | scale | shift] (4*C); updates running stats when asked. */
int sed_bn_finalize(const float* partial, int nblocks, int C, float count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, float* stats, int training,
                    int update_running, void* stream);

/* BN-apply + GLU (CNN.py:11-16) + Dropout (:90-91) + AvgPool2d (:96-98), fused.  y (B,T,F,C) -> out (B,T/PT,F/PF,C).
 * Dropout keeps element e iff (hash(e,seed)>>8) >= thr24; dscale = 1/(1-p).
 * split_bf16 != 0: the C x C gate linear of the 64/128-channel blocks runs on the split-bf16 MFMA (fp32-level accuracy). */
int sed_glu_fwd(const float* y, const float* stats, const float* Wg, const float* bg, float* out, int B, int T, int F,
                int C, int PT, int PF, unsigned seed, unsigned thr24, float dscale, const unsigned* seed_dev, int split_bf16,
                void* stream);

/* Floats of scratch sed_glu_bwd needs (per-workgroup partial sums, reduced in a fixed order; 0 = none). */
long long sed_glu_bwd_scratch_floats(int B, int T, int F, int C, int PT, int PF);

/* Backward of the above: gout -> dz = dL/d(xhat) (B,T,F,C) and dWg (C,C), dbg, dgamma, dbeta (C) (overwritten). */
int sed_glu_bwd(const float* y, const float* stats, const float* gamma, const float* beta, const float* Wg,
                const float* bg, const float* gout, float* dz, float* dWg, float* dbg, float* dgamma, float* dbeta,
                float* scratch, int B, int T, int F, int C, int PT, int PF, unsigned seed, unsigned thr24, float dscale,
                const unsigned* seed_dev, int split_bf16, void* stream);

/* BatchNorm backward apply in place: dz -> dy = dL/d(conv output); dbias (C) = conv-bias gradient. */
int sed_bn_bwd_apply(const float* y, float* dz, const float* stats, const float* gamma, const float* dgamma,
                     const float* dbeta, float* dbias, long long npix, int C, int training, void* stream);

/* Floats of scratch sed_conv_wgrad needs (per-workgroup partial gradients, reduced in a fixed order). */
long long sed_conv_wgrad_scratch_floats(int B, int T, int F, int CIN, int COUT);

/* Conv weight gradient: x (B,T,F,CIN), dy (B,T,F,COUT) -> dW (COUT,CIN,3,3); dWp = scratch (see above). */
int sed_conv_wgrad(const float* x, const float* dy, float* dWp, float* dW, int B, int T, int F, int CIN, int COUT,
                   void* stream);

/* Same contract with the pixel contraction of the wide layers (CIN >= 64) on the split-bf16 MFMA (x = hi + lo in bf16, three
 * bf16 MFMAs per product, fp32 accumulate: ~8e-6 relative). */
int sed_conv_wgrad_bf16x3(const float* x, const float* dy, float* dWp, float* dW, int B, int T, int F, int CIN, int COUT,
                          void* stream);

/* Layer-0 weight gradient: x (B,T,F) (+ SpecAugment bounds or null) -> dW (16,1,3,3).  fuse_bn = 0: dyz = dy (B,T,F,16).
 * fuse_bn = 1 (training mode): dyz = dz from sed_glu_bwd and the BatchNorm backward (sed_bn_bwd_apply) is applied while
 * loading (y, stats, gamma, dgamma, dbeta as there); dbias (16) receives the (zero) conv-bias gradient. */
int sed_conv0_wgrad(const float* x, const int* bounds, const float* dyz, const float* y, const float* stats,
                    const float* gamma, const float* dgamma, const float* dbeta, float* dW, float* dbias, int B, int T,
                    int F, int COUT, int fuse_bn, int training, void* stream);

/* ---- K7: bidirectional GRU (desed_task/nnet/RNN.py:19-30) ---------------------------------------------------- */

/* Plain GEMM on f32 MFMA: C[M][N] = opA(A) * opB(B) + bias[N].  transA: A stored [K][M]; transB: B stored [N][K].
 * split_k > 1 or accumulate != 0 adds atomically into C (caller zeroes C for split_k).  ld* in floats. */
int sed_gemm(const float* A, const float* Bm, const float* bias, float* Cm, int M, int N, int K, int lda, int ldb,
             int ldc, int transA, int transB, int split_k, int accumulate, void* stream);

/* Two same-shape GEMMs (the two GRU directions) in one launch: (A0,B0,bias0 -> C0), (A1,B1,bias1 -> C1). */
int sed_gemm_pair(const float* A0, const float* A1, const float* B0, const float* B1, const float* bias0,
                  const float* bias1, float* C0, float* C1, int M, int N, int K, int lda, int ldb, int ldc, int transA,
                  int transB, int split_k, int accumulate, void* stream);

/* The same two contracts with split-bf16 products (x = hi + lo in bf16, three bf16 MFMAs per product, fp32 accumulate:
 * ~8e-6 relative, fp32 in/out).  Falls back to the exact-f32 kernels when an operand is not 16-byte aligned. */
int sed_gemm_bf16x3(const float* A, const float* Bm, const float* bias, float* Cm, int M, int N, int K, int lda, int ldb,
                    int ldc, int transA, int transB, int split_k, int accumulate, void* stream);
/* C[M][N] = A[M][K] . [B0 ; B1] (row-major, no transposes): B is two tensors stacked along K, rows [0, ksplit) from B0 and
 * the rest from B1, ksplit % 32 == 0 -- dX of a bidirectional GRU layer.  16-byte aligned operands. */
int sed_gemm_kcat(const float* A, const float* B0, const float* B1, float* Cm, int M, int N, int K, int ksplit, int lda,
                  int ldb, int ldc, void* stream);
int sed_gemm_kcat_bf16x3(const float* A, const float* B0, const float* B1, float* Cm, int M, int N, int K, int ksplit, int lda,
                         int ldb, int ldc, void* stream);
/* sed_gemm_kcat_bf16x3 with a deterministic split-K: the slices go to `scratch` (sed_gemm_splitk_scratch_floats(M, N, K, split_k)
 * floats) as dense partials and are summed in slice order -- no atomics, C needs no zero fill.  C 16-byte aligned, N % 4 == 0. */
int sed_gemm_kcat_splitk_bf16x3(const float* A, const float* B0, const float* B1, float* Cm, int M, int N, int K, int ksplit,
                                int lda, int ldb, int ldc, int split_k, float* scratch, void* stream);
int sed_gemm_pair_bf16x3(const float* A0, const float* A1, const float* B0, const float* B1, const float* bias0,
                         const float* bias1, float* C0, float* C1, int M, int N, int K, int lda, int ldb, int ldc,
                         int transA, int transB, int split_k, int accumulate, void* stream);
/* The same pair with a deterministic split-K: slices land as dense partials in `scratch` (sed_gemm_splitk_scratch_floats() floats)
 * and are summed in slice order; C0 / C1 are overwritten (no zero fill, no atomics).  The weight gradients of a BiGRU layer
 * (torch.nn.GRU backward, desed_task/nnet/RNN.py:19-30): dW_ih[d] = dgi[d]^T x, dW_hh[d] = dgh[d]^T h_prev[d]. */
long long sed_gemm_splitk_scratch_floats(int M, int N, int K, int split_k);
int sed_gemm_pair_splitk_bf16x3(const float* A0, const float* A1, const float* B0, const float* B1, float* C0, float* C1,
                                int M, int N, int K, int lda, int ldb, int ldc, int transA, int transB, int split_k,
                                float* scratch, void* stream);
/* One product with the same deterministic split-K (`cat_tf` weight gradient of the embedding recipes, CRNN.py:283-296 backward:
 * dW = dy^T . [x | emb] over K = B T rows).  scratch: sed_gemm_splitk_scratch_floats(M, N, K, split_k) floats. */
int sed_gemm_splitk_bf16x3(const float* A, const float* Bm, float* Cm, int M, int N, int K, int lda, int ldb, int ldc,
                           int transA, int transB, int split_k, float* scratch, void* stream);

/* Column sums (bias gradients): out[n] = sum_m X[m*ld + n] for n < nsplit, out1[n-nsplit] for nsplit <= n < N. */
int sed_colsum(const float* X, float* out, float* out1, int nsplit, int M, int N, int ld, void* stream);

/* GRU recurrence of one layer, both directions: gi (B,T,2,3H) = W_ih x + b_ih per direction; whh0/whh1 (3H,H),
 * bhh0/bhh1 (3H) = forward / reverse direction; out (B,T,2H); saved (B,T,2,4,H) = r,z,n,hn or null.
 * H = 128 (2023 recipe) or 192 (2024 recipe's n_RNN_cell); other widths return SED_ERR_UNSUPPORTED. */
int sed_gru_fwd(const float* gi, const float* whh0, const float* whh1, const float* bhh0, const float* bhh1,
                float* out, float* saved, int B, int T, int H, void* stream);

/* Backward recurrence: dout (B,T,2H) -> dgi, dgh (B,T,2,3H) and hprev (B,T,2,H).  dbi0/dbi1, dbh0/dbh1 (3H each, forward /
 * reverse direction; either pair may be null) receive the bias gradients = column sums of dgi / dgh (overwritten): every (clip,
 * direction) writes its own record into `scratch` (2 B x 6 H floats, required when any bias gradient is asked for) and a small
 * kernel adds the clips in order -- deterministic, no atomics. */
int sed_gru_bwd(const float* dout, const float* out, const float* saved, const float* whh0, const float* whh1,
                float* dgi, float* dgh, float* hprev, float* dbi0, float* dbi1, float* dbh0, float* dbh1, int B, int T, int H,
                float* scratch, void* stream);
/* The bias-gradient half of sed_gru_bwd on its own: sed_gru_bwd with all four bias pointers null and a non-null scratch only
 * leaves the records; this sums them (same order, same bits).  Nothing on the backward chain reads the bias gradients -- the
 * optimizer does -- so the launcher runs this beside the chain, next to the weight-gradient GEMMs. */
int sed_gru_bias_reduce(const float* scratch, float* dbi0, float* dbi1, float* dbh0, float* dbh1, int B, int H, void* stream);

/* ---- K8 + K9: attention-pooling head (desed_task/nnet/CRNN.py:152-178, dropout :304) and losses ---------------- */

/* x (B,T,D), D = 256 or 384 (= 2 * n_RNN_cell), NC = 10 or 27 -> strong (B,T,NC) = sigmoid(dense), psoft (B,T,NC) = softmax over classes of dense_softmax,
 * weak (B,NC) = sum_t(strong*clamp(psoft)) / sum_t(clamp(psoft)), den (B,NC) = the denominators.
 * classes_valid (B,NC) bytes or null: the `classes_mask` of the multi-data-set recipes (CRNN.py:157-176; non-zero = the clip's
 * data set annotates the class): other classes cannot be attended to and their strong / weak outputs are 0.  pad_mask (B,T)
 * bytes or null (non-zero = padded frame): its attention logits are filled with -1e30 (CRNN.py:161-162). */
int sed_head_fwd(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, float* strong,
                 float* psoft, float* weak, float* den, int B, int T, int D, int NC, unsigned seed, unsigned thr24,
                 float dscale, const unsigned* seed_dev, const unsigned char* classes_valid, const unsigned char* pad_mask,
                 void* stream);

int sed_head_bwd(const float* x, const float* W1, const float* W2, const float* strong, const float* psoft,
                 const float* weak, const float* den, const float* d_strong, const float* d_weak, float* dx, float* dW1,
                 float* dW2, float* db1, float* db2, int B, int T, int D, int NC, unsigned seed, unsigned thr24,
                 float dscale, const unsigned* seed_dev, const unsigned char* classes_valid, const unsigned char* pad_mask,
                 float* scratch, void* stream);
/* floats of `scratch` for sed_head_bwd: one partial record (dW1 | dW2 | db1 | db2) per workgroup, summed in a fixed order. */
long long sed_head_bwd_scratch_floats(int B, int T, int D, int NC);
/* sed_head_bwd with dW1 = dW2 = db1 = db2 = null computes dx and leaves the records in `scratch`; this sums them (the second
 * half of sed_head_bwd, same bits) -- off the backward chain, like sed_gru_bias_reduce. */
int sed_head_bwd_reduce(const float* scratch, float* dW1, float* dW2, float* db1, float* db2, int B, int T, int D, int NC,
                        void* stream);

/* Mean-teacher losses of SEDTask4.training_step (recipes/dcase2023_task4_baseline/local/sed_trainer.py:309-342):
 * scalars[9] = BCE strong/weak (student), BCE strong/weak (teacher), MSE strong/weak, weight*(MSE_s + MSE_w), total, total again;
 * g_strong (B,T,NC), g_weak (B,NC)
 * = d(BCE_s + BCE_w + weight*(MSE_s + MSE_w)) / d(student outputs).  labels (B,NC,T); labels_weak (n_weak,NC).
 * selfsup_bce != 0: the two consistency terms are BCELoss(student, teacher) instead of MSELoss (`self_sup_loss: bce`, :99-100).
 * selfsup_from: the consistency terms average over clips [selfsup_from, B) (0 in the 2023 recipe; the 2024 recipe leaves its
 * MAESTRO clips out, dcase2024 local/sed_trainer_pretrained.py:337,399-406).  valid (B,NC) bytes or null: labels of classes a
 * clip's data set does not annotate count as 0 (:352-356).
 * work: 8 B + 1 floats of caller-owned scratch (per-clip partial sums + a ticket word); the ticket word work[8 B] must be 0 before
 * the first call and is 0 again after every call, so one zero-initialised buffer serves all later calls on the same stream.  The
 * sums are added in clip order by the last workgroup: deterministic, no zero-fill launch. */
int sed_mt_loss(const float* strong_s, const float* weak_s, const float* strong_t, const float* weak_t,
                const float* labels, const float* labels_weak, float* scalars, float* g_strong, float* g_weak, int B,
                int T, int NC, int n_strong, int n_weak, float weight, const float* weight_dev, int selfsup_bce,
                int selfsup_from, const unsigned char* valid, float* work, void* stream);
/* sed_mt_loss in two halves.  sed_mt_loss_records: the gradient seeds g_strong / g_weak (all the backward pass needs) and the per-clip
 * records in work[0 .. 8 B) -- no fence, no ticket, `scalars` untouched; sed_mt_loss_finish: scalars[0 .. 8] from the records, in clip
 * order (the same bits as the one-call form), on any stream ordered after the first half.  The launcher runs the second half beside
 * the backward chain, next to the EMA. */
int sed_mt_loss_records(const float* strong_s, const float* weak_s, const float* strong_t, const float* weak_t,
                        const float* labels, const float* labels_weak, float* scalars, float* g_strong, float* g_weak, int B,
                        int T, int NC, int n_strong, int n_weak, float weight, const float* weight_dev, int selfsup_bce,
                        int selfsup_from, const unsigned char* valid, float* work, void* stream);
int sed_mt_loss_finish(const float* work, float* scalars, int B, void* stream);

/* ---- K13: inference post-processing, recipes/dcase2023_task4_baseline/local/utils.py:16-73 ----- */

/* scipy.ndimage.median_filter(scores (T,NC), size=(win,1)) per clip (utils.py:55): median over time, 'reflect' boundary,
 * scores / out (B,T,NC) frame-major, 1 <= win <= 15.  Bit-exact (selection). */
int sed_median_filter(const float* scores, float* out, int B, int T, int NC, int win, void* stream);

/* `scores > thresholds[k]` (utils.py:62) + the contiguous-region search of ManyHotEncoder.decode_strong
 * (desed_task/utils/encoder.py:189-211): counts (n_thr,B,NC) int32 regions per (threshold, clip, class) and
 * events (n_thr,B,NC,max_events,2) int32 [onset_frame, offset_frame) pairs; max_events >= (T+1)/2.
 * true_len (B) int32 or null: frames at or past it are ignored (utils.py:48-50). */
int sed_threshold_events(const float* scores, const float* thresholds, const int* true_len, int* counts, int* events,
                         int B, int T, int NC, int n_thr, int max_events, void* stream);

/* ---- K14: embedding fusion before the recurrent stage, desed_task/nnet/CRNN.py:283-296 -------- */

/* z (B,T,C+E) = dropout(cat(x (B,T,C), adaptive_avg_pool1d(emb (B,E,Te), T)^T)) -- the input of `cat_tf`
 * (aggregation_type "pool1d"); pooling window of frame t = [floor(t*Te/T), ceil((t+1)*Te/T)).  Dropout mask = sed_keep over
 * the element index of z (thr24 = 0: off, dscale = 1); seed_dev as everywhere (null or one step-varying word).
 * tmask (B,4) int32 [x0, x1, e0, e1) or null: `dropstep_recurrent` (CRNN.py:288-294) -- output frames [x0, x1) of the CNN columns
 * and, independently, [e0, e1) of the embedding columns are zeroed before the dropout.  mode 0 = adaptive_avg_pool1d
 * (aggregation_type "pool1d"), 1 = nearest-exact interpolation (aggregation_type "interpolate", :271-279). */
int sed_embcat_fwd(const float* x, const float* emb, float* z, int B, int T, int Te, int C, int E, unsigned seed,
                   unsigned thr24, float dscale, const unsigned* seed_dev, const int* tmask, int mode, void* stream);
/* dx (M,C) = dzx (M,C) masked/scaled with the forward's dropout mask; dzx = the first C columns of dz = dy . W_cat_tf. */
int sed_embcat_bwd(const float* dzx, float* dx, int M, int C, int E, unsigned seed, unsigned thr24, float dscale,
                   const unsigned* seed_dev, const int* tmask, int T, void* stream);

/* `dropstep_recurrent` of a CRNN WITHOUT embeddings (CRNN.py:296-301): y = dropout(time_mask(x)) on (B,T,C); bounds (B,2) int32
 * [t0, t1) or null.  Diagonal operator: the backward is the same call on the gradient. */
int sed_dropstep(const float* x, float* y, const int* bounds, int B, int T, int C, unsigned seed, unsigned thr24, float dscale,
                 const unsigned* seed_dev, void* stream);

/* ---- K10 + K11: flat parameter arena ------------------------------------------------------------------------- */

/* SEDTask4.update_ema (sed_trainer.py:187-199) over the whole arena: teacher = alpha*teacher + (1-alpha)*student. */
int sed_ema_update(float* teacher, const float* student, long long n, float alpha, float one_minus_alpha,
                   const float* alpha_dev, void* stream);

/* torch.optim.Adam step (train_sed.py:199-201) over the whole arena; grad_scale folds in 1/world_size. */
int sed_adam_step(float* p, const float* g, float* m, float* v, long long n, float b1, float b2, float eps,
                  float step_size, float inv_bc2_sqrt, float grad_scale, const float* hyper_dev, void* stream);

/* Zero up to four small accumulator buffers in one launch (null / 0 entries are skipped). */
int sed_zero_buffers(float* p0, long long n0, float* p1, long long n1, float* p2, long long n2, float* p3, long long n3,
                     void* stream);

/* ---- frozen BEATs feature extractor (recipes/dcase2023_task4_baseline/local/beats/, inference only) ---------- */

/* torch.nn.Linear forward with an optional fused activation on the split-bf16 MFMA: C (M,N) = act(A (M,K) . W (N,K)^T + bias),
 * act 0 = none, 1 = exact GELU (backbone.py:279-283: the encoder layers' fc1).  16-byte aligned, K % 4 == 0. */
int sed_linear_bf16x3(const float* A, const float* W, const float* bias, float* Cm, int M, int N, int K, int act, void* stream);

/* Round 5: the same Linear for the large frozen-weight products of the BEATs encoder (backbone.py:214-276 fc1 / fc2, :286-330 the
 * q / k / v / out projections; M = 23 808 tokens at 48 clips).  The weight is split into bf16 hi / lo planes once
 * (sed_pack_weights_bf16x3: Wp = 2 * N * K bf16 bit patterns); 256 x 128 tiles, A fragments straight from HBM into registers, W tiles
 * double-buffered in LDS.  N % 128 == 0, K % 32 == 0. */
int sed_pack_weights_bf16x3(const float* W, unsigned short* Wp, int N, int K, void* stream);
int sed_linear_packed_bf16x3(const float* A, const unsigned short* Wp, const float* bias, float* C, int M, int N, int K, int act,
                             void* stream);

/* Round 6: the same Linear with BOTH operands as pre-split bf16 hi / lo planes in a K-tiled image that a workgroup copies into LDS by
 * LDS-DMA without touching a register (backbone.py:214-330 as above).  sed_split_tiles_bf16x3 turns a k-contiguous fp32 matrix
 * X (R, K) -- an activation (R = M tokens) or a frozen weight (R = N) -- into ceil(R / 256) * (K / 16) blocks of 16 KB: block
 * (r / 256, k / 16) = [hi | lo][256 rows][16 k], the 8-k octet o of row r stored at slot o ^ ((r >> 3) & 1); rows >= R are zero.
 * Xt holds 2 * ceil(R / 256) * 256 * K bf16 bit patterns.  sed_linear_tiles_bf16x3: C (M, N) = act(A . W^T + bias) from two such
 * images, 256 x 256 tiles, one persistent workgroup per CU, four LDS stages filled by DMA three K steps ahead.  N % 256 == 0,
 * K % 16 == 0, 16-byte aligned. */
int sed_split_tiles_bf16x3(const float* X, unsigned short* Xt, int R, int K, void* stream);
int sed_linear_tiles_bf16x3(const unsigned short* At, const unsigned short* Wt, const float* bias, float* C, int M, int N, int K,
                            int act, void* stream);
/* The same product as TWO partial sums over the two halves of K: C2 = [2][M][N], C2[0] = A[:, :K/2] . W[:, :K/2]^T + bias, C2[1] = the
 * other half; their sum is taken by the consumer (sed_layernorm_tiles' x2).  For the N = 768 layers of the encoder (backbone.py:279-283 fc2):
 * 279 output tiles on 256 CUs are two rounds, 558 half-K items three half-rounds.  (K / 16) even; no activation. */
int sed_linear_tiles_split2_bf16x3(const unsigned short* At, const unsigned short* Wt, const float* bias, float* C2, int M, int N, int K,
                                   void* stream);
/* The same product written as the K-tiled image of C (M, N) -- what the NEXT Linear takes as its activation (backbone.py:279-283: fc1's
 * GELU output is only ever read by fc2).  Ct holds 2 * ceil(M / 256) * 256 * N bf16 bit patterns; every row of the padded last panel
 * is written. */
int sed_linear_tiles_out_bf16x3(const unsigned short* At, const unsigned short* Wt, const float* bias, unsigned short* Ct, int M, int N,
                                int K, int act, void* stream);

/* torchaudio.compliance.kaldi.fbank(waveform * 2^15, num_mel_bins, 16 kHz, 25 ms frames, 10 ms shift) with that function's
 * defaults (povey window, pre-emphasis 0.97, DC removal, snip_edges, 512-point FFT, power spectrum, log) followed by
 * (x - norm_mean) * norm_inv -- BEATs.preprocess, BEATs.py:109-133.  audio (B,N) -> out (B, 1 + (N - 400) / 160, n_mels).
 * Host-built tables: window[400], tw[256] complex exp(-2 pi i k / 512), sparse Kaldi-mel bank (start, len, weights). */
int sed_kaldi_fbank(const float* audio, float* out, int B, int N, int n_mels, const float* window, const float* tw,
                    const int* fb_start, const int* fb_len, const float* fb_w, int fb_stride, float norm_mean, float norm_inv,
                    void* stream);

/* The gather of Conv2d(1, E, kernel = stride = P) (BEATs.py:98-104,153-156): fbank (B,M,F) -> patches (B * (M/P) * (F/P), P*P),
 * token order time-major / frequency-fastest; the patch embedding itself is sed_linear_bf16x3 on these rows. */
int sed_patchify(const float* fbank, float* patches, int B, int M, int F, int P, void* stream);

/* y = LayerNorm(alpha * res + x) * gamma + beta over the last dimension D (64 | D, D <= 1024); res may be null.  The post-LN /
 * deep-norm residual of the encoder layers (backbone.py:268-294) and the plain LayerNorms of BEATs.py:157, backbone.py:122. */
int sed_layernorm(const float* x, const float* res, float alpha, const float* gamma, const float* beta, float* y, int M, int D,
                  float eps, void* stream);
/* Round 6: the same LayerNorm, y written twice: fp32 (M, D) (the next sub-layer's residual) and as the K-tiled bf16 hi / lo image of
 * sed_split_tiles_bf16x3 (yt: 2 * ceil(M / 256) * 256 * D bf16 bit patterns; rows >= M are left as they are -- the Linear never stores
 * their products) that sed_linear_tiles_bf16x3 takes as its activation: the post-LN encoder's q / k / v projection reads the image
 * (backbone.py:286-330), the residual path the fp32 copy.  x2 (may be null) is added to x first: the second partial sum of
 * sed_linear_tiles_split2_bf16x3.  256 | D, D <= 1024. */
int sed_layernorm_tiles(const float* x, const float* x2, const float* res, float alpha, const float* gamma, const float* beta, float* y,
                        unsigned short* yt, int M, int D, float eps, void* stream);

/* y = x + GELU(bias + grouped Conv1d(x)): the convolutional position embedding (backbone.py:30-43,118-120; even kernel, padding
 * K/2, last output dropped).  x, y (B,T,D); wt (groups, K, D/groups co, D/groups ci) = the weight-normalised filter transposed
 * on the host once. */
int sed_posconv(const float* x, const float* wt, const float* bias, float* y, int B, int T, int D, int K, int groups, void* stream);
/* The same position convolution on the split-bf16 MFMA (default in beats.py).  wsplit: the weight-normalised filter as bf16 hi | lo
 * planes, (2, groups, K, 48 co, 48 ci), split once on the host (frozen weights). */
int sed_posconv_bf16x3(const float* x, const unsigned short* wsplit, const float* bias, float* y, int B, int T, int D, int K,
                       int groups, void* stream);

/* Multi-head self-attention of MultiheadAttention.forward (backbone.py:446-700, eval mode, no padding mask) on the output of ONE
 * fused q|k|v projection: qkv (B*T, 3*H*64) -> out (B*T, H*64).  relb (H, 2T-1) = relative position bias per offset s - t (the
 * bucket embedding of :390-444 gathered on the host), or null; grep_w (8,64), grep_b (8), grep_a (H) = the gate of :662-682, or
 * null for an ungated bias.  No T x T tensor is written to HBM. */
int sed_attention_relpos(const float* qkv, const float* relb, const float* grep_w, const float* grep_b, const float* grep_a,
                         float* out, int B, int T, int H, int head_dim, void* stream);

/* Tuning overrides for tests / sweep tools / A-B runs (no reference counterpart).  value 0 restores the built-in choice.  Keys:
 *   0  persistent-grid cap of the wide GLU kernels            1  128-channel GLU backward: 1 = 32x32x16 split tiling, 3 = exact f32
 *   2  channels per weight chunk of the split-bf16 conv       3  pixels per workgroup of the split-bf16 conv
 *   4  block-0 backward without the local centring constant  5  128-channel GLU forward: 1 = 32x32x16 tiling
 *   6  narrow weight gradients: 1 = exact-f32 all-taps kernel 7  workgroup cap of the all-taps weight gradients (tests)
 *   8  BEATs attention: 1 = vector-pipe kernel                9  wide weight gradients: 1 = one tap per workgroup
 *  10  KB of LDS a BiGRU recurrence workgroup claims (keeps workgroups of other streams off its CU; 0 = what the kernel needs)
 *  11  mel kernel: 1 = filterbank taps re-read from memory every frame instead of held in registers
 *  12  split-bf16 conv, single-chunk layers (CIN <= 32): n > 0 = n tiles per persistent workgroup, -1 = one tile per workgroup,
 *      0 = built-in choice (grid = the workgroups resident at once)
 *  (key 0 also caps the persistent grid of the block-0 backward kernel: tests)
 * Not for use while kernels are in flight on other threads. */
int sed_set_tuning(int key, int value);

/* Hardware self-test of the MFMA lane maps (no reference counterpart): C = A[M][K] * B[K][M], M = shape (32|16). */
int sed_selftest_mfma(const float* A, const float* Bm, float* C, int K, int shape, void* stream);

#ifdef __cplusplus
}
#endif
#endif
