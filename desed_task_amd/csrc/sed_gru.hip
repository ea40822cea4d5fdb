// K7: bidirectional GRU (desed_task/nnet/RNN.py:19-30 = nn.GRU(batch_first, bidirectional); gate order
// r,z,n; n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h0 = 0), forward and backward.
//
// The input projections (all time steps at once), dX and every weight gradient are GEMMs: by default the split-bf16 ones of
// sed_gemm_bf16.hip (three bf16 MFMAs per fp32 product; the dW pairs as deterministic split-K with dense per-slice partials); the
// exact-f32 MFMA GEMMs below (gemm_vec_kernel: 128 x 128 x 32 tiles, 16-byte loads, register prefetch) serve SED_GEMM_PRECISION=f32
// and operands that miss the 16-byte requirements.
// The recurrence itself is latency-bound: 156 dependent steps of a 128 -> 384 matvec.  It runs as one persistent workgroup per
// (clip, direction) -- 96 workgroups at batch 48 -- of 4 H threads (512 at H = 128, 768 at the 2024 recipe's H = 192): thread
// (j, quarter) owns hidden unit j and a QUARTER of the K range, i.e. 3 x H / 4 = 96 W_hh floats in VGPRs at H = 128 (at H = 192, 32 of a
// thread's 144 weights live in LDS as [block][thread] float4); packed v_pk_fma_f32; hardware-rate exp / rcp for the gates; the hidden
// state double-buffered in LDS with the four quarters 144 B apart (one ds_read_b128 touches 16 distinct banks), the quarters combined
// by two DPP quad_perm adds (sed_quad_sum); ONE barrier per step; nothing touches global memory inside the step loop -- gate inputs and
// results are staged through LDS in chunks of 8 steps (4 / 2 at H = 192); s_setprio 3 (measured neutral).  Exact fp32; the state never
// leaves the CU.  The forward saves r, z, n and (W_hn h + b_hn) for the backward recurrence, which mirrors the structure with W_hh^T
// in registers and also accumulates the four bias gradients in registers, off the dependent chain.
#include "sed_common.h"

#define GRU_H 128

// ---------------------------------------------------------------------------------------------
// generic GEMM  C[M][N] (+)= opA(A) * opB(B) (+ bias[N]),  f32 MFMA 32x32x2, 128 x (32*NTN) x 32 tiles
//   TA = 0: A is [M][K] (lda)      TA = 1: A is [K][M] (lda)
//   TB = 0: B is [K][N] (ldb)      TB = 1: B is [N][K] (ldb)
// grid.z = split-K slices; with more than one slice (or accumulate=1) results are atomically added to C.
// ---------------------------------------------------------------------------------------------
template <int TA, int TB, int NTN>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                   const float* __restrict__ bias, float* __restrict__ Cm, int M, int N, int K,
                                                   int lda, int ldb, int ldc, int k_per_slice, int atomic) {
    constexpr int BM = 128, BN = 32 * NTN, BK = 32, AP = BK + 1, BNP = BN + 1;
    __shared__ float As[BM * AP];
    __shared__ float Bs[BK * BNP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * k_per_slice, kend = min(K, kbeg + k_per_slice);
    f32x16 acc[NTN];
#pragma unroll
    for (int i = 0; i < NTN; ++i) acc[i] = f32x16_zero();

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
        // ---- stage A tile -> As[m][k] ----
        if (TA == 0) {
            for (int idx = tid; idx < BM * BK; idx += 256) {
                const int m = idx / BK, k = idx - m * BK;
                const int gm = m0 + m, gk = k0 + k;
                As[m * AP + k] = (gm < M && gk < kend) ? A[(size_t)gm * lda + gk] : 0.f;
            }
        } else {
            for (int idx = tid; idx < BM * BK; idx += 256) {
                const int k = idx / BM, m = idx - k * BM;
                const int gm = m0 + m, gk = k0 + k;
                As[m * AP + k] = (gm < M && gk < kend) ? A[(size_t)gk * lda + gm] : 0.f;
            }
        }
        // ---- stage B tile -> Bs[k][n] ----
        if (TB == 0) {
            for (int idx = tid; idx < BK * BN; idx += 256) {
                const int k = idx / BN, n = idx - k * BN;
                const int gn = n0 + n, gk = k0 + k;
                Bs[k * BNP + n] = (gn < N && gk < kend) ? Bm[(size_t)gk * ldb + gn] : 0.f;
            }
        } else {
            for (int idx = tid; idx < BK * BN; idx += 256) {
                const int n = idx / BK, k = idx - n * BK;
                const int gn = n0 + n, gk = k0 + k;
                Bs[k * BNP + n] = (gn < N && gk < kend) ? Bm[(size_t)gn * ldb + gk] : 0.f;
            }
        }
        __syncthreads();
        const float* ap = As + (32 * w + lo) * AP + hi;
#pragma unroll
        for (int k = 0; k < BK; k += 2) {
            const float av = ap[k];
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) acc[nt] = mfma32(av, Bs[(k + hi) * BNP + nt * 32 + lo], acc[nt]);
        }
    }
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        const int gn = n0 + nt * 32 + lo;
        if (gn < N) {
            const float bv = (bias != nullptr && blockIdx.z == 0) ? bias[gn] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + 32 * w + mfma32_row(r, lane);
                if (gm < M) {
                    float* dst = Cm + (size_t)gm * ldc + gn;
                    const float v = acc[nt][r] + bv;
                    if (atomic) atomicAdd(dst, v); else *dst = v;
                }
            }
        }
    }
}

// Vectorised variant (16-byte global loads, next K tile prefetched into registers under the MFMAs).
// Requires 16-byte aligned bases, lda/ldb % 4 == 0 and the contiguous extent of each operand % 4 == 0.
template <int TA, int TB, int NTN>
__global__ __launch_bounds__(256) void gemm_vec_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                       const float* __restrict__ bias, float* __restrict__ Cm, int M, int N, int K,
                                                       int lda, int ldb, int ldc, int k_per_slice, int atomic,
                                                       const float* __restrict__ A1, const float* __restrict__ B1,
                                                       const float* __restrict__ bias1, float* __restrict__ C1, int nbatch,
                                                       const float* __restrict__ Bsw, int ksw) {
    // Bsw != null: K-concatenated B -- rows k >= ksw come from Bsw (already offset by -ksw rows); ksw % 32 == 0
    constexpr int BM = 128, BN = 32 * NTN, BK = 32, AP = BK + 1, BNS = BN + 4;
    constexpr int AV = BM * BK / 4 / 256, BV = BK * BN / 4 / 256;     // float4 per thread per tile
    __shared__ float As[BM * AP];
    __shared__ __attribute__((aligned(16))) float Bs[BK * BNS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int zb = nbatch == 2 ? (blockIdx.z & 1) : 0, zs = nbatch == 2 ? (blockIdx.z >> 1) : blockIdx.z;
    if (zb) { A = A1; Bm = B1; bias = bias1; Cm = C1; }          // second problem of a batch of two
    const int kbeg = zs * k_per_slice, kend = min(K, kbeg + k_per_slice);
    f32x16 acc[NTN];
#pragma unroll
    for (int i = 0; i < NTN; ++i) acc[i] = f32x16_zero();
    float4 ra[AV], rb[BV];
    // (guarded loads are written as `v = 0; if (ok) v = load` -- a `ok ? *p : zero` select makes the compiler pick between
    // two ADDRESSES and emit scalar flat loads)
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    auto load_tile = [&](int k0) {
        const float* Bsrc = (Bsw != nullptr && k0 >= ksw) ? Bsw : Bm;
#pragma unroll
        for (int u = 0; u < AV; ++u) {
            const int i = tid + 256 * u;
            if (TA == 0) {
                const int m = i / (BK / 4), kq = i % (BK / 4);
                const int gm = m0 + m, gk = k0 + 4 * kq;
                ra[u] = zero4;
                if (gm < M && gk < kend) ra[u] = *(const float4*)(A + (size_t)gm * lda + gk);
            } else {
                const int k = i / (BM / 4), mq = i % (BM / 4);
                const int gm = m0 + 4 * mq, gk = k0 + k;
                ra[u] = zero4;
                if (gm < M && gk < kend) ra[u] = *(const float4*)(A + (size_t)gk * lda + gm);
            }
        }
#pragma unroll
        for (int u = 0; u < BV; ++u) {
            const int i = tid + 256 * u;
            if (TB == 0) {
                const int k = i / (BN / 4), nq = i % (BN / 4);
                const int gn = n0 + 4 * nq, gk = k0 + k;
                rb[u] = zero4;
                if (gn < N && gk < kend) rb[u] = *(const float4*)(Bsrc + (size_t)gk * ldb + gn);
            } else {
                const int n = i / (BK / 4), kq = i % (BK / 4);
                const int gn = n0 + n, gk = k0 + 4 * kq;
                rb[u] = zero4;
                if (gn < N && gk < kend) rb[u] = *(const float4*)(Bsrc + (size_t)gn * ldb + gk);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < AV; ++u) {
            const int i = tid + 256 * u;
            if (TA == 0) {
                const int m = i / (BK / 4), kq = i % (BK / 4);
                float* d = As + m * AP + 4 * kq;
                d[0] = ra[u].x; d[1] = ra[u].y; d[2] = ra[u].z; d[3] = ra[u].w;
            } else {
                const int k = i / (BM / 4), mq = i % (BM / 4);
                float* d = As + (4 * mq) * AP + k;
                d[0] = ra[u].x; d[AP] = ra[u].y; d[2 * AP] = ra[u].z; d[3 * AP] = ra[u].w;
            }
        }
#pragma unroll
        for (int u = 0; u < BV; ++u) {
            const int i = tid + 256 * u;
            if (TB == 0) {
                const int k = i / (BN / 4), nq = i % (BN / 4);
                *(float4*)(Bs + k * BNS + 4 * nq) = rb[u];
            } else {
                const int n = i / (BK / 4), kq = i % (BK / 4);
                float* d = Bs + (4 * kq) * BNS + n;
                d[0] = rb[u].x; d[BNS] = rb[u].y; d[2 * BNS] = rb[u].z; d[3 * BNS] = rb[u].w;
            }
        }
    };

    if (kbeg < kend) load_tile(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();                 // everyone finished reading the previous tile
        store_tile();
        __syncthreads();
        if (k0 + BK < kend) load_tile(k0 + BK);
        const float* ap = As + (32 * w + lo) * AP + hi;
#pragma unroll
        for (int k = 0; k < BK; k += 2) {
            const float av = ap[k];
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) acc[nt] = mfma32(av, Bs[(k + hi) * BNS + nt * 32 + lo], acc[nt]);
        }
    }
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        const int gn = n0 + nt * 32 + lo;
        if (gn < N) {
            const float bv = (bias != nullptr && zs == 0) ? bias[gn] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + 32 * w + mfma32_row(r, lane);
                if (gm < M) {
                    float* dst = Cm + (size_t)gm * ldc + gn;
                    const float v = acc[nt][r] + bv;
                    if (atomic) atomicAdd(dst, v); else *dst = v;
                }
            }
        }
    }
}

static int gemm_dispatch(const float* A, const float* Bm, const float* bias, float* Cm, const float* A1, const float* B1,
                         const float* bias1, float* C1, int nbatch, int M, int N, int K, int lda, int ldb, int ldc, int transA,
                         int transB, int split_k, int accumulate, hipStream_t s, const float* Bsw = nullptr, int ksw = 0) {
    if (M <= 0 || N <= 0 || K <= 0) return SED_OK;
    if (split_k < 1) split_k = 1;
    int kps = ((K + split_k - 1) / split_k + 31) / 32 * 32;
    split_k = (K + kps - 1) / kps;
    const int atomic = (split_k > 1 || accumulate) ? 1 : 0;
    int ntn = N > 64 ? 4 : 2;
    if (ntn == 4 && ((N + 127) / 128) * ((M + 127) / 128) * split_k * nbatch < 200) ntn = 2;     // too few workgroups for 256 CUs
    dim3 grid((N + 32 * ntn - 1) / (32 * ntn), (M + 127) / 128, split_k * nbatch);
    bool vec = ((uintptr_t)A % 16 == 0) && ((uintptr_t)Bm % 16 == 0) && lda % 4 == 0 && ldb % 4 == 0 &&
               ((transA ? M : K) % 4 == 0) && ((transB ? K : N) % 4 == 0) && kps % 4 == 0;
    if (nbatch == 2) vec = vec && ((uintptr_t)A1 % 16 == 0) && ((uintptr_t)B1 % 16 == 0);
#define GEMMV_CASE(ta, tb, nn) \
    if (vec && transA == ta && transB == tb && ntn == nn) { SED_LAUNCH((gemm_vec_kernel<ta, tb, nn>), grid, dim3(256), 0, s, A, Bm, bias, Cm, M, N, K, lda, ldb, ldc, kps, atomic, A1, B1, bias1, C1, nbatch, Bsw, ksw); return sed_check_launch(); }
    GEMMV_CASE(0, 0, 2) GEMMV_CASE(0, 0, 4) GEMMV_CASE(0, 1, 2) GEMMV_CASE(0, 1, 4) GEMMV_CASE(1, 0, 2) GEMMV_CASE(1, 0, 4)
#undef GEMMV_CASE
    if (Bsw != nullptr) return SED_ERR_UNSUPPORTED;      // the K-concatenated form needs the 16-byte path
    if (nbatch == 2) {          // scalar fallback: two plain launches
        int rc = gemm_dispatch(A, Bm, bias, Cm, nullptr, nullptr, nullptr, nullptr, 1, M, N, K, lda, ldb, ldc, transA, transB, split_k, accumulate, s);
        if (rc != SED_OK) return rc;
        return gemm_dispatch(A1, B1, bias1, C1, nullptr, nullptr, nullptr, nullptr, 1, M, N, K, lda, ldb, ldc, transA, transB, split_k, accumulate, s);
    }
#define GEMM_CASE(ta, tb, nn) \
    if (transA == ta && transB == tb && ntn == nn) { SED_LAUNCH((gemm_kernel<ta, tb, nn>), grid, dim3(256), 0, s, A, Bm, bias, Cm, M, N, K, lda, ldb, ldc, kps, atomic); return sed_check_launch(); }
    GEMM_CASE(0, 0, 2) GEMM_CASE(0, 0, 4) GEMM_CASE(0, 1, 2) GEMM_CASE(0, 1, 4) GEMM_CASE(1, 0, 2) GEMM_CASE(1, 0, 4)
#undef GEMM_CASE
    return SED_ERR_UNSUPPORTED;
}

// C[M][N] = opA(A)[M][K] * opB(B)[K][N] + bias.  accumulate != 0 adds into C (atomics); split_k > 1 requires
// the caller to have zeroed C (or accumulate).  Leading dimensions are in floats.
SED_API int sed_gemm(const float* A, const float* Bm, const float* bias, float* Cm, int M, int N, int K, int lda, int ldb,
                        int ldc, int transA, int transB, int split_k, int accumulate, void* stream) {
    return gemm_dispatch(A, Bm, bias, Cm, nullptr, nullptr, nullptr, nullptr, 1, M, N, K, lda, ldb, ldc, transA, transB, split_k,
                         accumulate, (hipStream_t)stream);
}
// Two same-shape problems (the two GRU directions) in ONE launch: (A0,B0,bias0 -> C0) and (A1,B1,bias1 -> C1).
SED_API int sed_gemm_pair(const float* A0, const float* A1, const float* B0, const float* B1, const float* bias0,
                             const float* bias1, float* C0, float* C1, int M, int N, int K, int lda, int ldb, int ldc, int transA,
                             int transB, int split_k, int accumulate, void* stream) {
    return gemm_dispatch(A0, B0, bias0, C0, A1, B1, bias1, C1, 2, M, N, K, lda, ldb, ldc, transA, transB, split_k, accumulate,
                         (hipStream_t)stream);
}

// C[M][N] = A[M][K] . [B0 ; B1]: B is two row-major tensors stacked along K (rows [0, ksplit) from B0, the rest from B1;
// ksplit % 32 == 0).  Exact-f32 MFMA; 16-byte aligned operands required.
SED_API int sed_gemm_kcat(const float* A, const float* B0, const float* B1, float* Cm, int M, int N, int K, int ksplit, int lda,
                             int ldb, int ldc, void* stream) {
    if (ksplit % 32 != 0 || ksplit <= 0 || ksplit >= K) return SED_ERR_ARG;
    return gemm_dispatch(A, B0, nullptr, Cm, nullptr, nullptr, nullptr, nullptr, 1, M, N, K, lda, ldb, ldc, 0, 0, 1, 0,
                         (hipStream_t)stream, B1 - (size_t)ksplit * ldb, ksplit);
}

// column sums: out[n] = sum_m X[m*ld + n], n < N  (bias gradients).  Deterministic since round 4 (it was zero fill + one float
// atomic per (column, row chunk): the last float atomics on a recipe path -- the `cat_tf` bias gradient of the embedding recipes):
// one workgroup owns 16 columns for ALL rows -- 64 row lanes per column (a wave reads 4 rows x 64 B), each summing its rows in
// order, combined by a fixed-order tree in LDS.  M ~ 10^4 rows x 64 B per workgroup: a few us, no scratch, no zero fill.
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ X, float* __restrict__ out, float* __restrict__ out1,
                                                      int nsplit, int M, int N, int ld) {
    __shared__ float part[64][16];
    const int col = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + col;
    float acc = 0.f;
    if (n < N)
        for (int r = lane; r < M; r += 64) acc += X[(size_t)r * ld + n];
    part[lane][col] = acc;
    __syncthreads();
#pragma unroll
    for (int h = 32; h > 0; h >>= 1) {
        if (lane < h) part[lane][col] += part[lane + h][col];
        __syncthreads();
    }
    if (lane == 0 && n < N) *(n < nsplit ? out + n : out1 + (n - nsplit)) = part[0][col];
}
// out[n] = sum_m X[m*ld + n] for n < nsplit, out1[n - nsplit] for nsplit <= n < N (out1 may be null when nsplit == N).
SED_API int sed_colsum(const float* X, float* out, float* out1, int nsplit, int M, int N, int ld, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (nsplit > N || (nsplit < N && out1 == nullptr)) return SED_ERR_ARG;
    if (N <= 0) return SED_OK;
    SED_LAUNCH(colsum_kernel, dim3((N + 15) / 16), dim3(1024), 0, s, X, out, out1, nsplit, M > 0 ? M : 0, N, ld);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------
// forward recurrence.  gi: (B, T, 2, 3H) input projections incl. b_ih; whh0/whh1: (3H, H) per direction; bhh0/bhh1: (3H)
// out: (B, T, 2H) ([fwd | bwd]); saved: (B, T, 2, 4, H) = r, z, n, hn   (null in inference)
//
// No global memory operation sits inside the step loop: the input projections of the NEXT chunk of 8 steps are
// fetched into registers at the start of a chunk and parked in LDS at its end, and the per-step results (h and
// the four saved gate tensors) are collected in an LDS buffer that is flushed, coalesced, one chunk later.
// (CDNA4's vmcnt counts stores too, so a single store in the loop would put HBM write latency on the chain.)
// ---------------------------------------------------------------------------------------------
#define GRU_CH 8
#define GRU_THREADS 512
// GRU_VARIANT: design / timing-ablation mask for tools/gru_variants.py.  Product build = 5:
//   1 = padded hidden-state quarters (121 -> 117 us per launch), 4 = quad stores of the step results (-> 112.6 us);
//   2 = two accumulators per gate (measured slower: 122.5 us, off);
//   8 = no hidden-state LDS reads, 16 = no FMAs, 32 = no exp/rcp, 64 = no output staging stores, 128 = no per-step barrier
//   (results are wrong under any of 8..128; numbers in DESIGN.md section 8).
#ifndef GRU_VARIANT
#define GRU_VARIANT 5
#endif
#define GRU_HPAD (GRU_VARIANT & 1)
#define GRU_ACC6 ((GRU_VARIANT & 2) != 0)
#define GRU_QSTORE ((GRU_VARIANT & 4) != 0)
#define GRU_NOREAD ((GRU_VARIANT & 8) != 0)
#define GRU_NOFMA ((GRU_VARIANT & 16) != 0)
#define GRU_NOTRANS ((GRU_VARIANT & 32) != 0)
#define GRU_NOOBUF ((GRU_VARIANT & 64) != 0)
#define GRU_NOBAR ((GRU_VARIANT & 128) != 0)
// GRU_STAMP (diagnostics build only: ONLY=sed_gru.hip python tools/build_variant.py grustamp -DGRU_STAMP; tools/gru_stamps.py): s_memtime
// at six points of a step -- top (behind the barrier), hidden state + gate inputs landed, FMAs done, quarters summed, gates done, results
// stored -- for the first and the last wave of workgroup 0, steps 16 .. 47, collected in LDS and dumped after the loop.  Every stamp
// waits for lgkmcnt(0) and fences the scheduler, so the phases it separates no longer overlap: an upper bound per phase, next to the
// unstamped total.  (profiles/r06_gru_phase_stamps.md)
#ifdef GRU_STAMP
__device__ unsigned long long* gru_stamp_buf;
SED_API int sed_gru_debug_set_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(gru_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? SED_OK : SED_ERR_LAUNCH;
}
#define GRU_TS(k)                                                                                      \
    {                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (stamp_on) {                                                                                \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                \
            if ((tid & 63) == 0 && ts_step >= 0 && ts_step < 32) s_ts[ts_wave][ts_step][k] = t_;       \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    }
#else
#define GRU_TS(k)
#endif
template <int H, int CH>
__global__ __launch_bounds__(4 * H) void gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ whh0,
                                                      const float* __restrict__ whh1, const float* __restrict__ bhh0,
                                                      const float* __restrict__ bhh1, float* __restrict__ out,
                                                      float* __restrict__ saved, int B, int T) {
    sed_wave_prio_high();
    constexpr int KH = H / 4, NT_ = 4 * H;                          // KH: K quarter per thread; thread (unit j, quarter)
    // H = 128 (the 2023 recipe): 512 threads, the thread's whole h slice is read before its FMAs (one LDS latency).  H = 192 (the
    // 2024 recipe's n_RNN_cell): 768 threads = 3 waves per SIMD = 170 VGPRs, of which the W_hh slice alone is 144: the h slice is
    // read in blocks of 4 float4 between the FMAs (HB), and the chunk is 4 steps (LDS).
    constexpr int HB = H <= 128 ? KH / 4 : 4;
    // H = 192: the last WL float4 blocks of the n gate's weight slice live in LDS ([block][thread] float4: lane-contiguous reads)
    // instead of registers -- 32 of the 144 weights per thread: the rest then (nearly) fits the 170-VGPR budget
    constexpr int WL = H <= 128 ? 0 : 8, WR = KH / 4 - WL;
    // result planes of a step sit OBP floats apart: with GRU_QSTORE the four lanes of a quad store to four planes at once, so
    // the plane pitch is H + 8 (quad lanes 8 banks apart) instead of H (same bank)
    constexpr int OBP = GRU_QSTORE ? H + 8 : H, OBS = 5 * OBP;
    constexpr int GI_F = CH * 3 * H, OB_F = CH * OBS;              // floats per chunk buffer
    // quarter q of h starts at q * (KH + 4) floats: the four quarters a wave reads with one ds_read_b128 fall into 16
    // distinct banks (at a 128-byte pitch they would share four)
    constexpr int HP = GRU_HPAD ? KH + 4 : KH;
    __shared__ __attribute__((aligned(16))) float hbuf[2][4 * HP];
#ifdef GRU_STAMP
    __shared__ unsigned long long s_ts[2][32][8];
    const bool stamp_on = blockIdx.x == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == (4 * H / 64 - 1));
    const int ts_wave = (threadIdx.x >> 6) == 0 ? 0 : 1;
    int ts_step = -16;
#endif
    SED_DYN_SMEM(smem);
    float* gis = (float*)smem;                 // [2][CH][3H]
    float* obuf = gis + 2 * GI_F;              // [2][CH][5H] = h | r | z | n | hn
    float4* wls = (float4*)(obuf + 2 * OB_F);  // [WL][NT_] float4 (H = 192 only)
    const int tid = threadIdx.x, j = tid >> 2, half = tid & 3;    // `half` = which K quarter
    const int b = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const float* W = dir ? whh1 : whh0;
    const float* bhh = dir ? bhh1 : bhh0;
    f32x2 wr[KH / 2], wz[KH / 2], wn[2 * WR];
#pragma unroll
    for (int k = 0; k < KH / 2; ++k) {
        wr[k] = *(const f32x2*)(W + (size_t)(0 * H + j) * H + half * KH + 2 * k);
        wz[k] = *(const f32x2*)(W + (size_t)(1 * H + j) * H + half * KH + 2 * k);
        if (k < 2 * WR) wn[k] = *(const f32x2*)(W + (size_t)(2 * H + j) * H + half * KH + 2 * k);
    }
#pragma unroll
    for (int q = 0; q < WL; ++q) wls[q * NT_ + tid] = *(const float4*)(W + (size_t)(2 * H + j) * H + half * KH + 4 * (WR + q));
    float br = bhh[j], bz = bhh[H + j], bn = bhh[2 * H + j];
    // the weights and biases are complete before the recurrence starts (see sed_pin: no vmcnt wait may sit in the step loop)
#pragma unroll
    for (int k = 0; k < KH / 2; ++k) { sed_pin(wr[k]); sed_pin(wz[k]); if (k < 2 * WR) sed_pin(wn[k]); }
    sed_pin(br); sed_pin(bz); sed_pin(bn);
    if (tid < H) hbuf[0][(tid / KH) * HP + tid % KH] = 0.f;
    float hprev = 0.f;
    const int nchunks = (T + CH - 1) / CH;
    constexpr int GV = (GI_F / 4 + NT_ - 1) / NT_;         // float4 of gi per thread per chunk
    float4 greg[GV];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int u = 0; u < GV; ++u) {
            const int e4 = tid + NT_ * u, s = e4 / (3 * H / 4), q = e4 - s * (3 * H / 4);
            const int step = c * CH + s;
            greg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e4 < GI_F / 4 && step < T) {
                const int t = dir ? T - 1 - step : step;
                greg[u] = *(const float4*)(gi + (((size_t)b * T + t) * 2 + dir) * 3 * H + 4 * q);
            }
        }
    };
    auto park_chunk = [&](int c) {
#pragma unroll
        for (int u = 0; u < GV; ++u)
            if (tid + NT_ * u < GI_F / 4) *(float4*)(gis + (c & 1) * GI_F + 4 * (tid + NT_ * u)) = greg[u];
    };
    auto flush_chunk = [&](int c) {
        constexpr int PER = 5 * H / 4;         // float4 per step: 32 of h + 128 of saved
        const float* ob = obuf + (c & 1) * OB_F;
        for (int e4 = tid; e4 < CH * PER; e4 += NT_) {
            const int s = e4 / PER, q = e4 - s * PER;
            const int step = c * CH + s;
            if (step >= T) break;
            const int t = dir ? T - 1 - step : step;
            const float4 v = *(const float4*)(ob + s * OBS + (q / (H / 4)) * OBP + 4 * (q % (H / 4)));
            if (q < H / 4) *(float4*)(out + ((size_t)b * T + t) * 2 * H + dir * H + 4 * q) = v;
            else if (saved) *(float4*)(saved + (((size_t)b * T + t) * 2 + dir) * 4 * H + 4 * (q - H / 4)) = v;
        }
    };
    load_chunk(0);
    park_chunk(0);
    __syncthreads();
    int cur = 0;
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) load_chunk(c + 1);
        if (c > 0) flush_chunk(c - 1);
        const float* gch = gis + (c & 1) * GI_F;
        float* och = obuf + (c & 1) * OB_F;
        const int nsteps = min(CH, T - c * CH);
        for (int s = 0; s < nsteps; ++s) {
            GRU_TS(0)
            const float gr = gch[s * 3 * H + j], gz = gch[s * 3 * H + H + j], gn = gch[s * 3 * H + 2 * H + j];
            // a block of this thread's h slice first (H = 128: all of it, one LDS latency, not one per read), then its FMAs
            f32x2 pr = {0.f, 0.f}, pz = {0.f, 0.f}, pn = {0.f, 0.f};
            f32x2 qr = {0.f, 0.f}, qz = {0.f, 0.f}, qn = {0.f, 0.f};      // GRU_ACC6: second accumulator per gate (shorter chains)
#pragma unroll
            for (int kb = 0; kb < KH / 4; kb += HB) {
                float4 hq[HB];
#pragma unroll
                for (int k = 0; k < HB; ++k) hq[k] = GRU_NOREAD ? make_float4(gr, gz, gn, gr) : *(const float4*)(hbuf[cur] + half * HP + 4 * (kb + k));
#ifdef GRU_STAMP
                if (kb == 0) GRU_TS(1)
#endif
#pragma unroll
                for (int k0 = 0; k0 < (GRU_NOFMA ? 1 : HB); ++k0) {
                    const int k = kb + k0;
                    const f32x2 lo2 = {hq[k0].x, hq[k0].y}, hi2 = {hq[k0].z, hq[k0].w};
                    f32x2 wn0, wn1;
                    if (k < WR) { wn0 = wn[2 * (k < WR ? k : 0)]; wn1 = wn[2 * (k < WR ? k : 0) + 1]; }
                    else { const float4 w4 = wls[(k - WR) * NT_ + tid]; wn0 = f32x2{w4.x, w4.y}; wn1 = f32x2{w4.z, w4.w}; }
                    pr = pk_fma(wr[2 * k], lo2, pr); pz = pk_fma(wz[2 * k], lo2, pz); pn = pk_fma(wn0, lo2, pn);
                    if (GRU_ACC6) { qr = pk_fma(wr[2 * k + 1], hi2, qr); qz = pk_fma(wz[2 * k + 1], hi2, qz); qn = pk_fma(wn1, hi2, qn); }
                    else { pr = pk_fma(wr[2 * k + 1], hi2, pr); pz = pk_fma(wz[2 * k + 1], hi2, pz); pn = pk_fma(wn1, hi2, pn); }
                }
                if (HB < KH / 4) sed_sched_fence();                       // keeps the later blocks' reads from being hoisted (VGPRs)
            }
            if (GRU_ACC6) { pr += qr; pz += qz; pn += qn; }
#ifdef GRU_STAMP
            sed_pin(pr); sed_pin(pz); sed_pin(pn);
            GRU_TS(2)
#endif
            float ar = pr.x + pr.y, az = pz.x + pz.y, an = pn.x + pn.y;
            ar = sed_quad_sum(ar); az = sed_quad_sum(az); an = sed_quad_sum(an);      // the four K-quarters (DPP, not ds_bpermute)
#ifdef GRU_STAMP
            sed_pin(ar); sed_pin(az); sed_pin(an);
            GRU_TS(3)
#endif
            const float r = GRU_NOTRANS ? (gr + ar + br) * 0.01f : sed_fast_sigmoid(gr + ar + br);
            const float z = GRU_NOTRANS ? (gz + az + bz) * 0.01f : sed_fast_sigmoid(gz + az + bz);
            const float hn = an + bn;
            const float n = GRU_NOTRANS ? (gn + r * hn) * 0.01f : sed_fast_tanh(gn + r * hn);
            float hnew = (1.0f - z) * n + z * hprev;
#ifdef GRU_STAMP
            sed_pin(hnew);
            GRU_TS(4)
#endif
            hprev = hnew;
            if (GRU_QSTORE) {
                // every lane of the quad stores one of the five results (all four hold them): two LDS stores on the chain, not five
                float* o = och + s * OBS;
                const float v4 = half == 0 ? hnew : half == 1 ? r : half == 2 ? z : n;
                if (!GRU_NOOBUF) o[half * OBP + j] = v4;
                if (half == 0) {
                    hbuf[cur ^ 1][(j / KH) * HP + j % KH] = hnew;
                    if (!GRU_NOOBUF) o[4 * OBP + j] = hn;
                }
            } else if (half == 0) {
                hbuf[cur ^ 1][(j / KH) * HP + j % KH] = hnew;
                float* o = och + s * OBS;
                if (!GRU_NOOBUF) { o[j] = hnew; o[OBP + j] = r; o[2 * OBP + j] = z; o[3 * OBP + j] = n; o[4 * OBP + j] = hn; }
            }
            cur ^= 1;
            GRU_TS(5)
            if (!GRU_NOBAR) __syncthreads();
#ifdef GRU_STAMP
            GRU_TS(6)
            ++ts_step;
#endif
        }
        if (c + 1 < nchunks) park_chunk(c + 1);
        __syncthreads();
    }
    flush_chunk(nchunks - 1);
#ifdef GRU_STAMP
    __syncthreads();
    if (blockIdx.x == 0 && gru_stamp_buf)
        for (int i = tid; i < 2 * 32 * 8; i += NT_) gru_stamp_buf[i] = (&s_ts[0][0][0])[i];
#endif
}
// A recurrence workgroup claims (most of) its CU's LDS: a (clip, direction) is one latency-bound dependent chain, and any workgroup
// of another stream that lands on the same CU (the prefetched mel front-end: 24 KB of LDS each, the other model's head, the
// weight-gradient GEMMs) takes issue slots from it.  With sed_set_tuning(SED_TUNE_GRU_LDS_KB, kb) the launch asks for kb KB of dynamic
// LDS (never less than the kernel needs, never more than 156 KB), so that such workgroups only fit on the CUs no recurrence runs on.
static inline int gru_lds_claim(int smem) {
    int kb = sed_tuning[SED_TUNE_GRU_LDS_KB];
    if (kb <= 0) return smem;
    if (kb > 156) kb = 156;
    return kb * 1024 > smem ? kb * 1024 : smem;
}

SED_API int sed_gru_fwd(const float* gi, const float* whh0, const float* whh1, const float* bhh0, const float* bhh1,
                           float* out, float* saved, int B, int T, int H, void* stream) {
    if (H != 128 && H != 192) return SED_ERR_UNSUPPORTED;
    if (B <= 0 || T <= 0) return SED_OK;
#define GRU_FWD_CASE(h, ch)                                                                                                       \
    if (H == h) {                                                                                                                 \
        int smem = (2 * ch * 3 * h + 2 * ch * 5 * (GRU_QSTORE ? h + 8 : h)) * 4 + (h <= 128 ? 0 : 8 * 4 * h * 16);               \
        smem = gru_lds_claim(smem);                                                                                               \
        SED_MAX_SMEM((gru_fwd_kernel<h, ch>), smem);                                                                              \
        SED_LAUNCH((gru_fwd_kernel<h, ch>), dim3(2 * B), dim3(4 * h), smem, (hipStream_t)stream, gi, whh0, whh1, bhh0, bhh1, out, saved, B, T); \
    }
    GRU_FWD_CASE(128, GRU_CH) GRU_FWD_CASE(192, 4)
#undef GRU_FWD_CASE
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------
// backward recurrence.  dout: (B,T,2H) upstream gradient of the layer output; out/saved from the forward.
// Produces dgi (B,T,2,3H) = dL/d(W_ih x + b_ih), dgh (B,T,2,3H) = dL/d(W_hh h + b_hh) and
// hprev (B,T,2,H) = the hidden state each step consumed (for dW_hh = dgh^T hprev).
// Same chunked LDS staging as the forward: nothing touches global memory inside the step loop.
// ---------------------------------------------------------------------------------------------
template <int H, int CH>
__global__ __launch_bounds__(4 * H) void gru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                      const float* __restrict__ saved, const float* __restrict__ whh0,
                                                      const float* __restrict__ whh1, float* __restrict__ dgi,
                                                      float* __restrict__ dgh, float* __restrict__ hprev_out,
                                                      float* __restrict__ bpart, int B, int T) {
    sed_wave_prio_high();
    constexpr int KH = H / 4, NT_ = 4 * H, HB = H <= 128 ? KH / 4 : 2;        // HB: gate-vector block (see the forward)
    constexpr int WL = H <= 128 ? 0 : 8, WR = KH / 4 - WL;                    // float4 blocks of the n-gate slice in LDS / in registers
    // gbuf: three planes (da_r, da_z, dhn) of four K-quarters; the quarters start QP floats apart so that the four quarters a
    // wave reads with one ds_read_b128 hit distinct banks (GRU_HPAD), the planes GP apart so that the quad's three stores do.
    // obuf: seven result planes per step in the order dgi(r, z, n) | hprev | dgh(r, z, hn), OBP apart (GRU_QSTORE: the four
    // lanes of a quad store four planes with one instruction, 8 banks apart).
    constexpr int QP = GRU_HPAD ? KH + 4 : KH, GP = 4 * QP + (GRU_QSTORE ? 8 : 0);
    constexpr int OBP = GRU_QSTORE ? H + 8 : H, OBS = 7 * OBP;
    constexpr int IB_F = CH * 6 * H, OB_F = CH * OBS;
    __shared__ __attribute__((aligned(16))) float gbuf[2][3 * GP];
    SED_DYN_SMEM(smem);
    float* ibuf = (float*)smem;                // [2][CH][6H] = r | z | n | hn | hprev | dout
    float* obuf = ibuf + 2 * IB_F;             // [2][CH][7H] = dgi(3H) | dgh(3H) | hprev(H)
    float4* wls = (float4*)(obuf + 2 * OB_F);  // [WL][NT_] float4 (H = 192 only)
    const int tid = threadIdx.x, k = tid >> 2, half = tid & 3;    // `half` = which quarter of the gate rows
    const int b = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const float* W = dir ? whh1 : whh0;
    f32x2 wr[KH / 2], wz[KH / 2], wn[2 * WR];  // W^T slices: contributions of gate rows j in this thread's quarter to unit k
#pragma unroll
    for (int jj = 0; jj < KH / 2; ++jj) {
        const int j = half * KH + 2 * jj;
        wr[jj] = f32x2{W[(size_t)(0 * H + j) * H + k], W[(size_t)(0 * H + j + 1) * H + k]};
        wz[jj] = f32x2{W[(size_t)(1 * H + j) * H + k], W[(size_t)(1 * H + j + 1) * H + k]};
        if (jj < 2 * WR) wn[jj] = f32x2{W[(size_t)(2 * H + j) * H + k], W[(size_t)(2 * H + j + 1) * H + k]};
    }
#pragma unroll
    for (int q = 0; q < WL; ++q) {
        const int j = half * KH + 4 * (WR + q);
        wls[q * NT_ + tid] = make_float4(W[(size_t)(2 * H + j) * H + k], W[(size_t)(2 * H + j + 1) * H + k],
                                         W[(size_t)(2 * H + j + 2) * H + k], W[(size_t)(2 * H + j + 3) * H + k]);
    }
    // (the W^T slices are complete before the recurrence starts: see sed_pin)
#pragma unroll
    for (int jj = 0; jj < KH / 2; ++jj) { sed_pin(wr[jj]); sed_pin(wz[jj]); if (jj < 2 * WR) sed_pin(wn[jj]); }
    const int nchunks = (T + CH - 1) / CH;
    constexpr int IV = (IB_F / 4 + NT_ - 1) / NT_;      // float4 per thread per chunk: 3 (H = 128, 8 steps), 1.5 -> 2 guarded (H = 192, 4 steps)
    float4 ireg[IV];
    // chunk c covers reverse-order positions rs = c*CH .. c*CH+CH-1, forward step index = T-1-rs
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int u = 0; u < IV; ++u) {
            const int e4 = tid + NT_ * u, s = e4 / (6 * H / 4), q = e4 - s * (6 * H / 4);
            const int rs = c * CH + s;
            ireg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e4 < IB_F / 4 && rs < T) {
                const int step = T - 1 - rs;
                const int t = dir ? T - 1 - step : step, tp = dir ? t + 1 : t - 1;
                const size_t bt = (size_t)b * T + t;
                if (q < H) ireg[u] = *(const float4*)(saved + (bt * 2 + dir) * 4 * H + 4 * q);
                else if (q < H + H / 4) { if (step > 0) ireg[u] = *(const float4*)(out + ((size_t)b * T + tp) * 2 * H + dir * H + 4 * (q - H)); }
                else ireg[u] = *(const float4*)(dout + bt * 2 * H + dir * H + 4 * (q - H - H / 4));
            }
        }
    };
    auto park_chunk = [&](int c) {
#pragma unroll
        for (int u = 0; u < IV; ++u)
            if (tid + NT_ * u < IB_F / 4) *(float4*)(ibuf + (c & 1) * IB_F + 4 * (tid + NT_ * u)) = ireg[u];
    };
    auto flush_chunk = [&](int c) {
        constexpr int PER = 7 * H / 4;
        const float* ob = obuf + (c & 1) * OB_F;
        for (int e4 = tid; e4 < CH * PER; e4 += NT_) {
            const int s = e4 / PER, q = e4 - s * PER;
            const int rs = c * CH + s;
            if (rs >= T) break;
            const int step = T - 1 - rs;
            const int t = dir ? T - 1 - step : step;
            const size_t bt = (size_t)b * T + t;
            const int pl = q / (H / 4), w4 = 4 * (q - pl * (H / 4));
            const float4 v = *(const float4*)(ob + s * OBS + pl * OBP + w4);
            if (pl < 3) *(float4*)(dgi + (bt * 2 + dir) * 3 * H + pl * H + w4) = v;
            else if (pl == 3) *(float4*)(hprev_out + (bt * 2 + dir) * H + w4) = v;
            else *(float4*)(dgh + (bt * 2 + dir) * 3 * H + (pl - 4) * H + w4) = v;
        }
    };
    load_chunk(0);
    park_chunk(0);
    __syncthreads();
    float dh_carry = 0.f;
    float sb_r = 0.f, sb_z = 0.f, sb_n = 0.f, sb_hn = 0.f;     // bias gradients: sums over this clip's steps (off the chain)
    int cur = 0;
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) load_chunk(c + 1);
        if (c > 0) flush_chunk(c - 1);
        const float* ich = ibuf + (c & 1) * IB_F;
        float* och = obuf + (c & 1) * OB_F;
        const int nsteps = min(CH, T - c * CH);
        for (int s = 0; s < nsteps; ++s) {
            const float* in = ich + s * 6 * H;
            const float r = in[k], z = in[H + k], n = in[2 * H + k], hn = in[3 * H + k], hp = in[4 * H + k];
            const float dh = in[5 * H + k] + dh_carry;
            const float dn = dh * (1.0f - z);
            const float dzg = dh * (hp - n);
            const float da_n = dn * (1.0f - n * n);
            const float da_z = dzg * z * (1.0f - z);
            const float da_r = da_n * hn * r * (1.0f - r);
            const float dhn = da_n * r;
            sb_r += da_r; sb_z += da_z; sb_n += da_n; sb_hn += dhn;
            const int gk = (k / KH) * QP + k % KH;
            float* o = och + s * OBS;
            if (GRU_QSTORE) {
                // all four lanes of the quad hold the same values: three store instructions on the chain instead of ten
                const float g3 = half == 0 ? da_r : half == 1 ? da_z : dhn;
                if (half < 3) gbuf[cur][half * GP + gk] = g3;
                o[half * OBP + k] = half == 0 ? da_r : half == 1 ? da_z : half == 2 ? da_n : hp;
                if (half < 3) o[(4 + half) * OBP + k] = g3;
            } else if (half == 0) {
                gbuf[cur][gk] = da_r; gbuf[cur][GP + gk] = da_z; gbuf[cur][2 * GP + gk] = dhn;
                o[k] = da_r; o[OBP + k] = da_z; o[2 * OBP + k] = da_n; o[3 * OBP + k] = hp;
                o[4 * OBP + k] = da_r; o[5 * OBP + k] = da_z; o[6 * OBP + k] = dhn;
            }
            __syncthreads();
            const float* gv = gbuf[cur] + half * QP;
            f32x2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f}, p2 = {0.f, 0.f};     // three independent packed chains
#pragma unroll
            for (int qb = 0; qb < KH / 4; qb += HB) {
                float4 ga[HB], gc[HB], gd[HB];
#pragma unroll
                for (int q0 = 0; q0 < HB; ++q0) {
                    ga[q0] = *(const float4*)(gv + 4 * (qb + q0));
                    gc[q0] = *(const float4*)(gv + GP + 4 * (qb + q0));
                    gd[q0] = *(const float4*)(gv + 2 * GP + 4 * (qb + q0));
                }
#pragma unroll
                for (int q0 = 0; q0 < HB; ++q0) {
                    const int q4 = qb + q0;
                    const f32x2 a0 = {ga[q0].x, ga[q0].y}, a1 = {ga[q0].z, ga[q0].w};
                    const f32x2 c0 = {gc[q0].x, gc[q0].y}, c1 = {gc[q0].z, gc[q0].w};
                    const f32x2 d0 = {gd[q0].x, gd[q0].y}, d1 = {gd[q0].z, gd[q0].w};
                    f32x2 wn0, wn1;
                    if (q4 < WR) { wn0 = wn[2 * (q4 < WR ? q4 : 0)]; wn1 = wn[2 * (q4 < WR ? q4 : 0) + 1]; }
                    else { const float4 w4 = wls[(q4 - WR) * NT_ + tid]; wn0 = f32x2{w4.x, w4.y}; wn1 = f32x2{w4.z, w4.w}; }
                    p0 = pk_fma(wr[2 * q4], a0, p0); p1 = pk_fma(wz[2 * q4], c0, p1); p2 = pk_fma(wn0, d0, p2);
                    p0 = pk_fma(wr[2 * q4 + 1], a1, p0); p1 = pk_fma(wz[2 * q4 + 1], c1, p1); p2 = pk_fma(wn1, d1, p2);
                }
                if (HB < KH / 4) sed_sched_fence();
            }
            float acc = ((p0.x + p0.y) + (p1.x + p1.y)) + (p2.x + p2.y);
            acc = sed_quad_sum(acc);
            dh_carry = dh * z + acc;
            cur ^= 1;
        }
        __syncthreads();                       // all steps of the chunk done (gbuf/obuf/ibuf reads retired)
        if (c + 1 < nchunks) park_chunk(c + 1);
        __syncthreads();
    }
    flush_chunk(nchunks - 1);
    // db_ih = sum over (clip, step) of dgi, db_hh likewise of dgh: this (clip, direction)'s sums over its steps go to its own record
    // [db_ih (3H) | db_hh (3H)]; gru_bias_reduce_kernel adds the clips in order (round 3: these were the last float atomics of the step)
    if (bpart != nullptr) {
        float* rec = bpart + (size_t)blockIdx.x * 6 * H;
        if (half == 0) { rec[k] = sb_r; rec[H + k] = sb_z; rec[2 * H + k] = sb_n; }
        if (half == 1) { rec[3 * H + k] = sb_r; rec[4 * H + k] = sb_z; rec[5 * H + k] = sb_hn; }
    }
}
// bias gradients: out[dir][j] = sum over clips b of bpart[2 b + dir][j] in clip order, j < 6H = db_ih (3H) | db_hh (3H)
__global__ __launch_bounds__(256) void gru_bias_reduce_kernel(const float* __restrict__ bpart, float* __restrict__ dbi0,
                                                              float* __restrict__ dbi1, float* __restrict__ dbh0,
                                                              float* __restrict__ dbh1, int B, int H) {
    const int j = blockIdx.x * 256 + threadIdx.x, dir = blockIdx.y;
    if (j >= 6 * H) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 3 < B; b += 4) {
        s0 += bpart[((size_t)2 * b + dir) * 6 * H + j]; s1 += bpart[((size_t)2 * (b + 1) + dir) * 6 * H + j];
        s2 += bpart[((size_t)2 * (b + 2) + dir) * 6 * H + j]; s3 += bpart[((size_t)2 * (b + 3) + dir) * 6 * H + j];
    }
    for (; b < B; ++b) s0 += bpart[((size_t)2 * b + dir) * 6 * H + j];
    const float v = (s0 + s1) + (s2 + s3);
    float* dbi = dir ? dbi1 : dbi0;
    float* dbh = dir ? dbh1 : dbh0;
    if (j < 3 * H) { if (dbi) dbi[j] = v; }
    else if (dbh) dbh[j - 3 * H] = v;
}
SED_API int sed_gru_bwd(const float* dout, const float* out, const float* saved, const float* whh0, const float* whh1,
                           float* dgi, float* dgh, float* hprev, float* dbi0, float* dbi1, float* dbh0, float* dbh1, int B, int T,
                           int H, float* scratch, void* stream) {
    if (H != 128 && H != 192) return SED_ERR_UNSUPPORTED;
    if ((dbi0 == nullptr) != (dbi1 == nullptr) || (dbh0 == nullptr) != (dbh1 == nullptr)) return SED_ERR_ARG;
    const bool want_bias = dbi0 || dbh0;
    if (B <= 0 || T <= 0) {
        if (want_bias) sed_zero4((hipStream_t)stream, dbi0, dbi0 ? 3 * H : 0, dbi1, dbi1 ? 3 * H : 0, dbh0, dbh0 ? 3 * H : 0, dbh1, dbh1 ? 3 * H : 0);
        return SED_OK;
    }
    if (want_bias && scratch == nullptr) return SED_ERR_ARG;
    float* bpart = scratch;                       // (scratch without bias pointers: records only, summed later by sed_gru_bias_reduce)
#define GRU_BWD_CASE(h, ch)                                                                                                       \
    if (H == h) {                                                                                                                 \
        int smem = (2 * ch * 6 * h + 2 * ch * 7 * (GRU_QSTORE ? h + 8 : h)) * 4 + (h <= 128 ? 0 : 8 * 4 * h * 16);               \
        smem = gru_lds_claim(smem);                                                                                               \
        SED_MAX_SMEM((gru_bwd_kernel<h, ch>), smem);                                                                              \
        SED_LAUNCH((gru_bwd_kernel<h, ch>), dim3(2 * B), dim3(4 * h), smem, (hipStream_t)stream, dout, out, saved, whh0, whh1, dgi, dgh, \
                   hprev, bpart, B, T);                                                                                           \
    }
    GRU_BWD_CASE(128, GRU_CH) GRU_BWD_CASE(192, 2)
#undef GRU_BWD_CASE
    if (sed_check_launch() != SED_OK) return SED_ERR_LAUNCH;
    if (want_bias)
        SED_LAUNCH(gru_bias_reduce_kernel, dim3((6 * H + 255) / 256, 2), dim3(256), 0, (hipStream_t)stream, (const float*)bpart, dbi0, dbi1,
                   dbh0, dbh1, B, H);
    return sed_check_launch();
}
// The second half of sed_gru_bwd on its own: the bias gradients from the records a sed_gru_bwd call WITHOUT bias pointers left in
// `scratch`.  Nothing on the backward chain reads them (the optimizer does), so a caller can run this beside the chain.
SED_API int sed_gru_bias_reduce(const float* scratch, float* dbi0, float* dbi1, float* dbh0, float* dbh1, int B, int H, void* stream) {
    if (H != 128 && H != 192) return SED_ERR_UNSUPPORTED;
    if ((dbi0 == nullptr) != (dbi1 == nullptr) || (dbh0 == nullptr) != (dbh1 == nullptr)) return SED_ERR_ARG;
    if (!dbi0 && !dbh0) return SED_OK;
    if (B <= 0) { sed_zero4((hipStream_t)stream, dbi0, dbi0 ? 3 * H : 0, dbi1, dbi1 ? 3 * H : 0, dbh0, dbh0 ? 3 * H : 0, dbh1, dbh1 ? 3 * H : 0); return SED_OK; }
    if (scratch == nullptr) return SED_ERR_ARG;
    SED_LAUNCH(gru_bias_reduce_kernel, dim3((6 * H + 255) / 256, 2), dim3(256), 0, (hipStream_t)stream, scratch, dbi0, dbi1, dbh0, dbh1, B, H);
    return sed_check_launch();
}
