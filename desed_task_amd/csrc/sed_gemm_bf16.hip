// Split-bf16 ("bf16x3") variant of the K7 GEMMs (GRU input projections, dW_ih / dW_hh, dX): same contract as
// sed_gemm / sed_gemm_pair in sed_gru.hip, fp32 in / fp32 out / fp32 accumulate, but every product is issued as
// three v_mfma_f32_32x32x16_bf16 on operands split x = hi + lo (hi = bf16(x), lo = bf16(x - hi)): hi*hi + hi*lo +
// lo*hi, ~8e-6 relative on a dot product (see sed_common.h) at 3/16 of the f32-MFMA cost.  The f32 kernels sit at
// 27-40 % of the f32 MFMA peak on these shapes (K = 128..384 is only 4-12 K tiles per workgroup); with the cheaper
// MFMA the launches become streaming-bound.
//
// Tile 128 x (32*NTN) x 32, 4 waves (wave w: rows 32w..32w+31 x all columns).  Both operands are staged into LDS as
// bf16 hi / lo planes in [row][k] order (row stride 40 bf16 = 80 B: the 16-byte A/B fragment reads of a quarter wave
// hit 16 distinct 4-bank groups), whatever their layout in HBM:
//   k-contiguous operand (A with TA = 0, B with TB = 1): a float4 along k -> one 8-byte store per plane;
//   row-contiguous operand (A with TA = 1, B with TB = 0): two float4 along the row for k, k+1 -> four 4-byte stores
//   per plane (the transposition happens in the LDS write).
// The next K tile is prefetched into registers under the MFMAs of the current one.
#include "sed_common.h"

namespace {

constexpr int GB_BM = 128, GB_BK = 32, GB_RS = 40;

__device__ __forceinline__ void split4(const float4 v, uint2& h, uint2& l) {
    bf16_split2(v.x, v.y, h.x, l.x);
    bf16_split2(v.z, v.w, h.y, l.y);
}

// One operand tile of ROWS rows x 32 k.  KC = true: element (row, k) at base[row * ld + k]; false: base[k * ld + row].
// NV float4 per thread.  load(): global -> registers (zero outside [0, nrows) x [k0, kend)); store(): registers -> LDS planes.
template <int ROWS, bool KC>
struct OperandTile {
    static constexpr int NV = ROWS * GB_BK / 4 / 256;
    float4 r[NV];
    __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int row0, int nrows, int k0, int kend, int tid) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            r[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KC) {
                const int i = tid + 256 * u, row = i / (GB_BK / 4), kq = i % (GB_BK / 4);
                const int gr = row0 + row, gk = k0 + 4 * kq;
                if (gr < nrows && gk < kend) r[u] = *(const float4*)(base + (size_t)gr * ld + gk);
            } else {
                // pair p = (k pair kp, row quad rq); this thread's float4 #u: pair (tid + 256 * (u / 2)), k = 2 kp + (u & 1)
                const int p = tid + 256 * (u >> 1), kp = p & 3, rest = p >> 2;
                const int rq = rest % (ROWS / 4), kph = rest / (ROWS / 4);
                const int gk = k0 + 2 * (kp + 4 * kph) + (u & 1), gr = row0 + 4 * rq;
                if (gr < nrows && gk < kend) r[u] = *(const float4*)(base + (size_t)gk * ld + gr);
            }
        }
    }
    __device__ __forceinline__ void store(unsigned short* __restrict__ hi_plane, unsigned short* __restrict__ lo_plane, int tid) const {
        if (KC) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int i = tid + 256 * u, row = i / (GB_BK / 4), kq = i % (GB_BK / 4);
                uint2 h, l;
                split4(r[u], h, l);
                *(uint2*)(hi_plane + row * GB_RS + 4 * kq) = h;
                *(uint2*)(lo_plane + row * GB_RS + 4 * kq) = l;
            }
        } else {
#pragma unroll
            for (int u = 0; u < NV; u += 2) {
                const int p = tid + 256 * (u >> 1), kp = p & 3, rest = p >> 2;
                const int rq = rest % (ROWS / 4), kph = rest / (ROWS / 4);
                const int k = 2 * (kp + 4 * kph);
                uint2 h0, l0, h1, l1;                          // rows 4rq..4rq+3 at k (r[u]) and k+1 (r[u+1])
                split4(r[u], h0, l0);
                split4(r[u + 1], h1, l1);
                const unsigned hk[4] = {h0.x & 0xFFFFu, h0.x >> 16, h0.y & 0xFFFFu, h0.y >> 16};
                const unsigned hk1[4] = {h1.x & 0xFFFFu, h1.x >> 16, h1.y & 0xFFFFu, h1.y >> 16};
                const unsigned lk[4] = {l0.x & 0xFFFFu, l0.x >> 16, l0.y & 0xFFFFu, l0.y >> 16};
                const unsigned lk1[4] = {l1.x & 0xFFFFu, l1.x >> 16, l1.y & 0xFFFFu, l1.y >> 16};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    *(unsigned*)(hi_plane + (4 * rq + j) * GB_RS + k) = hk[j] | (hk1[j] << 16);
                    *(unsigned*)(lo_plane + (4 * rq + j) * GB_RS + k) = lk[j] | (lk1[j] << 16);
                }
            }
        }
    }
};

// (waves-per-SIMD hint 3: without it the allocator spreads the accumulators over 64 AGPRs next to 180 VGPRs -- two workgroups per CU;
//  with it 141 - 168 registers, no AGPRs, no spills: three workgroups per CU.  BEATs linears 17.16 -> 15.49 ms per 48 clips, same box.)
template <int TA, int TB, int NTN>
__global__ __launch_bounds__(256, 3) void gemm_bf16x3_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                          const float* __restrict__ bias, float* __restrict__ Cm, int M, int N, int K,
                                                          int lda, int ldb, int ldc, int k_per_slice, int atomic,
                                                          const float* __restrict__ A1, const float* __restrict__ B1,
                                                          const float* __restrict__ bias1, float* __restrict__ C1, int nbatch,
                                                          const float* __restrict__ Bsw, int ksw, int act, float* __restrict__ part) {
    // Bsw != null: K-concatenated B -- rows k >= ksw come from Bsw (already offset by -ksw rows); ksw % 32 == 0
    constexpr int BM = GB_BM, BN = 32 * NTN, BK = GB_BK, RS = GB_RS;
    __shared__ __attribute__((aligned(16))) unsigned short As[2 * BM * RS];     // hi plane, lo plane
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2 * BN * RS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int zb = nbatch == 2 ? (blockIdx.z & 1) : 0, zs = nbatch == 2 ? (blockIdx.z >> 1) : blockIdx.z;
    if (zb) { A = A1; Bm = B1; bias = bias1; Cm = C1; }          // second problem of a batch of two
    if (part) { Cm = part + (size_t)blockIdx.z * M * N; ldc = N; }  // split-K slices as dense [z][M][N] partials (plain stores), summed
                                                                    // in a fixed order by splitk_reduce_kernel instead of atomics
    const int kbeg = zs * k_per_slice, kend = min(K, kbeg + k_per_slice);
    f32x16 acc[NTN];
#pragma unroll
    for (int i = 0; i < NTN; ++i) acc[i] = f32x16_zero();
    OperandTile<BM, TA == 0> ta;
    OperandTile<BN, TB == 1> tb;

    if (kbeg < kend) { ta.load(A, lda, m0, M, kbeg, kend, tid); tb.load((Bsw && kbeg >= ksw) ? Bsw : Bm, ldb, n0, N, kbeg, kend, tid); }
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();                 // everyone finished reading the previous tile
        ta.store(As, As + BM * RS, tid);
        tb.store(Bs, Bs + BN * RS, tid);
        __syncthreads();
        if (k0 + BK < kend) { ta.load(A, lda, m0, M, k0 + BK, kend, tid); tb.load((Bsw && k0 + BK >= ksw) ? Bsw : Bm, ldb, n0, N, k0 + BK, kend, tid); }
        const unsigned short* ap = As + (32 * w + lo) * RS + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const s16x8 a_hi = *(const s16x8*)(ap + 16 * ks);
            const s16x8 a_lo = *(const s16x8*)(ap + BM * RS + 16 * ks);
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                const unsigned short* bp = Bs + (nt * 32 + lo) * RS + 16 * ks + 8 * hi;
                const s16x8 b_hi = *(const s16x8*)bp;
                const s16x8 b_lo = *(const s16x8*)(bp + BN * RS);
                acc[nt] = mfma32_bf16(a_lo, b_hi, acc[nt]);
                acc[nt] = mfma32_bf16(a_hi, b_lo, acc[nt]);
                acc[nt] = mfma32_bf16(a_hi, b_hi, acc[nt]);
            }
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
                    float v = acc[nt][r] + bv;
                    if (act == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));      // exact GELU (sed_linear_bf16x3)
                    if (atomic) atomicAdd(dst, v); else *dst = v;
                }
            }
        }
    }
}

}  // namespace

// defined in sed_gru.hip: the exact-f32 path, used when the operands do not meet the 16-byte requirements below
SED_API int sed_gemm(const float* A, const float* Bm, const float* bias, float* Cm, int M, int N, int K, int lda, int ldb,
                        int ldc, int transA, int transB, int split_k, int accumulate, void* stream);
SED_API int sed_gemm_pair(const float* A0, const float* A1, const float* B0, const float* B1, const float* bias0,
                             const float* bias1, float* C0, float* C1, int M, int N, int K, int lda, int ldb, int ldc, int transA,
                             int transB, int split_k, int accumulate, void* stream);

static int gemmb_dispatch(const float* A, const float* Bm, const float* bias, float* Cm, const float* A1, const float* B1,
                          const float* bias1, float* C1, int nbatch, int M, int N, int K, int lda, int ldb, int ldc, int transA,
                          int transB, int split_k, int accumulate, hipStream_t s, const float* Bsw = nullptr, int ksw = 0, int act = 0,
                          float* part = nullptr) {
    if (M <= 0 || N <= 0 || K <= 0) return SED_OK;
    bool ok = ((uintptr_t)A % 16 == 0) && ((uintptr_t)Bm % 16 == 0) && lda % 4 == 0 && ldb % 4 == 0 &&
              ((transA ? M : K) % 4 == 0) && ((transB ? K : N) % 4 == 0) && !(transA && transB);
    if (nbatch == 2) ok = ok && ((uintptr_t)A1 % 16 == 0) && ((uintptr_t)B1 % 16 == 0);
    if (!ok && (Bsw || act || part)) return SED_ERR_UNSUPPORTED;
    if (!ok) {
        if (nbatch == 2) return sed_gemm_pair(A, A1, Bm, B1, bias, bias1, Cm, C1, M, N, K, lda, ldb, ldc, transA, transB, split_k, accumulate, s);
        return sed_gemm(A, Bm, bias, Cm, M, N, K, lda, ldb, ldc, transA, transB, split_k, accumulate, s);
    }
    if (split_k < 1) split_k = 1;
    int kps = ((K + split_k - 1) / split_k + 31) / 32 * 32;
    split_k = (K + kps - 1) / kps;
    const int atomic = (!part && (split_k > 1 || accumulate)) ? 1 : 0;
    int ntn = N > 64 ? 4 : 2;
    if (ntn == 4 && ((N + 127) / 128) * ((M + 127) / 128) * split_k * nbatch < 400) ntn = 2;     // too few workgroups for 512 resident slots
    dim3 grid((N + 32 * ntn - 1) / (32 * ntn), (M + 127) / 128, split_k * nbatch);
#define GEMMB_CASE(ta, tb, nn) \
    if (transA == ta && transB == tb && ntn == nn) { SED_LAUNCH((gemm_bf16x3_kernel<ta, tb, nn>), grid, dim3(256), 0, s, A, Bm, bias, Cm, M, N, K, lda, ldb, ldc, kps, atomic, A1, B1, bias1, C1, nbatch, Bsw, ksw, act, part); return sed_check_launch(); }
    GEMMB_CASE(0, 0, 2) GEMMB_CASE(0, 0, 4) GEMMB_CASE(0, 1, 2) GEMMB_CASE(0, 1, 4) GEMMB_CASE(1, 0, 2) GEMMB_CASE(1, 0, 4)
#undef GEMMB_CASE
    return SED_ERR_UNSUPPORTED;
}

// Same contract as sed_gemm, split-bf16 products (fp32-level accuracy, ~8e-6 relative).
SED_API int sed_gemm_bf16x3(const float* A, const float* Bm, const float* bias, float* Cm, int M, int N, int K, int lda, int ldb,
                               int ldc, int transA, int transB, int split_k, int accumulate, void* stream) {
    return gemmb_dispatch(A, Bm, bias, Cm, nullptr, nullptr, nullptr, nullptr, 1, M, N, K, lda, ldb, ldc, transA, transB, split_k,
                          accumulate, (hipStream_t)stream);
}
// Same contract as sed_gemm_pair, split-bf16 products.
SED_API int sed_gemm_pair_bf16x3(const float* A0, const float* A1, const float* B0, const float* B1, const float* bias0,
                                    const float* bias1, float* C0, float* C1, int M, int N, int K, int lda, int ldb, int ldc,
                                    int transA, int transB, int split_k, int accumulate, void* stream) {
    return gemmb_dispatch(A0, B0, bias0, C0, A1, B1, bias1, C1, 2, M, N, K, lda, ldb, ldc, transA, transB, split_k, accumulate,
                          (hipStream_t)stream);
}

// Split-K pair without atomics: the slices are written as dense partials into `scratch` (sed_gemm_splitk_scratch_floats floats)
// and summed in slice order by one small kernel -- deterministic, C needs no zero fill, and at the BiGRU weight-gradient shapes
// (M = 384, N = 128 / 256, K = 7488, 26 slices) faster than 5 M fp32 atomics on 98 K addresses (31 / 43 us per pair).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ C0, float* __restrict__ C1,
                                                            int nslices, int M, int N, int ldc) {
    const int MN4 = M * N / 4, i = blockIdx.x * 256 + threadIdx.x, zb = blockIdx.y, nb = gridDim.y;      // nb: 2 = pair, 1 = single
    if (i >= MN4) return;
    const float4* src = (const float4*)part + (size_t)zb * MN4 + i;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int z = 0; z < nslices; ++z) {
        const float4 v = src[(size_t)nb * z * MN4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const int e = 4 * i, m = e / N, n = e - m * N;
    *(float4*)((zb ? C1 : C0) + (size_t)m * ldc + n) = acc;
}
static inline int splitk_slices(int K, int split_k) {
    if (split_k < 1) split_k = 1;
    const int kps = ((K + split_k - 1) / split_k + 31) / 32 * 32;
    return (K + kps - 1) / kps;
}
SED_API long long sed_gemm_splitk_scratch_floats(int M, int N, int K, int split_k) {
    return 2LL * splitk_slices(K, split_k) * M * N;
}
SED_API int sed_gemm_pair_splitk_bf16x3(const float* A0, const float* A1, const float* B0, const float* B1, float* C0, float* C1,
                                           int M, int N, int K, int lda, int ldb, int ldc, int transA, int transB, int split_k,
                                           float* scratch, void* stream) {
    if (M <= 0 || N <= 0) return SED_OK;
    if (!scratch || N % 4 != 0 || ldc % 4 != 0 || K <= 0) return SED_ERR_ARG;
    if ((((uintptr_t)C0 | (uintptr_t)C1) & 15) != 0) return SED_ERR_UNSUPPORTED;       // float4 stores in the reduce
    hipStream_t s = (hipStream_t)stream;
    const int rc = gemmb_dispatch(A0, B0, nullptr, C0, A1, B1, nullptr, C1, 2, M, N, K, lda, ldb, ldc, transA, transB, split_k, 0, s,
                                  nullptr, 0, 0, scratch);
    if (rc != SED_OK) return rc;
    SED_LAUNCH(splitk_reduce_kernel, dim3((M * N / 4 + 255) / 256, 2), dim3(256), 0, s, (const float*)scratch, C0, C1,
               splitk_slices(K, split_k), M, N, ldc);
    return sed_check_launch();
}

// One product with the same deterministic split-K (the `cat_tf` weight gradient of the embedding recipes: dW = dy^T . z over K = B T
// rows; until round 4 it accumulated its slices with float atomics into a zero-filled dW).  scratch: sed_gemm_splitk_scratch_floats.
SED_API int sed_gemm_splitk_bf16x3(const float* A, const float* Bm, float* Cm, int M, int N, int K, int lda, int ldb, int ldc,
                                   int transA, int transB, int split_k, float* scratch, void* stream) {
    if (M <= 0 || N <= 0) return SED_OK;
    if (!scratch || N % 4 != 0 || ldc % 4 != 0 || K <= 0) return SED_ERR_ARG;
    if (((uintptr_t)Cm & 15) != 0) return SED_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int rc = gemmb_dispatch(A, Bm, nullptr, Cm, nullptr, nullptr, nullptr, nullptr, 1, M, N, K, lda, ldb, ldc, transA, transB,
                                  split_k, 0, s, nullptr, 0, 0, scratch);
    if (rc != SED_OK) return rc;
    SED_LAUNCH(splitk_reduce_kernel, dim3((M * N / 4 + 255) / 256, 1), dim3(256), 0, s, (const float*)scratch, Cm, Cm,
               splitk_slices(K, split_k), M, N, ldc);
    return sed_check_launch();
}

// C[M][N] = A[M][K] . [B0 ; B1]: the B operand is two row-major tensors stacked along K (rows [0, ksplit) from B0, the rest
// from B1; ksplit % 32 == 0) -- dX of a bidirectional GRU layer, whose dgi rows hold both directions side by side.
SED_API int sed_gemm_kcat_bf16x3(const float* A, const float* B0, const float* B1, float* Cm, int M, int N, int K, int ksplit,
                                    int lda, int ldb, int ldc, void* stream) {
    if (ksplit % 32 != 0 || ksplit <= 0 || ksplit >= K) return SED_ERR_ARG;
    return gemmb_dispatch(A, B0, nullptr, Cm, nullptr, nullptr, nullptr, nullptr, 1, M, N, K, lda, ldb, ldc, 0, 0, 1, 0,
                          (hipStream_t)stream, B1 - (size_t)ksplit * ldb, ksplit);
}

// The same product with a deterministic split-K (dense per-slice partials in `scratch`, sed_gemm_splitk_scratch_floats floats, summed
// in slice order).  At the BiGRU dX shapes -- M = 7488, N = 128 / 256, K = 768 -- one slice is 118 / 236 workgroups walking 24
// dependent K tiles each with the chip half empty (37 - 46 us, on the backward chain between two recurrences); 6 / 3 slices are ~ 700
// workgroups of 4 / 8 tiles, three per CU.
SED_API int sed_gemm_kcat_splitk_bf16x3(const float* A, const float* B0, const float* B1, float* Cm, int M, int N, int K, int ksplit,
                                           int lda, int ldb, int ldc, int split_k, float* scratch, void* stream) {
    if (ksplit % 32 != 0 || ksplit <= 0 || ksplit >= K) return SED_ERR_ARG;
    if (M <= 0 || N <= 0) return SED_OK;
    if (!scratch || N % 4 != 0 || ldc % 4 != 0) return SED_ERR_ARG;
    if (((uintptr_t)Cm & 15) != 0) return SED_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int rc = gemmb_dispatch(A, B0, nullptr, Cm, nullptr, nullptr, nullptr, nullptr, 1, M, N, K, lda, ldb, ldc, 0, 0, split_k, 0, s,
                                  B1 - (size_t)ksplit * ldb, ksplit, 0, scratch);
    if (rc != SED_OK) return rc;
    SED_LAUNCH(splitk_reduce_kernel, dim3((M * N / 4 + 255) / 256, 1), dim3(256), 0, s, (const float*)scratch, Cm, Cm,
               splitk_slices(K, split_k), M, N, ldc);
    return sed_check_launch();
}

// torch.nn.Linear forward with an optional fused activation: C[M][N] = act(A[M][K] . W[N][K]^T + bias[N]), act 0 = none, 1 = exact
// GELU (the FFN of the BEATs encoder layers).  16-byte aligned operands, K % 4 == 0.
SED_API int sed_linear_bf16x3(const float* A, const float* W, const float* bias, float* Cm, int M, int N, int K, int act,
                                 void* stream) {
    if (act < 0 || act > 1) return SED_ERR_ARG;
    return gemmb_dispatch(A, W, bias, Cm, nullptr, nullptr, nullptr, nullptr, 1, M, N, K, K, K, N, 0, 1, 1, 0, (hipStream_t)stream,
                          nullptr, 0, act);
}
