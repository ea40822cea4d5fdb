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
    // k-contiguous staging: eight lanes cover the 32 k of a row; the two rows of a 16-lane ds_write_b64 group are r and r + 4, whose
    // 64-byte pieces sit 320 B = 16 banks (mod 32) apart -- with rows r and r + 1 (80 B apart) four banks of every store were hit
    // twice (lds_conflict 0.31 of the LDS cycles, profiles/r05i_pmc_beats_wait.md)
    static __device__ __forceinline__ int kc_row(int i) {
        const int grp = i >> 4;
        return ((grp >> 2) << 3) + (grp & 3) + ((i >> 1) & 4);
    }
    __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int row0, int nrows, int k0, int kend, int tid) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            r[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KC) {
                const int i = tid + 256 * u, row = kc_row(i), kq = i & 7;
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
                const int i = tid + 256 * u, row = kc_row(i), kq = i & 7;
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
                                                          const float* __restrict__ Bsw, int ksw, int act, float* __restrict__ part,
                                                          int walk_nt) {
    // Bsw != null: K-concatenated B -- rows k >= ksw come from Bsw (already offset by -ksw rows); ksw % 32 == 0
    constexpr int BM = GB_BM, BN = 32 * NTN, BK = GB_BK, RS = GB_RS;
    __shared__ __attribute__((aligned(16))) unsigned short As[2 * BM * RS];     // hi plane, lo plane
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2 * BN * RS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lo = lane & 31, hi = lane >> 5;
    // walk_nt > 0 (the BEATs linears: one problem, no K slices, 1-D grid): XCD-aware tile walk.  Workgroup g runs on XCD g & 7; with
    // the N tile as the fastest grid index the walk_nt workgroups that share a 128-row panel of A sat on as many XCDs, each L2 fetched
    // the panel for itself (FC2: 6 x 292 MB per launch -- the launch ran at HBM speed, not at the matrix pipe's).  Now XCD x takes the
    // panels x, x + 8, ... and its consecutive workgroups sweep one panel's N tiles.
    int tile_m = blockIdx.y, tile_n = blockIdx.x;
    if (walk_nt > 0) {
        const int ml = blockIdx.x >> 3;
        tile_m = (blockIdx.x & 7) + 8 * (ml / walk_nt);
        tile_n = ml % walk_nt;
        if (tile_m * BM >= M) return;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
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
                          float* part = nullptr, bool act_linear = false) {
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
    if (ntn == 4 && !transA && transB && N % 96 == 0) {
        // more than one round of the 768 resident workgroups (3 per CU): take the tile width whose last round is fuller.  BEATs out-proj /
        // FC2 (M = 23 808, N = 768): 1 116 tiles of 128 x 128 = 1.45 rounds, 1 488 of 128 x 96 = 1.94
        const long long rows = (M + 127) / 128, z = (long long)split_k * nbatch, slots = 768;
        const long long t4 = rows * ((N + 127) / 128) * z, t3 = rows * (N / 96) * z;
        if (sed_tuning[SED_TUNE_GEMM_NTN] == 3) ntn = 3;              // (tests: the 128 x 96 tile at small sizes)
        else if (t4 > slots) {
            const double e4 = (double)t4 / (double)(((t4 + slots - 1) / slots) * slots), e3 = (double)t3 / (double)(((t3 + slots - 1) / slots) * slots);
            if (e3 > e4 + 0.1) ntn = 3;
        }
    }
    dim3 grid((N + 32 * ntn - 1) / (32 * ntn), (M + 127) / 128, split_k * nbatch);
    int walk_nt = 0;
    if (act_linear && split_k * nbatch == 1 && grid.y >= 16) {        // (act_linear: the call came through sed_linear_bf16x3)
        walk_nt = (int)grid.x;
        grid = dim3(grid.x * ((grid.y + 7) / 8) * 8, 1, 1);
    }
#define GEMMB_CASE(ta, tb, nn) \
    if (transA == ta && transB == tb && ntn == nn) { SED_LAUNCH((gemm_bf16x3_kernel<ta, tb, nn>), grid, dim3(256), 0, s, A, Bm, bias, Cm, M, N, K, lda, ldb, ldc, kps, atomic, A1, B1, bias1, C1, nbatch, Bsw, ksw, act, part, walk_nt); return sed_check_launch(); }
    GEMMB_CASE(0, 0, 2) GEMMB_CASE(0, 0, 4) GEMMB_CASE(0, 1, 2) GEMMB_CASE(0, 1, 3) GEMMB_CASE(0, 1, 4) GEMMB_CASE(1, 0, 2) GEMMB_CASE(1, 0, 4)
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

// ---- round 5: the large frozen-weight linears of the BEATs encoder (M = 23 808 tokens, N, K in {768, 2304, 3072}) -------------------
// gemm_bf16x3_kernel above stages BOTH operands through LDS as split planes it forms on the fly: per 32-wide K tile 16 ds_write_b64
// per thread (~6 LDS cycles each) next to the fragment reads -- the LDS pipe, not the matrix pipe, set its pace (235 - 260 TFLOP/s =
// 28 - 31 % of the 833 TFLOP/s that three bf16 MFMAs per product allow).  Here
//   * W is split ONCE (frozen weights: sed_pack_weights_bf16x3 -> [hi | lo][N][K] bf16 planes), so its tiles travel HBM -> LDS as plain
//     16-byte copies, double-buffered: one workgroup barrier per K tile;
//   * A never touches LDS: a wave owns 64 rows x all 128 columns of the 256 x 128 tile, and the MFMA A fragment of lane (row, k-half) is
//     8 consecutive floats of that row -- two 16-byte global loads, split into hi / lo in registers (5 VALU per pair, hidden behind
//     the 24 MFMAs of the k-step), prefetched one k-step ahead;
//   * per k-step (16 k) a wave issues 24 MFMAs against 8 ds_read_b128 (B) + 4 global_load_dwordx4 (A);
//   * tiles are walked so that the workgroups of one XCD share an A row panel (its L2) while they sweep the N tiles.
namespace {
constexpr int LB_BM = 256, LB_BN = 128, LB_BK = 32, LB_RS = 40;     // LDS row pitch 40 bf16 = 80 B = 5 sixteen-byte slots (odd)

__device__ __forceinline__ void split8(const float4 p, const float4 q, s16x8& h, s16x8& l) {
    uint4 hh, ll;
    bf16_split2(p.x, p.y, hh.x, ll.x); bf16_split2(p.z, p.w, hh.y, ll.y);
    bf16_split2(q.x, q.y, hh.z, ll.z); bf16_split2(q.z, q.w, hh.w, ll.w);
    h = __builtin_bit_cast(s16x8, hh); l = __builtin_bit_cast(s16x8, ll);
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void linear_big_kernel(const float* __restrict__ A, const unsigned short* __restrict__ Wp,
                                                         const float* __restrict__ bias, float* __restrict__ Cm, int M, int N, int K,
                                                         int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][2 * LB_BN * LB_RS];        // [stage][hi | lo][n][k]
    const int tid = threadIdx.x, lane = tid & 63, w = sed_wave_uniform(tid >> 6), lo = lane & 31, hi = lane >> 5;
    // tile walk: workgroup g runs on XCD g & 7; XCD x takes the row panels tm = x (mod 8) and sweeps their N tiles back to back
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int panels_here = (tiles_m - xcd + 7) >> 3;
    for (int tile = slot; tile < panels_here * tiles_n; tile += per_xcd) {
        const int tm = xcd + 8 * (tile / tiles_n), tn = tile % tiles_n;
        const int m0 = tm * LB_BM, n0 = tn * LB_BN;
        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x16_zero();
        // B tile copy: 2 planes x 128 rows x 64 B = 1024 sixteen-byte chunks, 4 per thread: chunk c = tid + 256 u ->
        // plane c >> 9, row (c >> 2) & 127, 16-byte piece c & 3 of the row's 64 bytes
        // (the two rows of an 8-lane ds_write_b128 group are r and r + 4: 320 B apart = 16 banks mod 32, see OperandTile::kc_row)
        const int brow = (((tid >> 3) >> 2) << 3) + ((tid >> 3) & 3) + (tid & 4), bq = tid & 3;     // u = 0..3: plane = u >> 1, row = brow + 64 (u & 1)
        const unsigned short* bsrc = Wp + ((size_t)(n0 + brow)) * K + 8 * bq;
        const size_t bplane = (size_t)N * K, brow64 = (size_t)64 * K;
        const int bdst = brow * LB_RS + 8 * bq;
        uint4 b0, b1, b2, b3;
#define LB_BLOAD(k0_) { b0 = *(const uint4*)(bsrc + (k0_)); b1 = *(const uint4*)(bsrc + brow64 + (k0_)); \
                        b2 = *(const uint4*)(bsrc + bplane + (k0_)); b3 = *(const uint4*)(bsrc + bplane + brow64 + (k0_)); }
#define LB_BSTORE(st_) { unsigned short* d_ = &Bs[st_][bdst]; *(uint4*)d_ = b0; *(uint4*)(d_ + 64 * LB_RS) = b1; \
                         *(uint4*)(d_ + LB_BN * LB_RS) = b2; *(uint4*)(d_ + LB_BN * LB_RS + 64 * LB_RS) = b3; }
        // A fragments: rows m0 + 64 w + 32 rb + lo (clamped: rows past M are computed and dropped).  One 32-wide K tile of a row is
        // one 128-byte line; lane (row, k-half hi) needs its bytes [32 hi, 32 hi + 32) (k-step 0) and [64 + 32 hi, ...) (k-step 1).  All four
        // 16-byte loads of a row block go out TOGETHER, so a line is fetched into the CU once (requested per k-step, 768 MFMA-cycles apart,
        // every line came over from L2 twice: the eight waves' lines of one k-step alone are 64 KB) -- and they go out one row block ahead:
        // while block rb computes its 24 MFMAs, the other block's lines (same K tile, or the next one) are in flight.
        int r0 = m0 + 64 * w + lo, r1 = r0 + 32;
        r0 = r0 < M ? r0 : M - 1;
        r1 = r1 < M ? r1 : M - 1;
        const float* arow0 = A + (size_t)r0 * K + 8 * hi;
        const float* arow1 = A + (size_t)r1 * K + 8 * hi;
        float4 a0, a1, a2, a3;              // k-step 0: a0 a1, k-step 1: a2 a3 of the row block in flight
#define LB_ALOAD(row_, k_) { a0 = *(const float4*)((row_) + (k_)); a1 = *(const float4*)((row_) + (k_) + 4); \
                             a2 = *(const float4*)((row_) + (k_) + 16); a3 = *(const float4*)((row_) + (k_) + 20); }
#define LB_RB(rb_, next_row_, next_k_) { \
            s16x8 ah0, al0, ah1, al1; \
            split8(a0, a1, ah0, al0); \
            split8(a2, a3, ah1, al1); \
            LB_ALOAD(next_row_, next_k_) \
            sed_sched_fence(); \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { \
                const s16x8 ah = ks ? ah1 : ah0, al = ks ? al1 : al0; \
                s16x8 bh[4], bl[4]; \
                _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) { \
                    const unsigned short* bp = &Bs[stage][(nt * 32 + lo) * LB_RS + 16 * ks + 8 * hi]; \
                    bh[nt] = *(const s16x8*)bp; \
                    bl[nt] = *(const s16x8*)(bp + LB_BN * LB_RS); \
                } \
                _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) acc[rb_][nt] = mfma32_bf16(al, bh[nt], acc[rb_][nt]); \
                _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) acc[rb_][nt] = mfma32_bf16(ah, bl[nt], acc[rb_][nt]); \
                _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) acc[rb_][nt] = mfma32_bf16(ah, bh[nt], acc[rb_][nt]); \
            } }
        __syncthreads();                    // (the previous tile's last reads of both stages)
        LB_BLOAD(0)
        LB_ALOAD(arow0, 0)
        LB_BSTORE(0)
        __syncthreads();
        const int nk = K / LB_BK;
        for (int kt = 0; kt < nk; ++kt) {
            const int stage = kt & 1;
            if (kt + 1 < nk) LB_BLOAD((kt + 1) * LB_BK)
            const int k0 = kt * LB_BK;
            const int k1 = kt + 1 < nk ? k0 + LB_BK : k0;       // (past the end: the last tile again, unused)
            LB_RB(0, arow1, k0)
            LB_RB(1, arow0, k1)
            if (kt + 1 < nk) LB_BSTORE(stage ^ 1)
            __syncthreads();                // stage ^ 1 is complete; everybody is done with `stage` before iteration kt + 2 rewrites it
        }
#undef LB_RB
#undef LB_BLOAD
#undef LB_BSTORE
#undef LB_ALOAD
        // epilogue: bias (+ exact GELU); lane holds column n0 + 32 nt + lo, rows mfma32_row(r, lane)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int gn = n0 + nt * 32 + lo;
            const float bv = bias != nullptr ? bias[gn] : 0.f;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = m0 + 64 * w + 32 * rb + mfma32_row(r, lane);
                    float v = acc[rb][nt][r] + bv;
                    if (ACT == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                    if (gm < M) Cm[(size_t)gm * N + gn] = v;
                }
        }
    }
}

}  // namespace
// ---- round 6: 256 x 256 tiles, eight waves, both operands through double-buffered LDS, one K tile of register prefetch ----------------
// Where linear_big_kernel stood (profiles/r05j_pmc_beats_wait.md, r05a_pmc_linear.md): MFMA-busy 0.29 - 0.36 with the LDS pipe only ~30 %
// busy -- each wave waits for its OWN fragment-shaped A loads (fp32 rows straight from L2 / HBM, one row block = 24 MFMAs = ~0.3 us of
// cover per load) and the 256 x 128 tile asks L2 for 1.5 KB per k (43 flop / B: ~10 TB/s at the target rate, 5.4 x the algorithmic
// bytes from HBM).  Here:
//   * tile 256 x 256 x 32: 2 KB per k for twice the products (64 flop / B); a wave owns 128 x 64 of it (4 x 2 MFMA blocks: 12 fragment
//     reads per 24 MFMAs);
//   * A (fp32 in HBM) and W (packed hi / lo planes) both travel global -> registers -> LDS as whole 128-byte / 64-byte row pieces, one K
//     tile AHEAD: the loads of tile k + 2 are issued during the MFMAs of tile k, their registers are split (A) and parked in the other
//     LDS buffer during tile k + 1 -- a full K tile (48 MFMAs per wave, two waves per SIMD: ~1.3 us) of memory latency is covered;
//   * LDS planes are unpadded [row][32 k] bf16 (64-byte rows), the 16-byte octet o of row r stored at slot o ^ ((r >> 2) & 3): every
//     ds_read_b128 lane group of an MFMA fragment read ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and their upper twins) then covers the
//     64 banks exactly once, and the 8-lane row pieces of the staging stores alternate between the two bank halves;
//   * 2 buffers x (A hi | lo + W hi | lo) x 16 KB = 128 KB: one workgroup per CU, 8 waves, <= 256 VGPRs (128 of them accumulators);
//   * XCD-aware walk: the N tiles of a 256-row panel of A are consecutive workgroups of ONE XCD.
namespace {
constexpr int P_BM = 256, P_BN = 256, P_BK = 32, P_PLANE = 256 * 32;        // one plane of one operand: 256 rows x 32 k (ushorts)
__device__ __forceinline__ int p_off(int row, int oct) { return row * P_BK + ((oct ^ ((row >> 2) & 3)) << 3); }

template <int ACT>
__global__ __launch_bounds__(512, 1) void linear_p256_kernel(const float* __restrict__ A, const unsigned short* __restrict__ Wp,
                                                            const float* __restrict__ bias, float* __restrict__ Cm, int M, int N, int K,
                                                            int tiles_m, int tiles_n) {
    SED_DYN_SMEM(smem);                               // [2 buffers][A hi | A lo | W hi | W lo][256][32] bf16
    unsigned short* lds = (unsigned short*)smem;
    const int tid = threadIdx.x, lane = tid & 63, w = sed_wave_uniform(tid >> 6), lo = lane & 31, hi = lane >> 5;
    const int wr = w >> 2, wc = w & 3;                // wave: rows 128 wr .. + 127, columns 64 wc .. + 63 of the tile
    // tile walk: workgroup g runs on XCD g & 7; XCD x owns the row panels tm = x (mod 8) and sweeps their N tiles back to back
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tm = xcd + 8 * (slot / tiles_n), tn = slot % tiles_n;
    if (tm >= tiles_m) return;
    const int m0 = tm * P_BM, n0 = tn * P_BN;
    // staging maps.  A: thread -> rows (tid >> 3) + 64 u, float4 q = tid & 7 of the row's 32 floats (8 lanes = one 128-byte line).
    // W: thread -> 16-byte pieces p = tid + 512 u: plane p >> 10, row (p >> 2) & 255, octet p & 3 (4 lanes = one 64-byte row piece).
    const int arow = tid >> 3, aq = tid & 7;
    // (named registers, not arrays: an indexed array of prefetch registers captured by a lambda ends up in scratch memory)
    auto arow_ptr = [&](int u) {
        int r = m0 + arow + 64 * u;
        r = r < M ? r : M - 1;                        // rows past M: computed on a clamped row, never stored
        return A + (size_t)r * K + 4 * aq;
    };
    const float* const asrc0 = arow_ptr(0);
    const float* const asrc1 = arow_ptr(1);
    const float* const asrc2 = arow_ptr(2);
    const float* const asrc3 = arow_ptr(3);
    const int wrow = (tid >> 2) & 127, woct = tid & 3;          // piece u: plane = u >> 1, row = wrow + 128 (u & 1)
    const unsigned short* wsrc = Wp + (size_t)(n0 + wrow) * K + 8 * woct;
    const size_t wplane = (size_t)N * K, wrow128 = (size_t)128 * K;
    float4 ra0, ra1, ra2, ra3;
    uint4 rw0, rw1, rw2, rw3;
#define P256_LOAD_TILE(k0)                                                   \
    do {                                                                     \
        ra0 = *(const float4*)(asrc0 + (k0));                                \
        ra1 = *(const float4*)(asrc1 + (k0));                                \
        ra2 = *(const float4*)(asrc2 + (k0));                                \
        ra3 = *(const float4*)(asrc3 + (k0));                                \
        rw0 = *(const uint4*)(wsrc + (k0));                                  \
        rw1 = *(const uint4*)(wsrc + wrow128 + (k0));                        \
        rw2 = *(const uint4*)(wsrc + wplane + (k0));                         \
        rw3 = *(const uint4*)(wsrc + wplane + wrow128 + (k0));               \
    } while (0)
    const int aoff0 = p_off(arow, aq >> 1) + 4 * (aq & 1);      // rows arow + 64 u: (row >> 2) & 3 is the same for all four
    const int woff0 = p_off(wrow, woct);                        // rows wrow, wrow + 128: likewise
#define P256_PARK_A(base, u, r)                                              \
    do {                                                                     \
        uint2 h_, l_;                                                        \
        split4(r, h_, l_);                                                   \
        *(uint2*)((base) + aoff0 + 64 * (u) * P_BK) = h_;                    \
        *(uint2*)((base) + P_PLANE + aoff0 + 64 * (u) * P_BK) = l_;          \
    } while (0)
#define P256_PARK_TILE(buf)                                                  \
    do {                                                                     \
        unsigned short* pb_ = lds + (buf) * 4 * P_PLANE;                     \
        P256_PARK_A(pb_, 0, ra0);                                            \
        P256_PARK_A(pb_, 1, ra1);                                            \
        P256_PARK_A(pb_, 2, ra2);                                            \
        P256_PARK_A(pb_, 3, ra3);                                            \
        *(uint4*)(pb_ + 2 * P_PLANE + woff0) = rw0;                          \
        *(uint4*)(pb_ + 2 * P_PLANE + woff0 + 128 * P_BK) = rw1;             \
        *(uint4*)(pb_ + 3 * P_PLANE + woff0) = rw2;                          \
        *(uint4*)(pb_ + 3 * P_PLANE + woff0 + 128 * P_BK) = rw3;             \
    } while (0)
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16_zero();
    const int nk = K / P_BK;
    P256_LOAD_TILE(0);
    P256_PARK_TILE(0);
    if (nk > 1) P256_LOAD_TILE(P_BK);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned short* base = lds + (kt & 1) * 4 * P_PLANE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            s16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int off = p_off(64 * wc + 32 * j + lo, 2 * ks + hi);
                bh[j] = *(const s16x8*)(base + 2 * P_PLANE + off);
                bl[j] = *(const s16x8*)(base + 3 * P_PLANE + off);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = p_off(128 * wr + 32 * i + lo, 2 * ks + hi);
                ah[i] = *(const s16x8*)(base + off);
                al[i] = *(const s16x8*)(base + P_PLANE + off);
            }
            if (ks == 1 && kt + 1 < nk) {
                // tile kt + 1 (in registers since the previous iteration) -> the other buffer: its last readers passed the barrier
                // at the end of iteration kt - 1; then the loads of tile kt + 2 go out under the second half of this tile's MFMAs
                P256_PARK_TILE((kt + 1) & 1);
                if (kt + 2 < nk) P256_LOAD_TILE((kt + 2) * P_BK);
            }
            sed_mfma_prio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32_bf16(al[i], bh[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32_bf16(ah[i], bl[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32_bf16(ah[i], bh[j], acc[i][j]);
            sed_mfma_prio(0);
        }
        __syncthreads();                    // buffer (kt + 1) & 1 is complete; everybody is done reading buffer kt & 1
    }
    // epilogue: bias (+ exact GELU); lane holds column n0 + 64 wc + 32 j + lo, rows mfma32_row(r, lane) of each 32-row block
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int gn = n0 + 64 * wc + 32 * j + lo;
        const float bv = bias != nullptr ? bias[gn] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + 128 * wr + 32 * i + mfma32_row(r, lane);
                float v = acc[i][j][r] + bv;
                if (ACT == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                if (gm < M) Cm[(size_t)gm * N + gn] = v;
            }
    }
#undef P256_LOAD_TILE
#undef P256_PARK_A
#undef P256_PARK_TILE
}
}  // namespace

// ---- round 6, second form: the same 256 x 256 x 32 tile and LDS layout, hand-phased ("ping-pong") ---------------------------------------
// linear_p256_kernel measured 0.27 - 0.30 of the split-bf16 ceiling, like every kernel of the family (profiles/r06b_linear_shapes.txt):
// __syncthreads() phase-locks the two waves of a SIMD, so both read their fragments at once (matrix pipe idle) and then queue for the
// pipe together.  Here the eight waves are two GROUPS of four (one wave per SIMD each: waves 0-3 own tile rows 0-127, waves 4-7 rows
// 128-255) running ONE BARRIER APART: a K tile is four phases (k16 step ks x row half h), each "LDS reads (+ a share of the staging) |
// barrier | 12 MFMAs | barrier"; while group 0 issues its 12 MFMAs (384 matrix-pipe cycles) group 1 reads the fragments of its next
// phase, parks its share of the next K tile and issues its global loads, and vice versa (cdna_hip_programming.md, the 8-phase
// template; s_setprio 1 around the MFMA cluster).  Barriers are raw s_barrier (no vmcnt drain: the global loads of tile kt + 2 stay in
// flight across twelve of them).  Hazards, in intervals between barriers (eight per K tile; group 1 runs one interval late):
//   * every phase q = 0 .. 3 parks quarter q of tile kt + 1 into the other buffer (and re-loads its registers with tile kt + 2: two global
//     loads per phase -- four in one phase made that phase longer than the 12 MFMAs beside it); phases 0 - 2 are followed by "barrier,
//     s_waitcnt lgkmcnt(0), MFMAs, barrier", phase 3 waits for lgkmcnt(0) BEFORE its barrier, so group 1's last store (interval 7) is
//     published by the barrier that ends interval 7; the first read of tile kt + 1 is group 0's in interval 8;
//   * the other buffer's last readers (tile kt - 1) are group 1's reads of phase 3 in interval -1, retired by that same early wait
//     before the barrier that ends interval -1; the first store is group 0's in interval 0;
//   * the prefetch registers of a quarter are loaded one whole K tile (eight intervals, ~3 000 matrix-pipe cycles) before they are parked.
#ifndef PP_DIAG
#define PP_DIAG 0           // tools/build_variant.py -DPP_DIAG=<bits>: timing-only builds with a part of the loop removed (wrong results)
#endif
namespace {
template <int ACT>
__global__ __launch_bounds__(512, 1) void linear_pp_kernel(const float* __restrict__ A, const unsigned short* __restrict__ Wp,
                                                          const float* __restrict__ bias, float* __restrict__ Cm, int M, int N, int K,
                                                          int tiles_m, int tiles_n) {
    SED_DYN_SMEM(smem);                               // [2 buffers][A hi | A lo | W hi | W lo][256][32] bf16
    unsigned short* lds = (unsigned short*)smem;
    const int tid = threadIdx.x, lane = tid & 63, w = sed_wave_uniform(tid >> 6), lo = lane & 31, hi = lane >> 5;
    const int wr = w >> 2, wc = w & 3;                // wave: rows 128 wr .. + 127, columns 64 wc .. + 63 of the tile; group = wr
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tm = xcd + 8 * (slot / tiles_n), tn = slot % tiles_n;
    if (tm >= tiles_m) return;
    const int m0 = tm * P_BM, n0 = tn * P_BN;
    const int arow = tid >> 3, aq = tid & 7;
    auto arow_ptr = [&](int u) {
        int r = m0 + arow + 64 * u;
        r = r < M ? r : M - 1;
        return A + (size_t)r * ((PP_DIAG & 128) ? K + 64 : K) + 4 * aq;       // diag 128: padded row pitch (timing only)
    };
    const float* const asrc0 = arow_ptr(0);
    const float* const asrc1 = arow_ptr(1);
    const float* const asrc2 = arow_ptr(2);
    const float* const asrc3 = arow_ptr(3);
    const int wrow = (tid >> 2) & 127, woct = tid & 3;
    const int ldw = (PP_DIAG & 256) ? K + 128 : K;                           // diag 256: padded row pitch of the weight planes
    const unsigned short* wsrc = Wp + (size_t)(n0 + wrow) * ldw + 8 * woct;
    const size_t wplane = (size_t)N * ldw, wrow128 = (size_t)128 * ldw;
    float4 ra0, ra1, ra2, ra3;                        // half 0 of a K tile: ra0, ra1 (rows arow, + 64), rw0, rw1 (hi plane, rows wrow, + 128)
    uint4 rw0, rw1, rw2, rw3;                         // half 1: ra2, ra3 (rows + 128, + 192), rw2, rw3 (lo plane)
    // quarter q of a K tile = A rows arow + 64 q (ra_q) + W piece q (rw_q: plane q >> 1, rows wrow + 128 (q & 1))
#define PP_LOAD_Q(ra, rw, asrc, woff, k0) do { if (!(PP_DIAG & 32)) ra = *(const float4*)((asrc) + (k0));                      \
                                              if (!(PP_DIAG & 64)) rw = *(const uint4*)(wsrc + (woff) + (k0)); } while (0)
#define PP_LOAD_Q0(k0) PP_LOAD_Q(ra0, rw0, asrc0, (size_t)0, k0)
#define PP_LOAD_Q1(k0) PP_LOAD_Q(ra1, rw1, asrc1, wrow128, k0)
#define PP_LOAD_Q2(k0) PP_LOAD_Q(ra2, rw2, asrc2, wplane, k0)
#define PP_LOAD_Q3(k0) PP_LOAD_Q(ra3, rw3, asrc3, wplane + wrow128, k0)
    const int aoff0 = p_off(arow, aq >> 1) + 4 * (aq & 1);
    const int woff0 = p_off(wrow, woct);
#define PP_PARK_Q(pb_, q, ra, rw) do { uint2 h_, l_; split4(ra, h_, l_);                                                        \
                                       *(uint2*)((pb_) + aoff0 + 64 * (q) * P_BK) = h_;                                         \
                                       *(uint2*)((pb_) + P_PLANE + aoff0 + 64 * (q) * P_BK) = l_;                               \
                                       *(uint4*)((pb_) + (2 + ((q) >> 1)) * P_PLANE + woff0 + 128 * ((q) & 1) * P_BK) = rw; } while (0)
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16_zero();
    const int nk = K / P_BK, klast = K - P_BK;
    PP_LOAD_Q0(0); PP_LOAD_Q1(0); PP_LOAD_Q2(0); PP_LOAD_Q3(0);
    PP_PARK_Q(lds, 0, ra0, rw0); PP_PARK_Q(lds, 1, ra1, rw1); PP_PARK_Q(lds, 2, ra2, rw2); PP_PARK_Q(lds, 3, ra3, rw3);
    { const int k1 = nk > 1 ? P_BK : 0; PP_LOAD_Q0(k1); PP_LOAD_Q1(k1); PP_LOAD_Q2(k1); PP_LOAD_Q3(k1); }
    __syncthreads();
    if (wr == 1) sed_phase_barrier();               // group 1 runs one interval behind group 0
    // fragment addresses: row 128 wr + 32 i + lo (A) / 64 wc + 32 j + lo (W), octet 2 ks + hi, swizzled by (row >> 2) & 3 = (lo >> 2) & 3
    const int sw = (lo >> 2) & 3;
    const int fa0 = (128 * wr + lo) * P_BK + ((hi ^ sw) << 3), fa1 = (128 * wr + lo) * P_BK + (((2 + hi) ^ sw) << 3);
    const int fb0 = 2 * P_PLANE + (64 * wc + lo) * P_BK + ((hi ^ sw) << 3), fb1 = 2 * P_PLANE + (64 * wc + lo) * P_BK + (((2 + hi) ^ sw) << 3);
    s16x8 ah0, ah1, al0, al1, bh0, bh1, bl0, bl1;
#if defined(PP_DIAG) && (PP_DIAG & 8)
#define PP_BAR() sed_sched_fence()
#else
#define PP_BAR() sed_phase_barrier()
#endif
#define PP_READ_B(base, fb) do { bh0 = *(const s16x8*)((base) + (fb)); bh1 = *(const s16x8*)((base) + (fb) + 32 * P_BK);       \
                                 bl0 = *(const s16x8*)((base) + P_PLANE + (fb)); bl1 = *(const s16x8*)((base) + P_PLANE + (fb) + 32 * P_BK); } while (0)
#define PP_READ_A(base, fa, h) do { ah0 = *(const s16x8*)((base) + (fa) + (2 * (h)) * 32 * P_BK); ah1 = *(const s16x8*)((base) + (fa) + (2 * (h) + 1) * 32 * P_BK); \
                                    al0 = *(const s16x8*)((base) + P_PLANE + (fa) + (2 * (h)) * 32 * P_BK);                    \
                                    al1 = *(const s16x8*)((base) + P_PLANE + (fa) + (2 * (h) + 1) * 32 * P_BK); } while (0)
#define PP_MFMA(h, pre) do { if (pre) sed_wait_lds(); PP_BAR(); sed_wait_lds(); sed_mfma_prio(1); if (!(PP_DIAG & 1024)) {                                                             \
        acc[2 * (h)][0] = mfma32_bf16(al0, bh0, acc[2 * (h)][0]); acc[2 * (h)][1] = mfma32_bf16(al0, bh1, acc[2 * (h)][1]);     \
        acc[2 * (h) + 1][0] = mfma32_bf16(al1, bh0, acc[2 * (h) + 1][0]); acc[2 * (h) + 1][1] = mfma32_bf16(al1, bh1, acc[2 * (h) + 1][1]); \
        acc[2 * (h)][0] = mfma32_bf16(ah0, bl0, acc[2 * (h)][0]); acc[2 * (h)][1] = mfma32_bf16(ah0, bl1, acc[2 * (h)][1]);     \
        acc[2 * (h) + 1][0] = mfma32_bf16(ah1, bl0, acc[2 * (h) + 1][0]); acc[2 * (h) + 1][1] = mfma32_bf16(ah1, bl1, acc[2 * (h) + 1][1]); \
        acc[2 * (h)][0] = mfma32_bf16(ah0, bh0, acc[2 * (h)][0]); acc[2 * (h)][1] = mfma32_bf16(ah0, bh1, acc[2 * (h)][1]);     \
        acc[2 * (h) + 1][0] = mfma32_bf16(ah1, bh0, acc[2 * (h) + 1][0]); acc[2 * (h) + 1][1] = mfma32_bf16(ah1, bh1, acc[2 * (h) + 1][1]); \
        } sed_mfma_prio(0); PP_BAR(); } while (0)
    // one loop body for both buffers (a runtime buffer offset costs a handful of address adds per K tile; two specialised copies behind
    // a branch made the register allocator copy and spill the accumulators)
#define PP_IF(bit, stmt) do { if (!(PP_DIAG & (bit))) { stmt; } } while (0)
    if (PP_DIAG & 4) { PP_READ_B(lds, fb0); PP_READ_A(lds, fa0, 0); }
#if PP_DIAG & 2048
    // diag: TWO K tiles of prefetch registers in flight (only possible without the accumulators: use with bit 1024)
    float4 sa0 = ra0, sa1 = ra1, sa2 = ra2, sa3 = ra3; uint4 sw0 = rw0, sw1 = rw1, sw2 = rw2, sw3 = rw3;
    for (int kt = 0; kt < nk; ++kt) {
        unsigned short* other = lds + ((kt & 1) ^ 1) * 4 * P_PLANE;
        const int k3 = min((kt + 3) * P_BK, klast);
        if (kt & 1) {
            PP_PARK_Q(other, 0, ra0, rw0); PP_LOAD_Q0(k3); PP_MFMA(0, 0); PP_PARK_Q(other, 1, ra1, rw1); PP_LOAD_Q1(k3); PP_MFMA(1, 0);
            PP_PARK_Q(other, 2, ra2, rw2); PP_LOAD_Q2(k3); PP_MFMA(0, 0); PP_PARK_Q(other, 3, ra3, rw3); PP_LOAD_Q3(k3); PP_MFMA(1, 1);
        } else {
            PP_PARK_Q(other, 0, sa0, sw0); PP_LOAD_Q(sa0, sw0, asrc0, (size_t)0, k3); PP_MFMA(0, 0);
            PP_PARK_Q(other, 1, sa1, sw1); PP_LOAD_Q(sa1, sw1, asrc1, wrow128, k3); PP_MFMA(1, 0);
            PP_PARK_Q(other, 2, sa2, sw2); PP_LOAD_Q(sa2, sw2, asrc2, wplane, k3); PP_MFMA(0, 0);
            PP_PARK_Q(other, 3, sa3, sw3); PP_LOAD_Q(sa3, sw3, asrc3, wplane + wrow128, k3); PP_MFMA(1, 1);
        }
    }
#else
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned short* base = lds + (kt & 1) * 4 * P_PLANE;
        unsigned short* other = lds + ((kt & 1) ^ 1) * 4 * P_PLANE;
        const int k2 = (PP_DIAG & 16) ? (kt & 1) * P_BK : min((kt + 2) * P_BK, klast);      // past the end: a valid tile nobody uses
        // phase q parks quarter q of tile kt + 1 (loaded one K tile ago) and re-loads its registers with quarter q of tile kt + 2:
        // two global loads, one split and three LDS stores per phase beside the 4 - 8 fragment reads
        PP_IF(4, PP_READ_B(base, fb0)); PP_IF(4, PP_READ_A(base, fa0, 0)); PP_IF(2, PP_PARK_Q(other, 0, ra0, rw0)); PP_IF(1, PP_LOAD_Q0(k2)); PP_MFMA(0, 0);
        PP_IF(4, PP_READ_A(base, fa0, 1));                                 PP_IF(2, PP_PARK_Q(other, 1, ra1, rw1)); PP_IF(1, PP_LOAD_Q1(k2)); PP_MFMA(1, 0);
        PP_IF(4, PP_READ_B(base, fb1)); PP_IF(4, PP_READ_A(base, fa1, 0)); PP_IF(2, PP_PARK_Q(other, 2, ra2, rw2)); PP_IF(1, PP_LOAD_Q2(k2)); PP_MFMA(0, 0);
        PP_IF(4, PP_READ_A(base, fa1, 1));                                 PP_IF(2, PP_PARK_Q(other, 3, ra3, rw3)); PP_IF(1, PP_LOAD_Q3(k2)); PP_MFMA(1, 1);
    }
#endif
#undef PP_IF
    if (wr == 0) sed_phase_barrier();               // the barrier group 1 took at the top
    // epilogue: bias (+ exact GELU); lane holds column n0 + 64 wc + 32 j + lo, rows mfma32_row(r, lane) of each 32-row block
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int gn = n0 + 64 * wc + 32 * j + lo;
        const float bv = bias != nullptr ? bias[gn] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + 128 * wr + 32 * i + mfma32_row(r, lane);
                float v = acc[i][j][r] + bv;
                if (ACT == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                if ((PP_DIAG & 4096) ? (v == 1.2345e-30f) : (gm < M)) Cm[(size_t)gm * N + gn] = v;      // diag 4096: no C stores
            }
    }
#undef PP_LOAD_Q
#undef PP_LOAD_Q0
#undef PP_LOAD_Q1
#undef PP_LOAD_Q2
#undef PP_LOAD_Q3
#undef PP_PARK_Q
#undef PP_READ_A
#undef PP_READ_B
#undef PP_MFMA
#undef PP_BAR
}
}  // namespace

// ---- round 6, third form: both operands as pre-split planes in a K-tiled LDS image, every byte moved by LDS-DMA -------------------------
// What the two kernels above taught (profiles/r06_linear_diag.md): with the loads removed the hand-phased loop runs at the matrix pipe's
// pace (1.57 us per 32-deep K tile); the loads alone -- global_load_dwordx4 into VGPRs, 64 KB per K tile and CU -- take 0.7 (L2-hot) to
// 1.4 us per K tile; and together they ADD (2.67 us): load data returning into the VGPR file and the MFMAs do not overlap, whatever the
// schedule.  So here nothing returns into a register: activations and weights arrive as bf16 hi / lo planes already cut into the blocks
// a workgroup needs (sed_split_tiles_bf16x3: block (row panel, 16-deep K tile) = [hi | lo][256][16], 16 KB, swizzled for the fragment
// reads), and a K tile's two blocks are copied into one of FOUR 32 KB LDS stages by eight wave-instructions of LDS-DMA per wave pair
// -- contiguous 1 KB pieces, full cache lines, no staging registers, no ds_write, no VALU.  Same 256 x 256 tile, eight waves as two
// groups one barrier apart, 12 MFMAs per phase; a K tile is two phases.  DMA for tile kt + 3 is issued during tile kt (two pieces per
// wave and phase); before the barrier that ends its second read phase every wave waits until at most the eight youngest of its DMAs
// are outstanding (tiles kt + 2, kt + 3), i.e. its share of tile kt + 1 has landed; the first read of tile kt + 1 is one barrier later.
// The stage refilled during tile kt is tile kt - 1's, whose last reads (group 1, second phase) were retired by the lgkmcnt(0) of that
// same wait.  LDS: 4 x 32 KB.
namespace {
constexpr int T_BK = 16, T_PLANE = 256 * T_BK, T_BLOCK = 2 * T_PLANE, T_STAGE = 2 * T_BLOCK;      // ushorts: 4096, 8192 (16 KB), 16384 (32 KB)
__device__ __forceinline__ int t_off(int row, int oct) { return row * T_BK + ((oct ^ ((row >> 3) & 1)) << 3); }

// X (R, K) fp32 -> tile image: one workgroup per block; thread t copies the octets (row t / 2 + 128 u, slot t & 1), u = 0, 1
__global__ __launch_bounds__(256) void split_tiles_kernel(const float* __restrict__ X, unsigned short* __restrict__ Xt, int R, int K, int nkt) {
    const int blk = blockIdx.x, panel = blk / nkt, kt = blk - panel * nkt, tid = threadIdx.x;
    unsigned short* dst = Xt + (size_t)blk * T_BLOCK;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = (tid >> 1) + 128 * u, slot = tid & 1, oct = slot ^ ((row >> 3) & 1), r = panel * 256 + row;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f), q = p;
        if (r < R) {
            const float* src = X + (size_t)r * K + kt * T_BK + 8 * oct;
            p = *(const float4*)src;
            q = *(const float4*)(src + 4);
        }
        s16x8 h, l;
        split8(p, q, h, l);
        *(s16x8*)(dst + row * T_BK + 8 * slot) = h;
        *(s16x8*)(dst + T_PLANE + row * T_BK + 8 * slot) = l;
    }
}

// T_STAMP (diagnostics build only: ONLY=sed_gemm_bf16.hip python tools/build_variant.py tstamp -DT_STAMP; tools/linear_stamps.py): s_memtime
// at the seams of a phase -- start, DMA issued + fragments landed, first barrier passed, MFMAs issued, second barrier passed (second phase:
// + after the vmcnt wait) -- for wave 0 (group 0) and wave 4 (group 1) of workgroup 64, K tiles 8 .. 23, kept in the 32 KB of LDS the
// stages leave free and dumped after the loop.  Each stamp waits for lgkmcnt(0) (s_memtime is an SMEM read).
#ifdef T_STAMP
__device__ unsigned long long* t_stamp_buf;
SED_API int sed_linear_debug_set_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(t_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? SED_OK : SED_ERR_LAUNCH;
}
#define T_TS(k) do { __builtin_amdgcn_sched_barrier(0);                                                                          \
                     if (stamp_on && kt >= 8 && kt < 24) { const unsigned long long t_ = __builtin_amdgcn_s_memtime();           \
                                                           if (lane == 0) s_ts[((w >> 2) * 16 + kt - 8) * 16 + (k)] = t_; }      \
                     __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define T_TS(k) do { } while (0)
#endif

template <int ACT>
__global__ __launch_bounds__(512, 1) void linear_dma_kernel(const unsigned short* __restrict__ At, const unsigned short* __restrict__ Wt,
                                                           const float* __restrict__ bias, float* __restrict__ Cm, int M, int N, int K,
                                                           int tiles_m, int tiles_n) {
    SED_DYN_SMEM(smem);                               // [4 stages][A hi | A lo | W hi | W lo][256][16] bf16
    unsigned short* lds = (unsigned short*)smem;
    const int tid = threadIdx.x, lane = tid & 63, w = sed_wave_uniform(tid >> 6), lo = lane & 31, hi = lane >> 5;
    const int wr = w >> 2, wc = w & 3;                // wave: rows 128 wr .. + 127, columns 64 wc .. + 63 of the tile; group = wr
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tm = xcd + 8 * (slot / tiles_n), tn = slot % tiles_n;
    if (tm >= tiles_m) return;
    const int m0 = tm * P_BM, n0 = tn * P_BN, nk = K / T_BK;
    // DMA pieces of a stage: 32 pieces of 1 KB (0 - 15: the A block, 16 - 31: the W block); wave w moves pieces w, w + 8 (first phase) and
    // w + 16, w + 24 (second phase)
    const unsigned short* asrc = At + (size_t)tm * nk * T_BLOCK + w * 512 + lane * 8;
    const unsigned short* wsrc = Wt + (size_t)tn * nk * T_BLOCK + w * 512 + lane * 8;
#define T_DMA_A(kt_, stage_) do { const unsigned short* g_ = asrc + (size_t)(kt_) * T_BLOCK; unsigned short* l_ = lds + (stage_) * T_STAGE + w * 512; \
                                  sed_dma16(g_, l_); sed_dma16(g_ + 8 * 512, l_ + 8 * 512); } while (0)
#define T_DMA_W(kt_, stage_) do { const unsigned short* g_ = wsrc + (size_t)(kt_) * T_BLOCK; unsigned short* l_ = lds + (stage_) * T_STAGE + T_BLOCK + w * 512; \
                                  sed_dma16(g_, l_); sed_dma16(g_ + 8 * 512, l_ + 8 * 512); } while (0)
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16_zero();
    // prologue: tiles 0, 1, 2 (past the end: the last tile again -- a valid block nobody reads -- so that the DMA count per tile is fixed)
    {
        const int k1 = min(1, nk - 1), k2 = min(2, nk - 1);
        T_DMA_A(0, 0); T_DMA_W(0, 0); T_DMA_A(k1, 1); T_DMA_W(k1, 1); T_DMA_A(k2, 2); T_DMA_W(k2, 2);
    }
    SED_WAIT_VM_LDS(8);                               // tile 0 has landed (this wave's share)
    __syncthreads();
    if (wr == 1) sed_phase_barrier();                 // group 1 runs one interval behind group 0
    const int sw = (lo >> 3) & 1;
    const int fa = (128 * wr + lo) * T_BK + ((hi ^ sw) << 3), fb = T_BLOCK + (64 * wc + lo) * T_BK + ((hi ^ sw) << 3);
    s16x8 ah0, ah1, al0, al1, bh0, bh1, bl0, bl1;
#define T_READ_B(base) do { bh0 = *(const s16x8*)((base) + fb); bh1 = *(const s16x8*)((base) + fb + 32 * T_BK);                 \
                            bl0 = *(const s16x8*)((base) + T_PLANE + fb); bl1 = *(const s16x8*)((base) + T_PLANE + fb + 32 * T_BK); } while (0)
#define T_READ_A(base, h) do { ah0 = *(const s16x8*)((base) + fa + (2 * (h)) * 32 * T_BK); ah1 = *(const s16x8*)((base) + fa + (2 * (h) + 1) * 32 * T_BK); \
                               al0 = *(const s16x8*)((base) + T_PLANE + fa + (2 * (h)) * 32 * T_BK);                            \
                               al1 = *(const s16x8*)((base) + T_PLANE + fa + (2 * (h) + 1) * 32 * T_BK); } while (0)
#define T_MFMA(h) do { sed_mfma_prio(1);                                                                                        \
        acc[2 * (h)][0] = mfma32_bf16(al0, bh0, acc[2 * (h)][0]); acc[2 * (h)][1] = mfma32_bf16(al0, bh1, acc[2 * (h)][1]);     \
        acc[2 * (h) + 1][0] = mfma32_bf16(al1, bh0, acc[2 * (h) + 1][0]); acc[2 * (h) + 1][1] = mfma32_bf16(al1, bh1, acc[2 * (h) + 1][1]); \
        acc[2 * (h)][0] = mfma32_bf16(ah0, bl0, acc[2 * (h)][0]); acc[2 * (h)][1] = mfma32_bf16(ah0, bl1, acc[2 * (h)][1]);     \
        acc[2 * (h) + 1][0] = mfma32_bf16(ah1, bl0, acc[2 * (h) + 1][0]); acc[2 * (h) + 1][1] = mfma32_bf16(ah1, bl1, acc[2 * (h) + 1][1]); \
        acc[2 * (h)][0] = mfma32_bf16(ah0, bh0, acc[2 * (h)][0]); acc[2 * (h)][1] = mfma32_bf16(ah0, bh1, acc[2 * (h)][1]);     \
        acc[2 * (h) + 1][0] = mfma32_bf16(ah1, bh0, acc[2 * (h) + 1][0]); acc[2 * (h) + 1][1] = mfma32_bf16(ah1, bh1, acc[2 * (h) + 1][1]); \
        sed_mfma_prio(0); } while (0)
#ifdef T_STAMP
    unsigned long long* s_ts = (unsigned long long*)(lds + 4 * T_STAGE);
    const bool stamp_on = blockIdx.x == 64 && (w & 3) == 0;
#endif
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned short* base = lds + (kt & 3) * T_STAGE;
        const int k3 = min(kt + 3, nk - 1), s3 = (kt + 3) & 3;
        // first phase: A pieces of tile kt + 3, W fragments + A fragments of the upper 64 rows
        T_TS(0);
        T_READ_B(base); T_READ_A(base, 0);          // fragment reads FIRST: the four waves of a group issue their DMA pieces at the same moment and
        sed_sched_fence();                          // queue for the CU's one address pipe (~25 - 60 cycles a piece, r06g_linear_stamps)
        T_DMA_A(k3, s3);
        T_TS(1);
        sed_phase_barrier(); sed_wait_lds();
        T_TS(2);
        T_MFMA(0);
        T_TS(3);
        sed_phase_barrier();
        T_TS(4);
        // second phase: W pieces of tile kt + 3, A fragments of the lower 64 rows; then this wave's share of tile kt + 1 must have landed
        T_READ_A(base, 1);
        sed_sched_fence();
        T_DMA_W(k3, s3);
        T_TS(5);
        SED_WAIT_VM_LDS(8);
        T_TS(6);
        sed_phase_barrier();
        T_TS(7);
        T_MFMA(1);
        T_TS(8);
        sed_phase_barrier();
        T_TS(9);
    }
#ifdef T_STAMP
    __syncthreads();
    if (stamp_on && lane < 16 && t_stamp_buf != nullptr)
        for (int q = 0; q < 16; ++q) t_stamp_buf[((w >> 2) * 16 + q) * 16 + lane] = s_ts[((w >> 2) * 16 + q) * 16 + lane];
#endif
    if (wr == 0) sed_phase_barrier();                 // the barrier group 1 took at the top
    SED_WAIT_VM_LDS(0);                               // the clamped tail DMAs: nothing may be in flight when the workgroup's LDS is released
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int gn = n0 + 64 * wc + 32 * j + lo;
        const float bv = bias != nullptr ? bias[gn] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + 128 * wr + 32 * i + mfma32_row(r, lane);
                float v = acc[i][j][r] + bv;
                if (ACT == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                if (gm < M) Cm[(size_t)gm * N + gn] = v;
            }
    }
#undef T_DMA_A
#undef T_DMA_W
#undef T_READ_A
#undef T_READ_B
#undef T_MFMA
}
}  // namespace

SED_API int sed_split_tiles_bf16x3(const float* X, unsigned short* Xt, int R, int K, void* stream) {
    if (!X || !Xt || R < 0 || K < 0) return SED_ERR_ARG;
    if (R == 0 || K == 0) return SED_OK;
    if (K % T_BK != 0 || ((uintptr_t)X & 15) || ((uintptr_t)Xt & 15)) return SED_ERR_UNSUPPORTED;
    const long long blocks = (long long)((R + 255) / 256) * (K / T_BK);
    if (blocks > 0x7fffffffLL) return SED_ERR_UNSUPPORTED;
    SED_LAUNCH(split_tiles_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X, Xt, R, K, K / T_BK);
    return sed_check_launch();
}

SED_API int sed_linear_tiles_bf16x3(const unsigned short* At, const unsigned short* Wt, const float* bias, float* Cm, int M, int N, int K,
                                    int act, void* stream) {
    if (!At || !Wt || !Cm || M < 0 || N < 0 || K < 0) return SED_ERR_ARG;
    if (act < 0 || act > 1) return SED_ERR_ARG;
    if (M <= 0 || N <= 0) return SED_OK;
    if (N % P_BN != 0 || K % T_BK != 0 || K <= 0 || ((uintptr_t)At & 15) || ((uintptr_t)Wt & 15)) return SED_ERR_UNSUPPORTED;
    const int tm = (M + P_BM - 1) / P_BM, tn = N / P_BN;
    const long long grid_ll = 8LL * ((tm + 7) / 8) * tn;
    if (grid_ll > 0x7fffffffLL) return SED_ERR_UNSUPPORTED;
#ifdef T_STAMP
    constexpr int SMEM_T = 4 * T_STAGE * 2 + 4096;
#else
    constexpr int SMEM_T = 4 * T_STAGE * 2;
#endif
    if (act) { SED_MAX_SMEM((linear_dma_kernel<1>), SMEM_T); SED_LAUNCH((linear_dma_kernel<1>), dim3((unsigned)grid_ll), dim3(512), SMEM_T, (hipStream_t)stream, At, Wt, bias, Cm, M, N, K, tm, tn); }
    else { SED_MAX_SMEM((linear_dma_kernel<0>), SMEM_T); SED_LAUNCH((linear_dma_kernel<0>), dim3((unsigned)grid_ll), dim3(512), SMEM_T, (hipStream_t)stream, At, Wt, bias, Cm, M, N, K, tm, tn); }
    return sed_check_launch();
}

namespace {
__global__ __launch_bounds__(256) void pack_bf16x3_kernel(const float* __restrict__ W, unsigned short* __restrict__ Wp, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned short h, l;
    bf16_split(W[i], h, l);
    Wp[i] = h;
    Wp[n + i] = l;
}
}  // namespace

// Frozen weights W[N][K] -> Wp[2][N][K] bf16 bit patterns: plane 0 = bf16(w), plane 1 = bf16(w - plane 0)  (done once per weight).
SED_API int sed_pack_weights_bf16x3(const float* W, unsigned short* Wp, int N, int K, void* stream) {
    const long long n = (long long)N * K;
    if (n <= 0) return SED_OK;
    SED_LAUNCH(pack_bf16x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, Wp, n);
    return sed_check_launch();
}

// torch.nn.Linear forward on packed weights: C[M][N] = act(A[M][K] . W[N][K]^T + bias[N]) -- every Linear of the BEATs encoder
// (recipes/dcase2023_task4_baseline/local/beats/backbone.py:286-330 q/k/v/out projections, :214-276 fc1 (GELU) / fc2).
// N % 128 == 0, K % 32 == 0, 16-byte aligned A and Wp.
SED_API int sed_linear_packed_bf16x3(const float* A, const unsigned short* Wp, const float* bias, float* Cm, int M, int N, int K, int act,
                                     void* stream) {
    if (act < 0 || act > 1) return SED_ERR_ARG;
    if (M <= 0 || N <= 0) return SED_OK;
    if (N % LB_BN != 0 || K % LB_BK != 0 || K <= 0 || ((uintptr_t)A & 15) || ((uintptr_t)Wp & 15)) return SED_ERR_UNSUPPORTED;
    if (N % P_BN == 0 && sed_tuning[SED_TUNE_LINEAR_P256] != 1 && (sed_tuning[SED_TUNE_LINEAR_P256] >= 2 || N >= 2048)) {
        // the 256 x 256 kernel (round 6): one workgroup per tile, 128 KB of LDS
        const int tm = (M + P_BM - 1) / P_BM, tn = N / P_BN;
        const long long grid_ll = 8LL * ((tm + 7) / 8) * tn;
        if (grid_ll > 0x7fffffffLL) return SED_ERR_UNSUPPORTED;
        constexpr int SMEM_P = 2 * 4 * P_PLANE * 2;
        if (sed_tuning[SED_TUNE_LINEAR_P256] != 4) {       // the hand-phased form (4: the __syncthreads() form, kept for A/B runs)
            if (act) { SED_MAX_SMEM((linear_pp_kernel<1>), SMEM_P); SED_LAUNCH((linear_pp_kernel<1>), dim3((unsigned)grid_ll), dim3(512), SMEM_P, (hipStream_t)stream, A, Wp, bias, Cm, M, N, K, tm, tn); }
            else { SED_MAX_SMEM((linear_pp_kernel<0>), SMEM_P); SED_LAUNCH((linear_pp_kernel<0>), dim3((unsigned)grid_ll), dim3(512), SMEM_P, (hipStream_t)stream, A, Wp, bias, Cm, M, N, K, tm, tn); }
            return sed_check_launch();
        }
        if (act) { SED_MAX_SMEM((linear_p256_kernel<1>), SMEM_P); SED_LAUNCH((linear_p256_kernel<1>), dim3((unsigned)grid_ll), dim3(512), SMEM_P, (hipStream_t)stream, A, Wp, bias, Cm, M, N, K, tm, tn); }
        else { SED_MAX_SMEM((linear_p256_kernel<0>), SMEM_P); SED_LAUNCH((linear_p256_kernel<0>), dim3((unsigned)grid_ll), dim3(512), SMEM_P, (hipStream_t)stream, A, Wp, bias, Cm, M, N, K, tm, tn); }
        return sed_check_launch();
    }
    const int tiles_m = (M + LB_BM - 1) / LB_BM, tiles_n = N / LB_BN;
    long long tiles = (long long)tiles_m * tiles_n;
    int grid = tiles < 512 ? (int)tiles : 512;          // two resident workgroups per CU
    grid = (grid + 7) & ~7;
    if (act) SED_LAUNCH((linear_big_kernel<1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, A, Wp, bias, Cm, M, N, K, tiles_m, tiles_n);
    else SED_LAUNCH((linear_big_kernel<0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, A, Wp, bias, Cm, M, N, K, tiles_m, tiles_n);
    return sed_check_launch();
}

// torch.nn.Linear forward with an optional fused activation: C[M][N] = act(A[M][K] . W[N][K]^T + bias[N]), act 0 = none, 1 = exact
// GELU (the FFN of the BEATs encoder layers).  16-byte aligned operands, K % 4 == 0.
SED_API int sed_linear_bf16x3(const float* A, const float* W, const float* bias, float* Cm, int M, int N, int K, int act,
                                 void* stream) {
    if (act < 0 || act > 1) return SED_ERR_ARG;
    return gemmb_dispatch(A, W, bias, Cm, nullptr, nullptr, nullptr, nullptr, 1, M, N, K, K, K, N, 0, 1, 1, 0, (hipStream_t)stream,
                          nullptr, 0, act, nullptr, true);
}
