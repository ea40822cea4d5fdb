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
