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

// ---- round 6: both operands as pre-split planes in a K-tiled LDS image, every byte moved by LDS-DMA --------------------------------------
// Two 256 x 256 kernels with register staging came first (a __syncthreads() form and a hand-phased two-group form; commit 43bb868,
// timing-only builds and stamps in profiles/r06_linear_diag.md).  What they taught: with the loads removed the hand-phased loop runs at the matrix pipe's
// pace (1.57 us per 32-deep K tile); the loads alone -- global_load_dwordx4 into VGPRs, 64 KB per K tile and CU -- take 0.7 (L2-hot) to
// 1.4 us per K tile; and together they ADD (2.67 us), whatever the schedule.  (The explanation found last: the kernel runs at the board's
// 1 400 W power cap with the clock throttled to 2.0 GHz -- profiles/r06_linear_diag.md section 5 -- so work that overlaps in time is paid back as
// clock.)  What this kernel removes is work: nothing returns into a register -- activations and weights arrive as bf16 hi / lo planes already cut into the blocks
// a workgroup needs (sed_split_tiles_bf16x3: block (row panel, 16-deep K tile) = [hi | lo][256][16], 16 KB, swizzled for the fragment
// reads), and a K tile's two blocks are copied into one of FOUR 32 KB LDS stages by eight wave-instructions of LDS-DMA per wave pair
// -- contiguous 1 KB pieces, full cache lines, no staging registers, no ds_write, no VALU.  Same 256 x 256 tile, eight waves as two
// groups one barrier apart, 12 MFMAs per phase; a K tile is two phases.  DMA for tile kt + 3 is issued during tile kt (two pieces per
// wave and phase); before the barrier that ends its second read phase every wave waits until at most the eight youngest of its DMAs
// are outstanding (tiles kt + 2, kt + 3), i.e. its share of tile kt + 1 has landed; the first read of tile kt + 1 is one barrier later.
// The stage refilled during tile kt is tile kt - 1's, whose last reads (group 1, second phase) were retired by the lgkmcnt(0) of that
// same wait.  LDS: 4 x 32 KB.
namespace {
constexpr int P_BM = 256, P_BN = 256;                    // output tile
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
                     if (stamp_on && step >= 8 && step < 24) { const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
                                                           if (lane == 0) s_ts[((w >> 2) * 16 + step - 8) * 16 + (k)] = t_; }      \
                     __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define T_TS(k) do { } while (0)
#endif

// Persistent form: one workgroup per CU walks its output tiles (virtual block g, g + gridDim, ...; the XCD of a tile's row panel is
// g & 7 as before) as ONE stream of K steps -- the DMA cursor runs three steps ahead of the MFMAs across tile boundaries, so a tile's first
// blocks are in flight while the previous tile's accumulators are written.  The product is accumulated transposed (MFMA a = W fragment, b
// = A fragment: a lane then holds one row m and 4 consecutive columns n per register quad): C leaves as 16-byte stores, or (OUT = 1) as
// 8-byte hi / lo pieces straight into the K-tiled image the NEXT Linear reads (fc1's GELU output -> fc2), rows of the padded last panel
// included.
template <int ACT, int OUT>
__global__ __launch_bounds__(512, 1) void linear_dma_kernel(const unsigned short* __restrict__ At, const unsigned short* __restrict__ Wt,
                                                           const float* __restrict__ bias, void* __restrict__ Cout, int M, int N, int K,
                                                           int tiles_m, int tiles_n, int nvb, int skew, int ksplit) {
    SED_DYN_SMEM(smem);                               // [4 stages][A hi | A lo | W hi | W lo][256][16] bf16
    unsigned short* lds = (unsigned short*)smem;
    const int tid = threadIdx.x, lane = tid & 63, w = sed_wave_uniform(tid >> 6), lo = lane & 31, hi = lane >> 5;
    const int wr = w >> 2, wc = w & 3;                // wave: rows 128 wr .. + 127, columns 64 wc .. + 63 of the tile; group = wr
    const int gstride = gridDim.x;
    // virtual block -> tile: XCD x = vb & 7 owns the row panels tm = x (mod 8) and sweeps their N tiles back to back.  With ksplit = 2 a
    // work item is (tile, K half): items [0, nvb) are the first halves, [nvb, 2 nvb) the second (nvb is a multiple of 8: same XCD), each
    // writing its own partial C -- 279 tiles on 256 CUs are two rounds, 558 half-items three half-rounds (the N = 768 layers).
    const int nk = K / T_BK / ksplit, nwi = nvb * ksplit;          // K steps per work item
    auto valid = [&](int wi) { const int vb = wi >= nvb ? wi - nvb : wi; return (vb & 7) + 8 * ((vb >> 3) / tiles_n) < tiles_m; };
    auto next_vb = [&](int wi) { do wi += gstride; while (wi < nwi && !valid(wi)); return wi; };
    int cvb = (int)blockIdx.x;                        // compute cursor
    if (!valid(cvb)) cvb = next_vb(cvb);
    if (cvb >= nwi) return;
#ifndef SED_EMU
    if (skew > 0) {
        // Every tile takes the same time, so the CUs would all reach their epilogues together and write 64 MB of C in one burst (17 us at
        // HBM's write rate, matrix pipes idle: a quarter of the QKV launch).  The workgroups that walk one tile FEWER than the longest
        // walk have a tile's time to spare: they start late, by a fraction of a tile's time that depends on their row panel (the
        // workgroups sharing an A panel stay together, for its L2), and their epilogues then fall into the others' K loops.
        int mine = 0, vb0 = (int)blockIdx.x;
        for (int vb = valid(vb0) ? vb0 : next_vb(vb0); vb < nwi; vb = next_vb(vb)) ++mine;
        if (mine < skew) {
            const int grp = (int)(blockIdx.x & 7) + 8 * (int)((blockIdx.x >> 3) / tiles_n);
            const long long wait = (long long)nk * 2800 * (((grp * 5) & 31) + 1) / 36;
            const long long t0 = (long long)__builtin_amdgcn_s_memtime();
            while ((long long)__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(64);
        }
    }
#endif
    // DMA pieces of a stage: 32 pieces of 1 KB (0 - 15: the A block, 16 - 31: the W block); wave w moves pieces w, w + 8 (first phase) and
    // w + 16, w + 24 (second phase).  Uniform block pointer + one 32-bit lane offset: no vector arithmetic in the read phases.
    const unsigned voff = (unsigned)(w * 512 + lane * 8), voff2 = voff + 8 * 512;
    int dvb = cvb, dkt = 0;                           // DMA cursor (tile, K step); past the last tile it stays on the last block (never read)
    bool dlive = true;
    // first block of a work item's K range: row panel (or N tile) x all K steps of the matrix, + the item's half
    auto a_blocks = [&](int wi) { const int ks = wi >= nvb ? 1 : 0, vb = wi - ks * nvb;
                                  return At + ((size_t)((vb & 7) + 8 * ((vb >> 3) / tiles_n)) * ksplit + ks) * nk * T_BLOCK; };
    auto w_blocks = [&](int wi) { const int ks = wi >= nvb ? 1 : 0, vb = wi - ks * nvb;
                                  return Wt + ((size_t)((vb >> 3) % tiles_n) * ksplit + ks) * nk * T_BLOCK; };
    const unsigned short* da = a_blocks(dvb);
    const unsigned short* dw = w_blocks(dvb);
#define T_DMA_A(stage_) do { const unsigned short* g_ = da + (size_t)dkt * T_BLOCK; unsigned short* l_ = lds + (stage_) * T_STAGE + w * 512; \
                             sed_dma16(g_ + voff, l_); sed_dma16(g_ + voff2, l_ + 8 * 512); } while (0)
#define T_DMA_W(stage_) do { const unsigned short* g_ = dw + (size_t)dkt * T_BLOCK; unsigned short* l_ = lds + (stage_) * T_STAGE + T_BLOCK + w * 512; \
                             sed_dma16(g_ + voff, l_); sed_dma16(g_ + voff2, l_ + 8 * 512); } while (0)
#define T_DMA_NEXT() do { if (dlive && ++dkt == nk) { const int nv_ = next_vb(dvb);                                              \
                              if (nv_ < nwi) { dvb = nv_; dkt = 0; da = a_blocks(dvb); dw = w_blocks(dvb); }                  \
                              else { dkt = nk - 1; dlive = false; } } } while (0)
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16_zero();
    // prologue: the first three K steps of the stream (a fixed DMA count per step keeps the vmcnt arithmetic below valid)
    T_DMA_A(0); T_DMA_W(0); T_DMA_NEXT();
    T_DMA_A(1); T_DMA_W(1); T_DMA_NEXT();
    T_DMA_A(2); T_DMA_W(2); T_DMA_NEXT();
    SED_WAIT_VM_LDS(8);                               // step 0 has landed (this wave's share)
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
    // acc[im][jn] = (W rows 32 jn ..) x (A rows 32 im ..)^T: lane -> A row lo of block im, W rows (r & 3) + 8 (r >> 2) + 4 hi of block jn
#ifndef T_DIAG
#define T_DIAG 0            // timing-only builds: 1 no MFMAs, 2 no fragment reads, 4 no DMA (wrong results)
#endif
#define T_MFMA(h) do { sed_mfma_prio(1); if (!(T_DIAG & 1)) {                                                                    \
        acc[2 * (h)][0] = mfma32_bf16(bh0, al0, acc[2 * (h)][0]); acc[2 * (h)][1] = mfma32_bf16(bh1, al0, acc[2 * (h)][1]);     \
        acc[2 * (h) + 1][0] = mfma32_bf16(bh0, al1, acc[2 * (h) + 1][0]); acc[2 * (h) + 1][1] = mfma32_bf16(bh1, al1, acc[2 * (h) + 1][1]); \
        acc[2 * (h)][0] = mfma32_bf16(bl0, ah0, acc[2 * (h)][0]); acc[2 * (h)][1] = mfma32_bf16(bl1, ah0, acc[2 * (h)][1]);     \
        acc[2 * (h) + 1][0] = mfma32_bf16(bl0, ah1, acc[2 * (h) + 1][0]); acc[2 * (h) + 1][1] = mfma32_bf16(bl1, ah1, acc[2 * (h) + 1][1]); \
        acc[2 * (h)][0] = mfma32_bf16(bh0, ah0, acc[2 * (h)][0]); acc[2 * (h)][1] = mfma32_bf16(bh1, ah0, acc[2 * (h)][1]);     \
        acc[2 * (h) + 1][0] = mfma32_bf16(bh0, ah1, acc[2 * (h) + 1][0]); acc[2 * (h) + 1][1] = mfma32_bf16(bh1, ah1, acc[2 * (h) + 1][1]); } \
        sed_mfma_prio(0); } while (0)
#ifdef T_STAMP
    unsigned long long* s_ts = (unsigned long long*)(lds + 4 * T_STAGE);
    const bool stamp_on = blockIdx.x == 64 && (w & 3) == 0;
#endif
    int step = 0;
    for (;;) {                                        // tiles of this workgroup
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt, ++step) {
        const unsigned short* base = lds + (step & 3) * T_STAGE;
        const int s3 = (step + 3) & 3;
        // first phase: W fragments + A fragments of the upper 64 rows, then the A pieces of step + 3 (fragment reads FIRST: the four waves
        // of a group issue their DMA pieces at the same moment and queue for the CU's one address pipe, profiles/r06g_linear_stamps_qkv.log)
        T_TS(0);
        if (!(T_DIAG & 2) || step == 0) { T_READ_B(base); T_READ_A(base, 0); }
        sed_sched_fence();
        T_TS(10);
        if (!(T_DIAG & 4)) T_DMA_A(s3);
        T_TS(1);
        sed_phase_barrier(); sed_wait_lds();
        T_TS(2);
        T_MFMA(0);
        T_TS(3);
        sed_phase_barrier();
        T_TS(4);
        // second phase: A fragments of the lower 64 rows, the W pieces of step + 3; then this wave's share of step + 1 must have landed
        if (!(T_DIAG & 2)) T_READ_A(base, 1);
        sed_sched_fence();
        T_TS(11);
        if (!(T_DIAG & 4)) T_DMA_W(s3);
        T_TS(12);
        T_DMA_NEXT();
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
        // ---- a tile is complete: bias (+ exact GELU), store, next tile ------------------------------------------------------------------
        const int eks = cvb >= nvb ? 1 : 0, evb = cvb - eks * nvb;
        const int tm = (evb & 7) + 8 * ((evb >> 3) / tiles_n), tn = (evb >> 3) % tiles_n;
        const int m0 = tm * P_BM, n0 = tn * P_BN;
        float* Cpart = (float*)Cout + (size_t)eks * M * N;              // (ksplit = 2: the second K half's partial sums; its bias is zero)
        int elo = lo, ehi = hi;                       // opaque copies: the epilogue's address arithmetic stays HERE (hoisted out of the K loop it
        sed_pin(elo); sed_pin(ehi);                   // occupied ~30 registers across it and spilled into the loop)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gn = n0 + 64 * wc + 32 * jn + 8 * q + 4 * ehi;                // 4 consecutive columns
                const float4 bv = (bias != nullptr && eks == 0) ? *(const float4*)(bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int im = 0; im < 4; ++im) {
                    const int mrow = 128 * wr + 32 * im + elo;
                    float4 v = make_float4(acc[im][jn][4 * q] + bv.x, acc[im][jn][4 * q + 1] + bv.y, acc[im][jn][4 * q + 2] + bv.z,
                                           acc[im][jn][4 * q + 3] + bv.w);
                    if (ACT == 1) {
                        v.x = 0.5f * v.x * (1.0f + erff(v.x * 0.70710678118654752f)); v.y = 0.5f * v.y * (1.0f + erff(v.y * 0.70710678118654752f));
                        v.z = 0.5f * v.z * (1.0f + erff(v.z * 0.70710678118654752f)); v.w = 0.5f * v.w * (1.0f + erff(v.w * 0.70710678118654752f));
                    }
                    if (OUT == 0) {
                        if ((T_DIAG & 8) ? (v.x == 1.2345e-30f) : (m0 + mrow < M)) *(float4*)(Cpart + (size_t)(m0 + mrow) * N + gn) = v;
                    } else {
                        uint2 h_, l_;
                        split4(v, h_, l_);
                        unsigned short* d_ = (unsigned short*)Cout + ((size_t)tm * (N / T_BK) + (gn >> 4)) * T_BLOCK + t_off(mrow, q & 1) + 4 * ehi;
                        *(uint2*)d_ = h_;
                        *(uint2*)(d_ + T_PLANE) = l_;
                    }
                }
            }
        cvb = next_vb(cvb);
        if (cvb >= nwi) break;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x16_zero();
    }
    if (wr == 0) sed_phase_barrier();                 // the barrier group 1 took at the top
    SED_WAIT_VM_LDS(0);                               // the tail's dummy DMAs: nothing may be in flight when the workgroup's LDS is released
#ifdef T_STAMP
    __syncthreads();
    if (stamp_on && lane < 16 && t_stamp_buf != nullptr)
        for (int q = 0; q < 16; ++q) t_stamp_buf[((w >> 2) * 16 + q) * 16 + lane] = s_ts[((w >> 2) * 16 + q) * 16 + lane];
#endif
#undef T_DMA_A
#undef T_DMA_W
#undef T_DMA_NEXT
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

// ---- loader-wave form (round 6, last step) -----------------------------------------------------------------------------------------------------
// linear_dma_kernel's remaining costs were both VMEM side effects on the waves that own the matrix pipe (profiles/r06_linear_diag.md): a DMA
// piece stalls its issuing wave in the CU's address pipe (the wave is late at its barrier, the other group's MFMAs wait), and the C
// stores share the wave's one vmcnt with the DMA pieces, so the first "has step + 1 landed" wait of the next tile drains the stores'
// acknowledgements (~ 10 us per tile).  Here TWO MORE waves (8, 9) do nothing but move data: each issues 16 of a K step's 32 DMA pieces,
// four per interval, and performs the counted wait; the eight MFMA waves issue no vector-memory load at all -- their read phases are
// 4 - 8 ds_read_b128, and their C stores are never waited for.  Same stages, same barrier sequence (the loaders keep group 0's clock),
// same hazards: the loaders' wait for step + 1 sits before the barrier that ends the step's fourth interval, the first read of step + 1 is
// one barrier later; the stage refilled during a step was last read in group 1's second read phase of the step before, retired by
// that wave's lgkmcnt(0) before its barrier.  640 threads: three waves on two of the SIMDs, so <= 168 VGPRs.
namespace {
template <int ACT, int OUT>
__global__ __launch_bounds__(640, 1) void linear_ldr_kernel(const unsigned short* __restrict__ At, const unsigned short* __restrict__ Wt,
                                                           const float* __restrict__ bias, void* __restrict__ Cout, int M, int N, int K,
                                                           int tiles_m, int tiles_n, int nvb) {
    SED_DYN_SMEM(smem);                               // [4 stages][A hi | A lo | W hi | W lo][256][16] bf16
    unsigned short* lds = (unsigned short*)smem;
    const int tid = threadIdx.x, lane = tid & 63, w = sed_wave_uniform(tid >> 6), lo = lane & 31, hi = lane >> 5;
    const int nk = K / T_BK, gstride = gridDim.x;
    auto valid = [&](int vb) { return (vb & 7) + 8 * ((vb >> 3) / tiles_n) < tiles_m; };
    auto next_vb = [&](int vb) { do vb += gstride; while (vb < nvb && !valid(vb)); return vb; };
    int cvb = (int)blockIdx.x;
    if (!valid(cvb)) cvb = next_vb(cvb);
    if (cvb >= nvb) return;
    if (w >= 8) {
        // ---------------- loader waves ----------------
        int ntl = 0;
        for (int vb = cvb; vb < nvb; vb = next_vb(vb)) ++ntl;
        const int total = ntl * nk, L = w - 8;
        // pieces addressed by SGPRs alone (sed_dma16_tid): one resource per operand, the piece's byte offset in soffset
        const sed_rsrc ra = sed_make_rsrc_tid16(At, (unsigned)((size_t)tiles_m * nk * T_BLOCK * 2));
        const sed_rsrc rw = sed_make_rsrc_tid16(Wt, (unsigned)((size_t)tiles_n * nk * T_BLOCK * 2));
        int dvb = cvb, dkt = 0;                       // DMA cursor (tile, K step); past the last tile it stays on the last block (never read)
        bool dlive = true;
        unsigned da = (unsigned)((dvb & 7) + 8 * ((dvb >> 3) / tiles_n)) * (unsigned)nk, dw = (unsigned)((dvb >> 3) % tiles_n) * (unsigned)nk;   // first block of the tile
        // interval q of a step: pieces 8 q + 4 L .. + 3 of the stage (0 - 15: the A block, 16 - 31: the W block)
#define L_ISSUE(stage_, q_) do { const unsigned o_ = ((((q_) < 2 ? da : dw) + (unsigned)dkt) * T_BLOCK + (8 * ((q_) & 1) + 4 * L) * 512) * 2;             \
                                 unsigned short* l_ = lds + (stage_) * T_STAGE + (8 * (q_) + 4 * L) * 512;                                                 \
                                 if ((q_) < 2) { sed_dma16_tid(ra, o_, l_); sed_dma16_tid(ra, o_ + 1024, l_ + 512);                                        \
                                                 sed_dma16_tid(ra, o_ + 2048, l_ + 1024); sed_dma16_tid(ra, o_ + 3072, l_ + 1536); }                       \
                                 else { sed_dma16_tid(rw, o_, l_); sed_dma16_tid(rw, o_ + 1024, l_ + 512);                                                 \
                                        sed_dma16_tid(rw, o_ + 2048, l_ + 1024); sed_dma16_tid(rw, o_ + 3072, l_ + 1536); } } while (0)
#define L_NEXT() do { if (dlive && ++dkt == nk) { const int nv_ = next_vb(dvb);                                                    \
                          if (nv_ < nvb) { dvb = nv_; dkt = 0; da = (unsigned)((dvb & 7) + 8 * ((dvb >> 3) / tiles_n)) * (unsigned)nk; \
                                           dw = (unsigned)((dvb >> 3) % tiles_n) * (unsigned)nk; }                               \
                          else { dkt = nk - 1; dlive = false; } } } while (0)
#ifdef T_STAMP
        unsigned long long* s_ts = (unsigned long long*)(lds + 4 * T_STAGE);
#define L_TS(k) do { __builtin_amdgcn_sched_barrier(0);                                                                          \
                     if (blockIdx.x == 64 && w == 8 && step >= 8 && step < 24) { const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
                                                                      if (lane == 0) s_ts[(2 * 16 + step - 8) * 16 + (k)] = t_; }   \
                     __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define L_TS(k) do { } while (0)
#endif
        for (int p = 0; p < 3; ++p) { L_ISSUE(p, 0); L_ISSUE(p, 1); L_ISSUE(p, 2); L_ISSUE(p, 3); L_NEXT(); }
        SED_WAIT_VM_LDS(32);                          // step 0 has landed (this wave's half)
        sed_phase_barrier();                          // (start) everybody
#pragma unroll 1
        for (int step = 0; step < total; ++step) {
            const int s3 = (step + 3) & 3;
            L_TS(0); L_ISSUE(s3, 0); L_TS(1); sed_phase_barrier();
            L_TS(2); L_ISSUE(s3, 1); L_TS(3); sed_phase_barrier();
            L_TS(4); L_ISSUE(s3, 2); L_TS(5); sed_phase_barrier();
            L_TS(6); L_ISSUE(s3, 3); L_NEXT(); L_TS(7);
            SED_WAIT_VM_LDS(32);                      // at most the 32 youngest pieces (steps + 2, + 3) outstanding: step + 1 has landed
            L_TS(8);
            sed_phase_barrier();
            L_TS(9);
        }
        sed_phase_barrier();                          // the barrier group 1 took at the top
        SED_WAIT_VM_LDS(0);
#ifdef T_STAMP
        sed_phase_barrier();                          // (stamped build: the MFMA waves' dump barrier)
        if (blockIdx.x == 64 && w == 8 && lane < 16 && t_stamp_buf != nullptr)
            for (int q = 0; q < 16; ++q) t_stamp_buf[(2 * 16 + q) * 16 + lane] = s_ts[(2 * 16 + q) * 16 + lane];
#endif
#undef L_TS
#undef L_ISSUE
#undef L_NEXT
        return;
    }
    // ---------------- MFMA waves ----------------
    const int wr = w >> 2, wc = w & 3;                // wave: rows 128 wr .. + 127, columns 64 wc .. + 63 of the tile; group = wr
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16_zero();
    sed_phase_barrier();                              // (start)
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
        acc[2 * (h)][0] = mfma32_bf16(bh0, al0, acc[2 * (h)][0]); acc[2 * (h)][1] = mfma32_bf16(bh1, al0, acc[2 * (h)][1]);     \
        acc[2 * (h) + 1][0] = mfma32_bf16(bh0, al1, acc[2 * (h) + 1][0]); acc[2 * (h) + 1][1] = mfma32_bf16(bh1, al1, acc[2 * (h) + 1][1]); \
        acc[2 * (h)][0] = mfma32_bf16(bl0, ah0, acc[2 * (h)][0]); acc[2 * (h)][1] = mfma32_bf16(bl1, ah0, acc[2 * (h)][1]);     \
        acc[2 * (h) + 1][0] = mfma32_bf16(bl0, ah1, acc[2 * (h) + 1][0]); acc[2 * (h) + 1][1] = mfma32_bf16(bl1, ah1, acc[2 * (h) + 1][1]); \
        acc[2 * (h)][0] = mfma32_bf16(bh0, ah0, acc[2 * (h)][0]); acc[2 * (h)][1] = mfma32_bf16(bh1, ah0, acc[2 * (h)][1]);     \
        acc[2 * (h) + 1][0] = mfma32_bf16(bh0, ah1, acc[2 * (h) + 1][0]); acc[2 * (h) + 1][1] = mfma32_bf16(bh1, ah1, acc[2 * (h) + 1][1]); \
        sed_mfma_prio(0); } while (0)
#ifdef T_STAMP
    unsigned long long* s_ts = (unsigned long long*)(lds + 4 * T_STAGE);
    const bool stamp_on = blockIdx.x == 64 && (w & 3) == 0;
#endif
    int step = 0;
    for (;;) {                                        // tiles of this workgroup
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt, ++step) {
        const unsigned short* base = lds + (step & 3) * T_STAGE;
        T_TS(0);
        T_READ_B(base); T_READ_A(base, 0);
        T_TS(1);
        sed_phase_barrier(); sed_wait_lds();
        T_TS(2);
        T_MFMA(0);
        T_TS(3);
        sed_phase_barrier();
        T_TS(4);
        T_READ_A(base, 1);
        sed_wait_lds();                               // before the barrier: the stage is refilled once every reader has passed it
        T_TS(5);
        sed_phase_barrier();
        T_TS(6);
        T_MFMA(1);
        T_TS(7);
        sed_phase_barrier();
        T_TS(8);
    }
        const int tm = (cvb & 7) + 8 * ((cvb >> 3) / tiles_n), tn = (cvb >> 3) % tiles_n;
        const int m0 = tm * P_BM, n0 = tn * P_BN;
        int elo = lo, ehi = hi;
        sed_pin(elo); sed_pin(ehi);
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gn = n0 + 64 * wc + 32 * jn + 8 * q + 4 * ehi;                // 4 consecutive columns
                const float4 bv = bias != nullptr ? *(const float4*)(bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int im = 0; im < 4; ++im) {
                    const int mrow = 128 * wr + 32 * im + elo;
                    float4 v = make_float4(acc[im][jn][4 * q] + bv.x, acc[im][jn][4 * q + 1] + bv.y, acc[im][jn][4 * q + 2] + bv.z,
                                           acc[im][jn][4 * q + 3] + bv.w);
                    if (ACT == 1) {
                        v.x = 0.5f * v.x * (1.0f + erff(v.x * 0.70710678118654752f)); v.y = 0.5f * v.y * (1.0f + erff(v.y * 0.70710678118654752f));
                        v.z = 0.5f * v.z * (1.0f + erff(v.z * 0.70710678118654752f)); v.w = 0.5f * v.w * (1.0f + erff(v.w * 0.70710678118654752f));
                    }
                    if (OUT == 0) {
                        if (m0 + mrow < M) *(float4*)((float*)Cout + (size_t)(m0 + mrow) * N + gn) = v;
                    } else {
                        uint2 h_, l_;
                        split4(v, h_, l_);
                        unsigned short* d_ = (unsigned short*)Cout + ((size_t)tm * (N / T_BK) + (gn >> 4)) * T_BLOCK + t_off(mrow, q & 1) + 4 * ehi;
                        *(uint2*)d_ = h_;
                        *(uint2*)(d_ + T_PLANE) = l_;
                    }
                }
            }
        cvb = next_vb(cvb);
        if (cvb >= nvb) break;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x16_zero();
    }
    if (wr == 0) sed_phase_barrier();                 // the barrier group 1 took at the top
#ifdef T_STAMP
    sed_phase_barrier();
    if (stamp_on && lane < 16 && t_stamp_buf != nullptr)
        for (int q = 0; q < 16; ++q) t_stamp_buf[((w >> 2) * 16 + q) * 16 + lane] = s_ts[((w >> 2) * 16 + q) * 16 + lane];
#endif
#undef T_READ_A
#undef T_READ_B
#undef T_MFMA
}
}  // namespace

static int linear_tiles_launch(const unsigned short* At, const unsigned short* Wt, const float* bias, void* Cout, int M, int N, int K, int act,
                               int out_tiles, void* stream, int ksplit = 1) {
    if (!At || !Wt || !Cout || M < 0 || N < 0 || K < 0) return SED_ERR_ARG;
    if (act < 0 || act > 1) return SED_ERR_ARG;
    if (M <= 0 || N <= 0) return SED_OK;
    if (ksplit == 2 && (act != 0 || out_tiles || (K / T_BK) % 2 != 0)) return SED_ERR_UNSUPPORTED;
    if (N % P_BN != 0 || K % T_BK != 0 || K <= 0 || ((uintptr_t)At & 15) || ((uintptr_t)Wt & 15) || ((uintptr_t)Cout & 15) || ((uintptr_t)bias & 15))
        return SED_ERR_UNSUPPORTED;
    const int tm = (M + P_BM - 1) / P_BM, tn = N / P_BN;
    const long long nvb_ll = 8LL * ((tm + 7) / 8) * tn;
    if (nvb_ll > 0x7fffffffLL) return SED_ERR_UNSUPPORTED;
    const int nvb = (int)nvb_ll;
    // one persistent workgroup per CU (128 KB of LDS each); a multiple of 8 so that a workgroup stays on one XCD's row panels
    // Which kernel: the eight-wave form.  The loader-wave form is faster where it was measured alone (one long-K tile per CU: 1 451 vs 1 615 -
    // 1 748 ns per K step; the QKV shape 270 vs 295 - 329 us on the same boxes) and NOT in the extractor: with it the q / k / v + fc2 launches take
    // 7.1 - 7.3 instead of 7.45 - 7.5 ms per 48 clips and the launches after them run slower by as much (fc1 4.04 -> 4.2 - 4.3 ms, the layer
    // norms + 3 %): 17.82 - 17.86 vs 17.55 - 17.60 ms in every alternation (profiles/r06s_beats_ab_forms.txt, r06t_beats_ab_forms.txt) -- the
    // extractor runs at the chip's power budget (all-zero operands: the same loop 1.5 x faster, profiles/r06t_linear_period.txt).  Tuning key
    // "linear_tiles" (tests, A/B): 3 = no start skew, 5 = the loader-wave form, n > 8 = n & ~7 workgroups (odd n: loader-wave form).
    const int tune = sed_tuning[SED_TUNE_LINEAR_TILES];
    bool ldr = tune == 5 || (tune > 8 && (tune & 1));
    if ((long long)tm * (K / T_BK) * T_BLOCK * 2 > 0x7fffffffLL || (long long)tn * (K / T_BK) * T_BLOCK * 2 > 0x7fffffffLL) ldr = false;   // (its buffer resources count bytes in 31 bits)
    int grid = tune > 8 ? tune & ~7 : 256;
    if (grid > nvb * ksplit) grid = nvb * ksplit;
    // skew = the longest walk's tile count when some workgroups walk fewer (0: none do, or switched off with the tuning key = 3)
    const long long ntiles = (long long)tm * tn * ksplit;
    int skew = (ntiles > grid && ntiles % grid != 0) ? (int)((ntiles + grid - 1) / grid) : 0;
    if (sed_tuning[SED_TUNE_LINEAR_TILES] == 3) skew = 0;
#ifdef T_STAMP
    constexpr int SMEM_T = 4 * T_STAGE * 2 + 8192;
#else
    constexpr int SMEM_T = 4 * T_STAGE * 2;
#endif
#define T_LAUNCH(A_, O_) do { if (ldr && ksplit == 1) { SED_MAX_SMEM((linear_ldr_kernel<A_, O_>), SMEM_T);   \
        SED_LAUNCH((linear_ldr_kernel<A_, O_>), dim3((unsigned)grid), dim3(640), SMEM_T, (hipStream_t)stream, At, Wt, bias, Cout, M, N, K, tm, tn, nvb); } \
        else { SED_MAX_SMEM((linear_dma_kernel<A_, O_>), SMEM_T);                                                               \
        SED_LAUNCH((linear_dma_kernel<A_, O_>), dim3((unsigned)grid), dim3(512), SMEM_T, (hipStream_t)stream, At, Wt, bias, Cout, M, N, K, tm, tn, nvb, skew, ksplit); } } while (0)
    if (out_tiles) { if (act) T_LAUNCH(1, 1); else T_LAUNCH(0, 1); }
    else { if (act) T_LAUNCH(1, 0); else T_LAUNCH(0, 0); }
#undef T_LAUNCH
    return sed_check_launch();
}

SED_API int sed_linear_tiles_bf16x3(const unsigned short* At, const unsigned short* Wt, const float* bias, float* Cm, int M, int N, int K,
                                    int act, void* stream) {
    return linear_tiles_launch(At, Wt, bias, Cm, M, N, K, act, 0, stream);
}

SED_API int sed_linear_tiles_split2_bf16x3(const unsigned short* At, const unsigned short* Wt, const float* bias, float* C2, int M, int N, int K,
                                           void* stream) {
    return linear_tiles_launch(At, Wt, bias, C2, M, N, K, 0, 0, stream, 2);
}

SED_API int sed_linear_tiles_out_bf16x3(const unsigned short* At, const unsigned short* Wt, const float* bias, unsigned short* Ct, int M, int N,
                                        int K, int act, void* stream) {
    return linear_tiles_launch(At, Wt, bias, Ct, M, N, K, act, 1, stream);
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
