// K6, split-bf16 ("bf16x3") variant of the 3x3 convolution (forward and data gradient) of blocks 1-6.
//
// Same implicit-GEMM tiling as conv3x3_kernel (sed_conv.hip) but on v_mfma_f32_32x32x16_bf16, which runs at 16x the
// rate of the exact-f32 MFMA.  fp32 accuracy is kept by splitting BOTH operands into bf16 pairs x = hi + lo
// (hi = bf16(x), lo = bf16(x - hi)) and issuing three MFMAs per product: hi*hi + hi*lo + lo*hi (the lo*lo term is
// < 2^-16 relative and dropped).  Every product is exact in the fp32 accumulator; the only loss is the 2^-17
// representation error of the operands, i.e. ~8e-6 relative on a dot product -- measured in tests/ against the fp32
// oracle with the same tolerances as the f32 path.  Net MFMA cost: 3/16 of the f32 path.
//
// Activations are split while the halo patch is staged into LDS (two bf16 planes, pixel-row stride CK+8 elements so the
// 16-byte fragment reads of a wave are conflict-free); weights are pre-split and pre-transposed once per forward by
// pack_weights_bf16_kernel into [tap][cin-chunk][plane][cout][cin] slabs that are copied verbatim into LDS.
#include "sed_common.h"
#include <stdlib.h>

#define CONVB_THREADS(COUT) ((COUT) >= 64 ? 512 : 256)

template <int CIN, int COUT, int TF, int MP = 128, int CKT = 32>
struct ConvBCfg {
    static constexpr int TR = MP / TF;
    // PWL = logical patch width; PW = its LDS pitch in pixels.  The A-fragment ds_read_b128 of a wave is served in lane groups
    // {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (MI355X_MICROARCH.md); with a pixel pitch of RSS / 8 = 5 (3) sixteen-byte slots a group is
    // conflict-free iff its 16 pixels have distinct (row * PW + col) mod 16.  Round 6 (conv_lane_pixel below): for TF <= 16 the two lane
    // groups take different tile rows; that works with PW = TF + 2 at TF = 16 and 4 and needs PW = 12 at TF = 8, PW = 6 at TF = 2.
#ifdef CONVB_OLDMAP     // A/B builds (tools/build_variant.py oldmap -DCONVB_OLDMAP): the lane = pixel map of rounds 2 - 5
    static constexpr int PWL = TF + 2, PW = TF + 2, PH = TR + 2, PP = PW * PH, PPL = PWL * PH;
#else
    static constexpr int PWL = TF + 2, PW = TF == 8 ? 12 : (TF == 2 ? 6 : TF + 2), PH = TR + 2, PP = PW * PH, PPL = PWL * PH;
#endif
    static constexpr int CK = CIN < CKT ? CIN : CKT;     // channels per chunk (CKT = 32: two MFMA k-steps of 16; 16: one)
    static constexpr int RSS = CK + 8;                   // LDS row stride in bf16 elements (80 B / 48 B)
    static constexpr int NCH = CIN / CK;
    static constexpr int NT = (COUT + 31) / 32;
    static constexpr int NCOL = NT * 32;                 // weight rows kept in LDS (padded to the MFMA tile)
    static constexpr int WM = MP / 32;
    static constexpr int THREADS = CONVB_THREADS(COUT);
    static constexpr int WN = THREADS / 64 / WM;
    static constexpr int NTW = NT / WN;
    static_assert(NTW >= 1 && NTW * WN == NT, "wave split must tile COUT");
    static_assert(THREADS % (CK / 4) == 0, "a thread keeps one channel quad for all its patch elements");
    static constexpr int PATCH_S = 2 * PP * RSS;         // shorts: hi plane | lo plane
    static constexpr int WBUF_S = 2 * NCOL * RSS;        // shorts per weight buffer: hi | lo
    static constexpr int SMEM = (PATCH_S + 3 * WBUF_S) * 2 + 64;    // patch + one kernel row (3 taps) of weight slabs
    static constexpr int SMEM_BNB = SMEM + 4 * CIN * 4;             // + the four per-channel BatchNorm-backward constants
    static constexpr int SLAB = 2 * COUT * CK;           // shorts per (tap, chunk) slab in global memory
};

// W (COUT, CIN, 3, 3) fp32 -> forward slabs Wf[tap][cc][plane][co][ci_local] and data-gradient slabs
// Wd[tap'][ccd][plane][ci][co_local] (taps flipped, channels transposed), bf16 hi/lo planes.
struct PackBJobs {
    const float* W[8];
    unsigned short* Wf[8];
    unsigned short* Wd[8];
    int cout[8], cin[8], start[9];
    int ckf[8], ckd[8];                                  // channels per chunk of the forward / data-gradient slabs (convb_ck)
    int n;
};
// Thread i of [0, tot): one element of the FORWARD slabs, i = (tap, co, ci) with ci fastest -- a wave's stores are runs of CK
// consecutive bf16 (32 - 64 B) per plane; thread tot + i: one element of the DATA-GRADIENT slabs, i = (tap, ci, co) with co fastest,
// same.  The reads are then strided (36 B resp. 36 CIN B apart: the 2 MB of weights sit in L2); until round 4 the element order of W
// was walked instead -- coalesced reads, but four 2-byte stores per thread each into a cache line of its own (2.1 M partial-line
// writes: most of the launch's 12 us, at the head of the step's critical path).
__device__ __forceinline__ void pack_weights_bf16_one(const PackBJobs& jobs, int i) {
    const int tot = jobs.start[jobs.n];
    const bool dgrad = i >= tot;
    if (dgrad) i -= tot;
    if (i >= tot) return;
    int j = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) j += (q < jobs.n && i >= jobs.start[q]) ? 1 : 0;
    if (dgrad && !jobs.Wd[j]) return;
    const int r = i - jobs.start[j], COUT = jobs.cout[j], CIN = jobs.cin[j];
    const int tap = r / (COUT * CIN), rem = r - tap * (COUT * CIN);
    unsigned short hi, lo;
    if (!dgrad) {
        const int co = rem / CIN, ci = rem - co * CIN;
        bf16_split(jobs.W[j][((size_t)co * CIN + ci) * 9 + tap], hi, lo);
        const int CK = jobs.ckf[j], NCH = CIN / CK, cc = ci / CK, cl = ci % CK;
        unsigned short* d = jobs.Wf[j] + ((size_t)(tap * NCH + cc) * 2 * COUT + co) * CK + cl;
        d[0] = hi;
        d[(size_t)COUT * CK] = lo;
    } else {
        const int ci = rem / COUT, co = rem - ci * COUT;                // tap: the data-gradient slab index (kernel flipped: source tap 8 - tap)
        bf16_split(jobs.W[j][((size_t)co * CIN + ci) * 9 + (8 - tap)], hi, lo);
        const int CK = jobs.ckd[j], NCH = COUT / CK, cc = co / CK, cl = co % CK;
        unsigned short* d = jobs.Wd[j] + ((size_t)(tap * NCH + cc) * 2 * CIN + ci) * CK + cl;
        d[0] = hi;
        d[(size_t)CIN * CK] = lo;
    }
}
__global__ __launch_bounds__(256) void pack_weights_bf16_kernel(PackBJobs jobs) {
    pack_weights_bf16_one(jobs, blockIdx.x * 256 + threadIdx.x);
}
// The CNN's prologue in ONE launch (round 4): everything a forward needs before its first convolution and that depends on nothing
// but the weights, a seed and the input -- the weight packs (blocks [0, pack_blocks)), the SpecAugment bands of the batch (one block)
// and, for a caller whose input buffer is rewritten before the backward pass reads it (the pipelined step's hand-over buffer), a
// private copy of the input (the remaining blocks, grid-stride over 16-byte words).  They were three dependent launches of 4 - 12 us
// at the head of the step's critical path.
struct PrologueExtra {
    int* bounds; int B, nb, f_param, n_freq, t_param, n_time; uint32_t seed; const unsigned* seed_dev;
    const float4* src; float4* dst; size_t n4; const float* src_tail; float* dst_tail; int ntail;
    int pack_blocks, copy_blocks;
};
__global__ __launch_bounds__(256) void cnn_prologue_bf16_kernel(PackBJobs jobs, PrologueExtra ex) {
    const int blk = blockIdx.x;
    if (blk < ex.pack_blocks) { pack_weights_bf16_one(jobs, blk * 256 + threadIdx.x); return; }
    if (blk == ex.pack_blocks) {
        if (ex.bounds) {
            const uint32_t seed = ex.seed + (ex.seed_dev ? *ex.seed_dev : 0u);
            for (int b = threadIdx.x; b < ex.B; b += 256) sed_specaug_draw(ex.bounds, b, ex.nb, ex.f_param, ex.n_freq, ex.t_param, ex.n_time, seed);
        }
        if ((int)threadIdx.x < ex.ntail) ex.dst_tail[threadIdx.x] = ex.src_tail[threadIdx.x];
        return;
    }
    const size_t stride = (size_t)ex.copy_blocks * 256;
    for (size_t i = (size_t)(blk - ex.pack_blocks - 1) * 256 + threadIdx.x; i < ex.n4; i += stride) ex.dst[i] = ex.src[i];
}
// Channels per weight chunk for a (CIN -> COUT) contraction: the packing and the kernel dispatch must agree, so both ask here.
static inline int convb_ck(int CIN, int COUT) {
    const int e = sed_tuning[SED_TUNE_CONVB_CK];       // tuning override (tools/convb_mp_sweep.py)
    // 64 <-> 128 channels: 16-channel chunks halve the LDS stage (68 KB at 256 pixels) and fit 128 VGPRs, so two workgroups
    // share a CU: 72.8 -> 63.0 us (forward) and 63.3 -> 56.8 us (data gradient).  128 -> 128 gains nothing (59.6 vs 60.9 us).
    int ck = ((CIN == 64 && COUT == 128) || (CIN == 128 && COUT == 64)) ? 16 : 32;
    if (e && CIN >= 32) ck = e == 16 ? 16 : 32;
    return CIN < ck ? CIN : ck;
}
static int packb_jobs(PackBJobs& jobs, int n, const void* const* W, void* const* Wf, void* const* Wd, const int* cout, const int* cin) {
    int tot = 0;
    for (int j = 0; j < n; ++j) {
        jobs.W[j] = (const float*)W[j]; jobs.Wf[j] = (unsigned short*)Wf[j]; jobs.Wd[j] = Wd ? (unsigned short*)Wd[j] : nullptr;
        jobs.cout[j] = cout[j]; jobs.cin[j] = cin[j]; jobs.start[j] = tot;
        jobs.ckf[j] = convb_ck(cin[j], cout[j]); jobs.ckd[j] = convb_ck(cout[j], cin[j]);
        tot += cout[j] * cin[j] * 9;
    }
    for (int j = n; j < 8; ++j) { jobs.W[j] = nullptr; jobs.Wf[j] = nullptr; jobs.Wd[j] = nullptr; jobs.cout[j] = 0; jobs.cin[j] = 0; jobs.start[j] = tot; jobs.ckf[j] = 1; jobs.ckd[j] = 1; }
    jobs.start[n] = tot;
    jobs.start[8] = tot;
    jobs.n = n;
    return tot;
}
static inline bool packb_any_dgrad(const PackBJobs& jobs) {
    for (int j = 0; j < jobs.n; ++j) if (jobs.Wd[j]) return true;
    return false;
}
// n <= 8 layers; Wf / Wd buffers of 9*CIN*COUT*4 BYTES each (same size as the fp32 packs).
SED_API int sed_conv_pack_multi_bf16(int n, const void* const* W, void* const* Wf, void* const* Wd, const int* cout,
                                        const int* cin, void* stream) {
    if (n < 1 || n > 8) return SED_ERR_ARG;
    PackBJobs jobs;
    const int tot = packb_jobs(jobs, n, W, Wf, Wd, cout, cin) * (packb_any_dgrad(jobs) ? 2 : 1);
    SED_LAUNCH(pack_weights_bf16_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, jobs);
    return sed_check_launch();
}
// The same packs + (bounds != null) the seeded SpecAugment bands of sed_specaug_bounds_seeded + (copy_dst != null) copy_n floats
// copy_src -> copy_dst (both 16-byte aligned), one launch.  n == 0: no packs.
SED_API int sed_cnn_prologue_bf16(int n, const void* const* W, void* const* Wf, void* const* Wd, const int* cout, const int* cin,
                                     int* bounds, int B, int nb, int f_param, int n_freq, int t_param, int n_time, unsigned seed,
                                     const unsigned* seed_dev, const float* copy_src, float* copy_dst, long long copy_n, void* stream) {
    if (n < 0 || n > 8) return SED_ERR_ARG;
    if (bounds && B > 0 && nb != 1 && nb != B) return SED_ERR_ARG;
    if (copy_dst && (!copy_src || copy_n < 0 || ((uintptr_t)copy_src & 15) || ((uintptr_t)copy_dst & 15))) return SED_ERR_ARG;
    PackBJobs jobs;
    const int tot = packb_jobs(jobs, n, W, Wf, Wd, cout, cin) * (packb_any_dgrad(jobs) ? 2 : 1);
    PrologueExtra ex;
    ex.bounds = (bounds && B > 0) ? bounds : nullptr; ex.B = B; ex.nb = nb; ex.f_param = f_param; ex.n_freq = n_freq;
    ex.t_param = t_param; ex.n_time = n_time; ex.seed = (uint32_t)seed; ex.seed_dev = seed_dev;
    const size_t n4 = copy_dst ? (size_t)copy_n / 4 : 0;
    ex.src = (const float4*)copy_src; ex.dst = (float4*)copy_dst; ex.n4 = n4;
    ex.ntail = copy_dst ? (int)(copy_n % 4) : 0;
    ex.src_tail = copy_src ? copy_src + 4 * n4 : nullptr; ex.dst_tail = copy_dst ? copy_dst + 4 * n4 : nullptr;
    ex.pack_blocks = (tot + 255) / 256;
    // four 16-byte words per thread: enough blocks to stream 15 MB in a few us, few enough not to delay the pack blocks
    size_t cb = (n4 + 1023) / 1024;
    ex.copy_blocks = (int)(cb > 4096 ? 4096 : cb);
    if (ex.pack_blocks == 0 && !ex.bounds && n4 == 0 && ex.ntail == 0) return SED_OK;
    SED_LAUNCH(cnn_prologue_bf16_kernel, dim3(ex.pack_blocks + 1 + ex.copy_blocks), dim3(256), 0, (hipStream_t)stream, jobs, ex);
    return sed_check_launch();
}

// CONVB_ABL: timing-ablation mask for tools/convb_variants.py (0 in the product build): 1 = no weight-slab global loads,
// 2 = no MFMAs, 4 = no patch global loads, 8 = no per-tap barrier, 16 = no LDS fragment reads, 32 = no output stores
// (results are wrong under any of them).
#ifndef CONVB_ABL
#define CONVB_ABL 0
#endif

// BNB (data gradient of a training-mode block, round 3): the BatchNorm backward is applied while the operand is staged.  `x` is
// then dz = dL/d(xhat) from the GLU backward, `bnb.ybn` the block's saved pre-BN conv output, and what goes into the MFMA planes is
// dy = istd (dz - m1 - (ybn - mean) istd m2), m1 = gamma dbeta / n, m2 = gamma dgamma / n -- the expression of bn_bwd_apply_kernel
// (sed_glu.hip), operation for operation, so dy has the same bits as the separate in-place pass it replaces (6 launches, 186 us
// and 0.93 GB of HBM traffic per step).  Every workgroup also writes the dy of its OWN pixels (the patch without its halo) to
// `bnb.dy_out` for the weight-gradient kernel that runs next.  Zero padding stays zero: out-of-image pixels are not transformed.
struct ConvBnb {
    const float* ybn;       // (B, T, F, CIN) pre-BN conv output saved by the forward
    const float* stats;     // mean[CIN] | invstd[CIN]
    const float* gamma;
    const float* dgamma;
    const float* dbeta;
    float* dy_out;          // (B, T, F, CIN)
    float* dbias;           // conv-bias gradient: identically zero in training mode
    float inv_count;
};

// A-fragment row i (0 .. 31) of a wave -> (row, col) of its pixel inside the wave's 32-pixel block of the tile (32 / TF tile rows of TF
// pixels).  TF = 32: the identity.  TF <= 16: the rows are dealt to the two ds_read_b128 lane groups (see ConvBCfg) -- group 0 takes the even
// rows (TF = 2: the first eight), group 1 the odd ones -- so that every 16-lane group reads 16 distinct 16-byte bank slots.  The epilogue
// maps the accumulator rows through the same function.  (Until round 6: pixel = i; lds_conflict 0.26 - 0.40 of the LDS cycles at TF <= 16,
// profiles/r05k_pmc_wait.md.)
template <int TF>
__device__ __forceinline__ void conv_lane_pixel(int i, int& prow, int& pcol) {
    if (TF >= 32) { prow = 0; pcol = i; return; }
#ifdef CONVB_OLDMAP
    prow = i / TF; pcol = i % TF; return;
#endif
    const int qd = i >> 2, gsel = (0x96 >> qd) & 1, g = 4 * (qd >> 1) + (i & 3);      // lane group and position (0 .. 15) inside it
    if (TF == 2) { prow = 8 * gsel + (g >> 1); pcol = g & 1; }
    else { prow = gsel + 2 * (g / TF); pcol = g % TF; }
}

template <int CIN, int COUT, int TF, bool STATS, int MP = 128, int CKT = 32, bool BNB = false>
__global__ __launch_bounds__(CONVB_THREADS(COUT)) void conv3x3_bf16_kernel(const float* __restrict__ x,
                                                                            const unsigned short* __restrict__ Wp,
                                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                                            float* __restrict__ partial, int B, int T, int F,
                                                                            ConvBnb bnb) {
    using Cfg = ConvBCfg<CIN, COUT, TF, MP, CKT>;
    constexpr int TR = Cfg::TR, PW = Cfg::PW, PWL = Cfg::PWL, PP = Cfg::PP, PPL = Cfg::PPL, CK = Cfg::CK, RSS = Cfg::RSS, NCH = Cfg::NCH, NT = Cfg::NT,
                  NTW = Cfg::NTW, THREADS = Cfg::THREADS, WM = Cfg::WM, NCOL = Cfg::NCOL, SLAB = Cfg::SLAB, WBUF_S = Cfg::WBUF_S;
    SED_DYN_SMEM(smem_raw);
    // (no integer round-trip on the LDS pointer: that would demote every LDS access to a flat_* instruction)
    unsigned short* patch = (unsigned short*)smem_raw;                  // 16-byte aligned; hi plane, then lo plane
    unsigned short* wbuf = patch + Cfg::PATCH_S;
    float* bnc = (float*)(smem_raw + Cfg::SMEM);                        // BNB: mean | istd | m1 | m2, CIN floats each
    const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) % WM, wn = (tid >> 6) / WM, lo = lane & 31, hi = lane >> 5;
    // Persistent workgroups: workgroup g walks tiles g, g + gridDim.x, ...  (grid == number of tiles: one tile each, as before).  The
    // halo patch and the first weight row of the NEXT tile are fetched into the (then idle) prefetch registers during the last
    // kernel row of the current one and stay in flight through its epilogue: the narrow layers (one or two cin chunks: 54 - 108 MFMAs
    // per wave and tile) otherwise spend more time waiting for their prologue loads than computing (SQ_WAIT_ANY 0.5 - 0.6 of the
    // wave cycles, profiles/r03f_pmc_wait.md).
    const int ftiles = F / TF, ttiles = (T + TR - 1) / TR, ntiles = B * ttiles * ftiles;
    constexpr int WROWS = TF >= 32 ? 1 : 32 / TF;                      // tile rows per wave (TF = 32: rows of 32 pixels, 32 w / TF = w)
    int lprow, lpcol;
    conv_lane_pixel<TF>(lo, lprow, lpcol);
    const int abase = (((TF >= 32 ? (32 * w) / TF : WROWS * w) + lprow) * PW + (TF >= 32 ? (32 * w) % TF : 0) + lpcol) * RSS + 8 * hi;   // this lane's A-fragment offset inside a plane

    f32x16 acc[NTW];

    // weight rows beyond COUT (narrow data-gradient outputs) stay zero for the whole kernel
    if (NCOL > COUT) {
        for (int i = tid; i < 3 * WBUF_S; i += THREADS) wbuf[i] = 0;
        __syncthreads();
    }
    // The weights stream through LDS one KERNEL ROW (3 taps x one cin chunk) at a time: the next row's three slabs are
    // fetched into registers at the start of an iteration and parked in LDS at its end, so their L2 latency hides behind
    // three taps of MFMAs (timing ablation, tools/convb_variants.py: with one tap per iteration the slab wait was 30-60 %
    // of the kernel).  The halo patch of the next cin chunk is prefetched the same way during the last row of a chunk.
    constexpr int WPIECES = SLAB / 8;                                   // 16-byte pieces per slab
    constexpr int WV = (WPIECES + THREADS - 1) / THREADS;
    constexpr int V = CK / 4;
    constexpr int NLD = (PPL * V + THREADS - 1) / THREADS;
    uint4 wreg[3 * WV];
    float4 ld[NLD];
    float4 ldy[BNB ? NLD : 1];
    if (BNB) {
        for (int c = tid; c < CIN; c += THREADS) {
            const float g = bnb.gamma[c];
            bnc[c] = bnb.stats[c];
            bnc[CIN + c] = bnb.stats[CIN + c];
            bnc[2 * CIN + c] = g * bnb.dbeta[c] * bnb.inv_count;
            bnc[3 * CIN + c] = g * bnb.dgamma[c] * bnb.inv_count;
        }
        if (blockIdx.x == 0 && tid < CIN && bnb.dbias != nullptr) bnb.dbias[tid] = 0.f;
        // (visible to store_patch through the first __syncthreads below: the first store_patch of chunk 0 runs before it, so
        //  the constants of chunk 0 are read after an explicit barrier here)
        __syncthreads();
    }
    auto w_dst = [&](int buf, int piece) -> unsigned short* {
        const int plane = piece / (COUT * CK / 8), rem = piece - plane * (COUT * CK / 8);
        const int co = rem / (CK / 8), pc = rem - co * (CK / 8);
        return wbuf + buf * WBUF_S + plane * NCOL * RSS + co * RSS + 8 * pc;
    };
    auto load_row = [&](int cc, int r) {
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
            const uint4* src = (const uint4*)(Wp + ((size_t)(3 * r + t3) * NCH + cc) * SLAB);
#pragma unroll
            for (int i = 0; i < WV; ++i) {
                const int piece = tid + THREADS * i;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (piece < WPIECES && !(CONVB_ABL & 1)) v = src[piece];
                wreg[t3 * WV + i] = v;
            }
        }
    };
    auto store_row = [&]() {
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
            for (int i = 0; i < WV; ++i) {
                const int piece = tid + THREADS * i;
                if (piece < WPIECES) *(uint4*)w_dst(t3, piece) = wreg[t3 * WV + i];
            }
    };
    auto load_patch = [&](int cc, int pb, int pt0, int pf0) {           // (pb, pt0, pf0): the tile the patch belongs to
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = tid + THREADS * u;
            const int pix = idx / V, v = idx - pix * V;
            const int i = pix / PWL, j = pix - i * PWL;
            const int t = pt0 - 1 + i, f = pf0 - 1 + j;
            ld[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BNB) ldy[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < PPL * V && t >= 0 && t < T && f >= 0 && f < F && !(CONVB_ABL & 4)) {
                const size_t off = (((size_t)pb * T + t) * F + f) * CIN + cc * CK + 4 * v;
                ld[u] = *(const float4*)(x + off);
                if (BNB) ldy[u] = *(const float4*)(bnb.ybn + off);
            }
        }
    };
    // (single-chunk layers only, CIN <= 32: on the wider ones the prefetch registers held across the epilogue cost an occupancy step
    //  -- 128 -> 206 registers on the 64 -> 128 layer -- and their prologue is a small part of a tile anyway)
    constexpr bool PERS = CIN <= 32;
    bool primed = false;
    int tile_it = blockIdx.x;
    do {
    const int tile = PERS ? tile_it : (int)blockIdx.x;
    const int ft = tile % ftiles, tt = (tile / ftiles) % ttiles, b = tile / (ftiles * ttiles);
    const int t0 = tt * TR, f0 = ft * TF;
    const int tile_n = PERS ? tile + (int)gridDim.x : ntiles;
    int nb = 0, nt0 = 0, nf0 = 0;
    if (PERS && tile_n < ntiles) { nb = tile_n / (ftiles * ttiles); nt0 = ((tile_n / ftiles) % ttiles) * TR; nf0 = (tile_n % ftiles) * TF; }
    auto store_patch = [&](int cc) {    // split into bf16 hi / lo planes (this tile: b, t0, f0)
        float4 bmean, bistd, bm1, bm2;
        if (BNB) {                      // this thread's channel quad of the chunk is the same for all its patch elements
            const int c0 = cc * CK + 4 * (tid % V);
            bmean = *(const float4*)(bnc + c0);
            bistd = *(const float4*)(bnc + CIN + c0);
            bm1 = *(const float4*)(bnc + 2 * CIN + c0);
            bm2 = *(const float4*)(bnc + 3 * CIN + c0);
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = tid + THREADS * u;
            if (idx < PPL * V) {
                const int pixl = idx / V, v = idx - pixl * V;
                const int i = pixl / PWL, j = pixl - i * PWL, pix = i * PW + j;       // logical patch pixel -> its LDS slot
                if (BNB) {
                    const int t = t0 - 1 + i, f = f0 - 1 + j;
                    if (t >= 0 && t < T && f >= 0 && f < F) {
                        float4 g = ld[u];
                        const float4 yv = ldy[u];
                        g.x = bistd.x * (g.x - bm1.x - (yv.x - bmean.x) * bistd.x * bm2.x);
                        g.y = bistd.y * (g.y - bm1.y - (yv.y - bmean.y) * bistd.y * bm2.y);
                        g.z = bistd.z * (g.z - bm1.z - (yv.z - bmean.z) * bistd.z * bm2.z);
                        g.w = bistd.w * (g.w - bm1.w - (yv.w - bmean.w) * bistd.w * bm2.w);
                        ld[u] = g;
                        if (i >= 1 && i <= TR && j >= 1 && j <= TF)        // this workgroup's own pixels: dy for the weight gradient
                            *(float4*)(bnb.dy_out + (((size_t)b * T + t) * F + f) * CIN + cc * CK + 4 * v) = g;
                    }
                }
                uint2 hv, lv;
                bf16_split2(ld[u].x, ld[u].y, hv.x, lv.x);
                bf16_split2(ld[u].z, ld[u].w, hv.y, lv.y);
                *(uint2*)(patch + pix * RSS + 4 * v) = hv;
                *(uint2*)(patch + PP * RSS + pix * RSS + 4 * v) = lv;
            }
        }
    };

#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[nt] = f32x16_zero();
    if (!PERS || !primed) {                 // the first tile of this workgroup: nothing was prefetched
        load_patch(0, b, t0, f0);
        load_row(0, 0);
        primed = true;
    }
    store_patch(0);
    store_row();
    __syncthreads();
#pragma unroll 1
    for (int cc = 0; cc < NCH; ++cc) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const bool next_chunk = r == 2 && cc + 1 < NCH;
            const bool more = r < 2 || cc + 1 < NCH;
            if (more) load_row(r < 2 ? cc : cc + 1, r < 2 ? r + 1 : 0);
            if (next_chunk) load_patch(cc + 1, b, t0, f0);
            if (PERS && !more && tile_n < ntiles) {  // last kernel row of the tile: the next tile's prologue loads go out now
                load_row(0, 0);
                load_patch(0, nb, nt0, nf0);
            }
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
                const unsigned short* wb = wbuf + t3 * WBUF_S;
                const unsigned short* ap = patch + abase + (r * PW + t3) * RSS;
#pragma unroll
                for (int ks = 0; ks < CK / 16; ++ks) {
                    s16x8 a_hi, a_lo;
                    if (CONVB_ABL & 16) { a_hi = (s16x8)(short)(lane + ks); a_lo = (s16x8)(short)(lane + t3); }
                    else { a_hi = *(const s16x8*)(ap + 16 * ks); a_lo = *(const s16x8*)(ap + PP * RSS + 16 * ks); }
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const unsigned short* bp = wb + ((wn * NTW + nt) * 32 + lo) * RSS + 16 * ks + 8 * hi;
                        s16x8 b_hi, b_lo;
                        if (CONVB_ABL & 16) { b_hi = (s16x8)(short)(lane + nt); b_lo = (s16x8)(short)(lane + r); }
                        else { b_hi = *(const s16x8*)bp; b_lo = *(const s16x8*)(bp + NCOL * RSS); }
                        if (CONVB_ABL & 2) { acc[nt][0] += (float)(a_lo[0] + b_hi[0] + a_hi[1] + b_lo[1]); continue; }
                        acc[nt] = mfma32_bf16(a_lo, b_hi, acc[nt]);
                        acc[nt] = mfma32_bf16(a_hi, b_lo, acc[nt]);
                        acc[nt] = mfma32_bf16(a_hi, b_hi, acc[nt]);
                    }
                }
            }
            if (!(CONVB_ABL & 8)) __syncthreads();      // every wave is done with this row's slabs (and, at r == 2, the patch)
            if (more) store_row();
            if (next_chunk) store_patch(cc + 1);
            if (!(CONVB_ABL & 8)) __syncthreads();
        }
    }
    // ---- epilogue: bias, store, per-channel partial statistics (identical to the f32 kernel) ----
    float* red = (float*)wbuf;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int co = (wn * NTW + nt) * 32 + lo;
        const float bv = (bias != nullptr && co < COUT) ? bias[co] : 0.f;
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int eprow, epcol;
            conv_lane_pixel<TF>(mfma32_row(r, lane), eprow, epcol);
            const int t = t0 + (TF >= 32 ? (32 * w) / TF : WROWS * w) + eprow, f = f0 + (TF >= 32 ? (32 * w) % TF : 0) + epcol;
            if (t < T && co < COUT) {
                const float v = acc[nt][r] + bv;
                if (!(CONVB_ABL & 32)) y[(((size_t)b * T + t) * F + f) * COUT + co] = v;
                s += v;
                s2 += v * v;
            }
        }
        if (STATS) {
            s += __shfl_xor(s, 32);
            s2 += __shfl_xor(s2, 32);
            if (hi == 0) {
                red[(w * 2 + 0) * (NT * 32) + co] = s;
                red[(w * 2 + 1) * (NT * 32) + co] = s2;
            }
        }
    }
    if (STATS) {
        __syncthreads();
        if (tid < 2 * COUT) {
            const int which = tid / COUT, co = tid - which * COUT;
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < WM; ++ww) v += red[(ww * 2 + which) * (NT * 32) + co];
            partial[(size_t)tid * ntiles + tile] = v;         // [2*COUT][ntiles], see bn_finalize_kernel
        }
    }
    if (PERS && STATS && NCOL > COUT) {     // (the statistics epilogue wrote `red` over the zero rows of a narrow output: restore them)
        __syncthreads();
        for (int i = tid; i < 3 * WBUF_S; i += THREADS) wbuf[i] = 0;
    }
    if constexpr (!PERS) break;             // (compile time: no back edge, straight-line code -- a loop, even a one-trip one, keeps
                                            //  the prefetch registers alive and cost the 64 -> 128 layer its second workgroup per CU)
    __syncthreads();                        // `red` (inside wbuf) and the patch are free again before the next tile parks its operands
    tile_it += (int)gridDim.x;
    } while (tile_it < ntiles);
}

// Grid of the persistent single-chunk layers (CIN <= 32; the wide layers keep one tile per workgroup).  Built-in choice: exactly the
// workgroups that are resident at once (occupancy x CUs, queried per kernel), so that every workgroup starts immediately and walks
// ceil / floor(ntiles / grid) tiles -- with a fixed number of tiles per workgroup the last partial round of workgroups sets the
// launch time (same-box sweep, ms per step: one tile 3.46, four tiles 3.40, five / six 3.44 - 3.45, eight 3.25 - 3.28 vs 3.22 for four
// on another box: the optimum moves with ntiles / resident slots).  sed_set_tuning key 12: n > 0 = n tiles per workgroup, -1 = off.
template <class K>
static inline int convb_persistent_grid(K kern, int threads, int smem, int ntiles, int CIN) {
    if (CIN > 32) return ntiles;            // (not compiled as persistent: see the kernel)
    const int e = sed_tuning[SED_TUNE_CONVB_TPW];
    if (e < 0) return ntiles;
    if (e > 0) return (ntiles + e - 1) / e;
#ifdef SED_EMU
    (void)kern; (void)threads; (void)smem;
    return ntiles;                          // (the CPU emulator has no occupancy: tests set the key)
#else
    static int slots = 0;                   // per instantiation (K is part of the template signature)
    if (slots == 0) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, threads, (size_t)smem) != hipSuccess || per_cu < 1)
            slots = -1;
        else
            slots = per_cu * prop.multiProcessorCount;
    }
    if (slots < 0 || ntiles <= 2 * slots) return ntiles;        // short grids: the hardware's own round-robin is as good
    return slots;
#endif
}

template <int CIN, int COUT, int TF, int MP = 128, int CKT = 32, bool BNB = false>
static int launch_convb(const float* x, const unsigned short* Wp, const float* bias, float* y, float* partial, int B, int T, int F,
                        hipStream_t s, ConvBnb bnb = ConvBnb()) {
    using Cfg = ConvBCfg<CIN, COUT, TF, MP, CKT>;
    const int ntiles = B * ((T + Cfg::TR - 1) / Cfg::TR) * (F / TF);
    if constexpr (BNB) {
        if constexpr (CIN >= COUT) {         // data gradients only (a block's convolution never narrows in the forward direction)
            SED_MAX_SMEM((conv3x3_bf16_kernel<CIN, COUT, TF, false, MP, CKT, true>), Cfg::SMEM_BNB);
            const int nblk = convb_persistent_grid(conv3x3_bf16_kernel<CIN, COUT, TF, false, MP, CKT, true>, Cfg::THREADS, Cfg::SMEM_BNB, ntiles, CIN);
            SED_LAUNCH((conv3x3_bf16_kernel<CIN, COUT, TF, false, MP, CKT, true>), dim3(nblk), dim3(Cfg::THREADS), Cfg::SMEM_BNB, s, x, Wp, bias, y, partial, B, T, F, bnb);
            return sed_check_launch();
        } else {
            return SED_ERR_UNSUPPORTED;
        }
    } else if (partial) {
        SED_MAX_SMEM((conv3x3_bf16_kernel<CIN, COUT, TF, true, MP, CKT>), Cfg::SMEM);
        const int nblk = convb_persistent_grid(conv3x3_bf16_kernel<CIN, COUT, TF, true, MP, CKT>, Cfg::THREADS, Cfg::SMEM, ntiles, CIN);
        SED_LAUNCH((conv3x3_bf16_kernel<CIN, COUT, TF, true, MP, CKT>), dim3(nblk), dim3(Cfg::THREADS), Cfg::SMEM, s, x, Wp, bias, y, partial, B, T, F, bnb);
    } else {
        SED_MAX_SMEM((conv3x3_bf16_kernel<CIN, COUT, TF, false, MP, CKT>), Cfg::SMEM);
        const int nblk = convb_persistent_grid(conv3x3_bf16_kernel<CIN, COUT, TF, false, MP, CKT>, Cfg::THREADS, Cfg::SMEM, ntiles, CIN);
        SED_LAUNCH((conv3x3_bf16_kernel<CIN, COUT, TF, false, MP, CKT>), dim3(nblk), dim3(Cfg::THREADS), Cfg::SMEM, s, x, Wp, bias, y, partial, B, T, F, bnb);
    }
    return sed_check_launch();
}

// Pixels per workgroup for the split-bf16 kernel.  The MFMA phase per (chunk, tap) step is 5x shorter than in the f32
// kernel, so the per-step costs (weight-slab streaming, barrier) must be amortised over more pixels: 256-pixel tiles
// (each wave 32 px x all couts) wherever that still leaves >= ~230 workgroups.
static inline int convb_mp(int F, int CIN, int COUT) {
    const int e = sed_tuning[SED_TUNE_CONVB_MP];       // tuning override (tools/convb_mp_sweep.py)
    if (e && COUT >= 64) return e;
    if (COUT < 64) return 128;
    if (CIN <= 32) return 128;          // one cin chunk: a 256-px patch + the 3-tap weight row leaves one workgroup per CU (44 vs 57 us)
    if (CIN == 128 && COUT == 128 && F <= 2) return 64;
    if (CIN == 128 && COUT == 128 && F <= 4) return 128;
    return 256;
}
SED_API int sed_conv_fwd_blocks_bf16(int B, int T, int F, int CIN, int COUT) {
    if (CIN == 1) return B * ((T + 15) / 16);
    const int TF = F >= 32 ? 32 : F;
    const int TR = convb_mp(F, CIN, COUT) / TF;
    return B * ((T + TR - 1) / TR) * (F / TF);
}

template <bool BNB>
static int convb_dispatch(const float* x, const void* Wp, const float* bias, float* y, float* partial, int B, int T, int F,
                          int CIN, int COUT, hipStream_t s, ConvBnb bnb) {
    if (B <= 0 || T <= 0) return SED_OK;
    const int TF = F >= 32 ? 32 : F;
    if (F % TF != 0 || (F & (F - 1)) != 0 || F < 2) return SED_ERR_UNSUPPORTED;
    const unsigned short* W = (const unsigned short*)Wp;
    const int MP = convb_mp(F, CIN, COUT);
    const int CK = CIN >= 32 ? convb_ck(CIN, COUT) : 32;      // narrower inputs are a single chunk either way
#define CONVB_CASE16(ci, co, tf, mp) \
    if (CIN == ci && COUT == co && TF == tf && MP == mp && CK == 16) return launch_convb<ci, co, tf, mp, 16, BNB>(x, W, bias, y, partial, B, T, F, s, bnb);
    // 16-channel weight chunks (half the LDS per workgroup): the wide production shapes
    CONVB_CASE16(32, 64, 32, 128) CONVB_CASE16(64, 128, 16, 128) CONVB_CASE16(64, 128, 16, 256)
    CONVB_CASE16(128, 128, 8, 128) CONVB_CASE16(128, 128, 8, 256) CONVB_CASE16(128, 128, 4, 64) CONVB_CASE16(128, 128, 4, 128)
    CONVB_CASE16(128, 128, 2, 64) CONVB_CASE16(128, 128, 2, 128) CONVB_CASE16(64, 32, 32, 128)
    CONVB_CASE16(128, 64, 16, 128) CONVB_CASE16(128, 64, 16, 256)
    // small-shape variants used by the unit tests / other n_mels
    CONVB_CASE16(64, 128, 4, 256) CONVB_CASE16(64, 128, 8, 256) CONVB_CASE16(64, 128, 32, 256)
    CONVB_CASE16(128, 64, 4, 256) CONVB_CASE16(128, 64, 8, 256) CONVB_CASE16(128, 64, 32, 256)
#undef CONVB_CASE16
    if (CK == 16) return SED_ERR_UNSUPPORTED;
#define CONVB_CASE(ci, co, tf, mp) \
    if (CIN == ci && COUT == co && TF == tf && MP == mp) return launch_convb<ci, co, tf, mp, 32, BNB>(x, W, bias, y, partial, B, T, F, s, bnb);
    // production shapes of the 2023 recipe (forward, then data gradient)
    CONVB_CASE(16, 32, 32, 128)
    CONVB_CASE(32, 64, 32, 128) CONVB_CASE(32, 64, 32, 256)
    CONVB_CASE(64, 128, 16, 128) CONVB_CASE(64, 128, 16, 256)
    CONVB_CASE(128, 128, 8, 128) CONVB_CASE(128, 128, 8, 256)
    CONVB_CASE(128, 128, 4, 64) CONVB_CASE(128, 128, 4, 128) CONVB_CASE(128, 128, 4, 256)
    CONVB_CASE(128, 128, 2, 64) CONVB_CASE(128, 128, 2, 128)
    CONVB_CASE(32, 16, 32, 128) CONVB_CASE(64, 32, 32, 128)
    CONVB_CASE(128, 64, 16, 128) CONVB_CASE(128, 64, 16, 256)
    // small-shape variants used by the unit tests / other n_mels
    CONVB_CASE(16, 32, 16, 128) CONVB_CASE(32, 64, 8, 256) CONVB_CASE(32, 64, 16, 256) CONVB_CASE(64, 128, 4, 256) CONVB_CASE(64, 128, 8, 256)
    CONVB_CASE(64, 128, 32, 256) CONVB_CASE(128, 128, 16, 256) CONVB_CASE(128, 128, 32, 256)
    CONVB_CASE(32, 16, 16, 128) CONVB_CASE(64, 32, 8, 128) CONVB_CASE(64, 32, 16, 128) CONVB_CASE(128, 64, 4, 256) CONVB_CASE(128, 64, 8, 256)
    CONVB_CASE(128, 64, 32, 256)
#undef CONVB_CASE
    return SED_ERR_UNSUPPORTED;
}

// Same contract as sed_conv3x3 with Wp from sed_conv_pack_multi_bf16; the partial layout is sed_conv_fwd_blocks_bf16's.
SED_API int sed_conv3x3_bf16x3(const float* x, const void* Wp, const float* bias, float* y, float* partial, int B, int T, int F,
                                  int CIN, int COUT, void* stream) {
    return convb_dispatch<false>(x, Wp, bias, y, partial, B, T, F, CIN, COUT, (hipStream_t)stream, ConvBnb());
}

// Data gradient of a training-mode block with the BatchNorm backward folded into its operand staging (see ConvBnb above):
// dz (B,T,F,CIN) = dL/d(xhat), ybn = the block's saved pre-BN conv output, stats = mean | invstd, Wd = the data-gradient pack.
// Writes dx (B,T,F,COUT), dy_out (B,T,F,CIN) = dL/d(conv output) for the weight gradient, and dbias[CIN] = 0.  dy_out must not alias dz.
SED_API int sed_conv3x3_bf16x3_bnbwd(const float* dz, const float* ybn, const float* stats, const float* gamma, const float* dgamma,
                                        const float* dbeta, const void* Wd, float* dx, float* dy_out, float* dbias, int B, int T, int F,
                                        int CIN, int COUT, void* stream) {
    if (!dz || !ybn || !stats || !gamma || !dgamma || !dbeta || !dy_out || dy_out == dz) return SED_ERR_ARG;
    ConvBnb bnb;
    bnb.ybn = ybn; bnb.stats = stats; bnb.gamma = gamma; bnb.dgamma = dgamma; bnb.dbeta = dbeta; bnb.dy_out = dy_out; bnb.dbias = dbias;
    bnb.inv_count = 1.0f / (float)((size_t)B * T * F);
    return convb_dispatch<true>(dz, Wd, nullptr, dx, nullptr, B, T, F, CIN, COUT, (hipStream_t)stream, bnb);
}
