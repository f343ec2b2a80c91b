// gemm.hip -- D = alpha * A * B^T + beta * C on the gfx950 matrix cores, fp64 and fp32.
//
// This one kernel carries the O(n^2 m) terms of the path outside the one-launch factorisation: the candidate
// solve V^T = B^T L^-T, the posterior covariance V^T V, the factor updates and the launch-sequence Cholesky.
// All are "NT" products with both operands contiguous along k, so one staging path serves.
//
// Shape: 128 x 128 output tile per 256-thread workgroup (4 waves as 2 x 2, each wave 64 x 64 =
// 4 x 4 MFMA tiles of 16 x 16).
//   fp64: v_mfma_f64_16x16x4_f64   (C/D: row = (lane>>4) + 4*reg, col = lane&15)
//   fp32: v_mfma_f32_16x16x4_f32   (C/D: row = (lane>>4)*4 + reg, col = lane&15)
// Both take ONE scalar of A and of B per lane (A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15]).
// Because the sum over k is order independent, lane group g = lane>>4 is handed the 16-byte chunk g of the
// row piece instead (2 f64 / 4 f32 consecutive k per chunk): one ds_read_b128 then feeds 2 (f64) or 4 (f32)
// MFMA k-steps, and A and B use the same permutation so products pair up.
//
// The kernel (gemm_nt_kernel_dma4): k advances 64 bytes per row per step (8 f64 / 16 f32); the operand pieces go
// global -> LDS by DMA (global_load_lds_dwordx4, no staging registers, no ds_write pass) into FOUR 16 KB stages,
// three k-tiles in flight while one is multiplied, with hand-placed s_waitcnt vmcnt(8/4/0) + s_barrier per k-tile;
// two workgroups per CU (64 KB LDS, 178 VGPRs each).
// LDS image per operand and stage: [128 rows][4 chunks of 16 B], chunk index XOR ((row>>2)&2).
// Round 1 measured its predecessors against it -- register-staged with one k-tile of prefetch (64.6 TFLOP/s at fp64
// 4096^3), two-stage LDS-DMA, fragment double-buffering, s_setprio around the MFMAs, 5 stages, 3 workgroups per CU --
// all slower (profiles/r01_gemm_ab_f64.txt); they are no longer in the source.
//   fp64 4096^3: 71.0 TFLOP/s (90 % of 78.6); candidate solve in situ 68 TFLOP/s.
//
// Requirements (the library pads every matrix to multiples of 128 with zeros / identity):
//   m % 128 == 0, n % 128 == 0, k % 128 == 0, leading dimensions multiples of 4 elements,
//   16-byte aligned base pointers.  In-place use (D aliasing A) is safe iff n == 128: a
//   workgroup then reads exactly the rows it later overwrites and finishes reading first.
#include "common.h"
#include <hip/hip_ext.h>
#include "mfma.h"
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>

namespace algp {

template <typename T>
struct GemmArgs {
    const T* A;
    const T* B;
    const T* C;
    T* D;
    int64_t lda, ldb, ldc, ldd;
    int64_t sA, sB, sC, sD;        // batch strides in elements (blockIdx.y = batch index)
    int tiles_m, tiles_n, ktiles;
    T alpha, beta;
    int lower_only;
    int ktri;                      // lower_only products of an UPPER-triangular operand with itself (X X^T, X = L^-T): row tile bm of
                                   // X is zero left of column 128 bm, so tile (bm, bn <= bm) sums over k >= 128 bm only
    int kcut;                      // B is LOWER triangular by 128 x 128 tiles (X * inv(L_JJ)^T with an explicit block inverse): column
                                   // tile bn sums over k < 128 (bn + 1) only
    // STATS kernels only (the launches that write a column tile of V^T for the last time): per output row, the sums over
    // the tile's 128 columns of d^2 and of d * stat_w[column] go to stat_out[2 bn + 0 / 1][row] (second index: stat_ld apart)
    const T* stat_w;
    T* stat_out;
    int64_t stat_ld;
};

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

// sum over the 16 lanes of a DPP row (the lanes that share lane >> 4), left in every lane: four rotate-and-add steps on
// the VALU (row_ror 8, 4, 2, 1), no LDS round trip (the same butterfly through __shfl_xor is 8 ds_bpermute per double with
// a wait each: 10 us per output tile in the epilogue below)
template <int CTRL>
__device__ __forceinline__ float dpp_row(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_row(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <typename T>
__device__ __forceinline__ T row16_sum(T x) {
    x += dpp_row<0x128>(x);
    x += dpp_row<0x124>(x);
    x += dpp_row<0x122>(x);
    x += dpp_row<0x121>(x);
    return x;
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA staging with FOUR stages of 64-byte rows (4 x 16 KB, two workgroups per CU) and hand-placed waits, so
// that the loads of THREE k-tiles (8 f64 / 16 f32 wide each) are in flight while one is multiplied.  Per k-tile and wave: s_waitcnt vmcnt(8) (own DMA of this tile landed, two
// younger tiles may still fly), s_barrier (everybody's DMA landed, everybody is done reading the stage
// that is refilled next), 4 global_load_lds for tile kt+3, then 8 ds_read_b128 + 32 MFMAs.
// LDS image per operand and stage: [128 rows][4 chunks of 16 B], chunk index XOR ((row>>2)&2).  A ds_read_b128 is served
// in four groups of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32 (MI355X_MICROARCH.md, LDS), NOT the
// four quarter-waves -- and a group is one LDS cycle when its lanes touch 16 distinct 16-byte slots of the 256-byte bank
// row: with lane = row fr + 16 * chunk fg that holds for this XOR (tools/pmc_lds.sh: SQ_LDS_BANK_CONFLICT 0).  Rounds 1-4
// XORed with (row>>2)&3, conflict-free for quarter-waves and two-way for the real groups: half of all LDS-array cycles were
// conflict cycles (2.4e10 of 4.8e10 in the bench run) -- at no measurable cost in time (164.8-165.2 vs 165.2 ms per step, A/B on
// one box): the LDS array is busy for a tenth of the kernel either way.
// ---------------------------------------------------------------------------------------------
#ifndef ALGP_GEMM_XOR_MASK
#define ALGP_GEMM_XOR_MASK 2                                       // 3: the LDS image of rounds 1-4 (for the traffic A/B of round 6)
#endif
template <typename T, bool STATS = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel_dma4(GemmArgs<T> g) {
    constexpr int NST = 4;
    using F = MF<T>;
    using acc_t = typename F::acc_t;
    using chunk_t = typename F::chunk_t;
    constexpr int EPC = F::EPC;
    constexpr int BK = 4 * EPC;                                    // elements per 64-byte row piece

    __shared__ __attribute__((aligned(1024))) char smem[NST * 16384];

    const int nwg = gridDim.x;
    int bm, bn;
    if (g.ktri) {
        // Tile row bm costs (bm + 1) tiles x K = k - 128 bm: a contiguous share of the tile list per XCD (below) would give the
        // XCD with the first rows several times the work of the last.  Instead XCD x takes the tile rows bm = x, x + 8, ...
        // (its rows' tiles share the row panel of X through its L2), longest K first; the grid holds 8 x the largest share
        // and the surplus workgroups of the other XCDs leave at once.
        int local = blockIdx.x >> 3;
        bm = blockIdx.x & 7;
        while (bm < g.tiles_m && local > bm) { local -= bm + 1; bm += 8; }
        if (bm >= g.tiles_m) return;
        bn = local;
    } else {
        int sid;
        {
            const int id = blockIdx.x, xcd = id & 7, q = nwg >> 3, r = nwg & 7;
            sid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        }
        if (g.lower_only) {
            bm = (int)((sqrt(8.0 * (double)sid + 1.0) - 1.0) * 0.5);
            while ((int64_t)(bm + 1) * (bm + 2) / 2 <= sid) ++bm;
            while ((int64_t)bm * (bm + 1) / 2 > sid) --bm;
            bn = sid - (int)((int64_t)bm * (bm + 1) / 2);
        } else {
            bm = sid / g.tiles_n;
            bn = sid - bm * g.tiles_n;
        }
    }
    const int64_t m0 = (int64_t)bm * 128, n0 = (int64_t)bn * 128;
    const int kskip = g.ktri ? bm * (128 / (4 * MF<T>::EPC)) : 0;   // 64-byte k-tiles this output tile leaves out
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t bz = blockIdx.y;

    // ---- DMA: wave w stages rows [32w, 32w+32) of each operand as two 16-row groups; lane l -> row l>>2 of
    // the group, LDS slot l&3, which must hold chunk (l&3) ^ ((row>>2)&2) = (l&3) ^ ((l>>4)&2)
    const int srow = lane >> 2;
    const int schunk = (lane & 3) ^ ((lane >> 4) & ALGP_GEMM_XOR_MASK);
    const T* Ag = g.A + bz * g.sA + (m0 + 32 * wave + srow) * g.lda + schunk * EPC + (int64_t)kskip * BK;
    const T* Bg = g.B + bz * g.sB + (n0 + 32 * wave + srow) * g.ldb + schunk * EPC + (int64_t)kskip * BK;
    auto stage = [&](int st, int kt) {
        char* As = smem + st * 16384 + wave * 2048;
        char* Bs = As + 8192;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((glb_vp)(Ag + (int64_t)(16 * i) * g.lda + (int64_t)kt * BK),
                                             (lds_vp)(As + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_vp)(Bg + (int64_t)(16 * i) * g.ldb + (int64_t)kt * BK),
                                             (lds_vp)(Bs + i * 1024), 16, 0, 0);
        }
    };

    // ---- fragment reads: row = w*64 + t*16 + (lane&15), chunk (lane>>4) ^ ((row>>2)&2) ----
    const int fr = lane & 15, fg = lane >> 4;
    const int coff = ((fg ^ ((fr >> 2) & ALGP_GEMM_XOR_MASK)) << 4);   // (row>>2)&2 == (fr>>2)&2: the row offsets are multiples of 16
    const int aoff = (wr * 64 + fr) * 64 + coff;
    const int boff = (wc * 64 + fr) * 64 + coff;

    acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;

    int nkt = g.ktiles * 2 - kskip;                                // g.ktiles counts 128-byte tiles
    if (g.kcut && (bn + 1) * (128 / BK) < nkt) nkt = (bn + 1) * (128 / BK);
    // prologue: tiles 0 .. NST-2 in flight
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nkt) stage(t, t);

    auto fread = [&](int st, chunk_t (&a)[4], chunk_t (&b)[4]) {
        const char* As = smem + st * 16384;
        const char* Bs = As + 8192;
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const chunk_t*>(As + aoff + t * 1024);
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const chunk_t*>(Bs + boff + t * 1024);
    };
    // column by column: the first four MFMAs need a[0..3] and b[0] only, the rest of b lands under them
    auto fmac = [&](const chunk_t (&a)[4], const chunk_t (&b)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < EPC; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = F::mfma(a[i][e], b[j][e], acc[i][j]);
    };
    // wait until this wave's DMA of a tile has landed while `younger` (0..2) later tiles may still fly, then meet
    auto arrive = [&](int younger) {
        if (younger >= 2) __builtin_amdgcn_s_waitcnt(0x0F78);                   // vmcnt(8)
        else if (younger >= 1) __builtin_amdgcn_s_waitcnt(0x0F74);              // vmcnt(4)
        else __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    int st = 0;                                                    // stage of tile kt; tile kt+NST-1 goes to st-1 (mod NST)
    for (int kt = 0; kt < nkt; ++kt) {
        arrive(nkt - 1 - kt);
        chunk_t a[4], b[4];
        fread(st, a, b);
        if (kt + NST - 1 < nkt) stage(st == 0 ? NST - 1 : st - 1, kt + NST - 1);   // the stage read in iteration kt-1
        fmac(a, b);
        st = (st + 1 == NST) ? 0 : st + 1;
    }

    const T alpha = g.alpha, beta = g.beta;
    const T* Cb = g.C + bz * g.sC;
    T* Db = g.D + bz * g.sD;
    if (beta != (T)0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            T cv[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gi = m0 + wr * 64 + i * 16 + F::row_of(lane, r);
#pragma unroll
                for (int j = 0; j < 4; ++j) cv[r][j] = Cb[gi * g.ldc + n0 + wc * 64 + j * 16 + fr];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gi = m0 + wr * 64 + i * 16 + F::row_of(lane, r);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    Db[gi * g.ldd + n0 + wc * 64 + j * 16 + fr] = alpha * acc[i][j][r] + beta * cv[r][j];
            }
        }
    } else if (!STATS) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gi = m0 + wr * 64 + i * 16 + F::row_of(lane, r);
#pragma unroll
                for (int j = 0; j < 4; ++j) Db[gi * g.ldd + n0 + wc * 64 + j * 16 + fr] = alpha * acc[i][j][r];
            }
    } else {
        // The tile goes out and, with it, its row statistics (the candidate solve's variance and mean, reference
        // utils.py:301-304: a second pass over V^T otherwise -- 8 GB at config 4).  A lane holds, of each of its 16 rows, the
        // four columns fr + 16 j of this wave's half: sum them, fold the 16 lanes of a row (DPP rotations inside the
        // 16-lane group), add the two column halves through LDS (free now), one store per row and statistic.  Fixed order:
        // the same bits in every run.  (Row by row, stores and sums together: kept for a second loop the 64 products
        // alpha * acc cost 20 spilled VGPRs.)
        T wv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[j] = g.stat_w[n0 + wc * 64 + j * 16 + fr];
        T* red = reinterpret_cast<T*>(smem);                       // [column half][row][2]
        __syncthreads();                                           // every wave is done with the last stage
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr * 64 + i * 16 + F::row_of(lane, r);
                T s2 = (T)0, sw = (T)0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const T d = alpha * acc[i][j][r];
                    Db[(m0 + row) * g.ldd + n0 + wc * 64 + j * 16 + fr] = d;
                    s2 += d * d;
                    sw += d * wv[j];
                }
                s2 = row16_sum<T>(s2);
                sw = row16_sum<T>(sw);
                if (fr == 0) {
                    red[(wc * 128 + row) * 2 + 0] = s2;
                    red[(wc * 128 + row) * 2 + 1] = sw;
                }
            }
        __syncthreads();
        const int row = tid >> 1, q = tid & 1;
        g.stat_out[(int64_t)(2 * bn + q) * g.stat_ld + m0 + row] = red[row * 2 + q] + red[(128 + row) * 2 + q];
    }
}


#ifndef ALGP_GEMM_T256
#define ALGP_GEMM_T256 0
#endif
#if ALGP_GEMM_T256
// ---------------------------------------------------------------------------------------------
// A/B form (round 6, EXPERIMENTS.md): 256 x 128 output tile per 512-thread workgroup, 8 waves as 4 x 2 with the per-wave
// tile unchanged (64 x 64), so the B stage is shared by twice the rows: 24 KB per 64-byte k-tile instead of 2 x 16 KB =
// 21.3 flop per L2->LDS byte instead of 16.  Four stages of 24 KB (96 KB, dynamic LDS), ONE workgroup per CU, 3 DMA
// instructions per thread and k-tile (vmcnt 6 / 3 / 0).  ALGP_GEMM_T256 == 2: waves 4-7 (the SIMD partners of waves
// 0-3) run half a k-tile behind -- after the barrier of tile kt they first finish tile kt-1's second 16 products from the
// fragments they hold, then read tile kt -- so that the two waves of a SIMD do not read LDS and wait in lockstep
// (MI355X_MICROARCH.md, "two waves that run the same program with one barrier per block: try a stagger").
// m need only be a multiple of 128: the last tile row may be half empty (its waves 4-7 re-read rows 0-127 and store nothing).
// ---------------------------------------------------------------------------------------------
template <typename T, bool STATS, bool STAGGER>
__global__ __launch_bounds__(512, 1) void gemm_nt_kernel_t256(GemmArgs<T> g) {
    constexpr int NST = 4, STB = 24576;
    using F = MF<T>;
    using acc_t = typename F::acc_t;
    using chunk_t = typename F::chunk_t;
    constexpr int EPC = F::EPC;
    constexpr int BK = 4 * EPC;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int nwg = gridDim.x;
    int sid;
    {
        const int id = blockIdx.x, xcd = id & 7, q = nwg >> 3, r = nwg & 7;
        sid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int bm = sid / g.tiles_n, bn = sid - bm * g.tiles_n;
    const int64_t m0 = (int64_t)bm * 256, n0 = (int64_t)bn * 128;
    const int rows_valid = (bm * 2 + 1 < g.tiles_m) ? 256 : 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const bool late = STAGGER && wave >= 4;

    const int srow = lane >> 2;
    const int schunk = (lane & 3) ^ ((lane >> 4) & 2);
    const int arow0 = (32 * wave >= rows_valid) ? 32 * wave - 128 : 32 * wave;
    const T* Ag = g.A + (m0 + arow0 + srow) * g.lda + schunk * EPC;
    const T* Bg = g.B + (n0 + 16 * wave + srow) * g.ldb + schunk * EPC;
    auto stage = [&](int st, int kt) {
        char* As = smem + st * STB + wave * 2048;
        char* Bs = smem + st * STB + 16384 + wave * 1024;
        __builtin_amdgcn_global_load_lds((glb_vp)(Ag + (int64_t)kt * BK), (lds_vp)(As), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_vp)(Ag + (int64_t)16 * g.lda + (int64_t)kt * BK), (lds_vp)(As + 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_vp)(Bg + (int64_t)kt * BK), (lds_vp)(Bs), 16, 0, 0);
    };
    const int fr = lane & 15, fg = lane >> 4;
    const int coff = ((fg ^ ((fr >> 2) & 2)) << 4);
    const int aoff = (wr * 64 + fr) * 64 + coff;
    const int boff = 16384 + (wc * 64 + fr) * 64 + coff;

    acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;

    int nkt = g.ktiles * 2;
    if (g.kcut && (bn + 1) * (128 / BK) < nkt) nkt = (bn + 1) * (128 / BK);
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nkt) stage(t, t);

    auto fread = [&](int st, chunk_t (&a)[4], chunk_t (&b)[4]) {
        const char* base = smem + st * STB;
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const chunk_t*>(base + aoff + t * 1024);
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const chunk_t*>(base + boff + t * 1024);
    };
    auto fmac_cols = [&](const chunk_t (&a)[4], const chunk_t (&b)[4], int j0) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < EPC; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j0 + jj] = F::mfma(a[i][e], b[j0 + jj][e], acc[i][j0 + jj]);
    };
    auto arrive = [&](int younger) {
        if (younger >= 2) __builtin_amdgcn_s_waitcnt(0x0F76);                   // vmcnt(6)
        else if (younger >= 1) __builtin_amdgcn_s_waitcnt(0x0F73);              // vmcnt(3)
        else __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    int st = 0;
    chunk_t a[4], b[4];
    if (!late) {
        for (int kt = 0; kt < nkt; ++kt) {
            arrive(nkt - 1 - kt);
            fread(st, a, b);
            if (kt + NST - 1 < nkt) stage(st == 0 ? NST - 1 : st - 1, kt + NST - 1);
            fmac_cols(a, b, 0);
            fmac_cols(a, b, 2);
            st = (st + 1 == NST) ? 0 : st + 1;
        }
    } else {
        for (int kt = 0; kt < nkt; ++kt) {
            arrive(nkt - 1 - kt);
            if (kt + NST - 1 < nkt) stage(st == 0 ? NST - 1 : st - 1, kt + NST - 1);
            if (kt > 0) fmac_cols(a, b, 2);                        // tile kt-1's second half, from registers
            fread(st, a, b);
            fmac_cols(a, b, 0);
            st = (st + 1 == NST) ? 0 : st + 1;
        }
        if (nkt > 0) fmac_cols(a, b, 2);
    }

    const T alpha = g.alpha, beta = g.beta;
    const T* Cb = g.C;
    T* Db = g.D;
    const bool live = wr * 64 < rows_valid;                        // wave-uniform
    if (beta != (T)0) {
        if (live) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                T cv[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t gi = m0 + wr * 64 + i * 16 + F::row_of(lane, r);
#pragma unroll
                    for (int j = 0; j < 4; ++j) cv[r][j] = Cb[gi * g.ldc + n0 + wc * 64 + j * 16 + fr];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t gi = m0 + wr * 64 + i * 16 + F::row_of(lane, r);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        Db[gi * g.ldd + n0 + wc * 64 + j * 16 + fr] = alpha * acc[i][j][r] + beta * cv[r][j];
                }
            }
        }
    } else if (!STATS) {
        if (live) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t gi = m0 + wr * 64 + i * 16 + F::row_of(lane, r);
#pragma unroll
                    for (int j = 0; j < 4; ++j) Db[gi * g.ldd + n0 + wc * 64 + j * 16 + fr] = alpha * acc[i][j][r];
                }
        }
    } else {
        T wv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[j] = g.stat_w[n0 + wc * 64 + j * 16 + fr];
        T* red = reinterpret_cast<T*>(smem);                       // [column half][row][2]
        __syncthreads();
        if (live) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wr * 64 + i * 16 + F::row_of(lane, r);
                    T s2 = (T)0, sw = (T)0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const T d = alpha * acc[i][j][r];
                        Db[(m0 + row) * g.ldd + n0 + wc * 64 + j * 16 + fr] = d;
                        s2 += d * d;
                        sw += d * wv[j];
                    }
                    s2 = row16_sum<T>(s2);
                    sw = row16_sum<T>(sw);
                    if (fr == 0) {
                        red[(wc * 256 + row) * 2 + 0] = s2;
                        red[(wc * 256 + row) * 2 + 1] = sw;
                    }
                }
        }
        __syncthreads();
        const int row = tid >> 1, q = tid & 1;
        if (row < rows_valid) g.stat_out[(int64_t)(2 * bn + q) * g.stat_ld + m0 + row] = red[row * 2 + q] + red[(256 + row) * 2 + q];
    }
}
#endif


#ifndef ALGP_GEMM_K96
#define ALGP_GEMM_K96 0
#endif
#if ALGP_GEMM_K96
// ---------------------------------------------------------------------------------------------
// A/B form (round 6, EXPERIMENTS.md): the dma4 workgroup (128 x 128, 4 waves, two per CU) with 96-byte k-tiles in THREE
// stages of 24 KB (72 KB): 48 MFMAs per barrier instead of 32 with the same 192 bytes per row in flight behind the tile
// being multiplied (the 128-byte / two-stage form of round 5 had 128).  A row piece is 6 chunks of 16 bytes; lane group g
// takes chunk g whole (ds_read_b128) and the g&1 half of chunk 4 + (g>>1) (ds_read_b64): 3 (f64) / 6 (f32) k-steps per
// lane group and tile, A and B permuted alike.  LDS image [128 rows][96 B], no swizzle (conflict-free for the b128 groups).
// K (a multiple of 128 elements = 1024 bytes) leaves 0, 32 or 64 bytes over: the FIRST tile is the partial one -- it loads
// bytes [0, 96) of each row like any other, the A fragments beyond the remainder are zeroed, and tile 1 starts at the remainder.
// ---------------------------------------------------------------------------------------------
template <typename T, bool STATS>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel_k96(GemmArgs<T> g) {
    constexpr int NST = 3, STB = 24576, OPB = 12288;
    using F = MF<T>;
    using acc_t = typename F::acc_t;
    using chunk_t = typename F::chunk_t;
    constexpr int EPC = F::EPC, HPC = EPC / 2;
    typedef T half_t __attribute__((ext_vector_type(HPC)));
    constexpr int TE = 6 * EPC;                                    // elements per 96-byte row piece
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int nwg = gridDim.x;
    int sid;
    {
        const int id = blockIdx.x, xcd = id & 7, q = nwg >> 3, r = nwg & 7;
        sid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int bm = sid / g.tiles_n, bn = sid - bm * g.tiles_n;
    const int64_t m0 = (int64_t)bm * 128, n0 = (int64_t)bn * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // ---- DMA: LDS slot L = 256 i + tid (i = 0..2) of each operand holds chunk L % 6 of row L / 6
    const T* Ag[3];
    const T* Bg[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int L = 256 * i + tid, row = L / 6, ch = L - 6 * row;
        Ag[i] = g.A + (m0 + row) * g.lda + ch * EPC;
        Bg[i] = g.B + (n0 + row) * g.ldb + ch * EPC;
    }
    int nk = g.ktiles * 8 * EPC;                                   // g.ktiles counts 128-byte tiles
    if (g.kcut && (bn + 1) * 128 < nk) nk = (bn + 1) * 128;
    const int rem = nk % TE;                                       // 0, 4 EPC or 2 EPC elements
    const int nkt = nk / TE + (rem ? 1 : 0);
    const int shift = rem ? TE - rem : 0;
    auto stage = [&](int st, int kt) {
        const int64_t off = kt ? (int64_t)kt * TE - shift : 0;
        char* As = smem + st * STB + wave * 1024;
        char* Bs = As + OPB;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            __builtin_amdgcn_global_load_lds((glb_vp)(Ag[i] + off), (lds_vp)(As + i * 4096), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_vp)(Bg[i] + off), (lds_vp)(Bs + i * 4096), 16, 0, 0);
        }
    };
    const int fr = lane & 15, fg = lane >> 4;
    const int aoff = (wr * 64 + fr) * 96, boff = OPB + (wc * 64 + fr) * 96;
    const int c16 = fg * 16, h8 = 64 + fg * 8;

    acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;

#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nkt) stage(t, t);

    int st = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) __builtin_amdgcn_s_waitcnt(0x0F76);      // vmcnt(6): this tile landed, the next may fly
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* base = smem + st * STB;
        chunk_t ac[4], bc[4];
        half_t ah[4], bh[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) ac[t] = *reinterpret_cast<const chunk_t*>(base + aoff + t * 1536 + c16);
#pragma unroll
        for (int t = 0; t < 4; ++t) bc[t] = *reinterpret_cast<const chunk_t*>(base + boff + t * 1536 + c16);
#pragma unroll
        for (int t = 0; t < 4; ++t) ah[t] = *reinterpret_cast<const half_t*>(base + aoff + t * 1536 + h8);
#pragma unroll
        for (int t = 0; t < 4; ++t) bh[t] = *reinterpret_cast<const half_t*>(base + boff + t * 1536 + h8);
        if (kt + NST - 1 < nkt) stage(st == 0 ? NST - 1 : st - 1, kt + NST - 1);   // the stage read in iteration kt-1
        if (kt == 0 && rem) {                                      // wave-uniform: the partial tile
            const T keep = ((fg + 1) * EPC <= rem) ? (T)1 : (T)0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) ac[t][e] *= keep;
#pragma unroll
                for (int e = 0; e < HPC; ++e) ah[t][e] = (T)0;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int e = 0; e < EPC; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = F::mfma(ac[i][e], bc[j][e], acc[i][j]);
#pragma unroll
            for (int e = 0; e < HPC; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = F::mfma(ah[i][e], bh[j][e], acc[i][j]);
        }
        st = (st + 1 == NST) ? 0 : st + 1;
    }

    const T alpha = g.alpha, beta = g.beta;
    const T* Cb = g.C;
    T* Db = g.D;
    if (beta != (T)0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            T cv[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gi = m0 + wr * 64 + i * 16 + F::row_of(lane, r);
#pragma unroll
                for (int j = 0; j < 4; ++j) cv[r][j] = Cb[gi * g.ldc + n0 + wc * 64 + j * 16 + fr];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gi = m0 + wr * 64 + i * 16 + F::row_of(lane, r);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    Db[gi * g.ldd + n0 + wc * 64 + j * 16 + fr] = alpha * acc[i][j][r] + beta * cv[r][j];
            }
        }
    } else if (!STATS) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gi = m0 + wr * 64 + i * 16 + F::row_of(lane, r);
#pragma unroll
                for (int j = 0; j < 4; ++j) Db[gi * g.ldd + n0 + wc * 64 + j * 16 + fr] = alpha * acc[i][j][r];
            }
    } else {
        T wv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[j] = g.stat_w[n0 + wc * 64 + j * 16 + fr];
        T* red = reinterpret_cast<T*>(smem);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr * 64 + i * 16 + F::row_of(lane, r);
                T s2 = (T)0, sw = (T)0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const T d = alpha * acc[i][j][r];
                    Db[(m0 + row) * g.ldd + n0 + wc * 64 + j * 16 + fr] = d;
                    s2 += d * d;
                    sw += d * wv[j];
                }
                s2 = row16_sum<T>(s2);
                sw = row16_sum<T>(sw);
                if (fr == 0) {
                    red[(wc * 128 + row) * 2 + 0] = s2;
                    red[(wc * 128 + row) * 2 + 1] = sw;
                }
            }
        __syncthreads();
        const int row = tid >> 1, q = tid & 1;
        g.stat_out[(int64_t)(2 * bn + q) * g.stat_ld + m0 + row] = red[row * 2 + q] + red[(128 + row) * 2 + q];
    }
}
#endif


#ifndef ALGP_GEMM_W4
#define ALGP_GEMM_W4 0
#endif
#if ALGP_GEMM_W4
// ---------------------------------------------------------------------------------------------
// gemm_nt_kernel_w4 (round 6): 256 x 128 output tile per 256-thread workgroup, ONE workgroup per CU, one wave per SIMD,
// each wave 128 x 64 = 8 x 4 MFMA tiles (256 accumulator registers: the AccVGPR half of the wave's 512).  The shape of the
// vendor's best fp64 kernel on this part (rocBLAS's MT128x256x16, 97 % of peak at 8192^3: tools/rocblas_yardstick.cpp),
// with this library's operand path: 128-byte k-tiles by LDS-DMA into THREE stages of 48 KB (A 256 rows + B 128 rows), two
// tiles in flight behind the one being multiplied, no staging registers and no ds_write pass.
//   * 128 MFMAs per wave between barriers (the dma4 kernel: 32), and the barrier sits INSIDE the MFMA stream: the products
//     of the second half of tile kt-1 (fragments in registers) are issued around it, so the matrix pipe has work while the
//     wave waits for its siblings and for its own DMA;
//   * every LDS read is issued behind an MFMA (one per product for the first 12 products of each half-tile) and lands under
//     the 64-cycle products in front of it -- no read burst + wait in front of the products; the 12 DMA instructions of
//     tile kt+2 likewise;
//   * 12 ds_read_b128 per 64 MFMAs (dma4: 8 per 32).
// LDS image per operand and stage: [rows][8 chunks of 16 B], chunk index XOR ((row >> 1) & 7): conflict-free for the 16-lane
// groups a ds_read_b128 is served in.  Lane group g = lane >> 4 takes chunk 4 h + g of half h: 2 (f64) / 4 (f32) k-steps.
// m a multiple of 128 (a last half tile row is masked), n and k multiples of 128; not for lower_only / ktri / batches.
// ---------------------------------------------------------------------------------------------
// the product with its accumulator updated IN PLACE in AccVGPRs ("+a"): as a builtin the three-address form lets the register
// allocator rotate the 32 accumulator tiles between loop iterations (224 v_accvgpr_mov per k-tile in the first build)
// (INV: the tile lives in arch VGPRs -- the last two tile rows do, so that the allocator has 64 AccVGPRs to spare: with all 256
// taken it still swapped a dozen tiles back and forth across the loop's back-edge)
template <bool INV>
__device__ __forceinline__ void w4_mfma(double a, double b, v4d& c) {
    if constexpr (INV) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <bool INV>
__device__ __forceinline__ void w4_mfma(float a, float b, v4f& c) {
    if constexpr (INV) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <typename T, bool STATS>
__global__ __launch_bounds__(256, 1) void gemm_nt_kernel_w4(GemmArgs<T> g) {
    constexpr int NST = 3, STB = 49152, BOFF = 32768;
    using F = MF<T>;
    using acc_t = typename F::acc_t;
    using chunk_t = typename F::chunk_t;
    constexpr int EPC = F::EPC;
    constexpr int BK = 8 * EPC;                                    // elements per 128-byte row piece
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int nwg = gridDim.x;
    int sid;
    {
        const int id = blockIdx.x, xcd = id & 7, q = nwg >> 3, r = nwg & 7;
        sid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int bm = sid / g.tiles_n, bn = sid - bm * g.tiles_n;
    const int64_t m0 = (int64_t)bm * 256, n0 = (int64_t)bn * 128;
    const int rows_valid = (bm * 2 + 1 < g.tiles_m) ? 256 : 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // ---- DMA: an instruction of a wave covers 8 rows x 128 bytes; lane l -> row l >> 3, LDS slot l & 7, which holds chunk (l & 7) ^ ((row >> 1) & 7)
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ ((srow >> 1) & 7);             // the rows of a group start at multiples of 8: (row >> 1) & 7 = ((srow >> 1) + 4 * i) & 7, see below
    const int arow0 = (64 * wave >= rows_valid) ? 64 * wave - 128 : 64 * wave;
    const T* Ag = g.A + (m0 + arow0 + srow) * g.lda;
    const T* Bg = g.B + (n0 + 32 * wave + srow) * g.ldb;
    int nkt = g.ktiles;                                            // 128-byte tiles
    if (g.kcut && (bn + 1) * (128 / BK) < nkt) nkt = (bn + 1) * (128 / BK);
    // row group i (8 rows) of the wave's share: rows 8 i .. 8 i + 7 -> (row >> 1) & 7 = (srow >> 1) ^ (4 * (i & 1)): i odd flips bit 2
    auto stage = [&](int st, int kt) {
        const int kk = kt < nkt ? kt : nkt - 1;                    // past the end: re-load the last tile (never read), the counts stay uniform
        char* As = smem + st * STB + wave * 8192;
        char* Bs = smem + st * STB + BOFF + wave * 4096;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((glb_vp)(Ag + (int64_t)(8 * i) * g.lda + (int64_t)kk * BK + (schunk ^ (4 * (i & 1))) * EPC),
                                             (lds_vp)(As + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_vp)(Bg + (int64_t)(8 * i) * g.ldb + (int64_t)kk * BK + (schunk ^ (4 * (i & 1))) * EPC),
                                             (lds_vp)(Bs + i * 1024), 16, 0, 0);
    };
    // ---- fragment reads: row = w * (128 | 64) + t * 16 + fr, chunk (4 h + fg) ^ ((row >> 1) & 7) with (row >> 1) & 7 = (fr >> 1) & 7
    const int fr = lane & 15, fg = lane >> 4;
    const int sw = (fr >> 1) & 7;
    const int aoff0 = (wr * 128 + fr) * 128 + ((fg ^ sw) << 4), aoff1 = (wr * 128 + fr) * 128 + (((4 + fg) ^ sw) << 4);
    const int boff0 = BOFF + (wc * 64 + fr) * 128 + ((fg ^ sw) << 4), boff1 = BOFF + (wc * 64 + fr) * 128 + (((4 + fg) ^ sw) << 4);

    acc_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;

    chunk_t a0[8], b0[4], a1[8], b1[4];
    // one product of the half-tile held in (a, b): k-step e of tile (i, j)
#ifndef W4_VROWS
#define W4_VROWS 8                                                 // tile rows >= this keep their accumulators in arch VGPRs (8: none)
#endif
#define W4_MFMA(a, b, q)                                                                                   \
    {                                                                                                      \
        const int e_ = (q) / 32, i_ = ((q) % 32) / 4, j_ = (q) % 4;                                        \
        if (i_ >= W4_VROWS) w4_mfma<true>(a[i_][e_ % EPC], b[j_][e_ % EPC], acc[i_][j_]);                  \
        else w4_mfma<false>(a[i_][e_ % EPC], b[j_][e_ % EPC], acc[i_][j_]);                                \
    }

    // ---- prologue: tiles 0 and 1 in flight; tile 0 landed; its first half into set 0; tile 2 on its way
    stage(0, 0);
    stage(1, 1);
    __builtin_amdgcn_s_waitcnt(0x0F7C);                            // vmcnt(12): tile 0 landed, tile 1 may fly
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 8; ++t) a0[t] = *reinterpret_cast<const chunk_t*>(smem + aoff0 + t * 2048);
#pragma unroll
    for (int t = 0; t < 4; ++t) b0[t] = *reinterpret_cast<const chunk_t*>(smem + boff0 + t * 2048);
    stage(2, 2);

    // phase B(kt): the products of tile kt's first half (set 0); its second half is read into set 1 behind them
#define W4_PHASE_B(stg)                                                                                    \
    {                                                                                                      \
        const char* base = smem + (stg) * STB;                                                             \
        _Pragma("unroll") for (int q = 0; q < 32 * EPC; ++q) {                                             \
            W4_MFMA(a0, b0, q);                                                                            \
            if (q < 8) a1[q] = *reinterpret_cast<const chunk_t*>(base + aoff1 + q * 2048);                 \
            else if (q < 12) b1[q - 8] = *reinterpret_cast<const chunk_t*>(base + boff1 + (q - 8) * 2048); \
            __builtin_amdgcn_sched_barrier(0);                                                             \
        }                                                                                                  \
    }
    int st = 0;                                                    // stage of the tile whose first half sits in set 0
    // nkt uniform iterations: the last one's hand-over waits for, reads and re-loads tiles past the end (stage() clamps them
    // to the last tile; what is read is never multiplied) -- ONE loop body, no peeled copies whose accumulators the register
    // allocator would have to reconcile with the loop's
    for (int kt = 0; kt < nkt; ++kt) {
        W4_PHASE_B(st);
        // ---- phase A(kt + 1): the products of tile kt's second half (set 1) with the hand-over in their middle: tile kt + 1 has
        // landed everywhere and everybody is done reading tile kt -> its stage takes tile kt + 3; tile kt + 1's first half into set 0
        const int st1 = (st + 1 == NST) ? 0 : st + 1;
        {
            const char* base = smem + st1 * STB;
#pragma unroll
            for (int q = 0; q < 32 * EPC; ++q) {
                W4_MFMA(a1, b1, q);
                if (q == 3) {
                    __builtin_amdgcn_s_waitcnt(0x0F7C);            // vmcnt(12): this wave's DMA of tile kt + 1 landed (tile kt + 2 may fly)
                    __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): this wave's reads of tile kt are done
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                if (q >= 4 && q < 12) a0[q - 4] = *reinterpret_cast<const chunk_t*>(base + aoff0 + (q - 4) * 2048);
                else if (q >= 12 && q < 16) b0[q - 12] = *reinterpret_cast<const chunk_t*>(base + boff0 + (q - 12) * 2048);
                __builtin_amdgcn_sched_barrier(0);
                if (q == 16) stage(st, kt + 3);
            }
        }
        st = st1;
    }
#undef W4_PHASE_B
#undef W4_MFMA
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // the surplus DMA of the last two iterations
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // matrix-pipe write -> VALU read of the accumulators (the asm products hide the hazard from the compiler)

    const T alpha = g.alpha, beta = g.beta;
    const T* Cb = g.C;
    T* Db = g.D;
    const bool live = wr * 128 < rows_valid;                       // wave-uniform
    if (beta != (T)0) {
        if (live) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                T cv[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t gi = m0 + wr * 128 + i * 16 + F::row_of(lane, r);
#pragma unroll
                    for (int j = 0; j < 4; ++j) cv[r][j] = Cb[gi * g.ldc + n0 + wc * 64 + j * 16 + fr];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t gi = m0 + wr * 128 + i * 16 + F::row_of(lane, r);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        Db[gi * g.ldd + n0 + wc * 64 + j * 16 + fr] = alpha * acc[i][j][r] + beta * cv[r][j];
                }
            }
        }
    } else if (!STATS) {
        if (live) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t gi = m0 + wr * 128 + i * 16 + F::row_of(lane, r);
#pragma unroll
                    for (int j = 0; j < 4; ++j) Db[gi * g.ldd + n0 + wc * 64 + j * 16 + fr] = alpha * acc[i][j][r];
                }
        }
    } else {
        T wv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[j] = g.stat_w[n0 + wc * 64 + j * 16 + fr];
        T* red = reinterpret_cast<T*>(smem);                       // [column half][row][2]
        __syncthreads();
        if (live) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wr * 128 + i * 16 + F::row_of(lane, r);
                    T s2 = (T)0, sw2 = (T)0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const T d = alpha * acc[i][j][r];
                        Db[(m0 + row) * g.ldd + n0 + wc * 64 + j * 16 + fr] = d;
                        s2 += d * d;
                        sw2 += d * wv[j];
                    }
                    s2 = row16_sum<T>(s2);
                    sw2 = row16_sum<T>(sw2);
                    if (fr == 0) {
                        red[(wc * 256 + row) * 2 + 0] = s2;
                        red[(wc * 256 + row) * 2 + 1] = sw2;
                    }
                }
        }
        __syncthreads();
        if (tid < rows_valid) {
            g.stat_out[(int64_t)(2 * bn + 0) * g.stat_ld + m0 + tid] = red[tid * 2 + 0] + red[(256 + tid) * 2 + 0];
            g.stat_out[(int64_t)(2 * bn + 1) * g.stat_ld + m0 + tid] = red[tid * 2 + 1] + red[(256 + tid) * 2 + 1];
        }
    }
}
#endif

template <typename T>
int gemm_nt_launch_batched(algp_ctx* c, int klass, int64_t m, int64_t n, int64_t k, T alpha, const T* A, int64_t lda,
                           int64_t sA, const T* B, int64_t ldb, int64_t sB, T beta, const T* C, int64_t ldc, int64_t sC,
                           T* D, int64_t ldd, int64_t sD, int lower_only, int batch, int ktri, const T* stat_w, T* stat_out,
                           int64_t stat_ld, int kcut) {
    if (m <= 0 || n <= 0 || batch <= 0) return ALGP_OK;
    if (stat_out && (beta != (T)0 || batch != 1 || lower_only || ktri))
        return fail(c, ALGP_ERR_BAD_ARG, "gemm_nt: row statistics need beta = 0, no batch");
    if (kcut && (k != n || lower_only || ktri)) return fail(c, ALGP_ERR_BAD_ARG, "gemm_nt: kcut needs k == n");
    if (m % 128 || n % 128 || k % 128 || k <= 0 || lda % 4 || ldb % 4 || sA % 4 || sB % 4)
        return fail(c, ALGP_ERR_BAD_ARG, "gemm_nt: operands must be padded to multiples of 128");
    if (lower_only && m != n) return fail(c, ALGP_ERR_BAD_ARG, "gemm_nt: lower_only needs a square output");
    if (batch > 65535) return fail(c, ALGP_ERR_BAD_ARG, "gemm_nt: batch too large");
    GemmArgs<T> g;
    g.A = A; g.B = B; g.C = C ? C : D; g.D = D;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldd = ldd;
    g.sA = sA; g.sB = sB; g.sC = sC; g.sD = sD;
    g.tiles_m = (int)(m / 128);
    g.tiles_n = (int)(n / 128);
    g.ktiles = (int)(k / (8 * MF<T>::EPC));
    g.alpha = alpha; g.beta = beta;
    g.lower_only = lower_only;
    g.ktri = ktri;
    g.stat_w = stat_w;
    g.stat_out = stat_out;
    g.stat_ld = stat_ld;
    g.kcut = kcut;
    if (ktri && (!lower_only || k != m)) return fail(c, ALGP_ERR_BAD_ARG, "gemm_nt: ktri needs a square lower-only product with k == m");
    const int64_t tiles = lower_only ? (int64_t)g.tiles_m * (g.tiles_m + 1) / 2 : (int64_t)g.tiles_m * g.tiles_n;
    if (tiles > 0x7fffffff) return fail(c, ALGP_ERR_BAD_ARG, "gemm_nt: grid too large");
    // ktri: tile row bm holds bm + 1 tiles of K = k - 128 bm: sum_bm (bm + 1)(tm - bm) = tm (tm + 1)(tm + 2) / 6 tile-steps of 128
    const double tm = (double)g.tiles_m;
    const double flops = ktri ? 2.0 * 128.0 * 128.0 * 128.0 * tm * (tm + 1.0) * (tm + 2.0) / 6.0 * batch
                         : kcut ? 2.0 * 128.0 * 128.0 * 128.0 * tm * (double)g.tiles_n * (g.tiles_n + 1.0) / 2.0 * batch
                              : 2.0 * 128.0 * 128.0 * (double)k * (double)tiles * batch;
    const double bytes = sizeof(T) * batch * ((double)tiles * 128.0 * 128.0 * (beta != (T)0 ? 2.0 : 1.0) +
                                              (double)k * 128.0 * (double)(g.tiles_m + g.tiles_n));
    {
        // $ALGP_LAUNCH_LOG=<file>: one line per GEMM launch, in enqueue order (class m n k lower_only batch ktri element-size kcut): joined with a
        // rocprofv3 kernel trace by dispatch order, it gives the trace the K its grid sizes do not show (tools/trace_shapes.py)
        static FILE* launch_log = getenv("ALGP_LAUNCH_LOG") ? fopen(getenv("ALGP_LAUNCH_LOG"), "w") : nullptr;
        if (launch_log) {
            fprintf(launch_log, "%d %lld %lld %lld %d %d %d %d %d\n", klass, (long long)m, (long long)n, (long long)k, lower_only, batch, ktri, (int)sizeof(T), kcut);
            fflush(launch_log);
        }
    }
    int64_t gx = tiles;
    if (ktri) {                                                    // 8 x the largest per-XCD share (rows x, x + 8, ... of XCD x)
        int64_t most = 0;
        for (int x = 0; x < 8; ++x) {
            int64_t cnt = 0;
            for (int64_t r = x; r < g.tiles_m; r += 8) cnt += r + 1;
            most = std::max(most, cnt);
        }
        gx = 8 * most;
    }
    const dim3 grid((unsigned)gx, (unsigned)batch);
    hipEvent_t ev_a, ev_b;
    const bool timed = prof_launch_events(c, klass, flops, bytes, &ev_a, &ev_b);
#if ALGP_GEMM_W4
    if (!lower_only && !ktri && batch == 1 && m >= 2048 && k >= 512) {
        static bool attr_set[2] = {false, false};
        if (!attr_set[stat_out ? 1 : 0]) {
            if (stat_out) ALGP_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel_w4<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152));
            else ALGP_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel_w4<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152));
            attr_set[stat_out ? 1 : 0] = true;
        }
        const dim3 g2((unsigned)(((g.tiles_m + 1) / 2) * g.tiles_n));
        if (stat_out) {
            if (timed) hipExtLaunchKernelGGL((gemm_nt_kernel_w4<T, true>), g2, dim3(256), 3 * 49152, c->cur, ev_a, ev_b, 0, g);
            else hipLaunchKernelGGL((gemm_nt_kernel_w4<T, true>), g2, dim3(256), 3 * 49152, c->cur, g);
        } else {
            if (timed) hipExtLaunchKernelGGL((gemm_nt_kernel_w4<T, false>), g2, dim3(256), 3 * 49152, c->cur, ev_a, ev_b, 0, g);
            else hipLaunchKernelGGL((gemm_nt_kernel_w4<T, false>), g2, dim3(256), 3 * 49152, c->cur, g);
        }
        ALGP_HIP(hipGetLastError());
        return ALGP_OK;
    }
#endif
#if ALGP_GEMM_K96
    if (!lower_only && !ktri && batch == 1 && m >= 2048) {
        static bool attr_set[2] = {false, false};
        if (!attr_set[stat_out ? 1 : 0]) {
            if (stat_out) ALGP_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel_k96<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 24576));
            else ALGP_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel_k96<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 24576));
            attr_set[stat_out ? 1 : 0] = true;
        }
        if (stat_out) {
            if (timed) hipExtLaunchKernelGGL((gemm_nt_kernel_k96<T, true>), grid, dim3(256), 3 * 24576, c->cur, ev_a, ev_b, 0, g);
            else hipLaunchKernelGGL((gemm_nt_kernel_k96<T, true>), grid, dim3(256), 3 * 24576, c->cur, g);
        } else {
            if (timed) hipExtLaunchKernelGGL((gemm_nt_kernel_k96<T, false>), grid, dim3(256), 3 * 24576, c->cur, ev_a, ev_b, 0, g);
            else hipLaunchKernelGGL((gemm_nt_kernel_k96<T, false>), grid, dim3(256), 3 * 24576, c->cur, g);
        }
        ALGP_HIP(hipGetLastError());
        return ALGP_OK;
    }
#endif
#if ALGP_GEMM_T256
    if (!lower_only && !ktri && batch == 1 && m >= 2048) {
        constexpr bool SG = ALGP_GEMM_T256 == 2;
        static bool attr_set[2] = {false, false};
        if (!attr_set[stat_out ? 1 : 0]) {
            if (stat_out) ALGP_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel_t256<T, true, SG>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 24576));
            else ALGP_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel_t256<T, false, SG>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 24576));
            attr_set[stat_out ? 1 : 0] = true;
        }
        const dim3 g2((unsigned)(((g.tiles_m + 1) / 2) * g.tiles_n));
        if (stat_out) {
            if (timed) hipExtLaunchKernelGGL((gemm_nt_kernel_t256<T, true, SG>), g2, dim3(512), 4 * 24576, c->cur, ev_a, ev_b, 0, g);
            else hipLaunchKernelGGL((gemm_nt_kernel_t256<T, true, SG>), g2, dim3(512), 4 * 24576, c->cur, g);
        } else {
            if (timed) hipExtLaunchKernelGGL((gemm_nt_kernel_t256<T, false, SG>), g2, dim3(512), 4 * 24576, c->cur, ev_a, ev_b, 0, g);
            else hipLaunchKernelGGL((gemm_nt_kernel_t256<T, false, SG>), g2, dim3(512), 4 * 24576, c->cur, g);
        }
        ALGP_HIP(hipGetLastError());
        return ALGP_OK;
    }
#endif
    if (stat_out) {
        if (timed) hipExtLaunchKernelGGL((gemm_nt_kernel_dma4<T, true>), grid, dim3(256), 0, c->cur, ev_a, ev_b, 0, g);
        else hipLaunchKernelGGL((gemm_nt_kernel_dma4<T, true>), grid, dim3(256), 0, c->cur, g);
    } else {
        if (timed) hipExtLaunchKernelGGL((gemm_nt_kernel_dma4<T, false>), grid, dim3(256), 0, c->cur, ev_a, ev_b, 0, g);
        else hipLaunchKernelGGL((gemm_nt_kernel_dma4<T, false>), grid, dim3(256), 0, c->cur, g);
    }
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}

template <typename T>
int gemm_nt_launch(algp_ctx* c, int klass, int64_t m, int64_t n, int64_t k, T alpha, const T* A,
                   int64_t lda, const T* B, int64_t ldb, T beta, const T* C, int64_t ldc, T* D,
                   int64_t ldd, int lower_only) {
    return gemm_nt_launch_batched<T>(c, klass, m, n, k, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, D, ldd, 0,
                                     lower_only, 1, 0, nullptr, nullptr, 0, 0);
}
// D = alpha A B^T for ONE column tile (n = 128), and per output row the sums of d^2 and d * w[column] over the tile
template <typename T>
int gemm_nt_launch_stats(algp_ctx* c, int klass, int64_t m, int64_t k, T alpha, const T* A, int64_t lda, const T* B, int64_t ldb,
                         T* D, int64_t ldd, const T* w, T* stat_out, int64_t stat_ld) {
    return gemm_nt_launch_batched<T>(c, klass, m, 128, k, alpha, A, lda, 0, B, ldb, 0, (T)0, nullptr, 0, 0, D, ldd, 0, 0, 1, 0, w,
                                     stat_out, stat_ld, 0);
}
// D (m x n) = A B^T with B (n x n) lower triangular by 128-tiles -- X_J = T inv(L_JJ)^T with the explicit inverse of a 512-column
// block: column tile c walks k < 128 (c + 1) only; stat_out (or null): the row statistics of every column tile as above
template <typename T>
int gemm_nt_launch_tri(algp_ctx* c, int klass, int64_t m, int64_t n, const T* A, int64_t lda, const T* B, int64_t ldb, T* D,
                       int64_t ldd, const T* w, T* stat_out, int64_t stat_ld) {
    return gemm_nt_launch_batched<T>(c, klass, m, n, n, (T)1, A, lda, 0, B, ldb, 0, (T)0, nullptr, 0, 0, D, ldd, 0, 0, 1, 0,
                                     stat_out ? w : nullptr, stat_out, stat_ld, 1);
}
template int gemm_nt_launch_tri<double>(algp_ctx*, int, int64_t, int64_t, const double*, int64_t, const double*, int64_t, double*,
                                        int64_t, const double*, double*, int64_t);
template int gemm_nt_launch_tri<float>(algp_ctx*, int, int64_t, int64_t, const float*, int64_t, const float*, int64_t, float*, int64_t,
                                       const float*, float*, int64_t);
template int gemm_nt_launch_stats<double>(algp_ctx*, int, int64_t, int64_t, double, const double*, int64_t, const double*, int64_t,
                                          double*, int64_t, const double*, double*, int64_t);
template int gemm_nt_launch_stats<float>(algp_ctx*, int, int64_t, int64_t, float, const float*, int64_t, const float*, int64_t, float*,
                                         int64_t, const float*, float*, int64_t);

template int gemm_nt_launch<double>(algp_ctx*, int, int64_t, int64_t, int64_t, double, const double*, int64_t,
                                    const double*, int64_t, double, const double*, int64_t, double*, int64_t, int);
template int gemm_nt_launch<float>(algp_ctx*, int, int64_t, int64_t, int64_t, float, const float*, int64_t,
                                   const float*, int64_t, float, const float*, int64_t, float*, int64_t, int);
template int gemm_nt_launch_batched<double>(algp_ctx*, int, int64_t, int64_t, int64_t, double, const double*, int64_t,
                                            int64_t, const double*, int64_t, int64_t, double, const double*, int64_t,
                                            int64_t, double*, int64_t, int64_t, int, int, int, const double*, double*, int64_t, int);
template int gemm_nt_launch_batched<float>(algp_ctx*, int, int64_t, int64_t, int64_t, float, const float*, int64_t,
                                           int64_t, const float*, int64_t, int64_t, float, const float*, int64_t, int64_t,
                                           float*, int64_t, int64_t, int, int, int, const float*, float*, int64_t, int);

// ---------------------------------------------------------------------------------------------
// MFMA fragment-layout probe (exact integer data, asymmetric B).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void mfma_probe_kernel(int* mismatches) {
    using F = MF<T>;
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kk = lane >> 4;
    const T a = (T)(i * 4 + kk + 1);               // A[i][k]
    const T b = (T)((kk + 1) * 17 + i * 3);        // B[k][j], j = lane&15
    typename F::acc_t acc;
    for (int r = 0; r < 4; ++r) acc[r] = (T)0;
    acc = F::mfma(a, b, acc);
    int bad = 0;
    for (int r = 0; r < 4; ++r) {
        const int row = F::row_of(lane, r), col = lane & 15;
        double want = 0;
        for (int k = 0; k < 4; ++k) want += (double)(row * 4 + k + 1) * (double)((k + 1) * 17 + col * 3);
        if ((double)acc[r] != want) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}

template <typename T>
int test_mfma_launch(algp_ctx* c, int* mismatches_dev) {
    hipLaunchKernelGGL(mfma_probe_kernel<T>, dim3(1), dim3(64), 0, c->cur, mismatches_dev);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int test_mfma_launch<double>(algp_ctx*, int*);
template int test_mfma_launch<float>(algp_ctx*, int*);


// ---------------------------------------------------------------------------------------------
// device-resident GEMM benchmark (pseudo-random operands, no host traffic)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void fill_random_kernel(T* p, int64_t n, unsigned seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)(i * 2654435761u) ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = (T)((int)(x & 0xffff) - 32768) * (T)(1.0 / 32768.0);
}

template <typename T>
int bench_gemm(algp_ctx* c, int64_t m, int64_t n, int64_t k, int lower_only, int beta_one, int reps,
               double* ms_out) {
    DevBuf a, b, cc;
    int rc = ensure(c, a, sizeof(T) * m * k);
    if (rc == ALGP_OK) rc = ensure(c, b, sizeof(T) * n * k);
    if (rc == ALGP_OK) rc = ensure(c, cc, sizeof(T) * m * n);
    if (rc != ALGP_OK) { hipFree(a.p); hipFree(b.p); hipFree(cc.p); return rc; }
    hipLaunchKernelGGL(fill_random_kernel<T>, dim3((unsigned)((m * k + 255) / 256)), dim3(256), 0, c->cur, (T*)a.p, m * k, 1u);
    hipLaunchKernelGGL(fill_random_kernel<T>, dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, c->cur, (T*)b.p, n * k, 2u);
    hipLaunchKernelGGL(fill_random_kernel<T>, dim3((unsigned)((m * n + 255) / 256)), dim3(256), 0, c->cur, (T*)cc.p, m * n, 3u);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const T beta = beta_one ? (T)1 : (T)0;
    for (int w = 0; w < 2 && rc == ALGP_OK; ++w)
        rc = gemm_nt_launch<T>(c, ALGP_PROF_GEMM_OTHER, m, n, k, (T)-1, (const T*)a.p, k, (const T*)b.p, k, beta,
                               (const T*)cc.p, n, (T*)cc.p, n, lower_only);
    hipEventRecord(e0, c->cur);
    for (int r = 0; r < reps && rc == ALGP_OK; ++r)
        rc = gemm_nt_launch<T>(c, ALGP_PROF_GEMM_OTHER, m, n, k, (T)-1, (const T*)a.p, k, (const T*)b.p, k, beta,
                               (const T*)cc.p, n, (T*)cc.p, n, lower_only);
    hipEventRecord(e1, c->cur);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / (reps > 0 ? reps : 1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(a.p); hipFree(b.p); hipFree(cc.p);
    c->dev_bytes -= (int64_t)(a.cap + b.cap + cc.cap);
    return rc;
}
template int bench_gemm<double>(algp_ctx*, int64_t, int64_t, int64_t, int, int, int, double*);
template int bench_gemm<float>(algp_ctx*, int64_t, int64_t, int64_t, int, int, int, double*);

}  // namespace algp
