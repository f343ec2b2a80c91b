// diag.h -- factor + invert one 128 x 128 SPD diagonal block inside one 256-thread workgroup.
//
// This is the serial link of the blocked Cholesky (reference: the LU inside np.linalg.inv /
// np.linalg.slogdet, utils.py:193, 300): every block column waits for it, so it is built for latency.
//
// The block is seen as 8 x 8 blocks of 16 x 16 ("leaf" size = one MFMA tile).
//   * The trailing part never sits in LDS: each wave keeps 9 of the 36 lower blocks in MFMA accumulator
//     registers (C layout, NEGATED so that the rank-16 updates are plain accumulating MFMAs), loaded straight
//     from global memory.  Blocks are dealt to waves cyclically in column-major order, so the blocks still
//     to update (a suffix of that order) are balanced over the waves at every step.
//   * Step p (16 columns): the owners store block column p to LDS; wave 0 factors the 16 x 16 leaf
//     wave-synchronously (no barrier, no LDS inside): lanes 0-15 own the rows of the leaf, lanes 16-31
//     own the columns of its inverse; the multipliers travel by v_readlane (scalar operands of the FMAs),
//     the pivot reciprocal by rcp + 2 Newton steps is the only transcendental on the dependency chain
//     (columns stay unscaled, 1/sqrt(d) is applied afterwards); then P_b = A_b X_pp^T on the matrix
//     cores for the blocks below, then the rank-16 update of the register-resident blocks.
//   * Inverse X = L^-1 by recursive doubling 16 -> 32 -> 64 -> 128: X_ba = -X_b (L_ba X_a), both
//     products on the matrix cores, the accumulator of the first is the B operand of the second.
// LDS image: the lower block triangle, 36 blocks of 16 x 17 elements (78 KB fp64 / 39 KB fp32); the strict
// upper part of a diagonal block holds the transposed leaf inverse, its diagonal lives in dinv[].
#pragma once
#include "common.h"
#include "mfma.h"

namespace algp {

// 1/x and 1/sqrt(x) from the hardware seed + Newton steps (a full IEEE fp64 divide / sqrt costs
// several hundred cycles and sits on the critical path of every pivot column)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}
__device__ __forceinline__ float fast_rcp(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    return r * (2.0f - x * r);
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}
__device__ __forceinline__ float fast_rsqrt(float x) {
    float r = __builtin_amdgcn_rsqf(x);
    return r * (1.5f - 0.5f * x * r * r);
}

// the links of fast_rcp, one at a time (diag_leaf interleaves them with independent work)
template <typename T>
struct RcpChain;
template <>
struct RcpChain<double> {
    static constexpr int NLINK = 4;
    static __device__ __forceinline__ double seed(double x) { return __builtin_amdgcn_rcp(x); }
    static __device__ __forceinline__ void step(int k, double x, double& r, double& t) {
        if ((k & 1) == 0) t = 2.0 - x * r;
        else r = r * t;
    }
};
template <>
struct RcpChain<float> {
    static constexpr int NLINK = 2;
    static __device__ __forceinline__ float seed(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ void step(int k, float x, float& r, float& t) {
        if (k == 0) t = 2.0f - x * r;
        else r = r * t;
    }
};

// value of lane `src` (a constant after unrolling) as a wave-uniform scalar
__device__ __forceinline__ double lane_bcast(double x, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_bcast(float x, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src));
}

__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Diagnostic builds only (tools/diag_test.hip): cycle stamps of thread 0 at phase boundaries.
#ifdef ALGP_POTRF_STAMPS
__device__ unsigned long long g_potrf_stamps[64];
#define ALGP_STAMP(k) do { if (threadIdx.x == 0) g_potrf_stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ALGP_STAMP(k) do { } while (0)
#endif

constexpr int DBS = 16 * 17;                                   // elements per stored block
__device__ __forceinline__ int LBLK(int I, int J) { return (I * (I + 1) / 2 + J) * DBS; }
__device__ __forceinline__ int LB(int i, int j) { return LBLK(i >> 4, j >> 4) + (i & 15) * 17 + (j & 15); }

template <typename T>
struct DiagShared {
    T S[36 * DBS];
    T dd[128];        // pivots d_j (before the square root)
    T dinv[128];      // 1 / L_jj
    double red[4];
    int bad;
};

// column-major enumeration of the 36 lower blocks: t -> (row block, column block)
__device__ __forceinline__ void diag_block_of(int t, int& bi, int& bk) {
    bk = 0;
    int first = 0;                                             // index of block (bk, bk)
    while (t >= first + (8 - bk)) { first += 8 - bk; ++bk; }
    bi = bk + (t - first);
}

// ---- the 16 x 16 leaf: factor A_pp = L L^T and invert L, one wave, no LDS traffic inside ----
// lanes 0-15 ("A lanes", row r): v[c] = A[r][c];  lanes 16-31 ("X lanes", column r of the inverse): v[] = 0.
// Column step j with pivot d_j: f = (e_j - v[j]) / d_j  (e_j = 1 only in X lane j), then v[c] += u_c f for c > j
// with u_c = A lane c's v[j] (wave-uniform).  In the A lanes this is the unscaled right-looking update
// a[c] -= a_rj a_cj / d_j; in the X lanes v[i] accumulates s_i = sum_k U_ik Z_k of Z = U^-1 (U = L diag(sqrt d)).
template <typename T>
__device__ __forceinline__ void diag_leaf(DiagShared<T>& sh, int p, int lane) {
    const int li = lane & 31, r = li & 15;                     // lanes 32-63 shadow lanes 0-31 and keep the pivots
    const bool isX = li >= 16;
    T* blk = sh.S + LBLK(p, p);
    T v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const T t = blk[r * 17 + c];
        v[c] = isX ? (T)0 : t;
    }
    T dmine = (T)1;                                            // lane 32 + j keeps pivot d_j
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const T d = lane_bcast(v[j], j);
        dmine = (lane == 32 + j) ? d : dmine;
        const T e = (li == 16 + j) ? (T)1 : (T)0;
        const T num = e - v[j];
        // The reciprocal (seed + Newton steps) is a chain of dependent operations and the only thing the next
        // column waits for; the column's broadcasts (scalar registers, independent of it) are issued between its
        // links, pinned there with scheduling barriers: a lone wave otherwise sits out every link's latency.
        constexpr int NLINK = RcpChain<T>::NLINK;
        const int npair = 15 - j, per = (npair + NLINK) / (NLINK + 1);
        T r = RcpChain<T>::seed(d), t = (T)0;
        int link = 0;
        T uc[16];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = j + 1; c < 16; ++c) {
            uc[c] = lane_bcast(v[j], c);
            if (per > 0 && (c - j) % per == 0 && link < NLINK) {
                __builtin_amdgcn_sched_barrier(0);
                RcpChain<T>::step(link++, d, r, t);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < NLINK; ++k)
            if (k >= link) RcpChain<T>::step(k, d, r, t);
        const T f = num * r;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) v[c] = fma_t(uc[c], f, v[c]);
        v[j] = isX ? num : v[j];                               // X lanes keep (e_j - s_j); A lanes keep u_rj
    }
    // one reciprocal square root per lane 32..47 instead of sixteen in every lane
    const T rsmine = fast_rsqrt(dmine);
    if (lane >= 32 && lane < 48) {
        sh.dinv[16 * p + lane - 32] = rsmine;
        sh.dd[16 * p + lane - 32] = dmine;
    }
    const unsigned long long badmask = __ballot(lane >= 32 && lane < 48 && !(dmine > (T)0));
    if (lane == 0 && badmask && sh.bad == 0) sh.bad = 16 * p + (__ffsll((long long)badmask) - 1 - 32) + 1;
    T rs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) rs[j] = lane_bcast(rsmine, 32 + j);
    if (lane < 16) {
        // L_rj = u_rj / sqrt(d_j) for j < r and L_rr = sqrt(d_r) = u_rr / sqrt(d_r) (an A lane keeps u_rr = d_r);
        // the upper part of the block belongs to the X lanes
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j <= r) blk[r * 17 + j] = v[j] * rs[j];
    } else if (lane < 32) {
        // X_ir = (e - s_i) / sqrt(d_i), stored transposed at (r, i) for i > r; the diagonal is dinv[]
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i > r) blk[r * 17 + i] = v[i] * rs[i];
    }
}

// B operand of a product with a leaf inverse TRANSPOSED, B[k][j] = X_II[j][k] (k <= j), from the upper storage
template <typename T>
__device__ __forceinline__ T leaf_inv_T(const DiagShared<T>& sh, int I, int k, int j) {
    const T* D = sh.S + LBLK(I, I);
    T b = (T)0;
    if (k < j) b = D[k * 17 + j];
    else if (k == j) b = sh.dinv[16 * I + j];
    return b;
}
// element X_II[i][k] (k <= i) of a leaf inverse
template <typename T>
__device__ __forceinline__ T leaf_inv(const DiagShared<T>& sh, int I, int i, int k) {
    const T* D = sh.S + LBLK(I, I);
    T a = (T)0;
    if (k < i) a = D[k * 17 + i];
    else if (k == i) a = sh.dinv[16 * I + i];
    return a;
}

// ---- inverse phase helpers (all operands in LDS; results held in registers until the caller's barrier) ----
// acc += L_KM X_MJ where X_MJ is a full block (M > J, row-major) or the leaf inverse (M == J)
template <typename T>
__device__ __forceinline__ void inv_accum_LX(const DiagShared<T>& sh, int K, int M, int J, typename MF<T>::acc_t& acc,
                                             int li, int lg) {
    using F = MF<T>;
    const T* Lkm = sh.S + LBLK(K, M);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int k = 4 * st + lg;
        T bv;
        if (M == J) {
            // B[k][b] = X_JJ[k][b] (k >= b)
            const T* D = sh.S + LBLK(J, J);
            bv = (T)0;
            if (k > li) bv = D[li * 17 + k];
            else if (k == li) bv = sh.dinv[16 * J + li];
        } else {
            bv = sh.S[LBLK(M, J) + k * 17 + li];
        }
        acc = F::mfma(Lkm[li * 17 + k], bv, acc);
    }
}
// out += X_IK * Tacc, where Tacc is an accumulator (row k = row_of(lane, st) pairs with A[m][k]); X_IK is a full
// block (I > K) or the leaf inverse (I == K)
template <typename T>
__device__ __forceinline__ void inv_accum_XT(const DiagShared<T>& sh, int I, int K, const typename MF<T>::acc_t& tacc,
                                             typename MF<T>::acc_t& out, int lane, int li) {
    using F = MF<T>;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int k = F::row_of(lane, st);
        T av;
        if (I == K) av = leaf_inv<T>(sh, I, li, k);
        else av = sh.S[LBLK(I, K) + li * 17 + k];
        out = F::mfma(av, tacc[st], out);
    }
}

// X_ba = -X_b (L_ba X_a) for one block column J of the off-diagonal part whose block rows are [r0, r0 + nb) and
// whose block columns are [c0, c0 + nb) (nb = 1, 2, 4): the results stay in outs[] (negated on store)
template <typename T, int NBK>
__device__ __forceinline__ void inv_column(const DiagShared<T>& sh, int r0, int c0, int J, typename MF<T>::acc_t (&outs)[NBK],
                                           int lane) {
    using F = MF<T>;
    const int li = lane & 15, lg = lane >> 4;
    typename F::acc_t tt[NBK];
#pragma unroll
    for (int a = 0; a < NBK; ++a) {
#pragma unroll
        for (int q = 0; q < 4; ++q) tt[a][q] = (T)0;
        for (int M = J; M < c0 + NBK; ++M) inv_accum_LX<T>(sh, r0 + a, M, J, tt[a], li, lg);
    }
#pragma unroll
    for (int a = 0; a < NBK; ++a) {
#pragma unroll
        for (int q = 0; q < 4; ++q) outs[a][q] = (T)0;
#pragma unroll
        for (int b = 0; b < NBK; ++b)
            if (b <= a) inv_accum_XT<T>(sh, r0 + a, r0 + b, tt[b], outs[a], lane, li);
    }
}
template <typename T, int NBK>
__device__ __forceinline__ void inv_store(DiagShared<T>& sh, int r0, int J, const typename MF<T>::acc_t (&outs)[NBK],
                                          int lane) {
    using F = MF<T>;
    const int li = lane & 15;
#pragma unroll
    for (int a = 0; a < NBK; ++a) {
        T* X = sh.S + LBLK(r0 + a, J);
#pragma unroll
        for (int q = 0; q < 4; ++q) X[F::row_of(lane, q) * 17 + li] = -outs[a][q];
    }
}

// ---- 16 x 16 block movers between the LDS image and global memory (one wave; 128-byte row segments in fp64) ----
// lane (column li, group lg) moves rows lg + 4q
template <typename T>
__device__ __forceinline__ void store_block_rowmajor(const T* blk, T* g, int64_t ld, int lane) {
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) st_wt(&g[(int64_t)(lg + 4 * q) * ld + li], blk[(lg + 4 * q) * 17 + li]);
}
// the lower triangle of a diagonal block of L (zeros above the diagonal: the LDS upper part holds the leaf inverse)
template <typename T>
__device__ __forceinline__ void store_block_lower(const T* blk, T* g, int64_t ld, int lane) {
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = lg + 4 * q;
        const T t = blk[row * 17 + li];
        st_wt(&g[(int64_t)row * ld + li], (li <= row) ? t : (T)0);
    }
}
// the leaf inverse X_II (transposed upper storage + dinv[]) as a dense lower-triangular block
template <typename T>
__device__ __forceinline__ void store_block_leafinv(const DiagShared<T>& sh, int I, T* g, int64_t ld, int lane) {
    const int li = lane & 15, lg = lane >> 4;
    const T* D = sh.S + LBLK(I, I);
    const T dv = sh.dinv[16 * I + li];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = lg + 4 * q;
        const T t = D[li * 17 + row];                          // X[row][li] for li < row
        st_wt(&g[(int64_t)row * ld + li], (li < row) ? t : (li == row ? dv : (T)0));
    }
}
template <typename T>
__device__ __forceinline__ void store_block_zero(T* g, int64_t ld, int lane) {
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) st_wt(&g[(int64_t)(lg + 4 * q) * ld + li], (T)0);
}

// Factor the 128 x 128 block at A (leading dimension lda) in place and write its inverse (dense, 128 x 128, ld 128,
// zeros above the diagonal) to inv_out.  Pipeline per 16-column step p (three barriers):
//   [owners: block column p -> LDS] B1 [wave 0: leaf p | waves 1-3: row p of L -> global, row p-1 of X -> global,
//   T_pJ = sum_K L_pK X_KJ for the inverse's row p] B2 [all: P_b = A_b X_pp^T; waves 1-3: X_pJ = -X_pp T_pJ] B3
//   [all: rank-16 update of the register-resident blocks].
// The inverse is complete one step after the factor: its rows ride in the shadow of the leaves.
template <typename T>
__device__ __forceinline__ void diag128_factor(DiagShared<T>& sh, T* A, int64_t lda, T* inv_out, double* logdet_acc,
                                               bool logdet_atomic, int* info, int64_t block_row0) {
    using F = MF<T>;
    using acc_t = typename F::acc_t;
    const int tid = threadIdx.x;
    const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) sh.bad = 0;
    ALGP_STAMP(0);
    // ---- load: global -> accumulator registers (negated), block t = 4u + wave of the column-major order ----
    acc_t acc[9];
    int bis[9], bks[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) {
        diag_block_of(4 * u + wave, bis[u], bks[u]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            acc[u][q] = -A[(int64_t)(16 * bis[u] + F::row_of(lane, q)) * lda + 16 * bks[u] + li];
    }
    ALGP_STAMP(1);
    acc_t tacc[3];                                             // waves 1-3: T_pJ for J = wave - 1 + 3m
    for (int p = 0; p < 8; ++p) {
        // (1) owners put block column p into LDS
#pragma unroll
        for (int u = 0; u < 9; ++u)
            if (bks[u] == p) {
                T* dst = sh.S + LBLK(bis[u], p);
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[F::row_of(lane, q) * 17 + li] = -acc[u][q];
            }
        __syncthreads();                                       // B1
        if (p == 0) ALGP_STAMP(2);
        ALGP_STAMP(8 + 6 * p + 0);
        if (wave == 0) {
            diag_leaf<T>(sh, p, lane);                         // (2)
        } else {
            const int w1 = wave - 1;
            // finished parts go out while the leaf runs: row p of L left of the diagonal, the diagonal block of
            // row p-1, row p-1 of the inverse
            for (int J = w1; J < p; J += 3)
                store_block_rowmajor<T>(sh.S + LBLK(p, J), A + (int64_t)(16 * p) * lda + 16 * J, lda, lane);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int J = w1 + 3 * m;
#pragma unroll
                for (int q = 0; q < 4; ++q) tacc[m][q] = (T)0;
                if (J < p)
                    for (int M = J; M < p; ++M) inv_accum_LX<T>(sh, p, M, J, tacc[m], li, lg);
            }
            if (p > 0) {
                const int I = p - 1;
                T* xo = inv_out + (int64_t)(16 * I) * 128;
                for (int J = w1; J < 8; J += 3) {
                    if (J < I) store_block_rowmajor<T>(sh.S + LBLK(I, J), xo + 16 * J, 128, lane);
                    else if (J == I) store_block_leafinv<T>(sh, I, xo + 16 * J, 128, lane);
                    else store_block_zero<T>(xo + 16 * J, 128, lane);
                }
                if (w1 == (I % 3))
                    store_block_lower<T>(sh.S + LBLK(I, I), A + (int64_t)(16 * I) * lda + 16 * I, lda, lane);
            }
        }
        if (p == 0) ALGP_STAMP(3);
        ALGP_STAMP(8 + 6 * p + 1);
        __syncthreads();                                       // B2
        ALGP_STAMP(8 + 6 * p + 2);
        // inverse row p: X_pJ = -X_pp T_pJ overwrites L_pJ (already stored; no later step reads row p of L)
        if (wave > 0) {
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int J = wave - 1 + 3 * m;
                if (J < p) {
                    acc_t o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = (T)0;
                    inv_accum_XT<T>(sh, p, p, tacc[m], o, lane, li);
                    T* X = sh.S + LBLK(p, J);
#pragma unroll
                    for (int q = 0; q < 4; ++q) X[F::row_of(lane, q) * 17 + li] = -o[q];
                }
            }
        }
        if (p == 7) break;
        // (3) P_b = A_b X_pp^T for the blocks below the leaf
        for (int b = p + 1 + wave; b < 8; b += 4) {
            T* Ab = sh.S + LBLK(b, p);
            acc_t pacc;
#pragma unroll
            for (int q = 0; q < 4; ++q) pacc[q] = (T)0;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int k = 4 * st + lg;
                pacc = F::mfma(Ab[li * 17 + k], leaf_inv_T<T>(sh, p, k, li), pacc);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) Ab[F::row_of(lane, q) * 17 + li] = pacc[q];
        }
        ALGP_STAMP(8 + 6 * p + 3);
        __syncthreads();                                       // B3
        if (p == 0) ALGP_STAMP(4);
        ALGP_STAMP(8 + 6 * p + 4);
        // (4) rank-16 update of the register-resident blocks right of the panel: (-C) += P_bi P_bk^T
#pragma unroll
        for (int u = 0; u < 9; ++u)
            if (bks[u] > p) {
                const T* Pa = sh.S + LBLK(bis[u], p);
                const T* Pb = sh.S + LBLK(bks[u], p);
#pragma unroll
                for (int st = 0; st < 4; ++st)
                    acc[u] = F::mfma(Pa[li * 17 + 4 * st + lg], Pb[li * 17 + 4 * st + lg], acc[u]);
            }
        if (p == 0) ALGP_STAMP(5);
        ALGP_STAMP(8 + 6 * p + 5);
    }
    ALGP_STAMP(6);
    __syncthreads();
    // ---- tail: the last diagonal block of L, row 7 of the inverse, log-determinant ----
    {
        T* xo = inv_out + (int64_t)(16 * 7) * 128;
        for (int J = wave; J < 8; J += 4) {
            if (J < 7) store_block_rowmajor<T>(sh.S + LBLK(7, J), xo + 16 * J, 128, lane);
            else store_block_leafinv<T>(sh, 7, xo + 16 * J, 128, lane);
        }
        if (wave == 2) store_block_lower<T>(sh.S + LBLK(7, 7), A + (int64_t)(16 * 7) * lda + 16 * 7, lda, lane);
        double v = (tid < 128) ? log((double)sh.dd[tid]) : 0.0;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((tid & 63) == 0) sh.red[tid >> 6] = v;
    }
    __syncthreads();
    if (tid == 0) {
        const double ld = sh.red[0] + sh.red[1];
        if (logdet_atomic) atomicAdd(logdet_acc, ld);
        else *logdet_acc = ld;
        if (sh.bad) atomicCAS(info, 0, (int)(block_row0 + sh.bad));
    }
    ALGP_STAMP(7);
    __syncthreads();
}

// A already holds a lower-triangular factor: write its inverse (dense, 128 x 128, ld 128) to inv_out.
template <typename T>
__device__ __forceinline__ void diag128_invert(DiagShared<T>& sh, const T* A, int64_t lda, T* inv_out) {
    using F = MF<T>;
    using acc_t = typename F::acc_t;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int VEC = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
            // ---- the factor is given: stage its lower block triangle, invert the eight leaves one column per thread ----
            vec_t tmp[4][16 / VEC];
    #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pr = tid + 256 * u, i = pr >> 3, J = pr & 7;
    #pragma unroll
                for (int v = 0; v < 16 / VEC; ++v)
                    tmp[u][v] = *reinterpret_cast<const vec_t*>(A + (int64_t)i * lda + 16 * J + v * VEC);
            }
    #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pr = tid + 256 * u, i = pr >> 3, J = pr & 7;
                if (J <= (i >> 4)) {
                    T* dst = sh.S + LB(i, 16 * J);
    #pragma unroll
                    for (int v = 0; v < 16 / VEC; ++v)
    #pragma unroll
                        for (int e = 0; e < VEC; ++e) dst[v * VEC + e] = tmp[u][v][e];
                }
            }
            __syncthreads();
            if (tid < 128) sh.dinv[tid] = (T)1 / sh.S[LB(tid, tid)];
            __syncthreads();
            if (tid < 128) {
                const int I = tid >> 4, c = tid & 15, base = I * 16;
                T* Db = sh.S + LBLK(I, I);
                // x[k] = 0 for k < c, so the sums run over all k < i with unconditional (broadcast) LDS loads
                T x[16];
    #pragma unroll
                for (int i = 0; i < 16; ++i) {
                    T sum = (T)0;
    #pragma unroll
                    for (int k = 0; k < i; ++k) sum += Db[i * 17 + k] * x[k];
                    x[i] = (i == c) ? sh.dinv[base + c] : ((i > c) ? -sum * sh.dinv[base + i] : (T)0);
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);      // all reads of the block have landed before its upper part is written
    #pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (i > c) Db[c * 17 + i] = x[i];
            }
    __syncthreads();

    // ---- inverse by recursive doubling; every level: products into registers, barrier, store, barrier ----
    {   // 16 -> 32: X_(2w+1, 2w) for wave w
        acc_t o1[1];
        inv_column<T, 1>(sh, 2 * wave + 1, 2 * wave, 2 * wave, o1, lane);
        __syncthreads();
        inv_store<T, 1>(sh, 2 * wave + 1, 2 * wave, o1, lane);
        __syncthreads();
    }
    {   // 32 -> 64: half m = wave >> 1, block column J = 4m + (wave & 1), block rows 4m + 2, 4m + 3
        const int m = wave >> 1, J = 4 * m + (wave & 1);
        acc_t o2[2];
        inv_column<T, 2>(sh, 4 * m + 2, 4 * m, J, o2, lane);
        __syncthreads();
        inv_store<T, 2>(sh, 4 * m + 2, J, o2, lane);
        __syncthreads();
    }
    {   // 64 -> 128: block column J = wave, block rows 4..7
        acc_t o4[4];
        inv_column<T, 4>(sh, 4, 0, wave, o4, lane);
        __syncthreads();
        inv_store<T, 4>(sh, 4, wave, o4, lane);
        __syncthreads();
    }
    // inverse out, 16 elements of a row at a time: left of the diagonal block X_IJ rows, inside it the
    // transposed upper storage with dinv on the diagonal, zeros to the right
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int pr = tid + 256 * u, i = pr >> 3, J = pr & 7, I = i >> 4, r = i & 15;
        T row[16];
        if (J < I) {
            const T* src = sh.S + LB(i, 16 * J);
#pragma unroll
            for (int e = 0; e < 16; ++e) row[e] = src[e];
        } else if (J == I) {
            const T* Db = sh.S + LBLK(I, I);
#pragma unroll
            for (int e = 0; e < 16; ++e) row[e] = (e < r) ? Db[e * 17 + r] : (e == r ? sh.dinv[i] : (T)0);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) row[e] = (T)0;
        }
#pragma unroll
        for (int v = 0; v < 16 / VEC; ++v) {
            vec_t o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = row[v * VEC + e];
            *reinterpret_cast<vec_t*>(inv_out + i * 128 + 16 * J + v * VEC) = o;
        }
    }
    __syncthreads();
}

template <typename T, bool FACTOR>
__device__ __forceinline__ void diag128_run(DiagShared<T>& sh, T* A, int64_t lda, T* inv_out, double* logdet_acc,
                                            bool logdet_atomic, int* info, int64_t block_row0) {
    if (FACTOR) diag128_factor<T>(sh, A, lda, inv_out, logdet_acc, logdet_atomic, info, block_row0);
    else diag128_invert<T>(sh, A, lda, inv_out);
}

}  // namespace algp
