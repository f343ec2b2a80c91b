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

// Per-lane choice between an LDS value t, the diagonal entry dv and zero, as in (k < i ? t : k == i ? dv : 0), written as
// t * m1 + dv * m2 with 0/1 masks: given the select, the compiler sinks the LDS read into a divergent branch (an
// exec-masked ds_read per case with a waitcnt each: 180 such branches in this kernel) instead of reading unconditionally.
// (The masked-out values are finite -- other entries of the same factor -- unless the block is not positive definite,
// which the leaves report anyway.)
template <typename T>
__device__ __forceinline__ T pick3(T t, bool use_t, T dv, bool use_d) {
    return fma_t(t, use_t ? (T)1 : (T)0, dv * (use_d ? (T)1 : (T)0));
}


// Diagnostic builds only (tools/diag_test.hip): cycle stamps of thread 0 at phase boundaries.
#ifdef ALGP_POTRF_STAMPS
__device__ unsigned long long g_potrf_stamps[64];
__device__ unsigned long long g_potrf_stamps_w[64];             // bulk wave 1 (threads 64..127), four per panel
#define ALGP_STAMP(k) do { if (threadIdx.x == 0) g_potrf_stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#define ALGP_STAMPW(k) do { if (threadIdx.x == 64) g_potrf_stamps_w[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ALGP_STAMP(k) do { } while (0)
#define ALGP_STAMPW(k) do { } while (0)
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
// gsrc != null (leaf 0 only): the block's rows come straight from global memory (leading dimension ld) instead of from
// the LDS image -- leaf 0 then runs while the other waves are still loading their blocks and staging block column 0.
template <typename T>
__device__ __forceinline__ void diag_leaf(DiagShared<T>& sh, int p, int lane, const T* gsrc = nullptr, int64_t ld = 0) {
    const int li = lane & 31, r = li & 15;                     // lanes 32-63 shadow lanes 0-31 and keep the pivots
    const bool isX = li >= 16;
    T* blk = sh.S + LBLK(p, p);
    T v[16];
    if (gsrc) {
        constexpr int VEC = 16 / sizeof(T);
        typedef T vec_t __attribute__((ext_vector_type(VEC)));
        const T* row = gsrc + (int64_t)r * ld;
#pragma unroll
        for (int c = 0; c < 16; c += VEC) {
            const vec_t t = *reinterpret_cast<const vec_t*>(row + c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[c + e] = isX ? (T)0 : t[e];
        }
    } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const T t = blk[r * 17 + c];
            v[c] = isX ? (T)0 : t;
        }
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

// ---- 16 x 16 block movers from the LDS image to global memory (one wave) ----
// lane l moves the four consecutive elements (row l >> 2, columns 4 (l & 3) ..) as 16-byte WRITE-THROUGH stores
// (buffer_store_dwordx4 ... sc1: one per lane in fp32, two in fp64) through a descriptor on the block's wave-uniform base;
// the per-lane part of the address is one 32-bit byte offset.  Rounds 1-3 stored four scalars per lane (rows lg + 4q,
// column li): four fabric writes where one does (a 4-byte sc1 store costs ~6x a 16-byte one per byte,
// MI355X_MICROARCH.md), and the leader's publish waits for all of them to drain.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ void store_row4(T* g, int64_t ld, int lane, T v0, T v1, T v2, T v3);
template <>
__device__ __forceinline__ void store_row4<float>(float* g, int64_t ld, int lane, float v0, float v1, float v2, float v3) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g, 0, 0x7ffffff0, 0x00020000);
    const int off = (int)(((uint32_t)(lane >> 2) * (uint32_t)ld + (uint32_t)(lane & 3) * 4u) * 4u);
    const u32x4 v = {__float_as_uint(v0), __float_as_uint(v1), __float_as_uint(v2), __float_as_uint(v3)};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
}
template <>
__device__ __forceinline__ void store_row4<double>(double* g, int64_t ld, int lane, double v0, double v1, double v2, double v3) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g, 0, 0x7ffffff0, 0x00020000);
    const int off = (int)(((uint32_t)(lane >> 2) * (uint32_t)ld + (uint32_t)(lane & 3) * 4u) * 8u);
    const u32x4 a = {(unsigned)__double2loint(v0), (unsigned)__double2hiint(v0), (unsigned)__double2loint(v1), (unsigned)__double2hiint(v1)};
    const u32x4 b = {(unsigned)__double2loint(v2), (unsigned)__double2hiint(v2), (unsigned)__double2loint(v3), (unsigned)__double2hiint(v3)};
    __builtin_amdgcn_raw_buffer_store_b128(a, rs, off, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(b, rs, off + 16, 0, 16);
}
template <typename T>
__device__ __forceinline__ void store_block_rowmajor(const T* blk, T* g, int64_t ld, int lane) {
    const T* src = blk + (lane >> 2) * 17 + (lane & 3) * 4;
    store_row4<T>(g, ld, lane, src[0], src[1], src[2], src[3]);
}
// the lower triangle of a diagonal block of L (zeros above the diagonal: the LDS upper part holds the leaf inverse)
template <typename T>
__device__ __forceinline__ void store_block_lower(const T* blk, T* g, int64_t ld, int lane) {
    const int r = lane >> 2, c = (lane & 3) * 4;
    const T* src = blk + r * 17 + c;
    const T t0 = src[0], t1 = src[1], t2 = src[2], t3 = src[3];
    store_row4<T>(g, ld, lane, c <= r ? t0 : (T)0, c + 1 <= r ? t1 : (T)0, c + 2 <= r ? t2 : (T)0, c + 3 <= r ? t3 : (T)0);
}
// the leaf inverse X_II (transposed upper storage + dinv[]) as a dense lower-triangular block: X[r][c] = D[c][r] for c < r
template <typename T>
__device__ __forceinline__ void store_block_leafinv(const DiagShared<T>& sh, int I, T* g, int64_t ld, int lane) {
    const int r = lane >> 2, c = (lane & 3) * 4;
    const T* D = sh.S + LBLK(I, I);
    const T dv = sh.dinv[16 * I + r];
    T v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const T t = D[(c + e) * 17 + r];
        v[e] = (c + e < r) ? t : (c + e == r ? dv : (T)0);
    }
    store_row4<T>(g, ld, lane, v[0], v[1], v[2], v[3]);
}
template <typename T>
__device__ __forceinline__ void store_block_zero(T* g, int64_t ld, int lane) {
    store_row4<T>(g, ld, lane, (T)0, (T)0, (T)0, (T)0);
}

// LDS traffic complete, then the workgroup barrier.  (Not __syncthreads(): that also drains vmcnt, and the bulk waves keep
// write-through stores of finished rows in flight across these barriers.)
__device__ __forceinline__ void diag_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- operand fragments of the 16 x 16 x 16 block products below.  One product is four MFMAs (k = 16 in steps of 4); the
// k index a lane feeds in step st is kperm(lg, st) = 4 lg + st for BOTH operands, so that a lane's four A values are
// contiguous in a row of the LDS image (two ds_read2 instead of four reads) -- any bijection of k works as long as A and
// B agree.  Reads are gathered for several independent products first, then their MFMAs are issued interleaved, then
// the results are written: a product chain is four dependent MFMAs (~32 cycles each) behind an LDS round trip, and
// the compiler cannot hoist the next product's LDS reads over the previous product's LDS writes (may alias).
template <typename T>
struct Frag { T v[4]; };
// A[i = li][k] of a row-major 16 x 17 block
template <typename T>
__device__ __forceinline__ Frag<T> frag_rows(const T* blk, int li, int lg, bool on) {
    Frag<T> f;
#pragma unroll
    for (int st = 0; st < 4; ++st) f.v[st] = blk[li * 17 + 4 * lg + st];
    if (!on) {                                                 // wave-uniform
#pragma unroll
        for (int st = 0; st < 4; ++st) f.v[st] = (T)0;
    }
    return f;
}
// B[k][j = li] of a row-major block (a column walk)
template <typename T>
__device__ __forceinline__ Frag<T> frag_cols(const T* blk, int li, int lg, bool on) {
    Frag<T> f;
#pragma unroll
    for (int st = 0; st < 4; ++st) f.v[st] = blk[(4 * lg + st) * 17 + li];
    if (!on) {                                                 // wave-uniform
#pragma unroll
        for (int st = 0; st < 4; ++st) f.v[st] = (T)0;
    }
    return f;
}
// B[k][j = li] = X_II[j][k] (k <= j): the leaf inverse TRANSPOSED, from the upper storage of diagonal block I
template <typename T>
__device__ __forceinline__ Frag<T> frag_leafinv_T(const DiagShared<T>& sh, int I, int li, int lg) {
    Frag<T> f;
    const T* D = sh.S + LBLK(I, I);
    const T dv = sh.dinv[16 * I + li];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int k = 4 * lg + st;
        f.v[st] = pick3<T>(D[k * 17 + li], k < li, dv, k == li);
    }
    return f;
}
// B[k][j = li] = X_II[k][j] (j <= k): the leaf inverse itself
template <typename T>
__device__ __forceinline__ Frag<T> frag_leafinv(const DiagShared<T>& sh, int I, int li, int lg, bool on) {
    Frag<T> f;
    const T* D = sh.S + LBLK(I, I);
    const T dv = sh.dinv[16 * I + li];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int k = 4 * lg + st;
        f.v[st] = pick3<T>(D[li * 17 + k], on && k > li, dv, on && k == li);
    }
    return f;
}
// B[k][j = li] = X_MJ[k][j] of the inverse under construction: a full block (M > J), the leaf inverse (M == J), zeros (M < J).
// M and J are wave-uniform, so the three cases are scalar branches: a full block -- most of them -- is four plain reads
// with no per-lane masks (the masked one-size-fits-all form spent ~60 VALU instructions per three fragments: a third of
// the time of the products of the inverse's rows, which overrun the leaf they are meant to hide behind from panel 4 on).
template <typename T>
__device__ __forceinline__ Frag<T> frag_xblock(const DiagShared<T>& sh, int M, int J, int li, int lg) {
    Frag<T> f;
    if (J < M) {
        const T* blk = sh.S + LBLK(M, J);
#pragma unroll
        for (int st = 0; st < 4; ++st) f.v[st] = blk[(4 * lg + st) * 17 + li];
    } else if (J == M) {
        f = frag_leafinv<T>(sh, J, li, lg, true);
    } else {
#pragma unroll
        for (int st = 0; st < 4; ++st) f.v[st] = (T)0;
    }
    return f;
}
template <typename T>
__device__ __forceinline__ void block_to_lds(T* dst, const typename MF<T>::acc_t& a, T sgn, int lane) {
    const int li = lane & 15;
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[MF<T>::row_of(lane, q) * 17 + li] = sgn * a[q];
}

// t (column-major enumeration of the 36 lower blocks) of diagonal block (I, I); block t belongs to bulk wave t % 3
__device__ __forceinline__ int diag_diag_index(int I) { return I * 8 - I * (I - 1) / 2; }
// The panel products P_b = A_bp X_pp^T (= L_bp, in place in the LDS image) of step p are shared out like this: block
// p+1 goes to the wave that holds block (p+1, p+1); the others, b = p+2 .. 7, go round the three remaining waves (wave 0
// included) in wave order, so each has at most two: b0 = p + 2 + slot and b0 + 3
__device__ __forceinline__ int diag_panel_slot(int wave, int own) { return wave < own ? wave : wave - 1; }
template <typename T>
__device__ __forceinline__ void diag_two_panel_blocks(DiagShared<T>& sh, int p, int slot, int lane) {
    using F = MF<T>;
    using acc_t = typename F::acc_t;
    const int li = lane & 15, lg = lane >> 4;
    const int b0 = p + 2 + slot, b1 = b0 + 3;
    if (b0 >= 8) return;
    const bool two = b1 < 8;
    T* A0 = sh.S + LBLK(b0, p);
    T* A1 = sh.S + LBLK(two ? b1 : b0, p);
    const Frag<T> x = frag_leafinv_T<T>(sh, p, li, lg);
    const Frag<T> a0 = frag_rows<T>(A0, li, lg, true), a1 = frag_rows<T>(A1, li, lg, two);
    acc_t c0, c1;
#pragma unroll
    for (int q = 0; q < 4; ++q) c0[q] = c1[q] = (T)0;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        c0 = F::mfma(a0.v[st], x.v[st], c0);
        c1 = F::mfma(a1.v[st], x.v[st], c1);
    }
    block_to_lds<T>(A0, c0, (T)1, lane);
    if (two) block_to_lds<T>(A1, c1, (T)1, lane);
}

// ---- wave 0: the eight leaves, nothing else on its way.  Between leaf p and leaf p+1 it only sits out the short
// section in which the others make block (p+1, p+1) final (and takes its share of the panel products).
template <typename T>
__device__ __forceinline__ void diag_leaf_wave(DiagShared<T>& sh, const T* A, int64_t lda, int lane) {
    // Leaf 0 does not wait for the LDS image: its 16 x 16 block comes straight from global memory, so it runs beside the
    // other waves' loads of the 36 blocks and their staging of block column 0 (which leaves block (0, 0) to this wave);
    // barrier B0 (block column 0 in LDS) falls after it.
    for (int p = 0; p < 8; ++p) {
        ALGP_STAMP(8 + 3 * p + 0);
        if (p == 0) {
            diag_leaf<T>(sh, 0, lane, A, lda);
            diag_barrier();                                    // B0
        } else {
            diag_leaf<T>(sh, p, lane);
        }
        ALGP_STAMP(8 + 3 * p + 1);
        diag_barrier();                                        // B2(p): L_pp, X_pp are in LDS
        if (p < 6) diag_two_panel_blocks<T>(sh, p, diag_panel_slot(0, 1 + diag_diag_index(p + 1) % 3), lane);
        ALGP_STAMP(8 + 3 * p + 2);
        diag_barrier();                                        // Bx(p): block (p+1, p+1) is final and in LDS
    }
}

// ---- waves 1-3 (w1 = 0..2): the 36 blocks of the trailing part in accumulator registers (12 each, NEGATED so that the
// rank-16 updates are plain accumulating MFMAs), the panel products, the updates, the inverse and all global stores --
// everything except the leaves, and all of it but a short section per panel while wave 0 runs the next leaf.
// W1 is a template parameter so that, once the loops over the 12 blocks are unrolled, each block's coordinates are
// constants (no index arrays in scalar registers, no address arithmetic at run time).
template <typename T, int W1>
__device__ __forceinline__ void diag_bulk_wave(DiagShared<T>& sh, T* A, int64_t lda, T* inv_out, int lane) {
    constexpr int w1 = W1;
    using F = MF<T>;
    using acc_t = typename F::acc_t;
    const int li = lane & 15, lg = lane >> 4;
    acc_t acc[12];
    int bis[12], bks[12];
    // row_of(lane, q) = row_of(lane, 0) + rstep q: one 32-bit per-lane offset, the rest is wave-uniform (see the block movers)
    const int rstep = F::row_of(0, 1) - F::row_of(0, 0);
    const uint32_t vo = (uint32_t)F::row_of(lane, 0) * (uint32_t)lda + (uint32_t)li;
#pragma unroll
    for (int u = 0; u < 12; ++u) {
        diag_block_of(3 * u + w1, bis[u], bks[u]);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[u][q] = -(A + (int64_t)(16 * bis[u] + rstep * q) * lda + 16 * bks[u])[vo];
    }
    // block column 0 -> LDS (block (0, 0) is wave 0's: its leaf reads it from global memory and writes L_00, X_00 there)
#pragma unroll
    for (int u = 0; u < 12; ++u)
        if (bks[u] == 0 && bis[u] != 0) block_to_lds<T>(sh.S + LBLK(bis[u], 0), acc[u], (T)-1, lane);
    diag_barrier();                                            // B0
    acc_t tacc[3];                                             // T_pJ = sum_K L_pK X_KJ of the inverse's row p, J = w1 + 3 m
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) tacc[m][q] = (T)0;
    for (int p = 0; p < 8; ++p) {
        ALGP_STAMPW(4 * p + 0);
        diag_barrier();                                        // B2(p): leaf p is done
        ALGP_STAMPW(4 * p + 1);
        // ================= the short section wave 0 waits for =================
        const int own = p < 7 ? 1 + diag_diag_index(p + 1) % 3 : 0;
        const bool mine = own == w1 + 1;
        // inverse row p: X_pJ = -X_pp T_pJ (J = w1, w1+3, w1+6 below p) goes over L_pJ -- row p of L went to global memory in
        // the previous shadow and every product that reads it is done.  A[i][k] = X_pp[i][k], k = the row of T this lane holds.
        // (Round 4 measured two rearrangements of this section, neither kept: all of a wave's products as ONE batch of
        // reads / MFMAs / writes -- 1 900 -> 1 760 cycles where the wave owns block (p+1, p+1), 1 840 -> 2 130 where it does not,
        // block unchanged -- and the owner's last update fed from the accumulator of the transposed panel product.)
        acc_t xo[3];
        {
            T xa[4];
            const T* D = sh.S + LBLK(p, p);
            const T dv = sh.dinv[16 * p + li];
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int k = F::row_of(lane, st);
                xa[st] = pick3<T>(D[k * 17 + li], k < li, dv, k == li);     // X_pp[li][k], k <= li
            }
            acc_t pc;                                          // the owner's panel product rides in the same batch (zeros otherwise)
            const Frag<T> pa = frag_rows<T>(sh.S + LBLK(mine ? p + 1 : 7, p), li, lg, mine);
            const Frag<T> px = frag_leafinv_T<T>(sh, p, li, lg);
#pragma unroll
            for (int q = 0; q < 4; ++q) pc[q] = (T)0;
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) xo[m][q] = (T)0;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                pc = F::mfma(pa.v[st], px.v[st], pc);
#pragma unroll
                for (int m = 0; m < 3; ++m) xo[m] = F::mfma(xa[st], tacc[m][st], xo[m]);
            }
            if (mine) block_to_lds<T>(sh.S + LBLK(p + 1, p), pc, (T)1, lane);
#pragma unroll
            for (int m = 0; m < 3; ++m)
                if (w1 + 3 * m < p) block_to_lds<T>(sh.S + LBLK(p, w1 + 3 * m), xo[m], (T)-1, lane);
        }
        if (mine) {
            // the last update of block (p+1, p+1) with the L_(p+1)p just written, then the block to LDS for the next leaf
            const Frag<T> f = frag_rows<T>(sh.S + LBLK(p + 1, p), li, lg, true);
#pragma unroll
            for (int u = 0; u < 12; ++u)
                if (bis[u] == p + 1 && bks[u] == p + 1) {
#pragma unroll
                    for (int st = 0; st < 4; ++st) acc[u] = F::mfma(f.v[st], f.v[st], acc[u]);
                    block_to_lds<T>(sh.S + LBLK(p + 1, p + 1), acc[u], (T)-1, lane);
                }
        } else if (p < 6) {
            diag_two_panel_blocks<T>(sh, p, diag_panel_slot(w1 + 1, own), lane);
        }
        ALGP_STAMPW(4 * p + 2);
        diag_barrier();                                        // Bx(p)
        ALGP_STAMPW(4 * p + 3);
        // ================= in the shadow of leaf p+1 =================
        if (p < 7) {
            // rank-16 update with panel p of every block right of it (block (p+1, p+1) has had its own), block by
            // block (no LDS write in between: the compiler is free to run the next block's operand reads ahead)
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                if (!(bks[u] > p && !(bis[u] == p + 1 && bks[u] == p + 1))) continue;
                const Frag<T> fa = frag_rows<T>(sh.S + LBLK(bis[u], p), li, lg, true);
                const Frag<T> fb = frag_rows<T>(sh.S + LBLK(bks[u], p), li, lg, true);
#pragma unroll
                for (int st = 0; st < 4; ++st) acc[u] = F::mfma(fa.v[st], fb.v[st], acc[u]);
            }
            // column p+1 below its diagonal block goes to LDS for the next panel products
#pragma unroll
            for (int u = 0; u < 12; ++u)
                if (bks[u] == p + 1 && bis[u] > p + 1) block_to_lds<T>(sh.S + LBLK(bis[u], p + 1), acc[u], (T)-1, lane);
            // finished parts go out: row p+1 of L left of its diagonal block
            for (int J = w1; J <= p; J += 3)
                store_block_rowmajor<T>(sh.S + LBLK(p + 1, J), A + (int64_t)(16 * (p + 1)) * lda + 16 * J, lda, lane);
        }
        // row p of the inverse and the diagonal block of L
        {
            T* xg = inv_out + (int64_t)(16 * p) * 128;
            for (int J = w1; J < 8; J += 3) {
                if (J < p) store_block_rowmajor<T>(sh.S + LBLK(p, J), xg + 16 * J, 128, lane);
                else if (J == p) store_block_leafinv<T>(sh, p, xg + 16 * J, 128, lane);
                else store_block_zero<T>(xg + 16 * J, 128, lane);
            }
            if (w1 == (p % 3)) store_block_lower<T>(sh.S + LBLK(p, p), A + (int64_t)(16 * p) * lda + 16 * p, lda, lane);
        }
        if (p < 7) {
            // T_(p+1)J = sum_{M = J..p} L_(p+1)M X_MJ for the inverse's next row (X rows <= p are complete): M outermost,
            // the three J of this wave as independent chains; a term that does not exist (M < J) multiplies zeros
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) tacc[m][q] = (T)0;
            // (Reading term M + 1's fragments before issuing term M's MFMAs, by hand -- hipcc does not pipeline this loop, each
            // term sits out its LDS round trip in front of its 12 MFMAs -- was measured in round 4: the block got SLOWER,
            // 59 200 -> 67 100 cycles in fp32, 80 600 -> 103 000 in fp64: the branches of frag_xblock then wait per case.)
            for (int M = w1; M <= p; ++M) {
                const Frag<T> fl = frag_rows<T>(sh.S + LBLK(p + 1, M), li, lg, true);
                Frag<T> fx[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const int J = w1 + 3 * m;
                    fx[m] = frag_xblock<T>(sh, M, J, li, lg);
                }
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int m = 0; m < 3; ++m) tacc[m] = F::mfma(fl.v[st], fx[m].v[st], tacc[m]);
            }
        }
    }
}

// Factor the 128 x 128 block at A (leading dimension lda) in place and write its inverse (dense, 128 x 128, ld 128,
// zeros above the diagonal) to inv_out.  The eight 16 x 16 leaves are one dependency chain (every pivot waits for the
// one before it), so wave 0 does nothing else; between leaf p and leaf p+1 lie only L_(p+1)p = A_(p+1)p X_pp^T, the
// last rank-16 update of block (p+1, p+1) -- both by the wave that holds that block -- and two barriers.  Waves 1-3
// do the rest in the shadow of the running leaf (see diag_bulk_wave): 48.9 -> see profiles us per block in fp64.
// PIVOTS: instead of the block's log-determinant, the 128 pivots d_j go to pivots_out (as doubles) and the caller takes
// the logarithms elsewhere -- the one-launch factorisation does: log() in here costs a reduction and two barriers on the
// critical path of every column step, and its polynomial constants sat in registers across the whole kernel.
template <typename T, bool PIVOTS = false>
__device__ __forceinline__ void diag128_factor(DiagShared<T>& sh, T* A, int64_t lda, T* inv_out, double* logdet_acc,
                                               bool logdet_atomic, int* info, int64_t block_row0, double* pivots_out = nullptr) {
    // tid through an opaque statement: inside a caller's loop (the one-launch factorisation's leader) everything this
    // routine derives from the lane index -- three dozen per-lane masks and offsets -- is loop-invariant, and hoisted out
    // of the loop it stays live across the whole routine on top of its own peak: 17 VGPRs spilled to scratch, reloaded in
    // the middle of the leaf chain.  Recomputing them per call costs a few dozen VALU instructions.
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) sh.bad = 0;                                  // ordered before the leaves by barrier B0
    ALGP_STAMP(0);
    if (wave == 0) diag_leaf_wave<T>(sh, A, lda, lane);
    else if (wave == 1) diag_bulk_wave<T, 0>(sh, A, lda, inv_out, lane);
    else if (wave == 2) diag_bulk_wave<T, 1>(sh, A, lda, inv_out, lane);
    else diag_bulk_wave<T, 2>(sh, A, lda, inv_out, lane);
    ALGP_STAMP(6);
    // every pivot was written before the last barrier of the loop; every store of this workgroup is issued when this
    // returns (the caller drains them)
    if (PIVOTS) {
        if (tid < 128) pivots_out[tid] = (double)sh.dd[tid];
        if (tid == 0 && sh.bad) atomicCAS(info, 0, (int)(block_row0 + sh.bad));
        ALGP_STAMP(7);
        diag_barrier();                                        // sh is free for the next block
        return;
    }
    double v = (tid < 128) ? log((double)sh.dd[tid]) : 0.0;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((tid & 63) == 0) sh.red[tid >> 6] = v;
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
