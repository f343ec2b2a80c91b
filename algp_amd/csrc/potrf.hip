// potrf.hip -- blocked Cholesky, blocked triangular solves, and their 128 x 128 diagonal kernel.
//
// Replaces the reference's np.linalg.inv (utils.py:300: LU + getri) and np.linalg.slogdet
// (utils.py:193: LU) with one SPD factorisation S = L L^T kept on the device.
//
// Blocked right-looking Cholesky with block NB = 128:
//   for each block column kb:
//     potrf_diag : L_kk = chol(S_kk) in LDS, plus inv(L_kk) and sum(log pivots)        (1 workgroup)
//     panel      : L_ik = S_ik * inv(L_kk)^T           for all i > kb   (MFMA GEMM, in place)
//     trailing   : S_ij -= L_ik L_jk^T                 for i >= j > kb  (MFMA GEMM, lower tiles)
// Triangular solves never substitute element by element: every diagonal block is applied through
// its explicit 128 x 128 inverse (a GEMM), which keeps all O(n^2 m) work on the matrix cores.
#include "common.h"
#include "mfma.h"
#include "vecops.h"

namespace algp {

// Diagnostic builds only (tools/potrf_stamp.hip): cycle stamps of thread 0 at phase boundaries.
#ifdef ALGP_POTRF_STAMPS
__device__ unsigned long long g_potrf_stamps[64];
#define ALGP_STAMP(k) do { if (threadIdx.x == 0) g_potrf_stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ALGP_STAMP(k) do { } while (0)
#endif

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

// ---------------------------------------------------------------------------------------------
// Diagonal block: factor + invert a 128 x 128 SPD block inside one 256-thread workgroup's LDS.
//   S[128][129] holds the block (one element of padding per row: column walks hit distinct banks).
//   Factor, 16 columns at a time (8 panels):
//     - LDL-style column sweep restricted to the panel: for column j with pivot d_j = S[j][j],
//       S[i][k] -= S[i][j] S[k][j] / d_j for j < k < panel end, k <= i   (one barrier per column,
//       but only 16 columns wide and 4 waves deep);
//     - scale the panel: L[i][c] = S[i][c] / sqrt(d_c);
//     - rank-16 update of everything right of the panel with 4 x 4 register tiles (one barrier).
//   Inverse X = L^-1 by 16 x 16 blocks: the 8 diagonal blocks by per-column substitution (no
//   barrier), then block row I: T = sum_K L_IK X_KJ for all J < I, X_IJ = -X_II T (two barriers).
//   X[a][b] (a > b) is kept transposed in the free strict upper triangle, S[b][a]; X[a][a] = dinv[a].
// ---------------------------------------------------------------------------------------------
template <typename T, bool FACTOR>
__global__ __launch_bounds__(256) void potrf_diag_kernel(T* A, int64_t lda, T* inv_out, double* logdet_acc,
                                                          int* info, int64_t block_row0) {
    __shared__ T S[128 * 129];
    __shared__ T dd[128];
    __shared__ T dinv[128];
    __shared__ double red[4];
    __shared__ T prow[2 * 18];
    __shared__ int bad;
    const int tid = threadIdx.x;
    ALGP_STAMP(0);
    if (tid == 0) bad = 0;
    {
        // block load: 16-byte vectors, 8 loads in flight per thread (a load-per-iteration loop
        // serialises 64 memory round trips and alone costs >100 us)
        constexpr int VEC = 16 / sizeof(T);
        typedef T vec_t __attribute__((ext_vector_type(VEC)));
        constexpr int VPR = 128 / VEC;                  // vectors per row
        constexpr int NV = 128 * VPR / 256;             // vectors per thread
#pragma unroll
        for (int b0 = 0; b0 < NV; b0 += 8) {
            vec_t tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int v = tid + 256 * (b0 + u);
                tmp[u] = *reinterpret_cast<const vec_t*>(A + (int64_t)(v / VPR) * lda + (v % VPR) * VEC);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int v = tid + 256 * (b0 + u);
#pragma unroll
                for (int e = 0; e < VEC; ++e) S[(v / VPR) * 129 + (v % VPR) * VEC + e] = tmp[u][e];
            }
        }
    }
    __syncthreads();
    ALGP_STAMP(1);

    if (FACTOR) {
        for (int k0 = 0; k0 < 128; k0 += 16) {
            const int k1 = k0 + 16;
            if (k0 == 0) ALGP_STAMP(2);
            // ---- panel sweep: thread `rowid` owns row k0+rowid of the panel in registers; the pivot
            //      row travels through a double-buffered LDS line: one barrier per column ----
            const int rowid = tid;                       // threads >= 128-k0 idle here
            const bool active = rowid < 128 - k0;
            T a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = active ? S[(k0 + rowid) * 129 + k0 + c] : (T)0;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                T* line = prow + (jj & 1) * 18;
                // column jj of the diagonal 16 x 16 block (lower entries only: the upper triangle is
                // never kept valid) is gathered from the threads that own those rows
                if (rowid > jj && rowid < 16) line[rowid] = a[jj];
                if (rowid == jj) {
                    const T d = a[jj];
                    line[16] = fast_rcp(d);
                    dd[k0 + jj] = d;
                    if (!(d > (T)0) && bad == 0) bad = k0 + jj + 1;
                }
                __syncthreads();
                if (active && rowid > jj) {
                    const T ci = a[jj] * line[16];
#pragma unroll
                    for (int c = jj + 1; c < 16; ++c) a[c] -= ci * line[c];
                }
            }
            // ---- write the scaled panel back: L[i][c] = a[c] / sqrt(d_c) ----
            __syncthreads();                             // dd[k0..k1) complete
            if (tid < 16) dinv[k0 + tid] = fast_rsqrt(dd[k0 + tid]);
            __syncthreads();
            if (k0 == 0) ALGP_STAMP(3);
            if (active) {
                const int i = k0 + rowid;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    if (i > k0 + c) S[i * 129 + k0 + c] = a[c] * dinv[k0 + c];
                    else if (i == k0 + c) S[i * 129 + i] = dd[i] * dinv[i];        // sqrt(d) = d / sqrt(d)
                }
            }
            __syncthreads();
            if (k0 == 0) ALGP_STAMP(4);
            // ---- rank-16 update of everything right of the panel on the matrix cores: one 16 x 16
            //      output tile = 4 MFMAs (K = 16); operands are 16 consecutive rows at one column
            //      (row stride 129: conflict free); C tiles are read-modify-written in the C layout ----
            const int r = 128 - k1;
            if (r > 0) {
                using F = MF<T>;
                const int lane = tid & 63, wave = tid >> 6;
                const int nb16 = r >> 4, ntile16 = nb16 * (nb16 + 1) / 2;
                for (int t = wave; t < ntile16; t += 4) {
                    int ti = 0;
                    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
                    const int tk = t - ti * (ti + 1) / 2;
                    const int i0 = k1 + 16 * ti, c0 = k1 + 16 * tk;
                    typename F::acc_t acc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = (T)0;
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const T av = S[(i0 + (lane & 15)) * 129 + k0 + 4 * st + (lane >> 4)];
                        const T bv = S[(c0 + (lane & 15)) * 129 + k0 + 4 * st + (lane >> 4)];
                        acc = F::mfma(av, bv, acc);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        T* dst = S + (i0 + F::row_of(lane, q)) * 129 + c0 + (lane & 15);
                        *dst -= acc[q];
                    }
                }
                __syncthreads();
                if (k0 == 0) ALGP_STAMP(5);
            }
        }
        ALGP_STAMP(6);
        {
            double v = (tid < 128) ? log((double)dd[tid]) : 0.0;
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if ((tid & 63) == 0) red[tid >> 6] = v;
        }
        __syncthreads();
        if (tid == 0) {
            atomicAdd(logdet_acc, red[0] + red[1]);
            if (bad) atomicCAS(info, 0, (int)(block_row0 + bad));
        }
        for (int e = tid; e < 128 * 128; e += 256) {
            const int i = e >> 7, j = e & 127;
            if (j <= i) A[(int64_t)i * lda + j] = S[i * 129 + j];
        }
    } else {
        if (tid < 128) dinv[tid] = (T)1 / S[tid * 129 + tid];
    }
    __syncthreads();
    ALGP_STAMP(7);

    // ---- inverse: diagonal 16 x 16 blocks, one column per thread, values kept in registers ----
    if (tid < 128) {
        const int base = (tid >> 4) * 16, c = tid & 15;
        T x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = (T)0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i == c) x[i] = dinv[base + c];
            if (i > c) {
                T sum = (T)0;
#pragma unroll
                for (int k = 0; k < i; ++k)
                    if (k >= c) sum += S[(base + i) * 129 + base + k] * x[k];
                x[i] = -sum * dinv[base + i];
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i > c) S[(base + c) * 129 + base + i] = x[i];
    }
    __syncthreads();
    ALGP_STAMP(8);
    // ---- block rows 1..7 on the matrix cores.  For block (I, J), J < I:
    //        T    = sum_{K=J}^{I-1} L_IK X_KJ      (A = L_IK rows r0+m; B[k][b] = X_KJ[k][b] = S[16J+b][16K+k])
    //        X_IJ = -X_II T                          (A = X_II; B = T straight from the accumulator: for the
    //                                                 k-step s the lane's register s IS row k of T in its column)
    //      X_JJ / X_II are lower triangular with the diagonal in dinv and zeros above. ----
    {
        using F = MF<T>;
        const int lane = tid & 63, wave = tid >> 6;
        const int li = lane & 15, lg = lane >> 4;
        for (int I = 1; I < 8; ++I) {
            const int r0 = 16 * I;
            for (int J = wave; J < I; J += 4) {
                const int b0 = 16 * J;
                typename F::acc_t acc;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = (T)0;
                // K = J: X_JJ[k][b] for k >= b only
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int k = 4 * st + lg;                        // row of X_JJ, b = li its column
                    const T av = S[(r0 + li) * 129 + b0 + k];
                    T bv = (T)0;
                    if (k > li) bv = S[(b0 + li) * 129 + b0 + k];
                    else if (k == li) bv = dinv[b0 + li];
                    acc = F::mfma(av, bv, acc);
                }
                for (int K = J + 1; K < I; ++K) {
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const int k = 16 * K + 4 * st + lg;
                        acc = F::mfma(S[(r0 + li) * 129 + k], S[(b0 + li) * 129 + k], acc);
                    }
                }
                // X_IJ = -X_II T : k-step s pairs A[m'][k] with the accumulator register s, whose row is
                // row_of(lane, s); A[m' = li][k] = X_II[li][k] = S[r0+k][r0+li] (k < li), dinv (k == li), 0 (k > li)
                typename F::acc_t out;
#pragma unroll
                for (int q = 0; q < 4; ++q) out[q] = (T)0;
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int k = F::row_of(lane, st);
                    T av = (T)0;
                    if (k < li) av = S[(r0 + k) * 129 + r0 + li];
                    else if (k == li) av = dinv[r0 + li];
                    out = F::mfma(av, acc[st], out);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) S[(b0 + li) * 129 + r0 + F::row_of(lane, q)] = -out[q];
            }
            __syncthreads();
            ALGP_STAMP(9 + I);
        }
    }
    for (int e = tid; e < 128 * 128; e += 256) {
        const int i = e >> 7, j = e & 127;
        T v = (T)0;
        if (j < i) v = S[j * 129 + i];
        else if (j == i) v = dinv[i];
        inv_out[i * 128 + j] = v;
    }
    ALGP_STAMP(17);
}

template <typename T>
int potrf_diag_launch(algp_ctx* c, T* A, int64_t lda, T* inv_out, double* logdet_acc, int* info,
                      int64_t block_row0) {
    ProfScope ps(c, ALGP_PROF_POTRF_DIAG, 128.0 * 128.0 * 128.0, sizeof(T) * 3.0 * 128.0 * 128.0);
    hipLaunchKernelGGL((potrf_diag_kernel<T, true>), dim3(1), dim3(256), 0, c->cur, A, lda, inv_out, logdet_acc,
                       info, block_row0);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int potrf_diag_launch<double>(algp_ctx*, double*, int64_t, double*, double*, int*, int64_t);
template int potrf_diag_launch<float>(algp_ctx*, float*, int64_t, float*, double*, int*, int64_t);

template <typename T>
int trinv_diag_launch(algp_ctx* c, const T* A, int64_t lda, T* inv_out) {
    hipLaunchKernelGGL((potrf_diag_kernel<T, false>), dim3(1), dim3(256), 0, c->cur, const_cast<T*>(A), lda, inv_out,
                       (double*)nullptr, (int*)nullptr, (int64_t)0);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int trinv_diag_launch<double>(algp_ctx*, const double*, int64_t, double*);
template int trinv_diag_launch<float>(algp_ctx*, const float*, int64_t, float*);

// ---------------------------------------------------------------------------------------------
// Two-level blocking.  The GEMM tile is 128 wide, but updating with K = 128 re-reads and re-writes
// the whole trailing matrix (Cholesky) or re-streams all solved columns of X (TRSM) once per 128
// columns: that is HBM-bound (16 flop/byte).  So blocks of WB = 512 columns are processed as a unit:
// inside a block the 128-wide steps touch only the block's own columns; everything outside is
// updated once per block with K = 512 (Cholesky) / produced by one n = 512 GEMM whose four column
// tiles share each A row-panel through the XCD's L2 (TRSM).
// ---------------------------------------------------------------------------------------------
constexpr int WB = 512;

// factor block column [j0, j0+w) over all rows >= j0, 128 columns at a time (launches go to c->cur)
template <typename T>
static int chol_panel(algp_ctx* c, T* A, int64_t npad, int64_t ld, T* invD, double* logdet_acc, int* info,
                      int64_t j0, int64_t w) {
    for (int64_t k0 = j0; k0 < j0 + w; k0 += NB) {
        T* Akk = A + k0 * ld + k0;
        T* inv = invD + (k0 / NB) * NB * NB;
        ALGP_TRY(potrf_diag_launch<T>(c, Akk, ld, inv, logdet_acc, info, k0));
        const int64_t mrem = npad - (k0 + NB);
        if (mrem <= 0) continue;
        T* P = A + (k0 + NB) * ld + k0;                          // rows below the diagonal block
        ALGP_TRY(gemm_nt_launch<T>(c, ALGP_PROF_GEMM_CHOL, mrem, NB, NB, (T)1, P, ld, inv, NB, (T)0, nullptr, 0, P,
                                   ld, 0));
        const int64_t wrem = j0 + w - (k0 + NB);                 // columns of this block still to do
        if (wrem > 0) {
            // A[k0+NB:, k0+NB : j0+w] -= P * P[0:wrem]^T   (K = 128, only inside the block column)
            T* Cw = A + (k0 + NB) * ld + (k0 + NB);
            ALGP_TRY(gemm_nt_launch<T>(c, ALGP_PROF_GEMM_CHOL, mrem, wrem, NB, (T)-1, P, ld, P, ld, (T)1, Cw, ld, Cw,
                                       ld, 0));
        }
    }
    return ALGP_OK;
}

// Right-looking blocked Cholesky, two-level (512 / 128).
//
// Look-ahead (factor block column J+1 on a second stream while the main stream applies block J to
// the rest) was built and measured in round 1 and is NOT used: the fp64 diagonal kernel needs 134 KB
// of LDS, i.e. an empty CU, and the trailing-update GEMM keeps two 64 KB workgroups on every CU, so
// the panel never starts before the GEMM drains (19.8 ms vs 17.6 ms serial at N = 10 000; reserving
// CUs with a CU-masked stream: 22.1 ms).  It becomes useful once the diagonal kernel fits beside a
// GEMM workgroup (<= 96 KB: lower-triangular 16 x 16 block storage).
template <typename T>
int cholesky_blocked(algp_ctx* c, T* A, int64_t npad, int64_t ld, T* invD, double* logdet_acc, int* info) {
    for (int64_t j0 = 0; j0 < npad; j0 += WB) {
        const int64_t w = (npad - j0 < WB) ? npad - j0 : WB;
        ALGP_TRY(chol_panel<T>(c, A, npad, ld, invD, logdet_acc, info, j0, w));
        const int64_t mrem = npad - (j0 + w);
        if (mrem > 0) {
            // trailing update with the whole block: A22 -= P_blk P_blk^T, K = w, lower tiles
            const T* Pb = A + (j0 + w) * ld + j0;
            T* A22 = A + (j0 + w) * ld + (j0 + w);
            ALGP_TRY(gemm_nt_launch<T>(c, ALGP_PROF_GEMM_CHOL, mrem, mrem, w, (T)-1, Pb, ld, Pb, ld, (T)1, A22, ld,
                                       A22, ld, 1));
        }
    }
    return ALGP_OK;
}
template int cholesky_blocked<double>(algp_ctx*, double*, int64_t, int64_t, double*, double*, int*);
template int cholesky_blocked<float>(algp_ctx*, float*, int64_t, int64_t, float*, double*, int*);

// X <- X L^-T, left-looking over 512-wide column blocks:
//   X_J <- X_J - X_{0:J} L_{J,0:J}^T            (one GEMM, n = 512)
//   inside J, 128 columns at a time: X_k <- (X_k - X_{J0:k} L_{k,J0:k}^T) inv(L_kk)^T
template <typename T>
int trsm_blocked(algp_ctx* c, int klass, T* X, int64_t mpad, int64_t ldx, const T* L, int64_t npad,
                 int64_t ldl, const T* invD, int64_t col_start) {
    // col_start (multiple of 128): columns [0, col_start) of X already hold the solution
    for (int64_t j0 = 0; j0 < npad; j0 += WB) {
        const int64_t w = (npad - j0 < WB) ? npad - j0 : WB;
        if (j0 + w <= col_start) continue;
        const int64_t cs = j0 > col_start ? j0 : col_start;         // first column of this block to solve
        T* Xj = X + j0;
        if (j0 > 0)
            ALGP_TRY(gemm_nt_launch<T>(c, klass, mpad, j0 + w - cs, j0, (T)-1, X, ldx, L + cs * ldl, ldl, (T)1, X + cs,
                                       ldx, X + cs, ldx, 0));
        for (int64_t k0 = cs; k0 < j0 + w; k0 += NB) {
            T* Xk = X + k0;
            if (k0 > j0)
                ALGP_TRY(gemm_nt_launch<T>(c, klass, mpad, NB, k0 - j0, (T)-1, Xj, ldx, L + k0 * ldl + j0, ldl, (T)1,
                                           Xk, ldx, Xk, ldx, 0));
            ALGP_TRY(gemm_nt_launch<T>(c, klass, mpad, NB, NB, (T)1, Xk, ldx, invD + (k0 / NB) * NB * NB, NB, (T)0,
                                       nullptr, 0, Xk, ldx, 0));
        }
    }
    return ALGP_OK;
}
template int trsm_blocked<double>(algp_ctx*, int, double*, int64_t, int64_t, const double*, int64_t, int64_t,
                                  const double*, int64_t);
template int trsm_blocked<float>(algp_ctx*, int, float*, int64_t, int64_t, const float*, int64_t, int64_t,
                                 const float*, int64_t);

// ---------------------------------------------------------------------------------------------
// Vector solves (HBM-bound: each reads the lower triangle of L once).
//   forward  b <- L^-1 b, right-looking: x_k = inv(L_kk) b_k ; b_{k+1:} -= L_{k+1:,k} x_k
//   backward b <- L^-T b, right-looking: x_k = inv(L_kk)^T b_k ; b_{0:k} -= L_{k,0:k}^T x_k
// ---------------------------------------------------------------------------------------------
template <typename T, bool TRANS>
__global__ __launch_bounds__(128) void diag_matvec_kernel(const T* inv, T* b) {
    // x = inv * b (TRANS: inv^T * b) for one 128-block, in place
    __shared__ T xb[128];
    const int t = threadIdx.x;
    xb[t] = b[t];
    __syncthreads();
    // the inverse block is lower triangular with explicit zeros above: walk all 128 terms so the
    // loads are independent and can be issued in batches (a data-dependent trip count serialises them)
    T s = (T)0;
    if (!TRANS) {
#pragma unroll 16
        for (int k = 0; k < 128; ++k) s += inv[t * 128 + k] * xb[k];
    } else {
#pragma unroll 16
        for (int k = 0; k < 128; ++k) s += inv[k * 128 + t] * xb[k];
    }
    b[t] = s;
}

// rows [0, mrem) of the panel P (mrem x 128, ld): out[r] -= dot(P[r][0:128], x)
template <typename T>
__global__ __launch_bounds__(256) void panel_gemv_kernel(const T* P, int64_t ld, int64_t mrem, const T* x, T* out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    const T x0 = x[2 * lane], x1 = x[2 * lane + 1];
    for (int64_t r = wave; r < mrem; r += nw) {
        const T* row = P + r * ld;
        T s = row[2 * lane] * x0 + row[2 * lane + 1] * x1;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) out[r] -= s;
    }
}

// columns [0, ncol) of the row panel R (128 x ncol, ld): out[cidx] -= sum_r R[r][cidx] * x[r]
template <typename T>
__global__ __launch_bounds__(256) void panel_gemv_t_kernel(const T* R, int64_t ld, int64_t ncol, const T* x, T* out) {
    __shared__ T xs[128];
    if (threadIdx.x < 128) xs[threadIdx.x] = x[threadIdx.x];
    __syncthreads();
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= ncol) return;
    T s = (T)0;
#pragma unroll 8
    for (int r = 0; r < 128; ++r) s += R[(int64_t)r * ld + col] * xs[r];
    out[col] -= s;
}

template <typename T>
int trsv_forward(algp_ctx* c, const T* L, int64_t npad, int64_t ldl, const T* invD, T* b) {
    const int64_t nblk = npad / NB;
    ProfScope ps(c, ALGP_PROF_TRSV, (double)npad * npad, sizeof(T) * 0.5 * (double)npad * npad);
    for (int64_t kb = 0; kb < nblk; ++kb) {
        hipLaunchKernelGGL((diag_matvec_kernel<T, false>), dim3(1), dim3(128), 0, c->cur, invD + kb * NB * NB,
                           b + kb * NB);
        const int64_t mrem = npad - (kb + 1) * NB;
        if (mrem > 0) {
            const int grid = (int)((mrem + 3) / 4 < 1024 ? (mrem + 3) / 4 : 1024);
            hipLaunchKernelGGL(panel_gemv_kernel<T>, dim3(grid), dim3(256), 0, c->cur,
                               L + (kb + 1) * NB * ldl + kb * NB, ldl, mrem, b + kb * NB, b + (kb + 1) * NB);
        }
    }
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template <typename T>
int trsv_backward(algp_ctx* c, const T* L, int64_t npad, int64_t ldl, const T* invD, T* b) {
    const int64_t nblk = npad / NB;
    ProfScope ps(c, ALGP_PROF_TRSV, (double)npad * npad, sizeof(T) * 0.5 * (double)npad * npad);
    for (int64_t kb = nblk - 1; kb >= 0; --kb) {
        hipLaunchKernelGGL((diag_matvec_kernel<T, true>), dim3(1), dim3(128), 0, c->cur, invD + kb * NB * NB,
                           b + kb * NB);
        const int64_t ncol = kb * NB;
        if (ncol > 0) {
            hipLaunchKernelGGL(panel_gemv_t_kernel<T>, dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, c->cur,
                               L + kb * NB * ldl, ldl, ncol, b + kb * NB, b);
        }
    }
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int trsv_forward<double>(algp_ctx*, const double*, int64_t, int64_t, const double*, double*);
template int trsv_forward<float>(algp_ctx*, const float*, int64_t, int64_t, const float*, float*);
template int trsv_backward<double>(algp_ctx*, const double*, int64_t, int64_t, const double*, double*);
template int trsv_backward<float>(algp_ctx*, const float*, int64_t, int64_t, const float*, float*);

}  // namespace algp
