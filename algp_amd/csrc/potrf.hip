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
#include "vecops.h"

namespace algp {

// ---------------------------------------------------------------------------------------------
// Diagonal block: factor + invert a 128 x 128 SPD block inside one workgroup's LDS.
//   S[128][129] holds the block (padding 1 element/row: column walks hit distinct banks).
//   Phase 1 (LDL-style, one barrier per column): for column j with pivot d_j = S[j][j],
//            S[i][k] -= S[i][j] * S[k][j] / d_j  for j < k <= i.  Columns stay unscaled.
//   Phase 2: L[i][j] = S[i][j] / sqrt(d_j), L[j][j] = sqrt(d_j).
//   Phase 3: X = L^-1 row by row; X[i][j] (j < i) is kept transposed in the (free) strict upper
//            triangle S[j][i]; each of the 128 columns is reduced by 8 lanes.
// ---------------------------------------------------------------------------------------------
template <typename T, bool FACTOR>
__global__ __launch_bounds__(1024) void potrf_diag_kernel(T* A, int64_t lda, T* inv_out, double* logdet_acc,
                                                           int* info, int64_t block_row0) {
    __shared__ T S[128 * 129];
    __shared__ T dd[128];
    __shared__ T dinv[128];
    __shared__ double red[16];
    __shared__ int bad;
    const int tid = threadIdx.x;
    if (tid == 0) bad = 0;
    for (int e = tid; e < 128 * 128; e += 1024) {
        const int i = e >> 7, j = e & 127;
        S[i * 129 + j] = A[(int64_t)i * lda + j];
    }
    __syncthreads();

    if (FACTOR) {
    const int ty = tid >> 5, tx = tid & 31;
    for (int j = 0; j < 128; ++j) {
        const T d = S[j * 129 + j];
        if (tid == 0) {
            dd[j] = d;
            if (!(d > (T)0) && bad == 0) bad = j + 1;
        }
        const T rd = (T)1 / d;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int i = j + 1 + ty + 32 * a;
            if (i < 128) {
                const T ci = S[i * 129 + j] * rd;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int k = j + 1 + tx + 32 * b;
                    if (k <= i) S[i * 129 + k] -= ci * S[k * 129 + j];
                }
            }
        }
        __syncthreads();
    }

    // scale columns, take sqrt of pivots, accumulate log det
    if (tid < 128) {
        const T d = dd[tid];
        const T s = sqrt(d);
        dinv[tid] = (T)1 / s;
    }
    __syncthreads();
    for (int e = tid; e < 128 * 128; e += 1024) {
        const int i = e >> 7, j = e & 127;
        if (j < i) S[i * 129 + j] *= dinv[j];
        else if (j == i) S[i * 129 + i] = sqrt(dd[i]);
    }
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
    // write L (lower incl. diagonal) back
    for (int e = tid; e < 128 * 128; e += 1024) {
        const int i = e >> 7, j = e & 127;
        if (j <= i) A[(int64_t)i * lda + j] = S[i * 129 + j];
    }
    } else {
        if (tid < 128) dinv[tid] = (T)1 / S[tid * 129 + tid];
    }
    __syncthreads();   // everyone has read the lower part they need from S before the upper is reused

    // inverse, row by row: X[i][j] = -(sum_{k=j}^{i-1} L[i][k] X[k][j]) / L[i][i],  X[j][j] = dinv[j]
    const int col = tid >> 3, part = tid & 7;
    for (int i = 1; i < 128; ++i) {
        T sum = (T)0;
        if (col < i) {
            for (int k = col + part; k < i; k += 8) {
                const T xkj = (k == col) ? dinv[col] : S[col * 129 + k];
                sum += S[i * 129 + k] * xkj;
            }
        }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        sum += __shfl_xor(sum, 4, 64);
        if (part == 0 && col < i) S[col * 129 + i] = -sum * dinv[i];
        __syncthreads();
    }
    for (int e = tid; e < 128 * 128; e += 1024) {
        const int i = e >> 7, j = e & 127;
        T v = (T)0;
        if (j < i) v = S[j * 129 + i];
        else if (j == i) v = dinv[i];
        inv_out[i * 128 + j] = v;
    }
}

template <typename T>
int potrf_diag_launch(algp_ctx* c, T* A, int64_t lda, T* inv_out, double* logdet_acc, int* info,
                      int64_t block_row0) {
    ProfScope ps(c, ALGP_PROF_POTRF_DIAG, 128.0 * 128.0 * 128.0, sizeof(T) * 3.0 * 128.0 * 128.0);
    hipLaunchKernelGGL((potrf_diag_kernel<T, true>), dim3(1), dim3(1024), 0, c->stream, A, lda, inv_out, logdet_acc,
                       info, block_row0);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int potrf_diag_launch<double>(algp_ctx*, double*, int64_t, double*, double*, int*, int64_t);
template int potrf_diag_launch<float>(algp_ctx*, float*, int64_t, float*, double*, int*, int64_t);

template <typename T>
int trinv_diag_launch(algp_ctx* c, const T* A, int64_t lda, T* inv_out) {
    hipLaunchKernelGGL((potrf_diag_kernel<T, false>), dim3(1), dim3(1024), 0, c->stream, const_cast<T*>(A), lda, inv_out,
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

template <typename T>
int cholesky_blocked(algp_ctx* c, T* A, int64_t npad, int64_t ld, T* invD, double* logdet_acc, int* info) {
    for (int64_t j0 = 0; j0 < npad; j0 += WB) {
        const int64_t w = (npad - j0 < WB) ? npad - j0 : WB;       // block width (multiple of 128)
        // ---- factor the block column [j0, j0+w) over all rows >= j0, 128 columns at a time ----
        for (int64_t k0 = j0; k0 < j0 + w; k0 += NB) {
            T* Akk = A + k0 * ld + k0;
            T* inv = invD + (k0 / NB) * NB * NB;
            ALGP_TRY(potrf_diag_launch<T>(c, Akk, ld, inv, logdet_acc, info, k0));
            const int64_t mrem = npad - (k0 + NB);
            if (mrem <= 0) continue;
            T* P = A + (k0 + NB) * ld + k0;                          // rows below the diagonal block
            ALGP_TRY(gemm_nt_launch<T>(c, ALGP_PROF_GEMM_CHOL, mrem, NB, NB, (T)1, P, ld, inv, NB, (T)0, nullptr, 0,
                                       P, ld, 0));
            const int64_t wrem = j0 + w - (k0 + NB);                 // columns of this block still to do
            if (wrem > 0) {
                // A[k0+NB:, k0+NB : j0+w] -= P * P[0:wrem]^T   (K = 128, only inside the block column)
                T* Cw = A + (k0 + NB) * ld + (k0 + NB);
                ALGP_TRY(gemm_nt_launch<T>(c, ALGP_PROF_GEMM_CHOL, mrem, wrem, NB, (T)-1, P, ld, P, ld, (T)1, Cw, ld,
                                           Cw, ld, 0));
            }
        }
        // ---- trailing update with the whole block: A22 -= P_blk P_blk^T, K = w, lower tiles ----
        const int64_t mrem = npad - (j0 + w);
        if (mrem > 0) {
            T* Pb = A + (j0 + w) * ld + j0;
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
                 int64_t ldl, const T* invD) {
    for (int64_t j0 = 0; j0 < npad; j0 += WB) {
        const int64_t w = (npad - j0 < WB) ? npad - j0 : WB;
        T* Xj = X + j0;
        if (j0 > 0)
            ALGP_TRY(gemm_nt_launch<T>(c, klass, mpad, w, j0, (T)-1, X, ldx, L + j0 * ldl, ldl, (T)1, Xj, ldx, Xj, ldx,
                                       0));
        for (int64_t k0 = j0; k0 < j0 + w; k0 += NB) {
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
                                  const double*);
template int trsm_blocked<float>(algp_ctx*, int, float*, int64_t, int64_t, const float*, int64_t, int64_t,
                                 const float*);

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
    T s = (T)0;
    if (!TRANS) {
        for (int k = 0; k <= t; ++k) s += inv[t * 128 + k] * xb[k];
    } else {
        for (int k = t; k < 128; ++k) s += inv[k * 128 + t] * xb[k];
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
        hipLaunchKernelGGL((diag_matvec_kernel<T, false>), dim3(1), dim3(128), 0, c->stream, invD + kb * NB * NB,
                           b + kb * NB);
        const int64_t mrem = npad - (kb + 1) * NB;
        if (mrem > 0) {
            const int grid = (int)((mrem + 3) / 4 < 1024 ? (mrem + 3) / 4 : 1024);
            hipLaunchKernelGGL(panel_gemv_kernel<T>, dim3(grid), dim3(256), 0, c->stream,
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
        hipLaunchKernelGGL((diag_matvec_kernel<T, true>), dim3(1), dim3(128), 0, c->stream, invD + kb * NB * NB,
                           b + kb * NB);
        const int64_t ncol = kb * NB;
        if (ncol > 0) {
            hipLaunchKernelGGL(panel_gemv_t_kernel<T>, dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, c->stream,
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
