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
#include "diag.h"
#include "vecops.h"
#include <stdlib.h>

namespace algp {

// The 128 x 128 diagonal block (factor + inverse in one workgroup) lives in diag.h; these kernels only give it
// a launch of its own.  The dependency-driven Cholesky (chol_dag.hip) calls the same routine as one of its tasks.
template <typename T, bool FACTOR>
__global__ __launch_bounds__(256) void potrf_diag_kernel(T* A, int64_t lda, T* inv_out, double* logdet_acc, int* info,
                                                          int64_t block_row0) {
    __shared__ DiagShared<T> sh;
    diag128_run<T, FACTOR>(sh, A, lda, inv_out, logdet_acc, true, info, block_row0);
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

// `batch` independent 128 x 128 blocks (block b at A + b * sA): factor + inverse + log det (to logdet[b], added) + first bad
// pivot (info[b], 1-based) -- the per-path blocks of algp_score_paths
template <typename T>
__global__ __launch_bounds__(256) void potrf_diag_batched_kernel(T* A, int64_t sA, int64_t lda, T* inv_out, int64_t sInv, double* logdet,
                                                                  int* info) {
    __shared__ DiagShared<T> sh;
    const int64_t b = blockIdx.x;
    diag128_run<T, true>(sh, A + b * sA, lda, inv_out + b * sInv, logdet + b, true, info + b, 0);
}
template <typename T>
int potrf_diag_batched_launch(algp_ctx* c, T* A, int64_t sA, int64_t lda, T* inv_out, int64_t sInv, double* logdet, int* info, int batch) {
    if (batch <= 0) return ALGP_OK;
    ProfScope ps(c, ALGP_PROF_POTRF_DIAG, 128.0 * 128.0 * 128.0 * batch, sizeof(T) * 3.0 * 128.0 * 128.0 * batch);
    hipLaunchKernelGGL(potrf_diag_batched_kernel<T>, dim3((unsigned)batch), dim3(256), 0, c->cur, A, sA, lda, inv_out, sInv, logdet, info);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int potrf_diag_batched_launch<double>(algp_ctx*, double*, int64_t, int64_t, double*, int64_t, double*, int*, int);
template int potrf_diag_batched_launch<float>(algp_ctx*, float*, int64_t, int64_t, float*, int64_t, double*, int*, int);

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
constexpr int64_t TRSM_PUSH_TILES = 320;   // candidate solves of up to 320 x 128 rows run right-looking (trsm_rows_blocked)

static hipEvent_t sync_event(algp_ctx* c, size_t i);
template <typename T>
int gemm_splitk_sub(algp_ctx* c, int klass, int64_t m, int64_t n, int64_t k, const T* A, int64_t lda, const T* B,
                    int64_t ldb, T* C, int64_t ldc, DevBuf& scratch);

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

// Right-looking blocked Cholesky, two-level (512 / 128), as a sequence of launches: used for small matrices and
// for matrices too large for the dependency-driven launch's task list (chol_dag.hip), which is the default in
// between.  (Round 1 also measured a two-stream look-ahead of this sequence: no gain, 19.2 vs 18.9 ms at N = 10 000.)
template <typename T>
static int cholesky_serial(algp_ctx* c, T* A, int64_t npad, int64_t ld, T* invD, double* logdet_acc, int* info) {
    for (int64_t j0 = 0; j0 < npad; j0 += WB) {
        const int64_t w = (npad - j0 < WB) ? npad - j0 : WB;
        ALGP_TRY(chol_panel<T>(c, A, npad, ld, invD, logdet_acc, info, j0, w));
        const int64_t mrem = npad - (j0 + w);
        if (mrem > 0) {
            // trailing update with the whole block: A22 -= P_blk P_blk^T, K = w, lower tiles
            const T* Pb = A + (j0 + w) * ld + j0;
            T* A22 = A + (j0 + w) * ld + (j0 + w);
            ALGP_TRY(gemm_nt_launch<T>(c, ALGP_PROF_GEMM_CHOL_UPDATE, mrem, mrem, w, (T)-1, Pb, ld, Pb, ld, (T)1, A22, ld,
                                       A22, ld, 1));
        }
    }
    return ALGP_OK;
}

template <typename T>
int cholesky_dag(algp_ctx* c, T* A, int64_t npad, int64_t ld, T* invD, double* logdet_acc, int* info);

// ALGP_CHOL_DAG=0 selects the launch sequences for every size (A/B runs, tests of the fallback)
bool dag_enabled() {
    static const int dag = env_int("ALGP_CHOL_DAG", 1);
    return dag != 0;
}

template <typename T>
int cholesky_blocked(algp_ctx* c, T* A, int64_t npad, int64_t ld, T* invD, double* logdet_acc, int* info) {
    const int64_t nt = npad / NB;
    if (dag_enabled() && nt >= DAG_MIN_TILES && nt <= DAG_MAX_TILES) return cholesky_dag<T>(c, A, npad, ld, invD, logdet_acc, info);
    return cholesky_serial<T>(c, A, npad, ld, invD, logdet_acc, info);
}
template int cholesky_blocked<double>(algp_ctx*, double*, int64_t, int64_t, double*, double*, int*);
template int cholesky_blocked<float>(algp_ctx*, float*, int64_t, int64_t, float*, double*, int*);

// ---------------------------------------------------------------------------------------------
// Explicit inverses of the factor's full 512 x 512 diagonal blocks, for the left-looking candidate solve: with
// them the solve inside a block is ONE product, X_J = T_J inv(L_JJ)^T (gemm_nt_launch_tri), where the 128-column steps
// were seven short launches of one column tile each (15 ms of 162 per config-4 solve for 5 % of its flops).  Built from
// the 128-block inverses the factorisation leaves (invD) by two levels of recursive doubling,
//     X_(hi,lo) = - X_hi ( L_(hi,lo) X_lo ),
// every product as the library's NT GEMM, batched over the blocks: P^T = X_lo^T L_(hi,lo)^T needs X_lo transposed (a small
// kernel), the second product takes P^T as its B operand as it is.  ~0.25 ms per solve at N = 10 000 (19 blocks).
// Layout of `out`: block J at out + J * 512 * 512, row-major, leading dimension 512, zeros above the diagonal.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_tiles_kernel(const T* src, int64_t lds_, int64_t sstride, T* dst, int64_t ldd,
                                                              int64_t dstride) {
    __shared__ T t[32][33];
    const T* s = src + (int64_t)blockIdx.z * sstride;
    T* d = dst + (int64_t)blockIdx.z * dstride;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[ty + 8 * i][tx] = s[(int64_t)(r0 + ty + 8 * i) * lds_ + c0 + tx];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) d[(int64_t)(c0 + ty + 8 * i) * ldd + r0 + tx] = t[tx][ty + 8 * i];
}
// zero a block and put the four 128 x 128 diagonal inverses on its diagonal
template <typename T>
__global__ __launch_bounds__(256) void inv512_init_kernel(const T* invD4, T* out) {
    const T* src = invD4 + (int64_t)blockIdx.y * 4 * NB * NB;
    T* dst = out + (int64_t)blockIdx.y * WB * WB;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;     // element of the 512 x 512 block
    const int r = (int)(e / WB), col = (int)(e % WB);
    const int tr = r / NB, tc = col / NB;
    dst[e] = tr == tc ? src[(int64_t)tr * NB * NB + (r % NB) * NB + (col % NB)] : (T)0;
}
template <typename T>
static int build_inv512(algp_ctx* c, int klass, const T* L, int64_t ldl, const T* invD, int64_t npad, T* out, T* scr) {
    const int nb = (int)(npad / WB);                               // full blocks J = 0 .. npad / 512 - 1
    if (nb <= 0) return ALGP_OK;
    const T* L1 = L;                                                // block J = 0
    const T* D1 = invD;                                             // its first 128-block inverse
    const int64_t sL = (int64_t)WB * ldl + WB, sD = 4 * NB * NB, sO = (int64_t)WB * WB;
    T* DT = scr;                                                    // [nb][4][128 x 128]: the diagonal inverses transposed
    T* PT1 = DT + (int64_t)nb * 4 * NB * NB;                        // [nb][128 x 128]
    T* XT = PT1 + (int64_t)nb * NB * NB;                            // [nb][256 x 256]
    T* PT2 = XT + (int64_t)nb * 256 * 256;                          // [nb][256 x 256]
    hipLaunchKernelGGL(inv512_init_kernel<T>, dim3(WB * WB / 256, (unsigned)nb), dim3(256), 0, c->cur, D1, out);
    for (int t = 0; t < 4; ++t)
        hipLaunchKernelGGL(transpose_tiles_kernel<T>, dim3(4, 4, (unsigned)nb), dim3(256), 0, c->cur, D1 + (int64_t)t * NB * NB, (int64_t)NB,
                           sD, DT + (int64_t)t * NB * NB, (int64_t)NB, sD);
    ALGP_HIP(hipGetLastError());
    // level 1: tiles (1, 0) and (3, 2)
    for (int lo = 0; lo < 4; lo += 2) {
        const int hi = lo + 1;
        ALGP_TRY(gemm_nt_launch_batched<T>(c, klass, NB, NB, NB, (T)1, DT + (int64_t)lo * NB * NB, NB, sD, L1 + (int64_t)(NB * hi) * ldl + NB * lo,
                                           ldl, sL, (T)0, nullptr, 0, 0, PT1, NB, (int64_t)NB * NB, 0, nb));
        ALGP_TRY(gemm_nt_launch_batched<T>(c, klass, NB, NB, NB, (T)-1, D1 + (int64_t)hi * NB * NB, NB, sD, PT1, NB, (int64_t)NB * NB, (T)0,
                                           nullptr, 0, 0, out + (int64_t)(NB * hi) * WB + NB * lo, WB, sO, 0, nb));
    }
    // level 2: the 256 x 256 block (2..3, 0..1)
    hipLaunchKernelGGL(transpose_tiles_kernel<T>, dim3(8, 8, (unsigned)nb), dim3(256), 0, c->cur, out, (int64_t)WB, sO, XT, (int64_t)256,
                       (int64_t)256 * 256);
    ALGP_HIP(hipGetLastError());
    ALGP_TRY(gemm_nt_launch_batched<T>(c, klass, 256, 256, 256, (T)1, XT, 256, (int64_t)256 * 256, L1 + (int64_t)256 * ldl, ldl, sL, (T)0,
                                       nullptr, 0, 0, PT2, 256, (int64_t)256 * 256, 0, nb));
    ALGP_TRY(gemm_nt_launch_batched<T>(c, klass, 256, 256, 256, (T)-1, out + (int64_t)256 * WB + 256, WB, sO, PT2, 256, (int64_t)256 * 256, (T)0,
                                       nullptr, 0, 0, out + (int64_t)256 * WB, WB, sO, 0, nb));
    return ALGP_OK;
}
template <typename T>
static size_t inv512_scratch_elems(int64_t npad) {
    const int64_t nb = npad / WB;
    return nb <= 0 ? 0 : (size_t)nb * (4 * NB * NB + NB * NB + 2 * 256 * 256);
}

// X <- X L^-T, left-looking over 512-wide column blocks:
//   X_J <- X_J - X_{0:J} L_{J,0:J}^T            (one GEMM, n = 512)
//   inside J, 128 columns at a time: X_k <- (X_k - X_{J0:k} L_{k,J0:k}^T) inv(L_kk)^T
template <typename T>
static int trsm_rows_blocked(algp_ctx* c, int klass, T* X, int64_t mpad, int64_t ldx, const T* L, int64_t npad,
                     int64_t ldl, const T* invD, int64_t col_start, bool whole_solve, const T* stat_w = nullptr,
                     T* stat_out = nullptr, int64_t stat_ld = 0, const T* inv512 = nullptr, T* tmp512 = nullptr) {
    // inv512 / tmp512 (left-looking order, col_start == 0): the explicit inverses of the full 512-column blocks
    // (build_inv512) and mpad x 512 scratch, holding X_0 on entry: block J is then two launches -- T = X_J - X_{0:J} L_{J,0:J}^T
    // into the scratch, X_J = T inv(L_JJ)^T -- instead of eight.
    // stat_out (left-looking order only, col_start == 0: trsm_blocked decides): the launch that writes a column tile of X for
    // the last time also leaves the tile's row sums of x^2 and x * stat_w[column] at stat_out[(2 tile + 0 / 1) * stat_ld + row]
    // col_start (multiple of 128): columns [0, col_start) of X already hold the solution
    if (mpad <= 32 * NB) {
        // A short X (a few test points): the left-looking order below would walk K up to npad inside
        // one or two workgroups (launch-latency bound).  Right-looking instead: solve one 128-column
        // block, then update ALL remaining columns with K = 128 -- (npad-k)/128 column tiles in parallel.
        // It re-reads/re-writes the trailing columns each step, which is negligible for so few rows.
        if (col_start > 0)      // columns >= col_start still need the contributions of the kept ones
            ALGP_TRY(gemm_splitk_sub<T>(c, klass, mpad, npad - col_start, col_start, X, ldx, L + col_start * ldl, ldl,
                                        X + col_start, ldx, c->splitk));
        for (int64_t k0 = col_start; k0 < npad; k0 += NB) {
            T* Xk = X + k0;
            ALGP_TRY(gemm_nt_launch<T>(c, klass, mpad, NB, NB, (T)1, Xk, ldx, invD + (k0 / NB) * NB * NB, NB, (T)0,
                                       nullptr, 0, Xk, ldx, 0));
            const int64_t nrem = npad - (k0 + NB);
            if (nrem > 0)
                ALGP_TRY(gemm_nt_launch<T>(c, klass, mpad, nrem, NB, (T)-1, Xk, ldx, L + (k0 + NB) * ldl + k0, ldl, (T)1,
                                           Xk + NB, ldx, Xk + NB, ldx, 0));
        }
        return ALGP_OK;
    }
    if (whole_solve && col_start == 0 && mpad <= TRSM_PUSH_TILES * NB) {     // (never a row chunk of a larger solve)
        // A mid-sized X (a few thousand to ~40 000 rows: a rank's share of the candidates on 4-8 GPUs, a held-out set):
        // the left-looking order below launches (rows/128) x 4 tiles per block column, each walking a K of up to npad --
        // 132 tiles on 512 slots per row chunk at 12 500 rows (41 TFLOP/s fp64).  RIGHT-looking over 512-column
        // blocks instead: solve a block (inside it left-looking, as below), then push it into ALL columns to its right
        // with one K = 512 GEMM of (rows/128) x (columns left/128) tiles -- the machine is full until the last few
        // blocks, at the price of one read + write of the trailing columns per block (rows x npad^2 / 512 elements in
        // all: 20 GB at 12 500 x 10 000 fp64, 64 flop per byte).  Measured at N = 10 000, fp64 (left-looking -> this):
        // 5 000 rows 18.1 -> 13.2 ms, 12 500 rows 30.6 -> 23.4 ms (41 -> 53 TFLOP/s), 25 000 rows 45.9 -> 42.7 ms; equal at
        // 50 000 rows, where the three row-chunk streams of the left-looking order fill the machine as well.
        // Two streams: the short launches that solve block J+1 (and the push of block J into block J+1 alone, which they
        // wait for) run on the caller's stream BESIDE the big push of block J into the columns beyond, on a helper stream.
        //   main:   I(J) . record a[J] . wait b[J-1] . Q(J) . I(J+1) ...      I = solve inside the block, Q = push into the next block
        //   helper: wait a[J] . P(J) . record b[J]                            P = push into everything beyond the next block
        // Every column block receives its pushes in ascending J (P(J-1) before Q(J), both before I(J+1)).
        constexpr int64_t PW = WB;                                  // (1024-wide pushes: 25.2 ms at 12 500 rows, 41.9 at 25 000)
        hipStream_t sa = c->cur, sb = (c->cur == c->stream && c->stream2) ? c->stream2 : nullptr;
        int rc = ALGP_OK;
        int64_t jb = 0;
        for (int64_t j0 = 0; j0 < npad && rc == ALGP_OK; j0 += PW, ++jb) {
            const int64_t w = (npad - j0 < PW) ? npad - j0 : PW, j1 = j0 + w;
            T* Xj = X + j0;
            for (int64_t k0 = j0; k0 < j1 && rc == ALGP_OK; k0 += NB) {
                T* Xk = X + k0;
                if (k0 > j0)
                    rc = gemm_nt_launch<T>(c, klass, mpad, NB, k0 - j0, (T)-1, Xj, ldx, L + k0 * ldl + j0, ldl, (T)1, Xk, ldx, Xk,
                                           ldx, 0);
                if (rc == ALGP_OK)
                    rc = gemm_nt_launch<T>(c, klass, mpad, NB, NB, (T)1, Xk, ldx, invD + (k0 / NB) * NB * NB, NB, (T)0, nullptr, 0,
                                           Xk, ldx, 0);
            }
            if (j1 >= npad || rc != ALGP_OK) break;
            const int64_t wq = (npad - j1 < PW) ? npad - j1 : PW;           // width of the next block
            if (sb) {
                hipEvent_t ea = sync_event(c, 8 + 2 * (size_t)(jb & 3)), eb = sync_event(c, 9 + 2 * (size_t)(jb & 3));
                hipEventRecord(ea, sa);
                if (jb > 0) hipStreamWaitEvent(sa, sync_event(c, 9 + 2 * (size_t)((jb - 1) & 3)), 0);
                rc = gemm_nt_launch<T>(c, klass, mpad, wq, w, (T)-1, Xj, ldx, L + j1 * ldl + j0, ldl, (T)1, X + j1, ldx, X + j1, ldx, 0);
                if (rc == ALGP_OK && j1 + wq < npad) {
                    hipStreamWaitEvent(sb, ea, 0);
                    c->cur = sb;
                    rc = gemm_nt_launch<T>(c, klass, mpad, npad - (j1 + wq), w, (T)-1, Xj, ldx, L + (j1 + wq) * ldl + j0, ldl, (T)1,
                                           X + j1 + wq, ldx, X + j1 + wq, ldx, 0);
                    c->cur = sa;
                }
                hipEventRecord(eb, sb);
            } else {
                rc = gemm_nt_launch<T>(c, klass, mpad, npad - j1, w, (T)-1, Xj, ldx, L + j1 * ldl + j0, ldl, (T)1, X + j1, ldx,
                                       X + j1, ldx, 0);
            }
        }
        if (sb) {                                                   // the caller's stream continues after the last push
            hipEvent_t done = sync_event(c, 16);
            hipEventRecord(done, sb);
            hipStreamWaitEvent(sa, done, 0);
        }
        return rc;
    }
    const int64_t TWB = WB;                                     // outer block width
    for (int64_t j0 = 0; j0 < npad; j0 += TWB) {
        const int64_t w = (npad - j0 < TWB) ? npad - j0 : TWB;
        if (j0 + w <= col_start) continue;
        const int64_t cs = j0 > col_start ? j0 : col_start;         // first column of this block to solve
        T* Xj = X + j0;
        if (inv512 && tmp512 && w == TWB && col_start == 0) {
            // (block 0 has nothing to its left: the caller has copied its right-hand sides into the scratch)
            if (j0 > 0) ALGP_TRY(gemm_nt_launch<T>(c, klass, mpad, TWB, j0, (T)-1, X, ldx, L + j0 * ldl, ldl, (T)1, Xj, ldx, tmp512, TWB, 0));
            ALGP_TRY(gemm_nt_launch_tri<T>(c, klass, mpad, TWB, tmp512, TWB, inv512 + (j0 / TWB) * TWB * TWB, TWB, Xj, ldx,
                                           stat_w ? stat_w + j0 : nullptr, stat_out ? stat_out + 2 * (j0 / NB) * stat_ld : nullptr, stat_ld));
            continue;
        }
        if (j0 > 0)
            ALGP_TRY(gemm_nt_launch<T>(c, klass, mpad, j0 + w - cs, j0, (T)-1, X, ldx, L + cs * ldl, ldl, (T)1, X + cs,
                                       ldx, X + cs, ldx, 0));
        for (int64_t k0 = cs; k0 < j0 + w; k0 += NB) {
            T* Xk = X + k0;
            if (k0 > j0)
                ALGP_TRY(gemm_nt_launch<T>(c, klass, mpad, NB, k0 - j0, (T)-1, Xj, ldx, L + k0 * ldl + j0, ldl, (T)1,
                                           Xk, ldx, Xk, ldx, 0));
            if (stat_out)
                ALGP_TRY(gemm_nt_launch_stats<T>(c, klass, mpad, NB, (T)1, Xk, ldx, invD + (k0 / NB) * NB * NB, NB, Xk, ldx,
                                                 stat_w + k0, stat_out + 2 * (k0 / NB) * stat_ld, stat_ld));
            else
                ALGP_TRY(gemm_nt_launch<T>(c, klass, mpad, NB, NB, (T)1, Xk, ldx, invD + (k0 / NB) * NB * NB, NB, (T)0,
                                           nullptr, 0, Xk, ldx, 0));
        }
    }
    return ALGP_OK;
}

// X (npad x npad, holding the identity) <- L^-T, which is UPPER triangular: row r is zero left of
// column r, so only the rows above a column block ever take part: N^3/6 multiply-adds instead of the
// N^3/2 of a dense right-hand side.  RIGHT-looking over 512-wide column blocks -- a left-looking sweep
// has at most (rows above)/128 x 4 tiles per launch, each walking the whole K (measured: 14.5 ms at
// N = 9000, 204 workgroups on the last block); pushing each solved block into all trailing columns
// gives (j0+w)/128 x (npad-j0-w)/128 tiles with K = 512, the shape of the Cholesky's trailing update.
template <typename T>
int trinv_upper(algp_ctx* c, int klass, T* X, int64_t npad, int64_t ldx, const T* L, int64_t ldl, const T* invD) {
    for (int64_t j0 = 0; j0 < npad; j0 += WB) {
        const int64_t w = (npad - j0 < WB) ? npad - j0 : WB, j1 = j0 + w;
        for (int64_t k0 = j0; k0 < j1; k0 += NB) {
            T* Xk = X + k0;
            const int64_t rows = k0 + NB;                         // rows below are zero in this column block
            ALGP_TRY(gemm_nt_launch<T>(c, klass, rows, NB, NB, (T)1, Xk, ldx, invD + (k0 / NB) * NB * NB, NB, (T)0,
                                       nullptr, 0, Xk, ldx, 0));
            if (k0 + NB < j1)
                ALGP_TRY(gemm_nt_launch<T>(c, klass, rows, j1 - (k0 + NB), NB, (T)-1, Xk, ldx, L + (k0 + NB) * ldl + k0,
                                           ldl, (T)1, Xk + NB, ldx, Xk + NB, ldx, 0));
        }
        if (j1 < npad)
            ALGP_TRY(gemm_nt_launch<T>(c, klass, j1, npad - j1, w, (T)-1, X + j0, ldx, L + j1 * ldl + j0, ldl, (T)1,
                                       X + j1, ldx, X + j1, ldx, 0));
    }
    return ALGP_OK;
}

// The same inverse IN PLACE: XL holds a factor L in its lower triangle (as cholesky_blocked leaves it) and receives
// X = L^-T in its upper triangle, diagonal tiles included -- the strictly-lower tiles keep L, which the sweep still reads.
// One n x n buffer instead of two: what lets the MI criterion hold its two pool-wide inverses at config 4's own pool
// (110 000 sites: 2 x 96.8 GB; the three matrices of the out-of-place form were 290 GB).  Differences from trinv_upper:
// the tiles on and above the diagonal are prepared here (identity / zero: they hold S's upper half), and the push of a
// finished 512-column block into the trailing columns is cut by row tile inside the block -- there X's tiles left of the
// diagonal are L's, so row tile t of the block multiplies from its own diagonal column on.  Readers of the result must
// skip the strictly-lower tiles (rows_reduce_launch with tri_c0 >= 0).  Reference: agent.py:330-339 takes two pool-wide
// slogdets per candidate.
template <typename T>
__global__ __launch_bounds__(256) void tri_prep_kernel(T* A, int64_t ld) {
    const int64_t bi = blockIdx.y, bj = blockIdx.x;
    if (bi > bj) return;
    T* t = A + bi * NB * ld + bj * NB;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int r = e >> 7, q = e & 127;
        t[(int64_t)r * ld + q] = (bi == bj && r == q) ? (T)1 : (T)0;
    }
}
template <typename T>
int trinv_upper_inplace(algp_ctx* c, int klass, T* XL, int64_t npad, int64_t ld, const T* invD) {
    const unsigned nt = (unsigned)(npad / NB);
    if (nt > 65535) return fail(c, ALGP_ERR_BAD_ARG, "trinv_upper_inplace: matrix too large");
    hipLaunchKernelGGL(tri_prep_kernel<T>, dim3(nt, nt), dim3(256), 0, c->cur, XL, ld);
    ALGP_HIP(hipGetLastError());
    for (int64_t j0 = 0; j0 < npad; j0 += WB) {
        const int64_t w = (npad - j0 < WB) ? npad - j0 : WB, j1 = j0 + w;
        for (int64_t k0 = j0; k0 < j1; k0 += NB) {
            T* Xk = XL + k0;
            const int64_t rows = k0 + NB;
            ALGP_TRY(gemm_nt_launch<T>(c, klass, rows, NB, NB, (T)1, Xk, ld, invD + (k0 / NB) * NB * NB, NB, (T)0, nullptr, 0, Xk, ld, 0));
            if (k0 + NB < j1)
                ALGP_TRY(gemm_nt_launch<T>(c, klass, rows, j1 - (k0 + NB), NB, (T)-1, Xk, ld, XL + (k0 + NB) * ld + k0, ld, (T)1,
                                           Xk + NB, ld, Xk + NB, ld, 0));
        }
        if (j1 >= npad) break;
        // rows above the block: dense in X; rows inside it: from their own diagonal tile on
        if (j0 > 0)
            ALGP_TRY(gemm_nt_launch<T>(c, klass, j0, npad - j1, w, (T)-1, XL + j0, ld, XL + j1 * ld + j0, ld, (T)1, XL + j1, ld, XL + j1, ld, 0));
        for (int64_t r0 = j0; r0 < j1; r0 += NB)
            ALGP_TRY(gemm_nt_launch<T>(c, klass, NB, npad - j1, j1 - r0, (T)-1, XL + r0 * ld + r0, ld, XL + j1 * ld + r0, ld, (T)1,
                                       XL + r0 * ld + j1, ld, XL + r0 * ld + j1, ld, 0));
    }
    return ALGP_OK;
}
template int trinv_upper_inplace<double>(algp_ctx*, int, double*, int64_t, int64_t, const double*);
template int trinv_upper_inplace<float>(algp_ctx*, int, float*, int64_t, int64_t, const float*);

// C (lower tiles) <- X X^T for the upper-triangular X above: ONE launch in which output tile (a, b <= a) sums over the
// columns k >= 128 a only (row tile a of X is zero left of them; GemmArgs::ktri).  The tiles are dealt out by ascending a,
// i.e. longest K first, so the 512 resident workgroups end within one short tile of each other.  (Rounds 1-3: a sum over
// 512-wide column panels of X, 20 launches whose first few hold fewer tiles than the machine has slots.)
template <typename T>
int syrk_upper(algp_ctx* c, int klass, const T* X, int64_t npad, int64_t ldx, T* C, int64_t ldc) {
    return gemm_nt_launch_batched<T>(c, klass, npad, npad, npad, (T)1, X, ldx, 0, X, ldx, 0, (T)0, nullptr, ldc, 0, C, ldc, 0, 1, 1, 1);
}
template int syrk_upper<double>(algp_ctx*, int, const double*, int64_t, int64_t, double*, int64_t);
template int syrk_upper<float>(algp_ctx*, int, const float*, int64_t, int64_t, float*, int64_t);
template int trinv_upper<double>(algp_ctx*, int, double*, int64_t, int64_t, const double*, int64_t, const double*);
template int trinv_upper<float>(algp_ctx*, int, float*, int64_t, int64_t, const float*, int64_t, const float*);

// (A divide-and-conquer order of this sweep was measured in round 1: same time, 339 GB instead of 200 GB of HBM
// traffic per solve at N = 10 000, M = 100 000 -- the big off-diagonal blocks of L fall out of the 4 MB L2s -- and removed.)
template <typename T>
static int trsm_rows(algp_ctx* c, int klass, T* X, int64_t mpad, int64_t ldx, const T* L, int64_t npad,
                     int64_t ldl, const T* invD, int64_t col_start, bool whole_solve, const T* stat_w = nullptr,
                     T* stat_out = nullptr, int64_t stat_ld = 0, const T* inv512 = nullptr, T* tmp512 = nullptr) {
    return trsm_rows_blocked<T>(c, klass, X, mpad, ldx, L, npad, ldl, invD, col_start, whole_solve, stat_w, stat_out, stat_ld, inv512,
                                tmp512);
}

static hipEvent_t sync_event(algp_ctx* c, size_t i) {
    while (c->sync_events.size() <= i) {
        hipEvent_t e;
        hipEventCreateWithFlags(&e, hipEventDisableTiming);
        c->sync_events.push_back(e);
    }
    return c->sync_events[i];
}

// The rows of X (candidates) are independent, so a tall X is solved as row chunks on streams of their own (three by
// default, algp_debug_set_trsm_chunks / $ALGP_TRSM_CHUNKS): the tail of every big GEMM of one chunk (1044 workgroups
// over 512 resident slots: the last round is 4 % full) and its short HBM-bound in-block launches run beside the other
// chunks' MFMA-bound GEMMs -- 84 % of the fp64 matrix peak against 78.5 % with the launches back to back on one stream.
template <typename T>
int trsm_blocked(algp_ctx* c, int klass, T* X, int64_t mpad, int64_t ldx, const T* L, int64_t npad,
                 int64_t ldl, const T* invD, int64_t col_start, const T* stat_w, T* stat_out, int64_t stat_ld, bool* stats_done) {
    const int64_t tiles = mpad / NB;
    // row statistics ride along where every row goes through the left-looking order from its first column
    const bool stats = stat_out && stat_w && col_start == 0 && tiles > TRSM_PUSH_TILES;
    if (stats_done) *stats_done = stats;
    if (!stats) stat_out = nullptr;
    // the left-looking order from column 0 (the same condition): 512-block inverses, one scratch of mpad x 512 for all chunks
    // ($ALGP_TRSM_INV512=0: the 128-column steps inside a block, as in rounds 1-3)
    const T* inv512 = nullptr;
    T* tmp512 = nullptr;
    {
        const bool on = env_switch("ALGP_TRSM_INV512", true);                                      // read per call: tests flip it
        if (on && col_start == 0 && tiles > TRSM_PUSH_TILES && npad / WB >= 1 && c->cur == c->stream &&
            ensure(c, c->inv512, sizeof(T) * (size_t)(npad / WB) * WB * WB) == ALGP_OK &&
            ensure(c, c->inv512_scr, sizeof(T) * inv512_scratch_elems<T>(npad)) == ALGP_OK &&
            ensure(c, c->trsm_tmp, sizeof(T) * (size_t)mpad * WB) == ALGP_OK) {
            ALGP_TRY(build_inv512<T>(c, klass, L, ldl, invD, npad, (T*)c->inv512.p, (T*)c->inv512_scr.p));
            inv512 = (const T*)c->inv512.p;
            tmp512 = (T*)c->trsm_tmp.p;
            // block 0's right-hand sides into the scratch: its product with inv(L_00)^T cannot run in place (a workgroup
            // of column tile c reads the tiles left of it in the same rows)
            ALGP_HIP(hipMemcpy2DAsync(tmp512, sizeof(T) * WB, X, sizeof(T) * ldx, sizeof(T) * WB, (size_t)mpad, hipMemcpyDeviceToDevice, c->cur));
        }
    }
    hipStream_t streams[4] = {c->stream, c->stream2, c->stream3, c->stream4};
    int nch = c->trsm_chunks < 1 ? 1 : (c->trsm_chunks > 4 ? 4 : c->trsm_chunks);
    while (nch > 1 && (!streams[nch - 1] || tiles < 32 * nch)) --nch;
    if (col_start == 0 && tiles <= TRSM_PUSH_TILES) nch = 1;    // the right-looking order fills the machine by itself
    if (nch == 1 || c->cur != c->stream)
        return trsm_rows<T>(c, klass, X, mpad, ldx, L, npad, ldl, invD, col_start, true, stat_w, stat_out, stat_ld, inv512, tmp512);
    ALGP_HIP(hipEventRecord(sync_event(c, 0), streams[0]));
    int rc = ALGP_OK;
    int64_t r0 = 0;
    for (int k = nch - 1; k >= 0; --k) {                       // helper streams first, the main stream last
        const int64_t rows = (k == 0) ? mpad - r0 : (tiles / nch) * NB;
        if (k > 0) ALGP_HIP(hipStreamWaitEvent(streams[k], sync_event(c, 0), 0));
        c->cur = streams[k];
        if (rc == ALGP_OK)
            rc = trsm_rows<T>(c, klass, X + r0 * ldx, rows, ldx, L, npad, ldl, invD, col_start, false, stat_w,
                              stat_out ? stat_out + r0 : nullptr, stat_ld, inv512, tmp512 ? tmp512 + r0 * WB : nullptr);
        r0 += rows;
    }
    c->cur = streams[0];
    // the main stream continues only after every chunk is done (also on the error path)
    for (int k = 1; k < nch; ++k) {
        hipEventRecord(sync_event(c, (size_t)k), streams[k]);
        hipStreamWaitEvent(streams[0], sync_event(c, (size_t)k), 0);
    }
    return rc;
}
template int trsm_blocked<double>(algp_ctx*, int, double*, int64_t, int64_t, const double*, int64_t, int64_t,
                                  const double*, int64_t, const double*, double*, int64_t, bool*);
template int trsm_blocked<float>(algp_ctx*, int, float*, int64_t, int64_t, const float*, int64_t, int64_t,
                                 const float*, int64_t, const float*, float*, int64_t, bool*);

// ---------------------------------------------------------------------------------------------
// C -= X X^T for a short, very wide X (the R <= 512 new rows of a factor update against N kept columns).
// One launch with K = N would walk N/16 k-tiles inside 1-10 workgroups (5.3 ms at N = 50 000); instead the
// K range is cut into chunks, all chunk products run as ONE batched launch into a scratch stack, and a
// second kernel subtracts them from C in chunk order (fixed order: bitwise reproducible).
// ---------------------------------------------------------------------------------------------
// C (m x n) -= sum over chunks of the partial products P[q] (m x n each, contiguous)
template <typename T>
__global__ void sub_partials_rect_kernel(T* C, int64_t ldc, int64_t m, int64_t n, const T* P, int nchunks) {
    const int64_t i = (int64_t)blockIdx.y * 16 + (threadIdx.x >> 4), j = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
    if (i >= m || j >= n) return;
    T s = (T)0;
    for (int q = 0; q < nchunks; ++q) s += P[(int64_t)q * m * n + i * n + j];
    C[i * ldc + j] -= s;
}

template <typename T>
__global__ void sub_partials_kernel(T* C, int64_t ldc, int64_t m, const T* P, int nchunks) {
    const int64_t i = (int64_t)blockIdx.y * 16 + (threadIdx.x >> 4), j = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
    if (i >= m || j > i) return;                                   // lower triangle only
    T s = (T)0;
    for (int q = 0; q < nchunks; ++q) s += P[(int64_t)q * m * m + i * m + j];
    C[i * ldc + j] -= s;
}

template <typename T>
int syrk_skinny_sub(algp_ctx* c, int klass, const T* X, int64_t m, int64_t k, int64_t ldx, T* C, int64_t ldc,
                    DevBuf& scratch) {
    if (m <= 0 || k <= 0) return ALGP_OK;
    const int64_t kb = k / NB;                                     // 128-column blocks of X
    int64_t per = (kb + 127) / 128;                                // blocks per chunk: at most 128 chunks ...
    if (per < 4) per = 4;                                          // ... of at least K = 512
    const int64_t nfull = kb / per, rem = kb - nfull * per;
    if (m > 4 * NB || nfull < 2)                                   // not skinny, or nothing to split
        return gemm_nt_launch<T>(c, klass, m, m, k, (T)-1, X, ldx, X, ldx, (T)1, C, ldc, C, ldc, 1);
    ALGP_TRY(ensure(c, scratch, sizeof(T) * (size_t)(nfull + 1) * m * m));
    T* P = (T*)scratch.p;
    ALGP_TRY(gemm_nt_launch_batched<T>(c, klass, m, m, per * NB, (T)1, X, ldx, per * NB, X, ldx, per * NB, (T)0, nullptr, m,
                                       m * m, P, m, m * m, 1, (int)nfull));
    int nch = (int)nfull;
    if (rem > 0) {
        ALGP_TRY(gemm_nt_launch<T>(c, klass, m, m, rem * NB, (T)1, X + nfull * per * NB, ldx, X + nfull * per * NB, ldx,
                                   (T)0, nullptr, m, P + nfull * m * m, m, 1));
        ++nch;
    }
    hipLaunchKernelGGL(sub_partials_kernel<T>, dim3((unsigned)((m + 15) / 16), (unsigned)((m + 15) / 16)), dim3(256), 0,
                       c->cur, C, ldc, m, P, nch);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
// C (m x n) -= A B^T with few output tiles and a long K (the kept-column contribution to the new columns of a
// short X): the same split as syrk_skinny_sub -- one batched launch over K chunks, then a fixed-order subtraction.
// scratch must not alias A, B or C.  Falls back to the plain launch when there is nothing to gain.
template <typename T>
int gemm_splitk_sub(algp_ctx* c, int klass, int64_t m, int64_t n, int64_t k, const T* A, int64_t lda, const T* B,
                    int64_t ldb, T* C, int64_t ldc, DevBuf& scratch) {
    if (m <= 0 || n <= 0 || k <= 0) return ALGP_OK;
    const int64_t tiles = (m / NB) * (n / NB), kb = k / NB;
    int64_t want = tiles > 0 ? 512 / tiles : 1;                    // chunks that would fill the 512 workgroup slots
    if (want > 128) want = 128;
    int64_t per = want > 0 ? (kb + want - 1) / want : kb;
    if (per < 4) per = 4;
    const int64_t nfull = kb / per, rem = kb - nfull * per;
    if (tiles > 64 || nfull < 2)
        return gemm_nt_launch<T>(c, klass, m, n, k, (T)-1, A, lda, B, ldb, (T)1, C, ldc, C, ldc, 0);
    ALGP_TRY(ensure(c, scratch, sizeof(T) * (size_t)(nfull + 1) * m * n));
    T* P = (T*)scratch.p;
    ALGP_TRY(gemm_nt_launch_batched<T>(c, klass, m, n, per * NB, (T)1, A, lda, per * NB, B, ldb, per * NB, (T)0, nullptr, n,
                                       m * n, P, n, m * n, 0, (int)nfull));
    int nch = (int)nfull;
    if (rem > 0) {
        ALGP_TRY(gemm_nt_launch<T>(c, klass, m, n, rem * NB, (T)1, A + nfull * per * NB, lda, B + nfull * per * NB, ldb,
                                   (T)0, nullptr, n, P + nfull * m * n, n, 0));
        ++nch;
    }
    hipLaunchKernelGGL(sub_partials_rect_kernel<T>, dim3((unsigned)((n + 15) / 16), (unsigned)((m + 15) / 16)), dim3(256), 0,
                       c->cur, C, ldc, m, n, P, nch);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}

template int syrk_skinny_sub<double>(algp_ctx*, int, const double*, int64_t, int64_t, int64_t, double*, int64_t, DevBuf&);
template int syrk_skinny_sub<float>(algp_ctx*, int, const float*, int64_t, int64_t, int64_t, float*, int64_t, DevBuf&);

// ---------------------------------------------------------------------------------------------
// Vector solves (HBM-bound: each reads the lower triangle of L once): forward b <- L^-1 b and backward b <- L^-T b,
// each ONE launch of a workgroup per 128-block with flags between them (trsv_chain_kernel, trsv_chain_back_kernel), and the
// start-up of a forward substitution that resumes at row k (tail_gemv2_kernel).
// ---------------------------------------------------------------------------------------------
// rows [k, k + nrows) of L against the solved leading part: uo[r] -= L[k+r][0:k] . u[0:k] (same for w), one
// wave per row -- the start-up of a forward substitution that resumes at row k
// (Round 5: a workgroup per row, each wave a quarter of the columns with four independent 16-byte loads in flight per lane,
// the quarters added in wave order -- one wave per row walked its 400 KB of config 5's factor alone: 174 us per step.)
template <typename T>
__global__ __launch_bounds__(256) void tail_gemv2_kernel(const T* Lrows, int64_t ld, int64_t nrows, int64_t k, const T* u,
                                                         const T* w, T* uo, T* wo) {
    constexpr int VEC = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ T red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = blockIdx.x;
    const T* row = Lrows + r * ld;
    const int64_t nv = k / VEC, per = (nv + 3) / 4;                // k is a multiple of 128
    const int64_t v0 = wave * per, v1 = v0 + per < nv ? v0 + per : nv;
    T su = (T)0, sw = (T)0;
    int64_t v = v0 + lane;
    for (; v + 192 < v1; v += 256) {
        vec_t x[4], a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            x[q] = *reinterpret_cast<const vec_t*>(row + (v + 64 * q) * VEC);
            a[q] = *reinterpret_cast<const vec_t*>(u + (v + 64 * q) * VEC);
            b[q] = *reinterpret_cast<const vec_t*>(w + (v + 64 * q) * VEC);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                su += x[q][e] * a[q][e];
                sw += x[q][e] * b[q][e];
            }
    }
    for (; v < v1; v += 64) {
        const vec_t x = *reinterpret_cast<const vec_t*>(row + v * VEC);
        const vec_t a = *reinterpret_cast<const vec_t*>(u + v * VEC);
        const vec_t b = *reinterpret_cast<const vec_t*>(w + v * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            su += x[e] * a[e];
            sw += x[e] * b[e];
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        su += __shfl_down(su, o, 64);
        sw += __shfl_down(sw, o, 64);
    }
    if (lane == 0) {
        red[0][wave] = su;
        red[1][wave] = sw;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uo[r] -= (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        wo[r] -= (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

template <typename T>
int tail_gemv2_launch(algp_ctx* c, const T* L, int64_t ldl, int64_t k, int64_t npad, T* u, T* w) {
    const int64_t nrows = npad - k;
    if (nrows <= 0 || k <= 0) return ALGP_OK;
    ProfScope ps(c, ALGP_PROF_TRSV, 4.0 * nrows * k, sizeof(T) * (double)nrows * k);
    hipLaunchKernelGGL(tail_gemv2_kernel<T>, dim3((unsigned)nrows), dim3(256), 0, c->cur, L + k * ldl, ldl, nrows,
                       k, u, w, u + k, w + k);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int tail_gemv2_launch<double>(algp_ctx*, const double*, int64_t, int64_t, int64_t, double*, double*);
template int tail_gemv2_launch<float>(algp_ctx*, const float*, int64_t, int64_t, int64_t, float*, float*);

// ---------------------------------------------------------------------------------------------
// Forward substitution b <- L^-1 b as ONE launch (one or two right-hand sides at once).  The launch sequence above it
// in round 2 -- a 128-thread diagonal kernel and a panel GEMV per block, ~160 launches at N = 10 000 -- spent 0.9 ms per
// right-hand side on launch latency (every fit, and every commit of a pick another rank owns).
// Workgroup i (numbered by an arrival ticket, so that the blocks it waits for are held by workgroups that already run:
// no assumption about dispatch order or residency) owns block row i: it streams L_ij, j < i, as the z_j appear (a flag
// per block; the L segment of block j+1 is loaded while block j is multiplied and flag j+1 is awaited), then multiplies
// by the inverse of its diagonal block and publishes z_i write-through.  512 threads: four per row, 32 columns each.
// ---------------------------------------------------------------------------------------------
// A wait that runs into its time limit (2 s; a lost hand-off) abandons the launch: the abort word ends every other wait, and
// the context's sticky stall word (scal[SC_STALL]) tells the host at its next checked synchronisation -- or the next pick's
// status word -- that what this launch was producing is incomplete (ALGP_ERR_HIP "stalled", as the task-list Cholesky).
// skip_block: test hook (algp_debug_trsv_stall) -- that block's flag is never set.
template <typename T, int NR>
__global__ __launch_bounds__(512) void trsv_chain_kernel(const T* L, int64_t ld, const T* invD, T* b0, T* b1, int kb0, int nblk,
                                                         int* ctrl, int* sticky, unsigned long long spin_limit, int skip_block) {
    __shared__ T zs[NR][128];
    __shared__ T red[NR][4][128];
    __shared__ int s_i, s_ok;
    const int tid = threadIdx.x, r = tid & 127, q = tid >> 7;
    if (tid == 0) s_i = kb0 + atomicAdd(&ctrl[0], 1);
    __syncthreads();
    const int i = s_i;
    if (i >= nblk) return;
    int* flags = ctrl + 8;
    T* bs[2] = {b0, b1};
    T acc[NR];
#pragma unroll
    for (int v = 0; v < NR; ++v) acc[v] = (T)0;
    const T* Lrow = L + ((int64_t)i * 128 + r) * ld + q * 32;
    // Two 32-element buffers per thread (round 5; three before: 254 VGPRs + 16 spilled in fp64): `cur` holds the segment being
    // multiplied, `nxt` the next one -- the next tile's segment of L or, behind the last tile, this thread's 32 entries of the
    // inverse of the diagonal block -- and every prefetch is issued BEFORE the wait for the block's flag, so it flies while the
    // chain's previous block finishes.
    T cur[32], nxt[32];
    const T* Xrow = invD + (int64_t)i * 128 * 128 + r * 128 + q * 32;
    if (kb0 < i) {
#pragma unroll
        for (int e = 0; e < 32; ++e) cur[e] = Lrow[(int64_t)kb0 * 128 + e];
    } else {
#pragma unroll
        for (int e = 0; e < 32; ++e) cur[e] = Xrow[e];
    }
    for (int j = kb0; j < i; ++j) {
        if (j + 1 < i) {
#pragma unroll
            for (int e = 0; e < 32; ++e) nxt[e] = Lrow[(int64_t)(j + 1) * 128 + e];
        } else {
#pragma unroll
            for (int e = 0; e < 32; ++e) nxt[e] = Xrow[e];
        }
        if (tid == 0) {
            bool ok = false;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (unsigned spins = 0;; ++spins) {
                if (__hip_atomic_load(&flags[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = true; break; }
                if (__hip_atomic_load(&ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if ((spins & 255) == 255 && __builtin_amdgcn_s_memrealtime() - t0 > spin_limit) {       // 2 s at 100 MHz
                    atomicCAS(&ctrl[1], 0, i + 1);
                    __hip_atomic_store(sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            s_ok = ok ? 1 : 0;
        }
        __syncthreads();
        if (!s_ok) return;                                         // wave-uniform: the launch is being abandoned
        // z_j was stored write-through and is read past this CU's L1 (sc1): no fence on either side
        if (tid < 128) {
#pragma unroll
            for (int v = 0; v < NR; ++v) zs[v][tid] = __hip_atomic_load(bs[v] + (int64_t)j * 128 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 32; ++e) {
#pragma unroll
            for (int v = 0; v < NR; ++v) acc[v] += cur[e] * zs[v][q * 32 + e];
        }
#pragma unroll
        for (int e = 0; e < 32; ++e) cur[e] = nxt[e];
        __syncthreads();                                           // zs is rewritten in the next round
    }
    // x = b_i - sum_j L_ij z_j (the four column quarters of a row are added in a fixed order), then z_i = X_ii x
#pragma unroll
    for (int v = 0; v < NR; ++v) red[v][q][r] = acc[v];
    __syncthreads();
    if (tid < 128) {
#pragma unroll
        for (int v = 0; v < NR; ++v)
            zs[v][tid] = bs[v][(int64_t)i * 128 + tid] - (((red[v][0][tid] + red[v][1][tid]) + red[v][2][tid]) + red[v][3][tid]);
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NR; ++v) {
        T s = (T)0;
#pragma unroll
        for (int e = 0; e < 32; ++e) s += cur[e] * zs[v][q * 32 + e];          // cur: the inverse block's entries by now
        red[v][q][r] = s;
    }
    __syncthreads();
    if (tid < 128) {
#pragma unroll
        for (int v = 0; v < NR; ++v)
            st_wt(bs[v] + (int64_t)i * 128 + tid, ((red[v][0][tid] + red[v][1][tid]) + red[v][2][tid]) + red[v][3][tid]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && i != skip_block) __hip_atomic_store(&flags[i], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Backward substitution b <- L^-T b as ONE launch (alpha = L^-T z of every posterior mean, algp_get_alpha and MLL gradient;
// reference utils.py:301).  The mirror image of the kernel above: workgroup i (arrival ticket t -> block nblk-1-t, so the
// blocks it waits for are held by workgroups that already run) walks DOWN column block i of L -- the tiles L_ji, j > i, as
// the alpha_j appear -- accumulating sum_j L_ji^T alpha_j: thread (c, q) owns column c and the row quarter q of each tile
// (for a fixed row the 128 threads of a quarter read 128 contiguous elements), the next tile is loaded while the current
// one is multiplied and flag j-1 is awaited.  Then alpha_i = X_ii^T (z_i - sum), published write-through.
// (The launch sequence this replaces was ~160 launches: 0.9 ms at N = 10 000.)
template <typename T>
__global__ __launch_bounds__(512) void trsv_chain_back_kernel(const T* L, int64_t ld, const T* invD, T* b, int nblk, int* ctrl,
                                                              int* sticky, unsigned long long spin_limit, int skip_block) {
    __shared__ T zs[128];
    __shared__ T red[4][128];
    __shared__ int s_i, s_ok;
    const int tid = threadIdx.x, cc = tid & 127, q = tid >> 7;
    if (tid == 0) s_i = nblk - 1 - atomicAdd(&ctrl[0], 1);
    __syncthreads();
    const int i = __builtin_amdgcn_readfirstlane(s_i);             // wave-uniform, and told so: scalar address arithmetic below
    if (i < 0) return;
    int* flags = ctrl + 8;
    T acc = (T)0;
    // row 32 q of a tile in column block i: a wave-uniform base (q is: two waves per row quarter) + the lane's column, so that
    // the 32 row addresses of a tile segment live in scalar registers (as 32 per-lane 64-bit addresses they cost 64 VGPRs:
    // 256 + 14 spilled in fp64)
    const int qs = __builtin_amdgcn_readfirstlane(q);
    const T* Lblk = L + (int64_t)(qs * 32) * ld + (int64_t)i * 128;
#define Lcol_at(row) (Lblk + (int64_t)(row) * ld)[(unsigned)cc]
    // two buffers, prefetch before the flag wait, the inverse block's column behind the last tile: as in the forward kernel
    T cur[32], nxt[32];
    const T* Xcol = invD + (int64_t)i * 128 * 128 + (int64_t)(qs * 32) * 128 + cc;     // column cc of X_ii, rows 32 q ..
    if (i + 1 < nblk) {
#pragma unroll
        for (int e = 0; e < 32; ++e) cur[e] = Lcol_at((int64_t)(nblk - 1) * 128 + e);
    } else {
#pragma unroll
        for (int e = 0; e < 32; ++e) cur[e] = Xcol[e * 128];
    }
    for (int j = nblk - 1; j > i; --j) {
        if (j - 1 > i) {
#pragma unroll
            for (int e = 0; e < 32; ++e) nxt[e] = Lcol_at((int64_t)(j - 1) * 128 + e);
        } else {
#pragma unroll
            for (int e = 0; e < 32; ++e) nxt[e] = Xcol[e * 128];
        }
        if (tid == 0) {
            bool ok = false;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (unsigned spins = 0;; ++spins) {
                if (__hip_atomic_load(&flags[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = true; break; }
                if (__hip_atomic_load(&ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if ((spins & 255) == 255 && __builtin_amdgcn_s_memrealtime() - t0 > spin_limit) {
                    atomicCAS(&ctrl[1], 0, i + 1);
                    __hip_atomic_store(sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            s_ok = ok ? 1 : 0;
        }
        __syncthreads();
        if (!s_ok) return;                                         // wave-uniform: the launch is being abandoned
        if (tid < 128) zs[tid] = __hip_atomic_load(b + (int64_t)j * 128 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            acc += cur[e] * zs[q * 32 + e];
            if ((e & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // eight LDS operands live at a time, not thirty-two
        }
#pragma unroll
        for (int e = 0; e < 32; ++e) cur[e] = nxt[e];
        __syncthreads();                                           // zs is rewritten in the next round
    }
    red[q][cc] = acc;
    __syncthreads();
    if (tid < 128) zs[tid] = b[(int64_t)i * 128 + tid] - (((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid]);
    __syncthreads();
    {
        T s = (T)0;
#pragma unroll
        for (int e = 0; e < 32; ++e) s += cur[e] * zs[q * 32 + e];             // cur: the inverse block's column by now
        red[q][cc] = s;
    }
    __syncthreads();
    if (tid < 128) st_wt(b + (int64_t)i * 128 + tid, ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && i != skip_block) __hip_atomic_store(&flags[i], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#undef Lcol_at
}

// control words of a substitution launch (zeroed per launch) and the test hook's arguments
static int trsv_prepare(algp_ctx* c, int nblk, unsigned long long* spin_limit, int* skip_block) {
    const size_t bytes = sizeof(int) * (size_t)(nblk + 8);
    ALGP_TRY(ensure(c, c->trsv_ctrl, bytes));
    ALGP_HIP(hipMemsetAsync(c->trsv_ctrl.p, 0, bytes, c->cur));
    *spin_limit = 200000000ull;                                    // 2 s at 100 MHz
    *skip_block = -1;
    if (c->debug_trsv_stall_block >= 0) {                          // algp_debug_trsv_stall: one launch with a lost flag
        *skip_block = c->debug_trsv_stall_block;
        *spin_limit = 20000000ull;                                 // 0.2 s
        c->debug_trsv_stall_block = -1;
    }
    return ALGP_OK;
}

template <typename T>
static int trsv_chain_launch(algp_ctx* c, const T* L, int64_t npad, int64_t ldl, const T* invD, T* b0, T* b1, int64_t kb_start) {
    const int nblk = (int)(npad / NB), kb0 = (int)kb_start;
    if (kb0 >= nblk) return ALGP_OK;
    unsigned long long limit;
    int skip;
    ALGP_TRY(trsv_prepare(c, nblk, &limit, &skip));
    int* sticky = (int*)((double*)c->scal.p + SC_STALL);
    const int nr = b1 ? 2 : 1;
    ProfScope ps(c, ALGP_PROF_TRSV, nr * (double)npad * npad, sizeof(T) * 0.5 * (double)npad * npad);
    if (b1) hipLaunchKernelGGL((trsv_chain_kernel<T, 2>), dim3(nblk - kb0), dim3(512), 0, c->cur, L, ldl, invD, b0, b1, kb0, nblk, (int*)c->trsv_ctrl.p, sticky, limit, skip);
    else hipLaunchKernelGGL((trsv_chain_kernel<T, 1>), dim3(nblk - kb0), dim3(512), 0, c->cur, L, ldl, invD, b0, (T*)nullptr, kb0, nblk, (int*)c->trsv_ctrl.p, sticky, limit, skip);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}

template <typename T>
int trsv_forward(algp_ctx* c, const T* L, int64_t npad, int64_t ldl, const T* invD, T* b, int64_t kb_start) {
    return trsv_chain_launch<T>(c, L, npad, ldl, invD, b, (T*)nullptr, kb_start);
}
// two right-hand sides in the same pass over L
template <typename T>
int trsv_forward2(algp_ctx* c, const T* L, int64_t npad, int64_t ldl, const T* invD, T* b0, T* b1, int64_t kb_start) {
    return trsv_chain_launch<T>(c, L, npad, ldl, invD, b0, b1, kb_start);
}
template int trsv_forward2<double>(algp_ctx*, const double*, int64_t, int64_t, const double*, double*, double*, int64_t);
template int trsv_forward2<float>(algp_ctx*, const float*, int64_t, int64_t, const float*, float*, float*, int64_t);
template <typename T>
int trsv_backward(algp_ctx* c, const T* L, int64_t npad, int64_t ldl, const T* invD, T* b) {
    const int nblk = (int)(npad / NB);
    if (nblk <= 0) return ALGP_OK;
    unsigned long long limit;
    int skip;
    ALGP_TRY(trsv_prepare(c, nblk, &limit, &skip));
    ProfScope ps(c, ALGP_PROF_TRSV, (double)npad * npad, sizeof(T) * 0.5 * (double)npad * npad);
    hipLaunchKernelGGL(trsv_chain_back_kernel<T>, dim3(nblk), dim3(512), 0, c->cur, L, ldl, invD, b, nblk, (int*)c->trsv_ctrl.p,
                       (int*)((double*)c->scal.p + SC_STALL), limit, skip);
    ALGP_HIP(hipGetLastError());
    return ALGP_OK;
}
template int trsv_forward<double>(algp_ctx*, const double*, int64_t, int64_t, const double*, double*, int64_t);
template int trsv_forward<float>(algp_ctx*, const float*, int64_t, int64_t, const float*, float*, int64_t);
template int trsv_backward<double>(algp_ctx*, const double*, int64_t, int64_t, const double*, double*);
template int trsv_backward<float>(algp_ctx*, const float*, int64_t, int64_t, const float*, float*);

}  // namespace algp
