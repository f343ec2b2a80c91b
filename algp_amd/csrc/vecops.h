// launchers defined in vecops.hip
#pragma once
#include "common.h"
namespace algp {
template <typename T>
int cand_finalize_launch(algp_ctx* c, int64_t M, const int* ckind, const int64_t* cidx, const T* Cp, int64_t n_pool,
                         T prior_const, const T* extra, const T* ss, const T* dot, T ybar, T* dstat, T* mu,
                         unsigned char* alive);
template <typename T>
int score_launch(algp_ctx* c, int64_t M, const int* ckind, const unsigned char* alive, const T* dstat, double ss,
                 double delta, const double* extra, double* out);
int argmax_launch(algp_ctx* c, const double* s, int64_t M, double* out_val, int64_t* out_idx);
template <typename T>
int rows_reduce3_launch(algp_ctx* c, const T* Vt, int64_t rows, int64_t ldv, int64_t c0, int64_t c1, const T* u, const T* w,
                        T* out3, int64_t stride, int accumulate);
template <typename T>
int combine3_launch(algp_ctx* c, int64_t M, const T* acc, const T* tmp, int64_t stride, T ybar, T* ss, T* dot);
template <typename T>
int upper_gemv_launch(algp_ctx* c, const T* X, int64_t ld, int64_t n, const T* z, T* out);
template <typename T>
int zero_rows3_launch(algp_ctx* c, T* acc3, int64_t stride, const int64_t* rows, int64_t n);
template <typename T>
int zero_listed_rows_launch(algp_ctx* c, T* X, int64_t ldx, const int64_t* rows, int64_t n, int64_t ncols);
template <typename T>
int rowstat_combine_launch(algp_ctx* c, const T* stat, int64_t ld, int ntiles, int64_t rows, T* ss, T* dot);
template <typename T>
int uw_init_launch(algp_ctx* c, T* u, T* w, const T* y, int64_t k, int64_t n, int64_t npad);
template <typename T>
int uw_combine_launch(algp_ctx* c, T* z, const T* u, const T* w, T ybar, int64_t npad);
template <typename T>
int gather_rows_launch(algp_ctx* c, const T* src, int64_t lds, const int64_t* src_row, T* dst, int64_t ldd, int64_t nrows,
                       int64_t ncols, const int64_t* lrow = nullptr, const T* lscale = nullptr, const T* Lb = nullptr,
                       int64_t ldl = 0);
// lazy greedy refresh (vecops.hip): mode 0 = row pos, 1 = stale rows whose bound reaches scores[pos], 2 = all stale;
// pos_dev != null: the row index is read from the device (what an argmax kernel left there; < 0 = nothing to do)
template <typename T>
int lazy_refresh_launch(algp_ctx* c, int64_t M, int mode, int64_t pos, const LazyPick* picks, int npicks, const int* ckind,
                        const int64_t* cidx, const T* Xs, const T* Cp, int64_t n_pool, int DP, int kernel, T os, T noise,
                        const T* prevrows, int64_t ldv, T* Vt, T* dstat, int* fresh, const unsigned char* alive,
                        double* scores, double ss, double delta, const int64_t* pos_dev = nullptr);
// best_path block scoring: dH of every path from the resident rows of V^T (one workgroup per path, <= 64 sites each)
template <typename T>
int path_score_launch(algp_ctx* c, const int64_t* cpos, const int64_t* lpos, int npaths, int maxlen, const int64_t* cidx,
                      const T* Vt, int64_t ldv, int64_t ncols, const T* L, int64_t ldl, const T* varA, const T* Xs,
                      const T* Cp, int64_t n_pool, int DP, int kernel, double os, double noise, double sm, double* out);
// paths of 65 .. 256 sites: G = C_PP + sigma_m^2 I - Gram in place for a batch of ppad x ppad blocks (sites packed to the front
// of each path's ppad entries of cpos, -1 behind them); out[p] = P CONST + 1/2 logdet[p] (NaN where info[p] != 0)
template <typename T>
int path_assemble_launch(algp_ctx* c, const int64_t* cpos, int batch, int ppad, const int64_t* cidx, const T* Xs, const T* Cp,
                         int64_t n_pool, int DP, int kernel, double os, double noise, double sm, T* G);
int path_finish_launch(algp_ctx* c, const int64_t* cpos, int ppad, int batch, const double* logdet, const int* info, double* out);
// greedy commit bookkeeping on the device: scale of the appended row, pick record, winner retired, (d_c, scale) out
template <typename T>
int commit_finalize_launch(algp_ctx* c, const T* dsrc, int in_train, double ss, double delta, LazyPick* lp_out,
                           int64_t pool_idx, int64_t ncols, unsigned char* alive_local, double* score_local, double* out2);
int fresh_at_launch(algp_ctx* c, const int* fresh, const int64_t* idx, double* out);
// MI criterion: fold one committed pick into a resident inverse (its diagonal, its rank-1 list, its entropy), and score
template <typename T>
int mi_rank1_launch(algp_ctx* c, int64_t m, const T* col0, T* U, int64_t ldu, double* sgn, int q, int64_t cpos, int mode,
                    double dlt, T* diag, double* H, double* H_A, const LazyPick* pick);
template <typename T>
int mi_score_launch(algp_ctx* c, int64_t M, const int* ckind, const int64_t* cidx, const unsigned char* alive, const T* dstat,
                    double ss, double delta, const int64_t* posbar, const T* dP, const T* dQ, const double* Hs, double* out);
template <typename T>
int kgemv_launch(algp_ctx* c, int64_t M, const int64_t* qidx, const T* Xs, int DP, int64_t N, const int64_t* aidx,
                 const T* alpha, int kernel, T os, T ybar, T* mu);
template <typename T>
int pad_identity_launch(algp_ctx* c, T* A, int64_t n, int64_t npad, int64_t ld);
template <typename T>
int set_identity_launch(algp_ctx* c, T* A, int64_t npad, int64_t ld);
template <typename T>
int add_diag_launch(algp_ctx* c, T* A, int64_t n, int64_t ld, T v);
template <typename T>
int logdiag_launch(algp_ctx* c, const T* L, int64_t ld, int64_t n, double* out);
template <typename T>
int to_double_launch(algp_ctx* c, double* dst, const T* src, int64_t n);
// inverse of the NB x NB lower-triangular diagonal block at A (no factorisation)
template <typename T>
int trinv_diag_launch(algp_ctx* c, const T* A, int64_t lda, T* inv_out);
template <typename T>
int mll_grad_launch(algp_ctx* c, const T* Sinv, int64_t ld, int64_t N, const T* Xs, int DP, const int64_t* aidx,
                    const T* alpha, int kernel, T os, double* out_dev, double* partial);
}  // namespace algp
