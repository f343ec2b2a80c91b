// api_impl.h -- the typed host-side implementation behind the C ABI (include/algp_hip.h), declared once and defined by
// concern: api.hip (context, hyper-parameters, pool, train set, the stand-alone matrix entry points), api_factor.hip (the
// factor of the train set and its updates), api_candidates.hip (candidate solve and posterior), api_greedy.hip (scoring,
// picks, the sharded exchange, the MI criterion), api_paths.hip (best_path block scoring), api_fit.hip (MLL gradient, one
// fit iteration).  Every file defines its members of Impl<T> and instantiates Impl<float> / Impl<double> for them.
#pragma once
#include <limits.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <stdlib.h>

#include "common.h"
#include "vecops.h"

namespace algp {

void release(algp_ctx* c, DevBuf& b);
hipEvent_t sync_event_api(algp_ctx* c, size_t i);
KmatSrc make_src(algp_ctx* c);
int sync(algp_ctx* c);
int sync_checked(algp_ctx* c, const char* what);

template <typename T>
struct Impl {
    static T* p(DevBuf& b) { return (T*)b.p; }
    static int rescale_pool(algp_ctx* c);
    static int kernel_matrix(algp_ctx* c, const void* x1, int64_t n1, const void* x2, int64_t n2, const void* diag_add,
                             int add_lik, void* out);
    static int set_pool(algp_ctx* c, const void* x, int64_t n);
    static uint64_t train_sites_hash(const algp_ctx* c, const std::vector<int64_t>& idx);
    static int set_pool_cov(algp_ctx* c, const void* cov, int64_t n);
    static int set_train(algp_ctx* c, const int64_t* idx, int64_t N, const void* y, const void* var);

    // Rows that ride along with a factorisation as extra block rows of its task list (chol_dag.hip): P <- P L^-T comes out
    // of the same launch.  done: the launch took them (otherwise the caller solves them afterwards).
    struct Panel {
        T* P;
        int64_t ldp, mpad;
        int mode;                  // 1: dense rows (the candidates' B^T), 2: the identity (-> L^-T)
        bool done;
        T* inv_out = nullptr;      // mode 2: S^-1 = P P^T (lower tiles, ld = mpad) is enqueued on the helper stream right behind
        bool inv_enqueued = false; // the launch, beside the substitutions and read-backs that follow on the main stream
        int64_t z_row = -1;        // this row of P holds y - ybar (mode 1: a padding row; mode 2: a dense tile row behind the identity):
                                   // z^T = (y - ybar)^T L^-T comes out of the launch too
        int64_t short_rows = 0;    // mode 1: the first short_rows rows (a multiple of 128) leave their last column tile to the caller
                                   // (the tail kernel solves a narrow last tile's columns afterwards: fit_and_solve)
    };
    static bool panel_fits(int64_t npad, int64_t mpad);
    static int factor_resident(algp_ctx* c, T* A, int64_t n, int64_t npad, T* invD, int slot_logdet, int slot_info,
                               double* logdet, int64_t ld = 0, int64_t pivot_offset = 0, Panel* panel = nullptr);
    static int reserve_factor(algp_ctx* c, int64_t npad_need, int64_t keep_rows, int64_t keep_height = 0,
                              bool headroom = false);
    static bool vt_rows_for_new_sites(algp_ctx* c, int64_t Nb, int64_t p0, std::vector<int64_t>& src_row,
                                      std::vector<int64_t>& lrow, std::vector<T>& lscale, bool& any_second);
    static int exchange_new_rows(algp_ctx* c, int64_t Nb, int64_t p0, int* placed, int st_in = 0);
    static int factorize(algp_ctx* c, int incremental, Panel* panel = nullptr);
    static int finish_factor(algp_ctx* c, int64_t keep, int64_t p0, double ld_total, T* z_src = nullptr);
    static int factorize_from(algp_ctx* c, algp_ctx* src);
    static int need_alpha(algp_ctx* c);
    static int set_candidates(algp_ctx* c, const int64_t* idx, int64_t M, int prior_noise, const void* extra);

    // V^T = B^T L^-T for the candidate list, then pv / s / mu.  With `incremental`, the columns that
    // were solved against rows of the factor that are unchanged (same leading train rows, same
    // hyper-parameters, same candidate list) are kept and only the trailing column blocks are solved.
    // `alive` (M bytes, may be null) disables candidates (sites that became static-sampled).
    // Three parts, so that algp_fit_and_solve can put the factorisation between the first two and let the rows of B^T
    // ride along in its launch: solve_prepare (buffers, candidate kinds, B^T), the solve itself, solve_finish (row
    // statistics, bookkeeping).
    struct SolvePlan {
        int64_t keep = 0;                    // leading columns of V^T that stay
        std::vector<int> kind;               // per candidate: its train row (a unit right-hand side) or -1
        std::vector<int64_t> became_unit;
        bool carried_sums = false;           // solve_finish will carry the rows' sums from step to step (u, w form of z)
        bool rowstat_done = false;           // solve_run's launches left the rows' sums per column tile in c->rowstat
        int nseg = 0;                        // > 0: only the new columns are solved (tail.hip), as 1-2 ranges [seg_c0, seg_c0 + seg_w)
        int64_t seg_c0[2] = {0, 0};
        int seg_w[2] = {0, 0};
        bool seg_window = false;             // the one range straddles two 128-column blocks of the factor
    };
    static int solve_prepare(algp_ctx* c, int incremental, SolvePlan& pl);
    static int solve_run(algp_ctx* c, SolvePlan& pl);
    static int solve_finish(algp_ctx* c, int incremental, const unsigned char* alive_host, const SolvePlan& pl);
    static int solve_candidates(algp_ctx* c, int incremental, const unsigned char* alive_host);
    static int fit_and_solve(algp_ctx* c);
    static int get_posterior(algp_ctx* c, void* mu, void* var);
    static int get_posterior_cov(algp_ctx* c, void* cov_out, double* mi_out);
    static int posterior_mean(algp_ctx* c, const int64_t* idx, int64_t M, void* mu_out);
    static int build_set_matrix(algp_ctx* c, const int64_t* idx, int64_t m, const void* var, int64_t* mpad_out, T* dst = nullptr);
    static int set_entropy(algp_ctx* c, const int64_t* idx, int64_t m, const void* var, double* H);
    static int inverse_diag_resident(algp_ctx* c, int64_t m, int64_t mpad, void* diag_out);
    static int set_inverse_diag(algp_ctx* c, const int64_t* idx, int64_t m, const void* var, void* diag_out, double* H);
    static int upload_padded(algp_ctx* c, DevBuf& b, const void* A, int64_t rows, int64_t cols, int64_t rpad,
                             int64_t cpad);
    static int entropy_from_cov(algp_ctx* c, const void* cov, int64_t k, double* H, void* L_out, double* logdet);
    static int gemm_host(algp_ctx* c, int64_t m, int64_t n, int64_t k, double alpha, const void* A, const void* B,
                         double beta, const void* C, void* D);
    static int trsm_host(algp_ctx* c, const void* L, int64_t n, const void* B, int64_t m, void* X);
    static int mi_build(algp_ctx* c, double ss, double sm);
    static int mi_apply_pick(algp_ctx* c, int64_t q, double ss, double sm);
    static int mi_scores_enqueue(algp_ctx* c, double ss, double sm, double delta, double* dst);
    static int scores_enqueue(algp_ctx* c, int criterion, double static_std, double mobile_std, double* dst);
    static int scores(algp_ctx* c, int criterion, double static_std, double mobile_std, void* out, int out_is_device);
    static int argmax(algp_ctx* c, int64_t* local_pos, int64_t* pool_idx, double* value);

    // Row of V^T for a pool index that is not a local candidate (sharded scoring: the global winner lives on another
    // rank), entirely on the device: kernel-matrix row, forward substitution against the replicated factor, then the
    // entries appended by earlier picks through the SAME kernels a local row goes through (rows_reduce +
    // cand_finalize for the statistic, lazy_refresh for the picks), so the row and its statistic equal the owner's bit
    // for bit.  The statistic stays on the device (c->remote); nothing is read back here.
    struct RemoteSlots {            // one-row stand-ins for the per-candidate arrays, 64 bytes apart in c->remote
        int64_t* cidx;
        int* ckind;
        T *ss, *dot, *dstat, *mu;
        unsigned char* alive;
        int* fresh;
        double* score;
    };
    static RemoteSlots remote_slots(algp_ctx* c);
    static int remote_row(algp_ctx* c, int64_t pool_idx, int in_train);
    static int commit_enqueue(algp_ctx* c, int64_t pool_idx, double ss, double delta, const char* winner_payload = nullptr);
    static int commit_pick(algp_ctx* c, int64_t pool_idx, double static_std, double mobile_std);
    static int lazy_launch(algp_ctx* c, int mode, int64_t pos, double ss, double delta, const int64_t* pos_dev = nullptr);
    static int reset_lazy(algp_ctx* c);
    static int flush_lazy(algp_ctx* c);
    static int enqueue_local_best(algp_ctx* c, double ss, double delta);
    static int ensure_bounds(algp_ctx* c, int criterion, double static_std, double mobile_std, double ss, double delta);
    static int best_candidate(algp_ctx* c, int criterion, double static_std, double mobile_std, int64_t* local_pos,
                              int64_t* pool_idx, double* value);
    static int greedy_picks(algp_ctx* c, double static_std, double mobile_std, int k, int64_t* picks_out, double* ut_out);
    static int greedy(algp_ctx* c, int criterion, double static_std, double mobile_std, int k, const int64_t* forced,
                      int64_t* picks_out, double* ut_out);
    static int score_paths_big(algp_ctx* c, const std::vector<int64_t>& cpos, const std::vector<int64_t>& lpos, int npaths, int maxlen,
                               int maxused, double mobile_std, double* dH);
    static int score_paths(algp_ctx* c, const int64_t* sites, int npaths, int maxlen, double mobile_std, double* dH);
    static int mll_grad(algp_ctx* c, double* grad_out, bool have_X = false, bool inv_enqueued = false);
    static int fit_step(algp_ctx* c, double* mll_out, double* grad_out);
    static int get_alpha(algp_ctx* c, void* out);
    static int get_factor(algp_ctx* c, void* out);
    static int selftest(algp_ctx* c, int* mism);
};

}  // namespace algp

#define DISPATCH(c, call) ((c)->dtype == ALGP_F64 ? Impl<double>::call : Impl<float>::call)
#define CHECK_CTX(c) do { if (!(c)) return ALGP_ERR_BAD_ARG; (c)->err.clear(); hipSetDevice((c)->device); } while (0)
#define NEED_HYPERS(c) do { if (!(c)->hyp.set) return fail(c, ALGP_ERR_STATE, "call algp_set_hypers first"); } while (0)
#define FINISH(c, expr) do { int rc__ = (expr); if ((c)->prof_on) prof_collect(c); return rc__; } while (0)

