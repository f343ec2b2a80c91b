/*
 * algp_hip.h -- C ABI of libalgp_hip.so: the MI355X (gfx950) GP-inference hot path of sumitsk/algp.
 *
 * The reference has NO native/FFI layer (it is 8 Python files); its seam for this path is the
 * duck-typed Python surface
 *     GPR.cov_mat / GPR.set_train_data            (reference models.py:126-135, 161-181)
 *     predictive_distribution / entropy_from_cov  (reference utils.py:188-194, 293-319)
 *     Agent._post_update / greedy / best_path     (reference agent.py:89-90, 295-403)
 * plus NumPy inv / slogdet / dot / argmax.  This header is the boundary a maintainer would bind
 * (ctypes) to route those calls onto the GPU.  Each entry point cites the
 * reference code it replaces.
 *
 * Conventions
 *  - one opaque context per GPU; NOT thread-safe (one caller thread per ctx); every call is
 *    synchronous at the ABI (stream-ordered inside, stream synchronised before return);
 *  - all matrices row-major, contiguous; element type = the ctx dtype (ALGP_F32 / ALGP_F64),
 *    passed as void*; indices int64_t; utilities / entropies / log-dets are always double;
 *  - caller owns every host buffer; the library owns device memory inside the ctx;
 *  - return value: 0 = ALGP_OK, otherwise an ALGP_ERR_* code; algp_last_error() gives text.
 *    A non-positive pivot (reference: LinAlgError from inv, utils.py:300, or a silently wrong
 *    slogdet, utils.py:193) returns ALGP_ERR_NOT_PD and algp_last_pivot() the 1-based index.
 *  - the "pool" is the set of n field locations (reference env.X); train set and candidates are
 *    index lists into it, exactly like Agent's static_data/mobile_data bookkeeping.
 */
#ifndef ALGP_HIP_H
#define ALGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct algp_ctx algp_ctx;

enum { ALGP_F32 = 0, ALGP_F64 = 1 };
enum { ALGP_KERNEL_RBF = 0, ALGP_KERNEL_MATERN15 = 1 };          /* models.py:217-220 */
enum { ALGP_CRIT_ENTROPY = 0, ALGP_CRIT_MUTUAL_INFORMATION = 1 };  /* agent.py:128 */
enum {
    ALGP_OK = 0,
    ALGP_ERR_BAD_ARG = 1,
    ALGP_ERR_HIP = 2,
    ALGP_ERR_NOT_PD = 3,
    ALGP_ERR_OOM = 4,
    ALGP_ERR_STATE = 5,
    ALGP_ERR_NO_DEVICE = 6
};

/* profiling classes for algp_prof_get (HIP-event timed, see DESIGN.md "measurement") */
enum {
    ALGP_PROF_KMAT = 0,        /* kernel-matrix build (HBM-bound)                  */
    ALGP_PROF_GEMM_CHOL = 1,   /* MFMA GEMM launches inside the Cholesky            */
    ALGP_PROF_GEMM_TRSM = 2,   /* MFMA GEMM launches inside the candidate TRSM      */
    ALGP_PROF_POTRF_DIAG = 3,  /* 128x128 diagonal-block factor + inverse           */
    ALGP_PROF_TRSV = 4,        /* vector triangular solves                           */
    ALGP_PROF_ROWS = 5,        /* row reductions over V^T (variance, mean, updates) */
    ALGP_PROF_SCORE = 6,       /* score + argmax                                    */
    ALGP_PROF_GEMM_OTHER = 7,
    ALGP_PROF_CHOLESKY = 8,    /* wall time of whole train-set factorisations (kernel build + factor + z)  */
    ALGP_PROF_TRSM = 9,        /* wall time of whole candidate solves (row chunks overlap on 3 streams) */
    ALGP_PROF_GEMM_CHOL_UPDATE = 10, /* the Cholesky's K=512 trailing (rank-512) updates, a subset of GEMM_CHOL's work
                                      * counted here instead: the "dense panel update" of the blocked factorisation */
    ALGP_PROF_CHOL_DAG = 11,   /* the Cholesky as one dependency-driven launch (chol_dag.hip): the whole factorisation */
    ALGP_PROF_DAG_PANEL = 12,  /* the same launch carrying a row panel: factorisation + candidate solve (or + L^-T), or the solve alone */
    ALGP_PROF_TAIL_COLS = 13,  /* tail_cols_kernel: the new columns of V^T after an append (HBM-bound: s * M * N_old bytes) */
    ALGP_PROF_COUNT = 14
};

/* ---- lifecycle ------------------------------------------------------------------------- */
int algp_version(void);
int algp_device_count(void);
int algp_create(int device_id, int dtype, algp_ctx** out);
void algp_destroy(algp_ctx* ctx);
const char* algp_last_error(const algp_ctx* ctx);
int64_t algp_last_pivot(const algp_ctx* ctx);
/* Diagonal jitter the last algp_get_posterior_cov had to add to cov_xx and cov before the two log-determinants of its
 * MI term existed (0: none).  utils.py:314 takes slogdet of the noise-free K_xx, which is singular to working precision
 * on dense grids; the reference returns rounding noise there (sign dropped, utils.py:193), this library a regularised
 * value plus the jitter it used.  Factorisations of the TRAIN matrix never use a jitter: they fail with ALGP_ERR_NOT_PD. */
double algp_last_jitter(const algp_ctx* ctx);
int algp_dtype(const algp_ctx* ctx);

/* ---- hyper-parameters: ExactGPModel's D+2 scalars (models.py:206-254; names run.py:35-37) ---
 * k(x,x') = exp(log_outputscale) * exp(-1/2 sum_d ((x_d-x'_d)/exp(log_lengthscale[d]))^2)  (RBF)
 * sigma_n^2 = exp(log_noise) (models.py:180).                                                  */
int algp_set_hypers(algp_ctx* ctx, int kernel, int D, const double* log_lengthscale,
                    double log_outputscale, double log_noise);

/* ---- a1: GPR.cov_mat (models.py:161-181) -------------------------------------------------
 * out[n1*n2] (host) = K(x1,x2) ; x2 == NULL -> symmetric K(x1,x1) (models.py:169-170);
 * diag_add (len n1, may be NULL): += diag(white_noise_var) (models.py:175-176), symmetric only;
 * add_likelihood_var: += exp(log_noise) * I (models.py:179-180), symmetric only.              */
int algp_kernel_matrix(algp_ctx* ctx, const void* x1, int64_t n1, const void* x2, int64_t n2,
                       const void* diag_add, int add_likelihood_var, void* out);

/* ---- pool: env.X (agent.py:90) ------------------------------------------------------------
 * algp_set_pool: coordinates n x D (kernel evaluated on the fly on the device).
 * algp_set_pool_cov: an explicit n x n covariance that already contains sigma_n^2 on its
 *   diagonal -- literally Agent.cov_matrix (agent.py:90) -- for callers that hand one over.   */
int algp_set_pool(algp_ctx* ctx, const void* x, int64_t n);
int algp_set_pool_cov(algp_ctx* ctx, const void* cov, int64_t n);

/* ---- a2 + "GP-fit": GPR.set_train_data (models.py:126-135) + factorisation ----------------
 * idx[N] pool indices, y[N] targets (mean-centred inside: models.py:129-130), var[N] per-point
 * noise (may be NULL = 0).  algp_factorize builds S = C_AA + diag(var) (+ sigma_n^2 I unless the
 * pool is an explicit cov that already has it), factors S = L L^T (blocked, MFMA), solves
 * z = L^-1 (y-ybar), alpha = L^-T z, and accumulates log det S.  Replaces np.linalg.inv at
 * utils.py:300 and slogdet at utils.py:193.
 * A pool index may occur in more than one row: independent measurements of the same site, each
 * with its own var (cross entries are C(i,i), the site's first row stands for it as a candidate).
 * Keeping a site's static and mobile means as two rows gives the posterior of the reference's
 * fused row (agent.py:100-109) with log det S larger by exactly log(ss + sm) per such site, and
 * turns a re-measurement into an append for algp_factorize_update.  (Not with the MI criterion.)  */
int algp_set_train(algp_ctx* ctx, const int64_t* idx, int64_t N, const void* y, const void* var);
/* The GP's constant mean is the mean of the train targets (models.py:129-130).  enable != 0 replaces it by
 * `value` for the following algp_set_train calls -- needed when the rows are not one per site (a site kept
 * as two rows must not count twice in the reference's mean of the fused targets).                         */
int algp_set_constant_mean(algp_ctx* ctx, int enable, double value);
int algp_factorize(algp_ctx* ctx);
/* f1: like algp_factorize, but keeps the leading 128-row blocks of the resident
 * factor whose train rows (pool index, noise, order) and hyper-parameters are unchanged and rebuilds
 * only the rest; appending k sites to N costs O((128+k) N^2).  kept_rows (may be NULL) reports how
 * many rows were reused.  The reference refactorises from scratch at every step (agent.py:210).   */
int algp_factorize_update(algp_ctx* ctx, int64_t* kept_rows);
/* Take the factor of the SAME train set (indices, noise, order; same hyper-parameters, dtype and device) from
 * another context instead of computing it: an agent keeps one context per candidate set (the pool for
 * Agent.greedy, the held-out points for Agent.predict, agent.py:289-356) and both need chol(C_AA + D).  Rows this
 * context already holds for an unchanged leading part are kept (*kept_rows), the rest is a device-to-device copy;
 * z / MLL terms are computed for THIS context's targets.  Both pools must address the train sites by the same
 * indices.  ALGP_ERR_STATE when the source's factor belongs to another train set or other hyper-parameters.    */
int algp_factorize_from(algp_ctx* ctx, algp_ctx* src, int64_t* kept_rows);
/* algp_factorize + algp_solve_candidates back to back in one call (one ABI crossing per planning step; what bench.py's
 * step uses).  Needs algp_set_train and algp_set_candidates; same results as the two separate calls.  (Round 1 overlapped
 * the two on separate streams: slower, removed -- the factorisation is now a single dependency-driven launch.)          */
int algp_fit_and_solve(algp_ctx* ctx);
int algp_get_logdet(algp_ctx* ctx, double* logdet);          /* log det S                        */
int algp_get_entropy(algp_ctx* ctx, double* H);              /* N*CONST + 1/2 log det S (utils.py:188) */
int algp_get_alpha(algp_ctx* ctx, void* alpha_out);          /* N values                         */
int algp_get_factor(algp_ctx* ctx, void* L_out);             /* N x N lower, zeros above         */
int algp_get_mll(algp_ctx* ctx, double* mll);                /* -1/2 y0'alpha - 1/2 logdet - N/2 log 2pi */
/* ---- f2: gradient of the MLL for GPR.fit (models.py:137-159; the loss there is -MLL/N) ------
 * grad_out[D+2] = d MLL / d (log_lengthscale[0..D), log_outputscale, log_noise), NOT divided by N.
 * Needs a coordinate pool and a current factorisation.                                          */
int algp_get_mll_grad(algp_ctx* ctx, double* grad_out);
/* The device work of ONE iteration of GPR.fit (models.py:145-158: output = model(train_x); loss = -mll(output, train_y);
 * loss.backward()) in one call, for the current hyper-parameters and train set: = algp_factorize + algp_get_mll +
 * algp_get_mll_grad with the same values, but L^-T comes out of the launch that factors S (the identity rides along as a
 * row panel of the task list) instead of from a separate launch sequence.  mll / grad_out[D+2] may each be NULL. */
int algp_fit_step(algp_ctx* ctx, double* mll, double* grad_out);

/* ---- candidates / test points: predictive_distribution (utils.py:293-319), greedy's pool ---
 * idx[M] pool indices.  A candidate that is itself in the train set (a mobile-sampled site,
 * agent.py:318 only skips static ones) is detected by index equality.
 * prior_includes_noise: 1 -> prior var = C_jj = outputscale + sigma_n^2 (greedy, agent.py:90),
 *                       0 -> prior var = K_jj (+ extra_var[j]) (cov_xx of utils.py:297).
 * algp_solve_candidates: V^T = B^T L^-T by blocked TRSM on MFMA, then row reductions
 *   pv_j = prior_j - |V_j|^2, mu_j = ybar + V_j . z.                                            */
int algp_set_candidates(algp_ctx* ctx, const int64_t* idx, int64_t M, int prior_includes_noise,
                        const void* extra_var);
int algp_solve_candidates(algp_ctx* ctx);
/* f1: like algp_solve_candidates, but keeps the columns of V^T that were solved against unchanged
 * leading rows of the factor (same candidate list, same hyper-parameters) and solves only the
 * rest: exactly the columns an append added when there are at most 64 of them (*kept_cols is then the
 * old train size; one pass over V^T, HBM-bound), else the trailing 128-column blocks.  alive[M] (may
 * be NULL) disables candidates that became static-sampled.                                         */
int algp_solve_candidates_update(algp_ctx* ctx, const uint8_t* alive, int64_t* kept_cols);
/* disable / enable candidates after a solve (alive[M] bytes; 0 = scored as -inf, agent.py:318)      */
int algp_set_candidate_alive(algp_ctx* ctx, const uint8_t* alive);
int algp_get_posterior(algp_ctx* ctx, void* mu_out, void* var_out);      /* either may be NULL  */
/* full M x M posterior covariance (utils.py:305) and mi = H(cov_xx) - H(cov) (utils.py:314);
 * cov_out / mi_out may be NULL.                                                                */
int algp_get_posterior_cov(algp_ctx* ctx, void* cov_out, double* mi_out);
/* mean only, mu = ybar + K_xa alpha with K never materialised (utils.py:301)                   */
int algp_posterior_mean(algp_ctx* ctx, const int64_t* idx, int64_t M, void* mu_out);

/* ---- a7: Agent.greedy (agent.py:295-356) ---------------------------------------------------
 * algp_scores: utilities of every candidate under the current state
 *   entropy: CONST + 1/2 log(pv_j + ss) (unsampled) | 1/2 log(1 + delta s_jj) (mobile-sampled)
 *   (identical to the reference's ent_a - cond, agent.py:341).
 *   out: M doubles; out_is_device != 0 -> `out` is a device pointer (multi-GPU all-gather
 *   buffers owned by the caller).  Committed candidates get -inf (agent.py:314, 318).
 * algp_argmax: first maximum (np.argmax, agent.py:349) over the local scores.
 * algp_best_candidate: the local first maximum of the utilities without scoring every row again.
 *   Entropy gains never grow when more sites are sampled (submodularity), so a utility computed
 *   before the last picks is an upper bound; only the rows whose bound can still win are brought
 *   up to date.  Same winner and value as algp_scores + algp_argmax, bit for bit.  (MI: full scoring.)
 * algp_commit_pick: make pool index `pool_idx` static-sampled (agent.py:352-354): the rank-1 row
 *   append to V^T and to every candidate's pv / s.  The pick is recorded and the rows catch up on
 *   demand (all of them before algp_scores / algp_get_posterior read them).  The index need not be
 *   a local candidate (sharded scoring: every rank commits the global winner).
 * algp_greedy: k picks on one GPU.  utilities_out (k*M doubles, local candidate order) may be
 *   NULL; forced_picks (k pool indices) may be NULL.  With both NULL and the entropy criterion a pick
 *   is one host round trip (see algp_greedy_sharded: the same chain without the gather).
 * MI criterion (agent.py:330-339) is exact and single-GPU: it needs the pool-wide complement. */
int algp_scores(algp_ctx* ctx, int criterion, double static_std, double mobile_std, void* out,
                int out_is_device);
int algp_argmax(algp_ctx* ctx, int64_t* local_pos, int64_t* pool_idx, double* value);
int algp_best_candidate(algp_ctx* ctx, int criterion, double static_std, double mobile_std,
                        int64_t* local_pos, int64_t* pool_idx, double* value);
int algp_commit_pick(algp_ctx* ctx, int64_t pool_idx, double static_std, double mobile_std);
int algp_greedy(algp_ctx* ctx, int criterion, double static_std, double mobile_std, int k,
                const int64_t* forced_picks, int64_t* picks_out, double* utilities_out);

/* ---- a8 / f3: Agent.best_path (agent.py:358-403), entropy criterion, all paths at once -----------------------------
 * algp_score_paths: dH_out[p] = H(A u path_p) - H(A) for npaths enumerated paths, where path p adds a mobile reading
 * (noise mobile_std^2) at each of its distinct sites sites[p*maxlen + a] (pool indices, -1 = no site): the reference
 * takes one slogdet of the enlarged covariance per path (agent.py:386-399); here every path is the log-determinant of
 * its posterior block of at most 256 x 256 (config 5's paths run along field rows of up to ~250 sites, env.py:197-310),
 * computed from the rows of V^T that algp_solve_candidates left resident (all sites of all paths must be resident
 * candidates, no pick committed since the solve): up to 64 distinct sites per path in one LDS kernel, one workgroup per
 * path; 65 .. 256 with the paths' rows gathered, the Gram matrices as one batched MFMA product and the blocks factored as
 * 2 x 2 tiles of 128 (batches of paths sized to ~4 GB of scratch).  More than 256 distinct sites: ALGP_ERR_BAD_ARG.  A site that already is a train row
 * (a statically sampled site crossed by the path) receives a second row, which is the same GP as the reference's fused
 * noise (agent.py:100-109) up to the constant log(sigma_s^2 + sigma_m^2)/2 + CONST per such site (the caller's to
 * subtract, see algp_amd/agent.py); the caller leaves out sites that already have a mobile row (no new reading).  The MI criterion's path utility stays with algp_set_entropy. */
int algp_score_paths(algp_ctx* ctx, const int64_t* sites, int npaths, int maxlen, double mobile_std, double* dH_out);

/* ---- (e) multi-GPU: the loop over candidates (agent.py:317-347) cut into shards, one process and one ctx per GPU ----
 * Every rank factorises the same train set (algp_factorize / algp_fit_and_solve) and holds a share of the candidate list
 * (algp_set_candidates + a solve; ANY partition -- contiguous slices, the strided owner map of algp_comm_set_owners below --
 * and a share may be empty: the first maximum breaks ties by the smaller pool index whatever the shards).  The only communication of the path is ONE
 * all-gather per pick, issued by the library on the context's stream: each rank contributes 32 bytes -- (its best local
 * utility, that candidate's pool index, a status word, the candidate's statistic) -- plus that candidate's row of V^T
 * (Npad + 128 elements: ~80 KB at N = 10 000 fp64), so that a rank which does not own the winner copies the winner's row
 * instead of rebuilding it from the factor ($ALGP_GATHER_ROWS=0: the 32 bytes only, rows rebuilt).
 * algp_comm_unique_id: 128 opaque bytes (ncclUniqueId); one rank calls it, the caller hands them to the others.
 * algp_comm_init: joins this ctx to an RCCL communicator (xGMI) of `nranks` ranks as `rank` (collective: every rank
 *   calls it).  RCCL is opened with dlopen here; without it these return ALGP_ERR_HIP and nothing else is affected.
 * algp_comm_init_host: the same loop over a transport the CALLER owns (MPI, gloo, shared memory): `fn(user, send, recv,
 *   bytes_per_rank)` must all-gather `bytes_per_rank` bytes of host memory in rank order and return 0; it is called
 *   once per pick (twice in the rare extra round) by every rank; one stream synchronisation per pick, in front of it
 *   (the winner is then chosen on the host and only its row goes back to the device).
 *   This is also how two ranks can share one card.
 * Both reserve the exchange's buffers for the current train set; algp_set_train re-reserves them when its size changes.
 * algp_greedy_sharded: k picks (entropy criterion; the MI criterion does not shard).  Per pick, stream-ordered and with
 *   a single read-back (40 bytes): each rank's best candidate resolved on the device (argmax, refresh of the rows
 *   whose bound can still win, argmax), the all-gather, the first maximum in rank order (= np.argmax over the
 *   concatenated scores, agent.py:349, shards being contiguous in rank order), then the commit of the winner on every
 *   rank (enqueued, not waited for).  picks_out: k pool indices (equal on all ranks); utilities_out: their k utilities,
 *   or NULL.  A rank that cannot take part in a pick (no solve, an allocation failure, a failed pack launch, a stalled
 *   one-launch kernel) still joins the gather and reports its error code in the status word: EVERY rank then returns that
 *   code, nobody commits the pick, nobody hangs.  A commit that fails on one rank AFTER the exchange that chose the winner
 *   is reported in that rank's status word of its next pick (every rank returns it from that call); after the LAST pick of
 *   a call it is returned by that rank at once and reported again in the first exchange of its next call -- the one case
 *   in which ranks leave a call with different codes (closing it would take a second collective per call).
 * The active-learning LOOP on sharded candidates (agent.py:125-229: greedy :141 -> _add_samples :66-82 -> predict :196-210,
 * with the loop of agent.py:313-354 cut into shards) -- algp_comm_set_owners: owner[q] = the rank that holds pool site q as
 *   a candidate (-1: nobody; NULL clears the map), the same array on every rank, after algp_comm_init[_host] and
 *   algp_set_pool (which DROPS a map attached before it: the map belongs to its pool; the communicator stays -- it is
 *   joined once per context).  With a map attached algp_factorize_update becomes a COLLECTIVE whatever a rank finds locally
 *   -- a rank that keeps nothing of its factor (an earlier factorisation failed, other hyper-parameters) or cannot
 *   re-allocate it still enters the agreement and says so there; the agreement word also carries a hash of the whole map
 *   (every rank calls it with the same train set): the sites a step appends to the train set are candidates of one rank each, and that rank's row of V^T is
 *   the site's new row of the replicated factor left of the tail block -- so instead of solving those rows against the
 *   kept factor on every rank (38 ms per 256 rows at N = 50 000) the ranks exchange them: a 32-byte agreement word per
 *   rank (status, first changed row, train size, a hash of the plan), then ONE all-gather of cap rows of the kept width
 *   per rank, cap = the largest number of new sites any rank owns (every rank derives the same plan from the train set
 *   and the map).  A rank whose V^T cannot supply its rows (no resident solve for these hyper-parameters / this
 *   candidate list) says so in the agreement and EVERY rank falls back to the solve; an error (allocation, injected) is
 *   returned by every rank, the second collective is then not entered by anyone.  (What is left: a HIP failure of the
 *   pack launch or its copies BETWEEN the two collectives returns from that rank alone -- the buffers are reserved before
 *   the agreement, so this means a lost device, not a full one.)  With any map other than contiguous
 *   shards in rank order the pick's first maximum still equals np.argmax in pool order: equal utilities go to the smaller
 *   pool index.  The calls that follow -- algp_solve_candidates_update on the rank's shard, algp_greedy_sharded -- are
 *   unchanged.  algp_debug_counter(ctx, 1..4): rows of L the last factor update placed without a triangular solve, how many
 *   of them came from other ranks, row exchanges so far, agreed fall-backs so far.                                        */
typedef int (*algp_allgather_fn)(void* user, const void* send, void* recv, int64_t bytes_per_rank);
int algp_comm_set_owners(algp_ctx* ctx, const int32_t* owner, int64_t n_pool);
int algp_comm_unique_id(void* out128);
int algp_comm_init(algp_ctx* ctx, int nranks, int rank, const void* unique_id128);
int algp_comm_init_host(algp_ctx* ctx, int nranks, int rank, algp_allgather_fn fn, void* user);
int algp_comm_destroy(algp_ctx* ctx);
int algp_greedy_sharded(algp_ctx* ctx, int criterion, double static_std, double mobile_std, int k, int64_t* picks_out,
                        double* utilities_out);
/* ---- test hooks -----------------------------------------------------------------------------------------------------
 * Compiled in when ALGP_TEST_HOOKS is 1 -- the default of algp_amd/csrc/Makefile, and what this repository's tests and
 * bench.py (its one-stream leg, the strong-scaling emulation) need; `make TEST_HOOKS=0` builds the library without them
 * (tests/test_abi.py checks that such an object exports no algp_debug_* symbol). */
#ifndef ALGP_TEST_HOOKS
#define ALGP_TEST_HOOKS 1
#endif
#if ALGP_TEST_HOOKS
/* Test hooks of the exchange (no reference counterpart).  algp_debug_first_max: the reduction kernel of the gather on a
 * caller-made buffer of nranks triples (utility, pool index or -1, status) -> out5 = (utility, pool index, owner rank,
 * status, first rank with a non-zero status).  algp_debug_fail_next_pick: this rank reports `code` (an ALGP_ERR_* >= 2)
 * instead of a candidate in its next pick.  algp_debug_counter(ctx, 0): stream synchronisations issued so far.
 * algp_debug_set_trsm_chunks: row-chunk streams of the candidate solve, 1..4 (0: back to the default 3 / $ALGP_TRSM_CHUNKS);
 * every setting gives the same bits -- bench.py uses 1 to time the GEMM launches back to back.
 * algp_debug_dag_stall: in the next one-launch factorisation (chol_dag.hip) the task that draws `ticket` never publishes its
 * tile and the spin limit drops from 2 s to 0.2 s: its consumers run into the limit, the launch aborts itself and
 * algp_factorize returns ALGP_ERR_HIP ("stalled") -- the path a lost hand-off would take; the context stays usable. */
/* algp_debug_trsv_stall: in the next one-launch forward or backward substitution (potrf.hip) the workgroup of 128-block
 * `block` never sets its flag and the spin limit drops to 0.2 s: the launch abandons itself, the context's sticky stall
 * word is set and the call that reads the result back (algp_factorize, algp_get_alpha, algp_commit_pick, the next pick of
 * algp_greedy[_sharded]) returns ALGP_ERR_HIP ("stalled"); the context stays usable.
 * algp_debug_get_pick: pick number q since the last candidate solve as every rank committed it: row_out = the winner's row
 * of V^T (ncols_out elements, the context's dtype; ncols = Npad + q), d_out = its statistic when it was committed.  What a
 * rank receives in the pick's all-gather from the winner's owner; bench.py uses it to fabricate the seven absent ranks of
 * an eight-rank run on one GPU. */
/* algp_debug_fail_at: inject `code` (an ALGP_ERR_* >= 2; 0 disarms) into this rank's next greedy pick at `where`:
 * 0 = while resolving its best candidate (= algp_debug_fail_next_pick), 1 = in the commit of the winner, AFTER the exchange
 * that chose it (the code travels in this rank's status word of its next pick: every rank returns it from that call; after
 * the last pick of a call it is returned by this rank at once and reported again in its next call's first exchange),
 * 2 = the launch that packs its contribution counts as failed (ALGP_ERR_HIP in its status word, the gather still runs),
 * 3 = this rank's agreement word of its next sharded algp_factorize_update carries `code`: every rank returns it. */
int algp_debug_fail_at(algp_ctx* ctx, int where, int code);
int algp_debug_trsv_stall(algp_ctx* ctx, int block);
int algp_debug_get_pick(algp_ctx* ctx, int q, void* row_out, int64_t row_capacity, int64_t* ncols_out, double* d_out);
/* algp_debug_get_factor_rows: rows [row0, row0 + nrows) of the resident factor, columns [0, ncols), row-major into out (the
 * context's dtype) -- what the owners of a step's new train sites contribute to the row exchange; bench.py uses it to
 * fabricate the seven absent ranks of an eight-rank config-5 step on one GPU. */
int algp_debug_get_factor_rows(algp_ctx* ctx, int64_t row0, int64_t nrows, int64_t ncols, void* out);
int algp_debug_first_max(algp_ctx* ctx, const double* triples, int nranks, double out5[5]);
int algp_debug_set_trsm_chunks(algp_ctx* ctx, int chunks);
int algp_debug_fail_next_pick(algp_ctx* ctx, int code);
int algp_debug_dag_stall(algp_ctx* ctx, int ticket);
int64_t algp_debug_counter(algp_ctx* ctx, int which);
#endif /* ALGP_TEST_HOOKS */

/* ---- a5 / a8: entropy_from_cov (utils.py:188-194) and set entropies for best_path --------
 * algp_entropy_from_cov: k*CONST + 1/2 log det cov for a host k x k SPD matrix.
 * algp_set_entropy: H(C[idx,idx] + diag(var)) for pool indices (agent.py:386-387).              */
int algp_entropy_from_cov(algp_ctx* ctx, const void* cov, int64_t k, double* H);
int algp_set_entropy(algp_ctx* ctx, const int64_t* idx, int64_t m, const void* var, double* H);
/* diag((C[idx,idx] + diag(var))^-1), m values, and its entropy (MI terms agent.py:331-338)     */
int algp_set_inverse_diag(algp_ctx* ctx, const int64_t* idx, int64_t m, const void* var,
                          void* diag_out, double* H);

/* ---- dense building blocks on host matrices (parity tests; also usable on their own) ------
 * algp_cholesky: L (n x n lower, zeros above) of a host SPD matrix, optional log det.
 * algp_gemm_nt: D = alpha * A(m x k) * B(n x k)^T + beta * C(m x n).
 * algp_trsm_right_lt: X = B * L^-T for B (m x n), L (n x n lower) -- the candidate solve.     */
int algp_cholesky(algp_ctx* ctx, const void* A, int64_t n, void* L_out, double* logdet);
int algp_gemm_nt(algp_ctx* ctx, int64_t m, int64_t n, int64_t k, double alpha, const void* A,
                 const void* B, double beta, const void* C, void* D);
int algp_trsm_right_lt(algp_ctx* ctx, const void* L, int64_t n, const void* B, int64_t m,
                       void* X_out);
/* MFMA fragment-layout probe: runs one 16x16x4 MFMA per dtype on exact integer operands and
 * returns the number of mismatching outputs (0 expected).                                      */
int algp_selftest_mfma(algp_ctx* ctx, int* mismatches);
/* device-resident GEMM timing on pseudo-random operands (no host traffic): average ms per launch
 * of D = C - A B^T (beta_one) or D = -A B^T, m x n x k, lower tiles only or all.                      */
int algp_bench_gemm(algp_ctx* ctx, int64_t m, int64_t n, int64_t k, int lower_only, int beta_one, int reps,
                    double* ms_per_launch);

/* ---- resident-buffer access for benchmarks / multi-GPU plumbing --------------------------- */
int algp_sync(algp_ctx* ctx);
int64_t algp_device_bytes(const algp_ctx* ctx);

/* ---- profiling: HIP events around every launch of a class, on the ctx stream ------------- */
int algp_prof_enable(algp_ctx* ctx, int on);
int algp_prof_reset(algp_ctx* ctx);
int algp_prof_get(algp_ctx* ctx, int klass, double* ms, double* flops, double* bytes,
                  int64_t* launches);
/* In-kernel accounting of the one-launch Cholesky since the last algp_prof_reset (collected only while profiling is
 * enabled): out[0] = microseconds spent inside the rank-k update products ("panel update", reference: the LU inside
 * np.linalg.inv / slogdet, utils.py:193, 300), summed over workgroups; out[1] = their K = 128 steps (2 * 128^3 flop
 * each); out[2], out[3] = the same for the triangular-solve tile products (microseconds, count). Two workgroups
 * share a CU, so the update rate while computing is 2 * 128^3 * out[1] / (out[0] / 2 / CUs) flop/s. */
int algp_cholesky_task_stats(algp_ctx* ctx, double out[4]);

#ifdef __cplusplus
}
#endif
#endif /* ALGP_HIP_H */
