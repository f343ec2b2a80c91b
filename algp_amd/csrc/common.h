// Internal declarations shared by the HIP translation units of libalgp_hip.so (gfx950 only).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/algp_hip.h"

namespace algp {

constexpr int NB = 128;          // factor block / GEMM tile edge; every device matrix is padded to it
constexpr int MAXD = 8;          // coordinates are stored scaled by 1/lengthscale and zero-padded to DP
constexpr int MAX_APPEND = 128;  // rows a greedy run may append to V^T (ldv = Npad + MAX_APPEND)
constexpr double ENT_CONST = 1.4189385332046727;  // 1/2 log(2 pi e)  (reference utils.py:10)

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// slots (doubles) of ctx->scal, the context's device scalars
constexpr int SC_LOGDET = 0;
constexpr int SC_INFO = 1;       // int stored in a double slot
constexpr int SC_AUXLOGDET = 2;
constexpr int SC_AUXINFO = 3;
constexpr int SC_AMAXV = 4;
constexpr int SC_AMAXI = 5;
constexpr int SC_PROBE = 6;
constexpr int SC_COMMIT = 8;     // (d_c, scale) of the pick being committed
constexpr int SC_AMAXF = 10;     // fresh[argmax] of the lazy greedy
constexpr int SC_STALL = 12;     // int, sticky: a one-launch kernel with inter-workgroup hand-offs gave up (sync_checked)
constexpr int SC_GRAD = 16;      // 16..27: partial sums of the MLL gradient
constexpr int SC_COUNT = 32;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct ProfSlot {
    double ms = 0, flops = 0, bytes = 0;
    int64_t launches = 0;
};

struct PendingEvent {
    hipEvent_t a, b;
    int klass;
};

struct Hypers {
    int kernel = ALGP_KERNEL_RBF;
    int D = 0, DP = 2;
    double inv_ls[MAXD] = {0};
    double outputscale = 1.0, noise = 1.0;
    bool set = false;
};

struct PickRec {           // one committed greedy pick, host side (the device keeps a LazyPick with the row's scale)
    int64_t pool_idx;
    int in_train;          // winner was a mobile-sampled train site (rank-1 noise change)
};

struct DagCache {          // device copy of one task list of the dependency-driven Cholesky (chol_dag.hip)
    int64_t nt;
    int mt = 0, mode = 0, solve_only = 0, pshort = 0;   // the row panel carried below the factor (DagShape)
    DevBuf tasks;
    DevBuf init;           // template of the per-launch state for lists whose tiles do not all start at version 0 (or null)
    int ntasks;
    int workers;           // workgroups the schedule was simulated for = the grid it is launched with
};

struct LazyPick {          // device copy of a committed pick for the lazy greedy refresh (vecops.hip)
    int64_t pool_idx;
    int64_t ncols;         // columns of V^T the pick's dot product covers = the column its entry goes to
    double scale;
    int64_t in_train;
    double d;              // the winner's statistic (pv or s) when it was committed: what `scale` was computed from
};

}  // namespace algp

namespace algp {
// exp(x) for the kernel values, x <= 0 in every use (-r^2/2, -sqrt(3) r): ONE definition for every kernel that evaluates
// k(x, x') -- the matrix build, the lazy refresh's b', the path scores, the mean-only posterior, the MLL gradient -- so
// that a value recomputed later has the bits of the one stored earlier.  fp64: Cody-Waite reduction x = k ln2 + r,
// |r| <= ln2 / 2, Taylor to r^13 (truncation 4e-18), v_ldexp: ~20 instructions where the library routine takes ~55 (the
// matrix build was VALU-bound on it: profiles/r02_valu_by_kernel.json); within 1 ulp of the correctly rounded value on
// [-745, 0], exact at 0, underflows to 0 like it.  fp32: the library's.
__device__ __forceinline__ float kexp(float x) { return expf(x); }
__device__ __forceinline__ double kexp(double x) {
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821614599e-10;                          // 1/13!
    p = __builtin_fma(p, r, 2.0876756987868098979e-09);            // 1/12!
    p = __builtin_fma(p, r, 2.5052108385441718775e-08);            // 1/11!
    p = __builtin_fma(p, r, 2.7557319223985890653e-07);            // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985892511e-06);            // 1/9!
    p = __builtin_fma(p, r, 2.4801587301587301566e-05);            // 1/8!
    p = __builtin_fma(p, r, 1.9841269841269841253e-04);            // 1/7!
    p = __builtin_fma(p, r, 1.3888888888888889419e-03);            // 1/6!
    p = __builtin_fma(p, r, 8.3333333333333332177e-03);            // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664354e-02);            // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666665741e-01);            // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);
}
}  // namespace algp

struct algp_ctx {
    int device = 0;
    int dtype = ALGP_F64;
    size_t es = 8;
    hipStream_t stream = nullptr;    // main stream (all results are complete on it before an ABI call returns)
    hipStream_t stream2 = nullptr;   // helper streams: independent row chunks of the candidate solve overlap on them
    hipStream_t stream3 = nullptr, stream4 = nullptr;
    int trsm_chunks = 3;
    hipStream_t cur = nullptr;       // stream the launch helpers currently target
    std::vector<hipEvent_t> sync_events;
    std::string err;
    int64_t pivot = 0;
    double last_jitter = 0;          // diagonal jitter the last algp_get_posterior_cov needed for its MI term (0: none)
    algp::Hypers hyp;

    // pool
    int64_t n_pool = 0;
    bool pool_is_cov = false;
    algp::DevBuf Xs;        // n x DP scaled, zero padded coordinates
    algp::DevBuf Xraw;      // n x D raw coordinates (kept to rescale on a hyper change)
    algp::DevBuf Cp;        // n x n explicit covariance (pool_is_cov)
    std::vector<int64_t> pos_in_train;   // host: pool index -> position in train set or -1
    std::vector<uint64_t> site_hash;     // host: fingerprint of every pool site's coordinates (algp_factorize_from)

    // train set / factor
    int64_t N = 0, Npad = 0;
    int64_t Lld = 0;                     // leading dimension (= capacity) of L; >= Npad
    int64_t Nfact = 0;                   // rows of L that are valid for (fact_idx, fact_var)
    std::vector<int64_t> fact_idx;       // train set the resident factor was computed for
    std::vector<double> fact_var;
    std::vector<double> train_var_host;
    uint64_t fact_hyp_stamp = 0, hyp_stamp = 1;
    bool train_dirty = false;
    int64_t kept_rows_last = 0;
    std::vector<int64_t> train_idx;
    bool mean_override = false;          // algp_set_constant_mean: the GP's constant mean instead of mean(train y)
    double mean_value = 0;
    bool train_has_repeats = false;      // a pool index occurs in more than one train row
    algp::DevBuf Aidx, yA, varA, y0, L, invD, z, alpha, scal;   // scal: device doubles (logdet, info...)
    double ybar = 0, logdet = 0, yalpha = 0;
    bool factored = false;
    // z = L^-1 (y - ybar 1) = u - ybar w with u = L^-1 y, w = L^-1 1: both are prefix-stable under row appends, so a
    // factor update only extends them (uw_rows leading rows valid for fact_idx / fact_var / fact_y)
    algp::DevBuf yraw, uvec, wvec;
    std::vector<double> train_y_host, fact_y;
    int64_t uw_rows = 0;
    int64_t uw_stable = 0;               // leading entries of u, w unchanged since the row sums below were started
    // per candidate: sums over the kept columns [0, acc_cols) of V^T of v^2, v u, v w (incremental solve)
    algp::DevBuf acc3;
    algp::DevBuf rowstat;                // per column tile of V^T: row sums of v^2 and v z left by the solve's own launches
    algp::DevBuf splitk;                 // partial products of split-K launches (skinny solves)
    algp::DevBuf ldpart;                 // logdiag_kernel's per-wave partial sums
    algp::DevBuf tailPart;               // tail.hip: partial accumulators of the k-split form
    algp::DevBuf tailE;                  // tail.hip: inverse of the 128 x 128 window of L at the first new column (a range that straddles two blocks)
    algp::DevBuf inv512, inv512_scr, trsm_tmp;   // candidate solve: explicit inverses of the factor's 512-column blocks, their scratch, mpad x 512
    std::vector<algp::DagCache> dag_cache;   // task lists of the dependency-driven Cholesky, per matrix size
    algp::DevBuf dag_state;              // its per-launch tile versions / control words / per-block log-determinants
    algp::DevBuf trsv_ctrl;              // ticket + per-block flags of the one-launch forward substitution (potrf.hip)
    algp::DevBuf dag_stats;              // its in-kernel task accounting (while profiling is on), see algp_cholesky_task_stats
    int64_t acc_cols = 0, acc_M = -1;
    int64_t factor_rows_from_vt = 0;     // last factor update: rows of L taken from V^T instead of a triangular solve
    bool alpha_valid = false;            // alpha = L^-T z is computed on first use (scoring does not need it)

    // candidates
    int64_t M = 0, Mpad = 0, ldv = 0;
    int64_t ncols = 0;       // active columns of V^T (Npad + appended)
    int prior_noise = 1;
    std::vector<int64_t> cand_idx;
    std::vector<int64_t> cand_pos;       // host: pool index -> local candidate position or -1
    algp::DevBuf Cidx, ckind, cextra, Vt, dstat, mu, alive, scores, lrow, tvec, amax;
    std::vector<algp::PickRec> picks;
    algp::DevBuf prevrows;   // MAX_APPEND x ldv: the l-rows of committed picks
    algp::DevBuf remote;     // one-row stand-ins for the per-candidate arrays while a remote winner's row is rebuilt
    // lazy greedy: fresh[j] = number of committed picks already applied to row j of V^T / dstat[j]
    algp::DevBuf fresh, lazypicks;
    bool lazy_stale = false;             // some rows lag behind picks.size(): flush before reading the full state
    bool bounds_valid = false;           // c->scores = entropy utility of each row as of fresh[row] picks, for (lazy_ss, lazy_delta)
    double lazy_ss = 0, lazy_delta = 0;
    bool solved = false;
    int64_t ldv_cap = 0;                 // allocated leading dimension of V^T
    // what the resident V^T columns were solved for (incremental candidate solve)
    std::vector<int64_t> vt_fact_idx, vt_cand_idx;
    std::vector<double> vt_fact_var;
    std::vector<int> vt_kind;            // per candidate: position in the train set it was solved as, or -1
    uint64_t vt_hyp_stamp = 0;
    int vt_prior_noise = -1;
    bool vt_has_extra = false;
    int64_t kept_cols_last = 0;

    // MI criterion (agent.py:330-339): the triangular inverses of the two pool-wide matrices stay resident between picks
    // (api.hip mi_build / mi_apply_pick); miH = [H(A), H(Abar), H(all) | signs of the rank-1 terms of P | of Q]
    algp::DevBuf miXbar, miXall, miDP, miDQ, miPos, miU, miW, miCol, miH;
    std::vector<int64_t> mi_posbar;      // host: pool index -> row of the complement matrix when it was built (-1: sampled)
    bool mi_valid = false;
    int64_t mi_npicks = 0, mi_base = 0, mi_nbar = 0;   // picks folded in; picks.size() at the build; rank-1 terms of P so far
    int64_t mi_mb = 0, mi_mbpad = 0, mi_npad = 0;
    double mi_ss = 0, mi_sm = 0;

    // multi-GPU: transport of the sharded greedy loop's one all-gather (comm.hip): an RCCL communicator
    // (algp_comm_init), a caller-supplied host all-gather (algp_comm_init_host), or neither (one rank)
    void* comm = nullptr;
    algp_allgather_fn host_gather = nullptr;
    void* host_gather_user = nullptr;
    int comm_nranks = 1, comm_rank = 0;
    algp::DevBuf commbuf;    // [own payload | gathered payloads | winner record], see comm.hip
    void* comm_host = nullptr;           // pinned staging of the host transport: [own payload | gathered payloads]
    size_t comm_host_cap = 0;
    // the factor update's row exchange (comm.hip, api.hip: exchange_new_rows): which rank holds each pool site as a candidate
    // (algp_comm_set_owners; empty: no exchange), [own rows | gathered rows] on the device, pinned staging (host transport)
    std::vector<int32_t> site_owner;
    uint64_t site_owner_hash = 0;                              // FNV-1a of the whole map: part of the agreement word of a sharded factor update
    algp::DevBuf rowx;
    void* rowx_host = nullptr;
    size_t rowx_host_cap = 0;
    int64_t rows_from_peers = 0;         // last factor update: rows of L that arrived from other ranks
    int64_t row_exchanges = 0, row_fallbacks = 0;   // exchanges carried out / agreed fall-backs to the triangular solve, so far
    int debug_fail_next_rowx = 0;        // algp_debug_fail_at(3): this rank's next agreement word carries this code
    int pending_pick_error = 0;          // a commit that failed after an exchange: this rank's status word in its next pick
    std::string pending_pick_msg;
    int64_t n_syncs = 0;     // stream synchronisations issued by the library (algp_debug_counter)
    int debug_dag_stall_ticket = -1; // algp_debug_dag_stall: the next one-launch factorisation loses this ticket's publish
    int debug_trsv_stall_block = -1; // algp_debug_trsv_stall: the next one-launch substitution loses this block's flag
    int debug_fail_next_pick = 0;   // algp_debug_fail_next_pick: error code this rank reports in its next pick
    int debug_fail_next_commit = 0; // algp_debug_fail_at(1): the next commit of a greedy pick fails with this code (after the exchange)
    int debug_fail_next_pack = 0;   // algp_debug_fail_at(2): the next pick's pack launch counts as failed

    // scratch for auxiliary factorizations (entropy_from_cov, set entropies, MI terms, posterior cov)
    algp::DevBuf auxA, auxInv, auxW, auxIdx, auxVar, auxD, hostStage;

    // profiling
    bool prof_on = false;
    algp::ProfSlot prof[ALGP_PROF_COUNT];
    std::vector<algp::PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;
    int64_t dev_bytes = 0;
};

namespace algp {

int fail(algp_ctx* c, int code, const std::string& msg);
int ensure(algp_ctx* c, DevBuf& b, size_t bytes);
void prof_begin(algp_ctx* c, int klass, double flops, double bytes);
void prof_end(algp_ctx* c);
bool prof_launch_events(algp_ctx* c, int klass, double flops, double bytes, hipEvent_t* a, hipEvent_t* b);
void prof_collect(algp_ctx* c);
// un-nested wall-time span on the main stream (e.g. a whole candidate solve whose row chunks overlap on several streams)
void prof_span_begin(algp_ctx* c, int klass, double flops, double bytes);
void prof_span_end(algp_ctx* c);
void prof_span_end_on(algp_ctx* c, hipStream_t st);
void prof_span_begin2(algp_ctx* c, int klass, double flops, double bytes);   // second span, starts on c->cur
void prof_span_end2(algp_ctx* c);

// ---- environment switches: ALL of them, read through these two functions ------------------------------------------------
// Size-range fall-backs that are product paths for other sizes, each forced by its switch so that the tests can run both on
// one input:   ALGP_TAIL_COLS=0 (appended columns re-solved as whole 128-column blocks: what inputs below 2 048 rows take;
//   also: a from-scratch solve's narrow last tile goes through the sweep instead of the tail kernel),
//   ALGP_TAIL_SPLIT=0 (the tail kernel without its k-split), ALGP_SOLVE_DAG=0 (mid-sized solves as the right-looking push
//   instead of the task list), ALGP_FOLD=0 (fit and solve as two steps: what a candidate set beyond 51 200 rows takes),
//   ALGP_ROW_STATS=0 / ALGP_TRSM_INV512=0 (the chunked solve with a variance pass / with 128-column steps inside a block),
//   ALGP_FACTOR_FROM_VT=0 (new rows of an updated factor solved, not gathered), ALGP_LAZY_GREEDY=0 (every row scored before
//   every pick), ALGP_TRSM_CHUNKS=n (row-chunk streams of the big solve; bench.py's one-stream leg sets it by the ABI).
// Cross-check routes: ALGP_CHOL_DAG=0 (launch-sequence factorisation), ALGP_GATHER_ROWS=0 (remote commits rebuild the row).
// Tooling: ALGP_LAUNCH_LOG=<file> (tools/trace_shapes.py), ALGP_RCCL_PATH=<file> (which librccl to dlopen).
inline bool env_switch(const char* name, bool dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) != 0 : dflt;
}
inline int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

#define ALGP_HIP(call)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return algp::fail(c, ALGP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define ALGP_TRY(call)          \
    do {                        \
        int r_ = (call);        \
        if (r_ != ALGP_OK) return r_; \
    } while (0)

struct ProfScope {
    algp_ctx* c;
    ProfScope(algp_ctx* c_, int klass, double flops, double bytes) : c(c_) { prof_begin(c, klass, flops, bytes); }
    ~ProfScope() { prof_end(c); }
};

// ---- typed kernels (definitions in the .hip files; explicit instantiation for float, double) ----
struct KmatSrc {
    // generator of C(pi, pj): coordinates (scaled, padded) or explicit pool covariance
    const void* Xs;      // n x DP
    const void* Cp;      // n x n or null
    int64_t n_pool;
    int DP;
    int kernel;
    double outputscale;
    double noise;        // sigma_n^2 added when pool indices coincide (coords mode only)
};

// out[r][c] for r<rows_pad, c<cols_pad (ld = ldo):
//   r<rows && c<cols : value(ridx[r], cidx[c])  [+ diag terms when the pool indices coincide]
//   unit != null && unit[r] >= 0 : (c == unit[r]) ? 1 : 0      (e_j rows of B^T)
//   padding: identity_pad ? (r==c) : 0
template <typename T>
int kmat_launch(algp_ctx* c, const KmatSrc& s, const int64_t* ridx, int64_t rows, int64_t rows_pad,
                const int64_t* cidx, int64_t cols, int64_t cols_pad, const T* diag_add,
                int add_noise_on_equal, const int* unit, int identity_pad, T* out, int64_t ldo,
                int64_t ident_shift = 0, int64_t col_shift = 0);

// plain (non-pool) kernel matrix between two scaled coordinate arrays, for algp_kernel_matrix
template <typename T>
int kmat_xy_launch(algp_ctx* c, const T* xs1, int64_t n1, const T* xs2, int64_t n2, int symmetric,
                   const T* diag_add, double add_noise, T* out, int64_t ldo);
template <typename T>
int scale_coords_launch(algp_ctx* c, const T* x, int64_t n, T* xs);

// D = alpha A B^T + beta C.  lower_only: square output, only the tiles on/below the diagonal.
template <typename T>
int gemm_nt_launch(algp_ctx* c, int klass, int64_t m, int64_t n, int64_t k, T alpha, const T* A,
                   int64_t lda, const T* B, int64_t ldb, T beta, const T* C, int64_t ldc, T* D,
                   int64_t ldd, int lower_only);
// the same product for `batch` independent problems at element strides sA/sB/sC/sD (grid.y = batch)
template <typename T>
int gemm_nt_launch_batched(algp_ctx* c, int klass, int64_t m, int64_t n, int64_t k, T alpha, const T* A, int64_t lda,
                           int64_t sA, const T* B, int64_t ldb, int64_t sB, T beta, const T* C, int64_t ldc, int64_t sC,
                           T* D, int64_t ldd, int64_t sD, int lower_only, int batch, int ktri = 0, const T* stat_w = nullptr,
                           T* stat_out = nullptr, int64_t stat_ld = 0, int kcut = 0);
// D (m x n) = A B^T, B (n x n) lower triangular by 128-tiles (column tile c sums over k < 128 (c + 1)); stat_out or null: the
// row statistics of every column tile written (stat_out[(2 tile + 0 / 1) * stat_ld + row])
template <typename T>
int gemm_nt_launch_tri(algp_ctx* c, int klass, int64_t m, int64_t n, const T* A, int64_t lda, const T* B, int64_t ldb, T* D,
                       int64_t ldd, const T* w, T* stat_out, int64_t stat_ld);
// D = alpha A B^T for ONE column tile (n = 128) and, per output row, the tile's sums of d^2 and of d * w[column] to
// stat_out[row] and stat_out[stat_ld + row] (the candidate solve's last write of a column tile of V^T: its share of the
// variance and the mean without a second pass over V^T)
template <typename T>
int gemm_nt_launch_stats(algp_ctx* c, int klass, int64_t m, int64_t k, T alpha, const T* A, int64_t lda, const T* B, int64_t ldb,
                         T* D, int64_t ldd, const T* w, T* stat_out, int64_t stat_ld);
// C (m x m, lower tiles) -= X X^T for a short, very wide X (m <= 512 rows, k columns): the k range is cut into
// chunks that run as one batched launch, the partial products are summed in chunk order (deterministic)
template <typename T>
int syrk_skinny_sub(algp_ctx* c, int klass, const T* X, int64_t m, int64_t k, int64_t ldx, T* C, int64_t ldc,
                    algp::DevBuf& scratch);

// factor the NB x NB diagonal block at A (ld = lda) in place, write its inverse (NB x NB, ld NB),
// add sum(log pivot) to *logdet_acc, record first bad pivot (block_row0 + j + 1) in *info (atomicMin style).
template <typename T>
int potrf_diag_launch(algp_ctx* c, T* A, int64_t lda, T* inv_out, double* logdet_acc, int* info,
                      int64_t block_row0);

template <typename T>
int potrf_diag_batched_launch(algp_ctx* c, T* A, int64_t sA, int64_t lda, T* inv_out, int64_t sInv, double* logdet, int* info, int batch);

int comm_unique_id(void* out128, std::string* why);
int comm_init(algp_ctx* c, int nranks, int rank, const void* unique_id128);
void comm_destroy(algp_ctx* c);
int comm_init_host(algp_ctx* c, int nranks, int rank, algp_allgather_fn fn, void* user);
int comm_pick_exchange(algp_ctx* c, const double* val_dev, const int64_t* pos_dev, const int64_t* cidx_dev,
                       const int* fresh_dev, int npicks, int status, double* rec5, const char** winner_payload);
int comm_reserve(algp_ctx* c);
int comm_agree(algp_ctx* c, const double mine[4], std::vector<double>& all);
int comm_rows_reserve(algp_ctx* c, size_t bytes_per_rank);
int comm_rows_gather(algp_ctx* c, size_t bytes_per_rank, const size_t* used_bytes = nullptr);
size_t comm_payload_bytes(const algp_ctx* c);
int comm_debug_first_max(algp_ctx* c, const double* triples, int nranks, double* out5);
void dag_release(algp_ctx* c);   // frees the cached task lists of the dependency-driven Cholesky
// Cholesky of the npad x npad matrix A (ld), inverse diagonal blocks to invD (one dependency-driven launch, or the
// blocked right-looking launch sequence for very small / very large matrices)
template <typename T>
int cholesky_blocked(algp_ctx* c, T* A, int64_t npad, int64_t ld, T* invD, double* logdet_acc,
                     int* info);
// Sizes the one-launch task list (chol_dag.hip) serves: the factor's tile count, and the tile rows of a panel carried along
constexpr int64_t DAG_MIN_TILES = 8, DAG_MAX_TILES = 192, DAG_MAX_PANEL_TILES = 400;
bool dag_enabled();       // $ALGP_CHOL_DAG != 0
// The factorisation and P <- P L^-T in one launch (P: mpad rows riding along as extra block rows of the task list;
// mode 1: dense rows, mode 2: P = I of the factor's size -> L^-T); and the same solve against a factor that is final.
template <typename T>
// (mode 2: the identity's nt tile rows may be followed by dense tile rows -- mpad > npad)
int cholesky_dag_panel(algp_ctx* c, T* A, int64_t npad, int64_t ld, T* invD, double* logdet_acc, int* info, T* P, int64_t ldp,
                       int64_t mpad, int mode, int pshort = 0);
template <typename T>
int solve_dag_panel(algp_ctx* c, const T* L, int64_t npad, int64_t ld, const T* invD, int* info, T* P, int64_t ldp, int64_t mpad,
                    int mode, int pshort = 0);
// X (mpad x npad, ld ldx) <- X * L^-T, in place
template <typename T>
// stat_out (with stat_w; or null): where every row takes the left-looking order from column 0 (more than 320 tile rows), the
// launch that writes a column tile for the last time leaves the tile's row sums of x^2 and x * stat_w[column] at
// stat_out[(2 tile + 0 / 1) * stat_ld + row]; *stats_done says whether that happened (else: take them in a pass over X)
int trsm_blocked(algp_ctx* c, int klass, T* X, int64_t mpad, int64_t ldx, const T* L, int64_t npad,
                 int64_t ldl, const T* invD, int64_t col_start = 0, const T* stat_w = nullptr, T* stat_out = nullptr,
                 int64_t stat_ld = 0, bool* stats_done = nullptr);
// columns [c0, c0 + w) of X <- the solution's new columns after rows c0.. were appended to L (tail.hip): c0 a multiple of 16,
// w <= 64, inside one 128-column block whose explicit inverse is invD_blk; lrows = rows of L that exist
template <typename T>
int tail_cols_launch(algp_ctx* c, int klass, T* X, int64_t mpad, int64_t ldx, const T* L, int64_t ldl, int64_t lrows, const T* invD_blk,
                     int64_t c0, int w, const T* E_window = nullptr);
// X (npad x npad, holding the identity) <- L^-T (upper triangular; zero parts are never touched)
template <typename T>
int trinv_upper(algp_ctx* c, int klass, T* X, int64_t npad, int64_t ldx, const T* L, int64_t ldl, const T* invD);
template <typename T>
int trinv_upper_inplace(algp_ctx* c, int klass, T* XL, int64_t npad, int64_t ld, const T* invD);
// C (lower tiles, npad x npad) <- X X^T for that upper-triangular X
template <typename T>
int syrk_upper(algp_ctx* c, int klass, const T* X, int64_t npad, int64_t ldx, T* C, int64_t ldc);

// b <- L^-1 b (forward) and b <- L^-T b (backward) for one vector of length npad
template <typename T>
int trsv_forward(algp_ctx* c, const T* L, int64_t npad, int64_t ldl, const T* invD, T* b, int64_t kb_start = 0);
template <typename T>
int trsv_forward2(algp_ctx* c, const T* L, int64_t npad, int64_t ldl, const T* invD, T* b0, T* b1, int64_t kb_start = 0);
// u[k:] -= L[k:, 0:k] u[0:k] (and the same for w): resume two forward substitutions at row k
template <typename T>
int tail_gemv2_launch(algp_ctx* c, const T* L, int64_t ldl, int64_t k, int64_t npad, T* u, T* w);
template <typename T>
int trsv_backward(algp_ctx* c, const T* L, int64_t npad, int64_t ldl, const T* invD, T* b);

// row reductions over V^T: ss[j] = sum_r V[j][r]^2 (if ss), dot[j] = sum_r V[j][r]*w[r] (if w)
template <typename T>
int rows_reduce_launch(algp_ctx* c, const T* Vt, int64_t rows, int64_t ldv, int64_t ncols, const T* w, T* ss, T* dot,
                       int64_t tri_c0 = -1);

template <typename T>
int test_mfma_launch(algp_ctx* c, int* mismatches_dev);
template <typename T>
int bench_gemm(algp_ctx* c, int64_t m, int64_t n, int64_t k, int lower_only, int beta_one, int reps,
               double* ms_out);

}  // namespace algp
