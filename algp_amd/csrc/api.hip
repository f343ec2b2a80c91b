// api.hip -- extern "C" entry points of libalgp_hip.so (declared in include/algp_hip.h) and the
// host-side orchestration of the device kernels.  No torch types, no CPU fallback: every matrix, vector and
// per-candidate quantity comes from the HIP kernels in this directory; what the host computes is bookkeeping
// (index maps, the mean of the targets) and O(1) scalar combinations of values the kernels returned.
#include <limits.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <stdlib.h>

#include "common.h"
#include "vecops.h"

using namespace algp;

namespace algp {

int fail(algp_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

int ensure(algp_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return ALGP_OK;
    if (bytes == 0) bytes = 256;
    if (b.p) {
        hipStreamSynchronize(c->stream);
        hipFree(b.p);
        c->dev_bytes -= (int64_t)b.cap;
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t want = (bytes + 255) / 256 * 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        (void)hipGetLastError();
        return fail(c, ALGP_ERR_OOM, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
    }
    b.cap = want;
    c->dev_bytes += (int64_t)want;
    return ALGP_OK;
}

static void release(algp_ctx* c, DevBuf& b) {
    if (b.p) {
        hipFree(b.p);
        c->dev_bytes -= (int64_t)b.cap;
    }
    b.p = nullptr;
    b.cap = 0;
}

static hipEvent_t get_event(algp_ctx* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

// Profiling: one event pair per scope on the ctx stream; flops/bytes are the ALGORITHMIC figures
// of the launch (DESIGN.md "measurement").  Nested scopes: only the outermost records.
static thread_local int prof_depth = 0;
void prof_begin(algp_ctx* c, int klass, double flops, double bytes) {
    if (!c->prof_on) return;
    if (prof_depth++ > 0) return;
    PendingEvent pe;
    pe.a = get_event(c);
    pe.b = get_event(c);
    pe.klass = klass;
    c->prof[klass].flops += flops;
    c->prof[klass].bytes += bytes;
    c->prof[klass].launches += 1;
    hipEventRecord(pe.a, c->cur);
    c->pending.push_back(pe);
}
void prof_end(algp_ctx* c) {
    if (!c->prof_on) return;
    if (--prof_depth > 0) return;
    hipEventRecord(c->pending.back().b, c->cur);
}
// A scope that is exactly one kernel launch: the two events ride on the dispatch itself (hipExtLaunchKernelGGL: its own begin
// and end timestamps, the figures rocprofv3 reports) instead of two hipEventRecord barrier packets around it on the stream --
// with ~470 GEMM launches per candidate solve those packets cost the step 2 % (bench.py: ms_per_step vs ms_per_step_unprofiled).
// False: profiling is off or an outer scope is open -- launch plainly.
bool prof_launch_events(algp_ctx* c, int klass, double flops, double bytes, hipEvent_t* a, hipEvent_t* b) {
    if (!c->prof_on || prof_depth > 0) return false;
    PendingEvent pe;
    pe.a = get_event(c);
    pe.b = get_event(c);
    pe.klass = klass;
    c->prof[klass].flops += flops;
    c->prof[klass].bytes += bytes;
    c->prof[klass].launches += 1;
    c->pending.push_back(pe);
    *a = pe.a;
    *b = pe.b;
    return true;
}
static thread_local int span_index = -1;
void prof_span_begin(algp_ctx* c, int klass, double flops, double bytes) {
    if (!c->prof_on) return;
    PendingEvent pe;
    pe.a = get_event(c);
    pe.b = get_event(c);
    pe.klass = klass;
    c->prof[klass].flops += flops;
    c->prof[klass].bytes += bytes;
    c->prof[klass].launches += 1;
    hipEventRecord(pe.a, c->stream);
    span_index = (int)c->pending.size();
    c->pending.push_back(pe);
}
void prof_span_end_on(algp_ctx* c, hipStream_t st) {
    if (!c->prof_on || span_index < 0) return;
    hipEventRecord(c->pending[span_index].b, st);
    span_index = -1;
}
void prof_span_end(algp_ctx* c) { prof_span_end_on(c, c->stream); }
static thread_local int span2_index = -1;
void prof_span_begin2(algp_ctx* c, int klass, double flops, double bytes) {
    if (!c->prof_on) return;
    PendingEvent pe;
    pe.a = get_event(c);
    pe.b = get_event(c);
    pe.klass = klass;
    c->prof[klass].flops += flops;
    c->prof[klass].bytes += bytes;
    c->prof[klass].launches += 1;
    hipEventRecord(pe.a, c->cur);
    span2_index = (int)c->pending.size();
    c->pending.push_back(pe);
}
void prof_span_end2(algp_ctx* c) {
    if (!c->prof_on || span2_index < 0) return;
    hipEventRecord(c->pending[span2_index].b, c->stream);
    span2_index = -1;
}
void prof_collect(algp_ctx* c) {
    for (auto& pe : c->pending) {
        hipEventSynchronize(pe.b);
        float ms = 0;
        hipEventElapsedTime(&ms, pe.a, pe.b);
        c->prof[pe.klass].ms += ms;
        c->event_pool.push_back(pe.a);
        c->event_pool.push_back(pe.b);
    }
    c->pending.clear();
}

}  // namespace algp

// ---------------------------------------------------------------------------------------------
// typed implementation
// ---------------------------------------------------------------------------------------------
namespace {

hipEvent_t sync_event_api(algp_ctx* c, size_t i) {
    while (c->sync_events.size() <= i) {
        hipEvent_t e;
        hipEventCreateWithFlags(&e, hipEventDisableTiming);
        c->sync_events.push_back(e);
    }
    return c->sync_events[i];
}


KmatSrc make_src(algp_ctx* c) {
    KmatSrc s;
    s.Xs = c->Xs.p;
    s.Cp = c->pool_is_cov ? c->Cp.p : nullptr;
    s.n_pool = c->n_pool;
    s.DP = c->hyp.DP;
    s.kernel = c->hyp.kernel;
    s.outputscale = c->hyp.outputscale;
    s.noise = c->hyp.noise;
    return s;
}

int sync(algp_ctx* c) {
    ALGP_HIP(hipStreamSynchronize(c->stream));
    c->n_syncs++;
    return ALGP_OK;
}

// A synchronisation that also reads the sticky stall word (scal[SC_STALL]): one of the one-launch kernels that hand data
// between workgroups (forward / backward substitution, the solve-only task list) ran into its spin limit and abandoned
// its work since the word was last read -- whatever it was producing is incomplete.  The word is cleared again.
int sync_checked(algp_ctx* c, const char* what) {
    int stall = 0;
    int* d = (int*)((double*)c->scal.p + SC_STALL);
    ALGP_HIP(hipMemcpyAsync(&stall, d, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    ALGP_HIP(hipStreamSynchronize(c->stream));
    c->n_syncs++;
    if (stall != 0) {
        hipMemsetAsync(d, 0, sizeof(double), c->stream);
        return fail(c, ALGP_ERR_HIP, std::string(what) + ": a one-launch kernel stalled (a hand-off between its workgroups never "
                                     "arrived within the spin limit); the result is incomplete");
    }
    return ALGP_OK;
}

template <typename T>
struct Impl {
    static T* p(DevBuf& b) { return (T*)b.p; }

    static int rescale_pool(algp_ctx* c) {
        if (c->pool_is_cov || c->n_pool == 0 || !c->hyp.set) return ALGP_OK;
        ALGP_TRY(ensure(c, c->Xs, sizeof(T) * c->n_pool * c->hyp.DP));
        return scale_coords_launch<T>(c, (const T*)c->Xraw.p, c->n_pool, p(c->Xs));
    }

    static int kernel_matrix(algp_ctx* c, const void* x1, int64_t n1, const void* x2, int64_t n2, const void* diag_add,
                             int add_lik, void* out) {
        const int D = c->hyp.D, DP = c->hyp.DP;
        const bool sym = (x2 == nullptr);
        if (sym) n2 = n1;
        if (n1 == 0 || n2 == 0) return ALGP_OK;
        const int64_t ldo = round_up(n2, 4);
        DevBuf raw1, raw2, s1, s2, dv, o;
        int rc = ALGP_OK;
        auto cleanup = [&]() { release(c, raw1); release(c, raw2); release(c, s1); release(c, s2); release(c, dv); release(c, o); };
#define KM_TRY(x) do { rc = (x); if (rc != ALGP_OK) { cleanup(); return rc; } } while (0)
#define KM_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(c, ALGP_ERR_HIP, hipGetErrorString(e_)); } } while (0)
        KM_TRY(ensure(c, raw1, sizeof(T) * n1 * D));
        KM_TRY(ensure(c, s1, sizeof(T) * n1 * DP));
        KM_HIP(hipMemcpyAsync(raw1.p, x1, sizeof(T) * n1 * D, hipMemcpyHostToDevice, c->stream));
        KM_TRY(scale_coords_launch<T>(c, (const T*)raw1.p, n1, (T*)s1.p));
        if (!sym) {
            KM_TRY(ensure(c, raw2, sizeof(T) * n2 * D));
            KM_TRY(ensure(c, s2, sizeof(T) * n2 * DP));
            KM_HIP(hipMemcpyAsync(raw2.p, x2, sizeof(T) * n2 * D, hipMemcpyHostToDevice, c->stream));
            KM_TRY(scale_coords_launch<T>(c, (const T*)raw2.p, n2, (T*)s2.p));
        }
        if (sym && diag_add) {
            KM_TRY(ensure(c, dv, sizeof(T) * n1));
            KM_HIP(hipMemcpyAsync(dv.p, diag_add, sizeof(T) * n1, hipMemcpyHostToDevice, c->stream));
        }
        KM_TRY(ensure(c, o, sizeof(T) * n1 * ldo));
        KM_TRY(kmat_xy_launch<T>(c, (const T*)s1.p, n1, sym ? nullptr : (const T*)s2.p, n2, sym ? 1 : 0,
                                 (sym && diag_add) ? (const T*)dv.p : nullptr, (sym && add_lik) ? c->hyp.noise : 0.0,
                                 (T*)o.p, ldo));
        KM_HIP(hipMemcpy2DAsync(out, sizeof(T) * n2, o.p, sizeof(T) * ldo, sizeof(T) * n2, n1, hipMemcpyDeviceToHost,
                                c->stream));
        KM_HIP(hipStreamSynchronize(c->stream));
        cleanup();
#undef KM_TRY
#undef KM_HIP
        return ALGP_OK;
    }

    static int set_pool(algp_ctx* c, const void* x, int64_t n) {
        const int D = c->hyp.D;
        ALGP_TRY(ensure(c, c->Xraw, sizeof(T) * n * D));
        ALGP_HIP(hipMemcpyAsync(c->Xraw.p, x, sizeof(T) * n * D, hipMemcpyHostToDevice, c->stream));
        c->n_pool = n;
        c->pool_is_cov = false;
        // a fingerprint per site (FNV-1a over its coordinates' bytes): lets algp_factorize_from check that a factor
        // adopted from another context was computed for the same COORDINATES, not only the same indices
        c->site_hash.assign((size_t)n, 0);
        const unsigned char* bytes = (const unsigned char*)x;
        const size_t stride = sizeof(T) * (size_t)D;
        for (int64_t i = 0; i < n; ++i) {
            uint64_t h = 1469598103934665603ull;
            for (size_t b = 0; b < stride; ++b) h = (h ^ bytes[(size_t)i * stride + b]) * 1099511628211ull;
            c->site_hash[(size_t)i] = h;
        }
        ALGP_TRY(rescale_pool(c));
        return sync(c);
    }
    // fingerprint of the coordinates of a train set, in row order (0 for an explicit-covariance pool)
    static uint64_t train_sites_hash(const algp_ctx* c, const std::vector<int64_t>& idx) {
        if (c->pool_is_cov || c->site_hash.empty()) return 0;
        uint64_t h = 1469598103934665603ull;
        for (int64_t i : idx) h = (h ^ c->site_hash[(size_t)i]) * 1099511628211ull;
        return h;
    }

    static int set_pool_cov(algp_ctx* c, const void* cov, int64_t n) {
        ALGP_TRY(ensure(c, c->Cp, sizeof(T) * n * n));
        ALGP_HIP(hipMemcpyAsync(c->Cp.p, cov, sizeof(T) * n * n, hipMemcpyHostToDevice, c->stream));
        c->n_pool = n;
        c->pool_is_cov = true;
        c->site_hash.clear();
        return sync(c);
    }

    static int set_train(algp_ctx* c, const int64_t* idx, int64_t N, const void* y, const void* var) {
        const int64_t Npad = round_up(std::max<int64_t>(N, 1), NB);
        c->N = N;
        c->Npad = Npad;
        c->train_idx.assign(idx, idx + N);
        c->pos_in_train.assign(c->n_pool, -1);
        c->train_has_repeats = false;
        for (int64_t i = 0; i < N; ++i) {                      // a site measured more than once: its FIRST row stands for it
            if (c->pos_in_train[idx[i]] < 0) c->pos_in_train[idx[i]] = i;
            else c->train_has_repeats = true;
        }
        double ybar = 0;
        const T* yt = (const T*)y;
        for (int64_t i = 0; i < N; ++i) ybar += (double)yt[i];
        ybar = N > 0 ? ybar / (double)N : 0.0;
        if (c->mean_override) ybar = c->mean_value;             // algp_set_constant_mean
        c->ybar = ybar;
        std::vector<T> y0(Npad, (T)0), vv(Npad, (T)0);
        for (int64_t i = 0; i < N; ++i) y0[i] = (T)((double)yt[i] - ybar);
        if (var)
            for (int64_t i = 0; i < N; ++i) vv[i] = ((const T*)var)[i];
        ALGP_TRY(ensure(c, c->Aidx, sizeof(int64_t) * Npad));
        ALGP_TRY(ensure(c, c->y0, sizeof(T) * Npad));
        ALGP_TRY(ensure(c, c->varA, sizeof(T) * Npad));
        if (N > 0) ALGP_HIP(hipMemcpyAsync(c->Aidx.p, idx, sizeof(int64_t) * N, hipMemcpyHostToDevice, c->stream));
        ALGP_HIP(hipMemcpyAsync(c->y0.p, y0.data(), sizeof(T) * Npad, hipMemcpyHostToDevice, c->stream));
        std::vector<T> yr(Npad, (T)0);
        for (int64_t i = 0; i < N; ++i) yr[i] = yt[i];
        ALGP_TRY(ensure(c, c->yraw, sizeof(T) * Npad));
        ALGP_HIP(hipMemcpyAsync(c->yraw.p, yr.data(), sizeof(T) * Npad, hipMemcpyHostToDevice, c->stream));
        c->train_y_host.assign(N, 0.0);
        for (int64_t i = 0; i < N; ++i) c->train_y_host[i] = (double)yt[i];
        ALGP_HIP(hipMemcpyAsync(c->varA.p, vv.data(), sizeof(T) * Npad, hipMemcpyHostToDevice, c->stream));
        ALGP_TRY(sync(c));   // host vectors go out of scope
        c->train_var_host.assign(N, 0.0);
        if (var)
            for (int64_t i = 0; i < N; ++i) c->train_var_host[i] = (double)((const T*)var)[i];
        c->train_dirty = true;       // the resident factor (if any) still describes fact_idx / fact_var
        c->solved = false;
        if (c->comm || c->host_gather) ALGP_TRY(comm_reserve(c));   // the exchange's buffers follow the train set's size
        return ALGP_OK;
    }

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
    };
    static bool panel_fits(int64_t npad, int64_t mpad) {
        const int64_t nt = npad / NB, mt = mpad / NB;
        return dag_enabled() && nt >= DAG_MIN_TILES && nt <= DAG_MAX_TILES && mt >= 1 && mt <= DAG_MAX_PANEL_TILES;
    }

    // factor an npad x npad matrix already resident in A; returns logdet; NOT_PD -> error with pivot
    static int factor_resident(algp_ctx* c, T* A, int64_t n, int64_t npad, T* invD, int slot_logdet, int slot_info,
                               double* logdet, int64_t ld = 0, int64_t pivot_offset = 0, Panel* panel = nullptr) {
        if (ld == 0) ld = npad;
        double* sc = (double*)c->scal.p;
        ALGP_HIP(hipMemsetAsync(sc + slot_logdet, 0, 2 * sizeof(double), c->stream));
        if (panel && panel_fits(npad, panel->mpad)) {
            ALGP_TRY(cholesky_dag_panel<T>(c, A, npad, ld, invD, sc + slot_logdet, (int*)(sc + slot_info), panel->P, panel->ldp,
                                           panel->mpad, panel->mode));
            panel->done = true;
            if (panel->mode == 2 && panel->inv_out && c->stream2 && c->cur == c->stream) {
                hipEvent_t ready = sync_event_api(c, 20), done = sync_event_api(c, 21);
                ALGP_HIP(hipEventRecord(ready, c->stream));
                ALGP_HIP(hipStreamWaitEvent(c->stream2, ready, 0));
                c->cur = c->stream2;
                const int rc = syrk_upper<T>(c, ALGP_PROF_GEMM_OTHER, panel->P, npad, panel->ldp, panel->inv_out, npad);
                c->cur = c->stream;
                ALGP_HIP(hipEventRecord(done, c->stream2));
                ALGP_TRY(rc);
                panel->inv_enqueued = true;
            }
        } else {
            ALGP_TRY(cholesky_blocked<T>(c, A, npad, ld, invD, sc + slot_logdet, (int*)(sc + slot_info)));
        }
        double host[2];
        ALGP_HIP(hipMemcpyAsync(host, sc + slot_logdet, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        ALGP_TRY(sync(c));
        int info;
        memcpy(&info, &host[1], sizeof(int));
        if (info == INT_MIN)
            return fail(c, ALGP_ERR_HIP, "cholesky: the dependency-driven launch stalled (a task's inputs never arrived)");
        if (info != 0) {
            info += (int)pivot_offset;
            c->pivot = info;
            return fail(c, ALGP_ERR_NOT_PD,
                        "matrix is not positive definite: non-positive pivot at index " + std::to_string(info) +
                            " (1-based) of " + std::to_string(n));
        }
        *logdet = host[0];
        return ALGP_OK;
    }

    // make room for an Npad x Npad factor with leading dimension Lld >= Npad, keeping the first
    // `keep_rows` rows (and their inverse diagonal blocks) when the buffers have to grow -- and, of the rows
    // [keep_rows, keep_height), the part left of column keep_rows (rows of the partial last block whose
    // solved entries against the kept blocks stay valid)
    static int reserve_factor(algp_ctx* c, int64_t npad_need, int64_t keep_rows, int64_t keep_height = 0,
                              bool headroom = false) {
        if (c->Lld >= npad_need && c->L.p && c->invD.p) return ALGP_OK;
        // a caller that updates the factor incrementally gets 12.5 % headroom from the start: growing the buffer
        // later means a new allocation and a copy of the kept rows (0.5 s for the 20 GB factor of N = 50 000)
        const int64_t first = headroom ? npad_need + npad_need / 8 : npad_need;
        const int64_t newld = round_up(std::max<int64_t>(first, c->Lld + c->Lld / 4), NB);
        DevBuf nl, ni;
        int rc = ensure(c, nl, sizeof(T) * newld * newld);
        if (rc == ALGP_OK) rc = ensure(c, ni, sizeof(T) * newld * NB);
        if (rc != ALGP_OK) { release(c, nl); release(c, ni); return rc; }
        if (keep_rows > 0 && c->L.p) {
            hipError_t e = hipMemcpy2DAsync(nl.p, sizeof(T) * newld, c->L.p, sizeof(T) * c->Lld, sizeof(T) * keep_rows,
                                            std::max(keep_rows, keep_height), hipMemcpyDeviceToDevice, c->stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(ni.p, c->invD.p, sizeof(T) * keep_rows * NB, hipMemcpyDeviceToDevice, c->stream);
            if (e != hipSuccess) { release(c, nl); release(c, ni); return fail(c, ALGP_ERR_HIP, hipGetErrorString(e)); }
            hipStreamSynchronize(c->stream);
        }
        release(c, c->L);
        release(c, c->invD);
        c->L = nl;
        c->invD = ni;
        c->Lld = newld;
        return ALGP_OK;
    }

    // S = C_AA + D -> L.  With `incremental`, the leading 128-row blocks of the resident factor are
    // kept as long as the train set (indices, noise, in order) and the hyper-parameters agree with
    // what they were computed for; only the rows from the first changed block on are rebuilt:
    //   rows R of S regenerated, X = S[R, 0:Nb] L[0:Nb,0:Nb]^-T, S_RR -= X X^T, chol(S_RR).
    // Appending k sites to N therefore costs O((128 + k) N^2) instead of O(N^3 / 3).
    // Factor update: can the rows of the new train sites [p0, N) (left of the tail block, columns [0, Nb)) be
    // taken from the resident V^T?  Needs V^T solved for the same kept blocks and hyper-parameters, the
    // same candidate list, and every new site an ordinary candidate row.  src_row: V^T row per factor row
    // p0 .. Npad-1 (-1 = padding row, zero).
    static bool vt_rows_for_new_sites(algp_ctx* c, int64_t Nb, int64_t p0, std::vector<int64_t>& src_row,
                                      std::vector<int64_t>& lrow, std::vector<T>& lscale, bool& any_second) {
        static const bool on = env_switch("ALGP_FACTOR_FROM_VT", true);
        const int64_t N = c->N, Npad = c->Npad;
        if (!on || !c->Vt.p || c->vt_hyp_stamp != c->hyp_stamp || (int64_t)c->vt_fact_idx.size() < Nb || Nb <= 0) return false;
        if (c->vt_cand_idx != c->cand_idx || (int64_t)c->vt_kind.size() != c->M) return false;
        for (int64_t r = 0; r < Nb; ++r)
            if (c->vt_fact_idx[r] != c->train_idx[r] || c->vt_fact_var[r] != c->train_var_host[r]) return false;
        src_row.assign((size_t)(Npad - p0), -1);
        lrow.assign((size_t)(Npad - p0), -1);
        lscale.assign((size_t)(Npad - p0), (T)0);
        any_second = false;
        for (int64_t i = p0; i < N; ++i) {
            const int64_t q = c->train_idx[i], j = c->cand_pos[q];
            if (j < 0) return false;
            const int k = c->vt_kind[j];
            if (k >= 0) {
                // a further measurement of a site that already is train row k: its covariances with the old rows
                // are S[k, :] - var_k e_k^T, so its row is L[k, :] - var_k (e_k^T L^-T), and e_k^T L^-T is the unit
                // row V^T holds for that candidate
                if (k >= p0 || c->train_idx[k] != q || (int64_t)c->vt_fact_idx.size() <= k || c->vt_fact_idx[k] != q ||
                    c->vt_fact_var[k] != c->train_var_host[k])
                    return false;
                lrow[(size_t)(i - p0)] = k;
                lscale[(size_t)(i - p0)] = (T)c->train_var_host[k];
                any_second = true;
            }
            src_row[(size_t)(i - p0)] = j;
        }
        return true;
    }

    // The same rows when the candidates are sharded over ranks (a communicator and an owner map are attached): every new
    // train site is a candidate of exactly one rank, whose row of V^T (for a further reading of a site that is train row
    // k already: L[k, :] - var_k * its unit row) is what EVERY rank's replica of the factor needs.  All ranks hold the
    // same train set and owner map, so all compute the same plan -- owner and slot of every new row, cap = the largest
    // count any rank contributes -- and take part in: a 32-byte agreement (comm_agree), then one all-gather of cap rows
    // of Nb elements per rank.  *placed = 1: rows [p0, Npad) of L, columns [0, Nb), are in place; 0: the ranks agreed to
    // solve them against the kept factor instead (some rank's V^T cannot supply its rows).  An error code >= 2 of any
    // rank (an allocation that failed, ...) is returned by every rank.  Reference: agent.py:66-82 (the sites a step adds),
    // agent.py:313-354 (the loop whose shards own them).
    // st_in: what this rank found BEFORE the plan (0; 1 = it keeps nothing of its factor, Nb = 0; >= 2 = an allocation of
    // the factor itself failed): it travels in the agreement word like every later failure, so that no rank returns
    // from factorize_update before the agreement its peers are waiting in.
    static int exchange_new_rows(algp_ctx* c, int64_t Nb, int64_t p0, int* placed, int st_in = 0) {
        const int64_t N = c->N, Npad = c->Npad, ld = c->Lld;
        const int nr = c->comm_nranks, me = c->comm_rank;
        const int64_t nnew = N - p0, ntot = Npad - p0;
        *placed = 0;
        std::vector<int> owner((size_t)std::max<int64_t>(nnew, 0)), slot((size_t)std::max<int64_t>(nnew, 0));
        std::vector<int64_t> cnt((size_t)nr, 0);
        int st = st_in;
        uint64_t h = 1469598103934665603ull;
        auto mix = [&h](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
        mix((uint64_t)Nb);
        mix(c->site_owner_hash);                                             // the WHOLE owner map, not only the new sites' entries
        for (int64_t i = 0; i < nnew; ++i) {
            const int64_t q = c->train_idx[(size_t)(p0 + i)];
            const int o = c->site_owner[(size_t)q];
            mix((uint64_t)q);
            mix((uint64_t)(int64_t)o);
            if (o < 0 || o >= nr) { st = 1; owner[(size_t)i] = -1; continue; }     // nobody holds this site as a candidate
            owner[(size_t)i] = o;
            slot[(size_t)i] = (int)cnt[(size_t)o]++;
        }
        int64_t cap = 0;
        for (int r = 0; r < nr; ++r) cap = std::max(cap, cnt[(size_t)r]);
        // this rank's own rows: the checks of vt_rows_for_new_sites, for the sites it owns
        std::vector<int64_t> src_row((size_t)std::max<int64_t>(cap, 1), -1), lrow((size_t)std::max<int64_t>(cap, 1), -1);
        std::vector<T> lscale((size_t)std::max<int64_t>(cap, 1), (T)0);
        bool second = false;
        if (st == 0 && Nb > 0 && cnt[(size_t)me] > 0) {
            bool ok = c->Vt.p && c->vt_hyp_stamp == c->hyp_stamp && (int64_t)c->vt_fact_idx.size() >= Nb &&
                      c->vt_cand_idx == c->cand_idx && (int64_t)c->vt_kind.size() == c->M;
            for (int64_t r = 0; ok && r < Nb; ++r)
                ok = c->vt_fact_idx[(size_t)r] == c->train_idx[(size_t)r] && c->vt_fact_var[(size_t)r] == c->train_var_host[(size_t)r];
            for (int64_t i = 0; ok && i < nnew; ++i) {
                if (owner[(size_t)i] != me) continue;
                const int64_t q = c->train_idx[(size_t)(p0 + i)], j = c->cand_pos[(size_t)q];
                if (j < 0) { ok = false; break; }                            // the map says this rank, its candidate list does not
                const int k = c->vt_kind[(size_t)j];
                if (k >= 0) {
                    if (k >= p0 || c->train_idx[(size_t)k] != q || (int64_t)c->vt_fact_idx.size() <= k ||
                        c->vt_fact_idx[(size_t)k] != q || c->vt_fact_var[(size_t)k] != c->train_var_host[(size_t)k]) { ok = false; break; }
                    lrow[(size_t)slot[(size_t)i]] = k;
                    lscale[(size_t)slot[(size_t)i]] = (T)c->train_var_host[(size_t)k];
                    second = true;
                }
                src_row[(size_t)slot[(size_t)i]] = j;
            }
            if (!ok) st = 1;
        }
        const size_t rowbytes = sizeof(T) * (size_t)Nb;
        // sized by the factor's capacity, not by this step's Nb and cap: the buffers then stay put while the train set grows
        if (st <= 1 && cap > 0) {
            const int rc = comm_rows_reserve(c, sizeof(T) * (size_t)c->Lld * (size_t)std::max<int64_t>(16, round_up(cap, 8)));
            if (rc != ALGP_OK) st = rc;
        }
        if (st <= 1) {
            int rc = ensure(c, c->auxIdx, sizeof(int64_t) * 2 * (size_t)std::max<int64_t>(std::max(ntot, cap), 1));
            if (rc == ALGP_OK) rc = ensure(c, c->auxVar, sizeof(T) * (size_t)std::max<int64_t>(cap, 1) + 256);
            if (rc != ALGP_OK) st = rc;
        }
        if (c->debug_fail_next_rowx) {
            st = c->debug_fail_next_rowx;
            c->debug_fail_next_rowx = 0;
            c->err = "factorize_update: failure injected by algp_debug_fail_at";
        }
        const std::string local_err = c->err;
        double mine[4] = {(double)st, (double)p0, (double)N, 0.0};
        memcpy(&mine[3], &h, sizeof(h));
        std::vector<double> all;
        ALGP_TRY(comm_agree(c, mine, all));
        int worst = 0, bad_rank = -1;
        bool same = true;
        for (int r = 0; r < nr; ++r) {
            const double* t = &all[(size_t)r * 4];
            const int s_r = (t[0] == t[0] && t[0] >= 0 && t[0] <= 64) ? (int)t[0] : ALGP_ERR_HIP;
            if (s_r > worst) { worst = s_r; bad_rank = r; }
            if (t[1] != mine[1] || t[2] != mine[2] || memcmp(&t[3], &mine[3], 8) != 0) same = false;
        }
        if (worst >= 2) {
            if (st >= 2) return fail(c, st, local_err);
            return fail(c, worst, "factorize_update: rank " + std::to_string(bad_rank) + " failed with error " + std::to_string(worst) +
                                      " before the row exchange; no rank updated its factor");
        }
        if (worst == 1 || !same) {
            // every rank builds the rows itself: the solve against its kept blocks, or (a rank that keeps nothing) from scratch.
            // Not a fall-back when NO rank keeps anything: then there was nothing to exchange (the first factorisation of a run).
            bool any_kept = false;
            for (int r = 0; r < nr; ++r) any_kept = any_kept || all[(size_t)r * 4 + 1] >= (double)NB;
            if (any_kept) c->row_fallbacks += 1;
            return ALGP_OK;
        }
        if (cap > 0) {
            T* own = (T*)c->rowx.p;
            const size_t bytes = rowbytes * (size_t)cap;
            T* gathered = (T*)((char*)c->rowx.p + bytes);
            int64_t* d_src = (int64_t*)c->auxIdx.p;
            int64_t* d_lrow = d_src + std::max<int64_t>(std::max(ntot, cap), 1);
            ALGP_HIP(hipMemcpyAsync(d_src, src_row.data(), sizeof(int64_t) * (size_t)cap, hipMemcpyHostToDevice, c->stream));
            if (second) {
                ALGP_HIP(hipMemcpyAsync(d_lrow, lrow.data(), sizeof(int64_t) * (size_t)cap, hipMemcpyHostToDevice, c->stream));
                ALGP_HIP(hipMemcpyAsync(c->auxVar.p, lscale.data(), sizeof(T) * (size_t)cap, hipMemcpyHostToDevice, c->stream));
            }
            // slots this rank does not fill (it owns fewer than cap rows) are written as zeros: src_row = -1
            ALGP_TRY(gather_rows_launch<T>(c, p(c->Vt), c->ldv, d_src, own, Nb, cap, Nb, second ? d_lrow : nullptr,
                                           second ? (const T*)c->auxVar.p : nullptr, p(c->L), ld));
            ALGP_TRY(sync(c));                                               // src_row / lrow / lscale are host temporaries
            std::vector<size_t> used((size_t)nr);
            for (int r = 0; r < nr; ++r) used[(size_t)r] = rowbytes * (size_t)cnt[(size_t)r];
            ALGP_TRY(comm_rows_gather(c, bytes, used.data()));
            // scatter: factor row p0 + i <- the slot of its owner's contribution; padding rows are zero
            std::vector<int64_t> from((size_t)ntot, -1);
            int64_t peers = 0;
            for (int64_t i = 0; i < nnew; ++i) {
                from[(size_t)i] = (int64_t)owner[(size_t)i] * cap + slot[(size_t)i];
                peers += owner[(size_t)i] != me;
            }
            ALGP_HIP(hipMemcpyAsync(d_src, from.data(), sizeof(int64_t) * (size_t)ntot, hipMemcpyHostToDevice, c->stream));
            ALGP_TRY(gather_rows_launch<T>(c, gathered, Nb, d_src, p(c->L) + p0 * ld, ld, ntot, Nb));
            ALGP_TRY(sync(c));
            c->rows_from_peers = peers;
            c->row_exchanges += 1;
        } else if (ntot > 0) {
            ALGP_HIP(hipMemset2DAsync(p(c->L) + p0 * ld, sizeof(T) * (size_t)ld, 0, rowbytes, (size_t)ntot, c->stream));
            c->rows_from_peers = 0;
        }
        *placed = 1;
        return ALGP_OK;
    }

    static int factorize(algp_ctx* c, int incremental, Panel* panel = nullptr) {
        const int64_t N = c->N, Npad = c->Npad;
        int64_t keep = 0, p0 = 0;                                // rows of the resident factor to keep; unchanged leading rows
        if (incremental && c->factored && c->fact_hyp_stamp == c->hyp_stamp && c->Lld > 0) {
            const int64_t lim = std::min<int64_t>(N, c->Nfact);
            while (p0 < lim && c->fact_idx[p0] == c->train_idx[p0] && c->fact_var[p0] == c->train_var_host[p0]) ++p0;
            keep = p0 / NB * NB;
        }
        c->factored = false;
        c->solved = false;
        // candidates sharded over ranks (a transport and an owner map of this pool attached): the incremental call is a
        // COLLECTIVE whatever this rank finds locally -- a rank that keeps nothing (an earlier factorisation failed, other
        // hyper-parameters) or whose factor cannot be re-allocated says so in the agreement its peers enter
        const bool sharded = incremental && (c->comm || c->host_gather) && c->comm_nranks > 1 && !c->site_owner.empty() &&
                             (int64_t)c->site_owner.size() == c->n_pool;
        int pre = reserve_factor(c, Npad, keep, p0, incremental != 0);
        if (pre == ALGP_OK) pre = ensure(c, c->z, sizeof(T) * Npad);
        if (pre == ALGP_OK) pre = ensure(c, c->alpha, sizeof(T) * Npad);
        if (sharded && (pre != ALGP_OK || keep == 0)) {
            int placed_unused = 0;
            const int arc = exchange_new_rows(c, 0, p0, &placed_unused, pre != ALGP_OK ? pre : 1);
            if (arc != ALGP_OK) return arc;                                  // this rank's failure, or a peer's: the same code everywhere
        }
        ALGP_TRY(pre);
        const int64_t ld = c->Lld;
        KmatSrc s = make_src(c);
        double ld_total = 0;
        prof_span_begin(c, ALGP_PROF_CHOLESKY, keep == 0 ? (double)N * N * N / 3.0 : (double)(N - keep) * N * N,
                        sizeof(T) * (double)N * N);
        int frc = ALGP_OK;
        if (keep == 0) {
            frc = kmat_launch<T>(c, s, (const int64_t*)c->Aidx.p, N, Npad, (const int64_t*)c->Aidx.p, N, Npad,
                                 (const T*)c->varA.p, c->pool_is_cov ? 0 : 1, nullptr, 1, p(c->L), ld);
            if (frc == ALGP_OK)
                frc = factor_resident(c, p(c->L), N, Npad, p(c->invD), SC_LOGDET, SC_INFO, &ld_total, ld, 0, panel);
        } else {
            const int64_t Nb = keep, R = Npad - Nb;
            T* rows = p(c->L) + Nb * ld;
            // X = S[R, 0:Nb] L11^-T, the new rows of L left of the tail block.  A new train site that is a
            // resident candidate already has this row: it is the leading part of its row of V^T (both are
            // C[site, A] L^-T against the same kept blocks).  Then rows [Nb, p0) keep what they hold, rows
            // [p0, N) are gathered from V^T and only the R x R tail block of S is regenerated -- no
            // triangular solve against the kept factor (38 ms for 256 rows at N = 50 000).
            std::vector<int64_t> src_row, lrow;
            std::vector<T> lscale;
            bool second = false;
            // candidates sharded over ranks: the rows come from their owners (one exchange)
            int placed = 0;
            c->rows_from_peers = 0;
            if (sharded) {
                frc = exchange_new_rows(c, Nb, p0, &placed);
                if (frc != ALGP_OK) { prof_span_end(c); return frc; }
            }
            if (placed) {
                frc = kmat_launch<T>(c, s, (const int64_t*)c->Aidx.p + Nb, N - Nb, R, (const int64_t*)c->Aidx.p + Nb, N - Nb, R,
                                     (const T*)c->varA.p + Nb, c->pool_is_cov ? 0 : 1, nullptr, 1, rows + Nb, ld);
                c->factor_rows_from_vt = Npad - p0;
            } else if (!sharded && vt_rows_for_new_sites(c, Nb, p0, src_row, lrow, lscale, second)) {
                frc = kmat_launch<T>(c, s, (const int64_t*)c->Aidx.p + Nb, N - Nb, R, (const int64_t*)c->Aidx.p + Nb, N - Nb, R,
                                     (const T*)c->varA.p + Nb, c->pool_is_cov ? 0 : 1, nullptr, 1, rows + Nb, ld);
                const size_t nr = src_row.size();
                if (frc == ALGP_OK) frc = ensure(c, c->auxIdx, sizeof(int64_t) * 2 * std::max<size_t>(nr, 1));
                if (frc == ALGP_OK) frc = ensure(c, c->auxVar, sizeof(T) * std::max<size_t>(nr, 1) + 256);
                if (frc == ALGP_OK && nr > 0) {
                    int64_t* d_src = (int64_t*)c->auxIdx.p;
                    int64_t* d_lrow = d_src + nr;
                    hipMemcpyAsync(d_src, src_row.data(), sizeof(int64_t) * nr, hipMemcpyHostToDevice, c->stream);
                    if (second) {
                        hipMemcpyAsync(d_lrow, lrow.data(), sizeof(int64_t) * nr, hipMemcpyHostToDevice, c->stream);
                        hipMemcpyAsync(c->auxVar.p, lscale.data(), sizeof(T) * nr, hipMemcpyHostToDevice, c->stream);
                    }
                    frc = gather_rows_launch<T>(c, p(c->Vt), c->ldv, d_src, p(c->L) + p0 * ld, ld, (int64_t)nr, Nb,
                                                second ? d_lrow : nullptr, second ? (const T*)c->auxVar.p : nullptr,
                                                p(c->L), ld);
                    if (frc == ALGP_OK) frc = sync(c);               // the index vectors are host temporaries
                }
                c->factor_rows_from_vt = (int64_t)src_row.size();
            } else {
                c->factor_rows_from_vt = 0;
                // regenerate rows [Nb, Npad) of S (all columns), identity on the padded diagonal
                frc = kmat_launch<T>(c, s, (const int64_t*)c->Aidx.p + Nb, N - Nb, R, (const int64_t*)c->Aidx.p, N, Npad,
                                     (const T*)c->varA.p + Nb, c->pool_is_cov ? 0 : 1, nullptr, 1, rows, ld, Nb);
                // X = S[R, 0:Nb] L11^-T  (in place, against the kept blocks only)
                if (frc == ALGP_OK)
                    frc = trsm_blocked<T>(c, ALGP_PROF_GEMM_CHOL, rows, R, ld, p(c->L), Nb, ld, p(c->invD));
            }
            // S_RR -= X X^T
            if (frc == ALGP_OK)
                frc = syrk_skinny_sub<T>(c, ALGP_PROF_GEMM_CHOL, rows, R, Nb, ld, rows + Nb, ld, c->auxW);
            double ld_tail = 0;
            if (frc == ALGP_OK)
                frc = factor_resident(c, rows + Nb, N - Nb, R, p(c->invD) + Nb * NB, SC_LOGDET, SC_INFO, &ld_tail, ld, Nb);
            if (frc == ALGP_OK) {
                // log det over the whole diagonal (the kept blocks' share is not stored separately)
                double* sc = (double*)c->scal.p;
                hipMemsetAsync(sc + SC_AUXLOGDET, 0, sizeof(double), c->stream);
                frc = logdiag_launch<T>(c, p(c->L), ld, N, sc + SC_AUXLOGDET);
                if (frc == ALGP_OK) {
                    hipMemcpyAsync(&ld_total, sc + SC_AUXLOGDET, sizeof(double), hipMemcpyDeviceToHost, c->stream);
                    frc = sync(c);
                    ld_total *= 2.0;
                }
            }
        }
        prof_span_end(c);
        ALGP_TRY(frc);
        T* z_src = (panel && panel->done && panel->z_row >= 0) ? panel->P + panel->z_row * panel->ldp : nullptr;
        return finish_factor(c, keep, p0, ld_total, z_src);
    }

    // after L (rows >= keep new) is in place: log det, z = L^-1 (y - ybar), y0' S^-1 y0, bookkeeping
    static int finish_factor(algp_ctx* c, int64_t keep, int64_t p0, double ld_total, T* z_src = nullptr) {
        const int64_t N = c->N, Npad = c->Npad, ld = c->Lld;
        c->logdet = ld_total;
        if (keep > 0) {
            // z = u - ybar w, u = L^-1 y, w = L^-1 1: the leading entries of u and w only depend on the kept rows of
            // L (and their y), so the substitutions resume at the first changed block instead of row 0
            int64_t pu = 0;
            const int64_t lim = std::min<int64_t>(std::min<int64_t>(p0, c->uw_rows), (int64_t)c->fact_y.size());
            while (pu < lim && c->fact_y[pu] == c->train_y_host[pu]) ++pu;
            int64_t ku = std::min<int64_t>(keep, pu / NB * NB);
            const size_t need = sizeof(T) * (size_t)ld;
            if (!c->uvec.p || c->uvec.cap < need || !c->wvec.p || c->wvec.cap < need) {
                ALGP_TRY(ensure(c, c->uvec, need));
                ALGP_TRY(ensure(c, c->wvec, need));
                ku = 0;
            }
            T* u = p(c->uvec);
            T* w = p(c->wvec);
            ALGP_TRY(uw_init_launch<T>(c, u, w, (const T*)c->yraw.p, ku, N, Npad));
            ALGP_TRY(tail_gemv2_launch<T>(c, p(c->L), ld, ku, Npad, u, w));
            ALGP_TRY(trsv_forward2<T>(c, p(c->L), Npad, ld, p(c->invD), u, w, ku / NB));
            ALGP_TRY(uw_combine_launch<T>(c, p(c->z), u, w, (T)c->ybar, Npad));
            c->uw_rows = N;
            c->uw_stable = std::min(c->uw_stable, ku);
            c->fact_y = c->train_y_host;
        } else if (z_src) {
            // z rode along with the factorisation as a row of the candidates' panel (fit_and_solve): no substitution launch
            ALGP_HIP(hipMemcpyAsync(c->z.p, z_src, sizeof(T) * Npad, hipMemcpyDeviceToDevice, c->stream));
            ALGP_HIP(hipMemsetAsync(z_src, 0, sizeof(T) * Npad, c->stream));       // the row is a padding row of V^T again
            c->uw_rows = 0;
            c->uw_stable = 0;
        } else {
            ALGP_HIP(hipMemcpyAsync(c->z.p, c->y0.p, sizeof(T) * Npad, hipMemcpyDeviceToDevice, c->stream));
            ALGP_TRY(trsv_forward<T>(c, p(c->L), Npad, ld, p(c->invD), p(c->z)));
            c->uw_rows = 0;
            c->uw_stable = 0;
        }
        c->alpha_valid = false;                   // alpha = L^-T z: on first use (need_alpha)
        std::vector<T> zh(Npad);
        ALGP_HIP(hipMemcpyAsync(zh.data(), c->z.p, sizeof(T) * Npad, hipMemcpyDeviceToHost, c->stream));
        ALGP_TRY(sync_checked(c, "factorize: forward substitution"));
        double q = 0;
        for (int64_t i = 0; i < N; ++i) q += (double)zh[i] * (double)zh[i];
        c->yalpha = q;    // y0' S^-1 y0 = |L^-1 y0|^2
        c->factored = true;
        c->train_dirty = false;
        c->Nfact = N;
        c->fact_idx = c->train_idx;
        c->fact_var = c->train_var_host;
        c->fact_hyp_stamp = c->hyp_stamp;
        c->kept_rows_last = keep;
        return ALGP_OK;
    }

    // Take the factor of the same train set from another context of the same device (an agent keeps one context
    // per candidate set -- the pool for greedy, the held-out points for predict -- and both need the factor of
    // the sampled sites).  Rows this context already holds for an unchanged leading part are kept; the rest is a
    // device-to-device copy; z, MLL terms etc. are then computed for THIS context's targets.
    static int factorize_from(algp_ctx* c, algp_ctx* src) {
        const int64_t N = c->N, Npad = c->Npad;
        if (src == c) return fail(c, ALGP_ERR_BAD_ARG, "factorize_from: source and destination are the same context");
        if (src->dtype != c->dtype || src->device != c->device)
            return fail(c, ALGP_ERR_BAD_ARG, "factorize_from: contexts differ in dtype or device");
        if (!src->factored || src->train_dirty) return fail(c, ALGP_ERR_STATE, "factorize_from: the source holds no factor");
        const Hypers &a = c->hyp, &b = src->hyp;
        bool same = a.D == b.D && a.kernel == b.kernel && a.outputscale == b.outputscale && a.noise == b.noise;
        for (int d = 0; same && d < a.D; ++d) same = a.inv_ls[d] == b.inv_ls[d];
        if (!same) return fail(c, ALGP_ERR_STATE, "factorize_from: hyper-parameters differ");
        if (src->N != N || src->fact_idx != c->train_idx || src->fact_var != c->train_var_host)
            return fail(c, ALGP_ERR_STATE, "factorize_from: the source factor belongs to a different train set");
        if (c->pool_is_cov || src->pool_is_cov)
            return fail(c, ALGP_ERR_STATE, "factorize_from: needs coordinate pools on both sides (an explicit covariance cannot be compared)");
        if (train_sites_hash(src, src->fact_idx) != train_sites_hash(c, c->train_idx))
            return fail(c, ALGP_ERR_STATE, "factorize_from: the two pools hold different coordinates at the train indices");
        int64_t keep = 0, p0 = 0;
        if (c->factored && c->fact_hyp_stamp == c->hyp_stamp && c->Lld > 0) {
            const int64_t lim = std::min<int64_t>(N, c->Nfact);
            while (p0 < lim && c->fact_idx[p0] == c->train_idx[p0] && c->fact_var[p0] == c->train_var_host[p0]) ++p0;
            keep = p0 / NB * NB;
        }
        c->factored = false;
        c->solved = false;
        ALGP_TRY(reserve_factor(c, Npad, keep, 0));
        ALGP_TRY(ensure(c, c->z, sizeof(T) * Npad));
        ALGP_TRY(ensure(c, c->alpha, sizeof(T) * Npad));
        hipStreamSynchronize(src->stream);                       // the source's factor is complete
        if (Npad > keep) {
            ALGP_HIP(hipMemcpy2DAsync(p(c->L) + keep * c->Lld, sizeof(T) * c->Lld, (const T*)src->L.p + keep * src->Lld,
                                      sizeof(T) * src->Lld, sizeof(T) * Npad, Npad - keep, hipMemcpyDeviceToDevice, c->stream));
            ALGP_HIP(hipMemcpyAsync(p(c->invD) + keep * NB, (const T*)src->invD.p + keep * NB, sizeof(T) * (Npad - keep) * NB,
                                    hipMemcpyDeviceToDevice, c->stream));
        }
        return finish_factor(c, keep, p0, src->logdet);
    }

    static int need_alpha(algp_ctx* c) {
        if (c->alpha_valid) return ALGP_OK;
        ALGP_HIP(hipMemcpyAsync(c->alpha.p, c->z.p, sizeof(T) * c->Npad, hipMemcpyDeviceToDevice, c->stream));
        ALGP_TRY(trsv_backward<T>(c, p(c->L), c->Npad, c->Lld, p(c->invD), p(c->alpha)));
        c->alpha_valid = true;
        return ALGP_OK;
    }

    static int set_candidates(algp_ctx* c, const int64_t* idx, int64_t M, int prior_noise, const void* extra) {
        if (c->pool_is_cov && !prior_noise)
            return fail(c, ALGP_ERR_BAD_ARG, "an explicit pool covariance carries sigma_n^2 on its diagonal: prior_includes_noise must be 1");
        const int64_t Mpad = round_up(std::max<int64_t>(M, 1), NB);
        c->M = M;
        c->Mpad = Mpad;
        c->prior_noise = prior_noise;
        c->cand_idx.assign(idx, idx + M);
        c->cand_pos.assign(c->n_pool, -1);
        for (int64_t j = 0; j < M; ++j) c->cand_pos[idx[j]] = j;
        ALGP_TRY(ensure(c, c->Cidx, sizeof(int64_t) * Mpad));
        if (M > 0) ALGP_HIP(hipMemcpyAsync(c->Cidx.p, idx, sizeof(int64_t) * M, hipMemcpyHostToDevice, c->stream));
        if (extra) {
            ALGP_TRY(ensure(c, c->cextra, sizeof(T) * Mpad));
            ALGP_HIP(hipMemcpyAsync(c->cextra.p, extra, sizeof(T) * M, hipMemcpyHostToDevice, c->stream));
        } else {
            release(c, c->cextra);
        }
        ALGP_TRY(sync(c));
        c->solved = false;
        return ALGP_OK;
    }

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
    static int solve_prepare(algp_ctx* c, int incremental, SolvePlan& pl) {
        const int64_t N = c->N, Npad = c->Npad, M = c->M, Mpad = c->Mpad;
        pl.carried_sums = incremental && c->uw_rows == N && c->uvec.p && c->wvec.p;
        const int64_t ldv = Npad + MAX_APPEND;
        int64_t keep = 0;
        if (incremental && c->Vt.p && c->vt_hyp_stamp == c->hyp_stamp && c->vt_prior_noise == c->prior_noise &&
            !c->vt_has_extra && !c->cextra.p && c->vt_cand_idx == c->cand_idx) {
            const int64_t lim = std::min<int64_t>((int64_t)c->vt_fact_idx.size(), N);
            int64_t p0 = 0;
            while (p0 < lim && c->vt_fact_idx[p0] == c->fact_idx[p0] && c->vt_fact_var[p0] == c->fact_var[p0]) ++p0;
            keep = p0 / NB * NB;
            // Rows were appended behind p0 unchanged ones: the columns left of p0 stay as they are (L's old rows do not change),
            // so only [p0, N) has to be solved -- at 16-column granularity, as one or two ranges of at most 64 columns inside
            // a 128-column block of the factor (tail.hip): HBM-bound, where re-solving the whole open 128-block walks all of
            // V^T on the matrix cores at full tile width (28 -> 13 ms per step at N = 50 000 x 100 000 candidates).
            // $ALGP_TAIL_COLS=0: the 128-column blocks as before.  Small problems keep them too (nothing to gain).
            const bool tail_on = env_switch("ALGP_TAIL_COLS", true);                                 // read per call: tests flip it
            if (tail_on && p0 >= 2048 && Mpad >= 2048 && c->cur == c->stream) {
                // exactly the appended rows [p0, N) when there are at most 64 of them (tail.hip handles any first column; the
                // epilogue's inverse is that of a window of L around the range, solve_run); more than 64: from the 16-column
                // boundary below p0 to the one above N, as ranges of at most 64 columns
                const bool exact = N > p0 && N - p0 <= 64;
                const int64_t k16 = exact ? p0 : p0 / 16 * 16, c1 = exact ? N : round_up(N, 16);
                int n = 0;
                bool ok = c1 > k16;
                if (exact) {
                    pl.seg_c0[0] = p0;
                    pl.seg_w[0] = (int)(N - p0);
                    pl.seg_window = true;
                    n = 1;
                }
                // at most 64 new columns: ONE pass over V^T even where they straddle two 128-column blocks of the factor (the
                // epilogue then takes the inverse of the 128 x 128 window of L at (k16, k16), solve_run)
                if (ok && !exact && c1 - k16 <= 64 && k16 / NB != (c1 - 1) / NB && k16 + NB <= Npad) {
                    pl.seg_c0[0] = k16;
                    pl.seg_w[0] = (int)(c1 - k16);
                    pl.seg_window = true;
                    n = 1;
                }
                for (int64_t a = k16; a < c1 && ok && !pl.seg_window;) {
                    const int64_t b = std::min<int64_t>(c1, (a / NB + 1) * NB);
                    if (b - a > 64 || n == 2) { ok = false; break; }
                    pl.seg_c0[n] = a;
                    pl.seg_w[n] = (int)(b - a);
                    ++n;
                    a = b;
                }
                if (ok && n > 0) {
                    pl.nseg = n;
                    keep = k16;
                }
            }
        }
        // candidate kinds under the current train set
        std::vector<int>& kind = pl.kind;
        kind.assign(Mpad, -1);
        if (c->prior_noise)
            for (int64_t j = 0; j < M; ++j) kind[j] = (int)c->pos_in_train[c->cand_idx[j]];
        std::vector<int64_t>& became_unit = pl.became_unit;
        became_unit.clear();
        if (keep > 0) {
            // a kept column block is only valid for a row whose right-hand side is unchanged:
            //  - ordinary -> unit row e_pos with pos >= keep: the solution is zero before pos: zero the kept part;
            //  - anything else that changed: give up the reuse.
            for (int64_t j = 0; j < M && keep > 0; ++j) {
                const int was = c->vt_kind[j], now = kind[j];
                if (was == now) continue;
                if (was < 0 && now >= keep) became_unit.push_back(j);
                else if (!(was >= keep && now >= keep)) keep = 0;      // unit rows beyond `keep` are rebuilt anyway
            }
            if (keep == 0) became_unit.clear();
        }
        if (keep == 0) { pl.nseg = 0; pl.seg_window = false; }
        pl.keep = keep;
        c->solved = false;
        if (!c->Vt.p || c->ldv_cap < ldv || (keep == 0 && !incremental && c->ldv_cap != ldv) ||
            c->Vt.cap < sizeof(T) * Mpad * c->ldv_cap) {
            // (re)allocate; keep the valid columns when growing.  A caller that asks for reuse gets 12.5 %
            // headroom in the row stride from the start, so that a growing train set does not force a
            // re-layout (a 2-D copy of all of V^T) the first time it crosses a 128 boundary.
            const int64_t newcap = incremental ? round_up(ldv + ldv / 8, NB) : ldv;
            DevBuf nv;
            ALGP_TRY(ensure(c, nv, sizeof(T) * Mpad * newcap));
            if (keep > 0) {
                hipError_t e = hipMemcpy2DAsync(nv.p, sizeof(T) * newcap, c->Vt.p, sizeof(T) * c->ldv_cap, sizeof(T) * keep,
                                                Mpad, hipMemcpyDeviceToDevice, c->stream);
                if (e != hipSuccess) { release(c, nv); return fail(c, ALGP_ERR_HIP, hipGetErrorString(e)); }
                hipStreamSynchronize(c->stream);
            }
            release(c, c->Vt);
            c->Vt = nv;
            c->ldv_cap = newcap;
        }
        const int64_t ldc = c->ldv_cap;          // row stride of V^T
        c->ldv = ldc;
        ALGP_TRY(ensure(c, c->dstat, sizeof(T) * Mpad));
        ALGP_TRY(ensure(c, c->mu, sizeof(T) * Mpad));
        ALGP_TRY(ensure(c, c->tvec, sizeof(T) * 2 * Mpad));
        ALGP_TRY(ensure(c, c->alive, Mpad));
        ALGP_TRY(ensure(c, c->scores, sizeof(double) * Mpad));
        ALGP_TRY(ensure(c, c->lrow, sizeof(T) * ldc));
        ALGP_TRY(ensure(c, c->prevrows, sizeof(T) * MAX_APPEND * ldc));
        {
            // greedy semantics: a candidate that is a train site is the unit vector e_pos (its
            // noise changes); predictive semantics: it is an ordinary point at the same location
            ALGP_TRY(ensure(c, c->ckind, sizeof(int) * Mpad));
            ALGP_HIP(hipMemcpyAsync(c->ckind.p, kind.data(), sizeof(int) * Mpad, hipMemcpyHostToDevice, c->stream));
            if (!became_unit.empty() && keep > 0) {                      // one launch for all of them
                ALGP_TRY(ensure(c, c->auxIdx, sizeof(int64_t) * became_unit.size()));
                ALGP_HIP(hipMemcpyAsync(c->auxIdx.p, became_unit.data(), sizeof(int64_t) * became_unit.size(), hipMemcpyHostToDevice, c->stream));
                ALGP_TRY(zero_listed_rows_launch<T>(c, p(c->Vt), ldc, (const int64_t*)c->auxIdx.p, (int64_t)became_unit.size(), keep));
            }
            ALGP_TRY(sync(c));
        }
        KmatSrc s = make_src(c);
        // B^T: row j = C[cand_j, A] (ordinary) or e_pos (train-site candidate); zero padding.
        // Only columns >= keep are (re)generated and solved.
        return kmat_launch<T>(c, s, (const int64_t*)c->Cidx.p, M, Mpad, (const int64_t*)c->Aidx.p + keep, N - keep,
                              ldv - keep, nullptr, 0, c->prior_noise ? (const int*)c->ckind.p : nullptr, 0,
                              p(c->Vt) + keep, ldc, 0, keep);
    }

    // the solve proper, against the resident factor.  A from-scratch solve of 33 .. 400 tile rows (a rank's share of the
    // candidates on 4-8 GPUs, a held-out set) runs as ONE task-list launch (chol_dag.hip without the factorisation's
    // own tasks; $ALGP_SOLVE_DAG=0: the launch sequences of potrf.hip); everything else is trsm_blocked.
    static int solve_run(algp_ctx* c, SolvePlan& pl) {
        const bool solve_dag_on = env_switch("ALGP_SOLVE_DAG", true);                                // read per call: tests flip it
        const int64_t Npad = c->Npad, Mpad = c->Mpad, ldc = c->ldv, keep = pl.keep;
        prof_span_begin(c, ALGP_PROF_TRSM, (double)(Npad - keep) * (double)(Npad + keep) * (double)Mpad,
                        sizeof(T) * (double)Mpad * (double)Npad);
        int trc = ALGP_OK;
        if (pl.nseg > 0 && pl.seg_window) {
            // the inverse of a 128 x 128 window of L that contains the range (inv(D)[S, S] = inv(D[S, S]) for any diagonal range S of
            // a lower-triangular D): from the range's first column, or -- near the end of the factor -- the last 128 rows
            const int64_t w0 = std::min<int64_t>(pl.seg_c0[0], Npad - NB), o = pl.seg_c0[0] - w0;
            trc = ensure(c, c->tailE, sizeof(T) * NB * NB);
            if (trc == ALGP_OK) trc = trinv_diag_launch<T>(c, p(c->L) + w0 * c->Lld + w0, c->Lld, p(c->tailE));
            if (trc == ALGP_OK)
                trc = tail_cols_launch<T>(c, ALGP_PROF_TAIL_COLS, p(c->Vt), Mpad, ldc, p(c->L), c->Lld, Npad, (const T*)nullptr,
                                          pl.seg_c0[0], pl.seg_w[0], p(c->tailE) + o * NB + o);
        } else if (pl.nseg > 0) {
            for (int q = 0; q < pl.nseg && trc == ALGP_OK; ++q)
                trc = tail_cols_launch<T>(c, ALGP_PROF_TAIL_COLS, p(c->Vt), Mpad, ldc, p(c->L), c->Lld, Npad,
                                          p(c->invD) + (pl.seg_c0[q] / NB) * NB * NB, pl.seg_c0[q], pl.seg_w[q]);
        } else if (solve_dag_on && keep == 0 && Mpad / NB > 32 && panel_fits(Npad, Mpad) && c->cur == c->stream)
            trc = solve_dag_panel<T>(c, p(c->L), Npad, c->Lld, p(c->invD), (int*)((double*)c->scal.p + SC_STALL), p(c->Vt), ldc, Mpad, 1);
        else {
            // a from-scratch solve of more than 320 tile rows: its launches leave the rows' sums of v^2 and v z per column tile
            // (utils.py:301-304 needs nothing else of V^T), the 8 GB pass over V^T at config 4 falls away
            T* stat = nullptr;
            const bool stats_on = env_switch("ALGP_ROW_STATS", true);                                  // read per call: tests flip it
            if (stats_on && keep == 0 && !pl.carried_sums && ensure(c, c->rowstat, sizeof(T) * 2 * (size_t)(Npad / NB) * (size_t)Mpad) == ALGP_OK)
                stat = p(c->rowstat);
            trc = trsm_blocked<T>(c, ALGP_PROF_GEMM_TRSM, p(c->Vt), Mpad, ldc, p(c->L), Npad, c->Lld, p(c->invD), keep, p(c->z), stat,
                                  Mpad, &pl.rowstat_done);
        }
        prof_span_end(c);
        return trc;
    }

    static int solve_finish(algp_ctx* c, int incremental, const unsigned char* alive_host, const SolvePlan& pl) {
        const int64_t N = c->N, Npad = c->Npad, M = c->M, Mpad = c->Mpad, ldc = c->ldv, keep = pl.keep;
        T* ss = p(c->tvec);
        T* dot = ss + Mpad;
        if (pl.carried_sums) {
            // the factor update maintains z = u - ybar w: carry sum v^2, sum v u, sum v w over the finished column
            // blocks of V^T from step to step and read only the new columns (a full pass is 40 GB at N = 50 000)
            const size_t need = sizeof(T) * 6 * (size_t)Mpad;            // 3 running sums + 3 sums of the open tail
            bool ok = keep > 0 && c->acc3.p && c->acc3.cap >= need && c->acc_M == M && c->acc_cols > 0 &&
                      c->acc_cols <= keep && c->acc_cols <= c->uw_stable;
            if (!ok) {
                ALGP_TRY(ensure(c, c->acc3, need));
                ALGP_HIP(hipMemsetAsync(c->acc3.p, 0, sizeof(T) * 3 * (size_t)Mpad, c->stream));
                c->acc_cols = 0;
            }
            T* acc = p(c->acc3);
            T* tmp = acc + 3 * Mpad;
            if (!pl.became_unit.empty()) {                                // their kept columns were zeroed above: one launch
                const size_t nb = pl.became_unit.size();                  // (three 8-byte memsets per row before: ~5 us each)
                ALGP_TRY(ensure(c, c->auxIdx, sizeof(int64_t) * nb));
                ALGP_HIP(hipMemcpyAsync(c->auxIdx.p, pl.became_unit.data(), sizeof(int64_t) * nb, hipMemcpyHostToDevice, c->stream));
                ALGP_TRY(zero_rows3_launch<T>(c, acc, Mpad, (const int64_t*)c->auxIdx.p, (int64_t)nb));
            }
            const int64_t fin = N / NB * NB;                              // column blocks no later append can touch
            if (fin > c->acc_cols)
                ALGP_TRY(rows_reduce3_launch<T>(c, p(c->Vt), M, ldc, c->acc_cols, fin, p(c->uvec), p(c->wvec), acc, Mpad, 1));
            ALGP_TRY(rows_reduce3_launch<T>(c, p(c->Vt), M, ldc, fin, Npad, p(c->uvec), p(c->wvec), tmp, Mpad, 0));
            ALGP_TRY(combine3_launch<T>(c, M, acc, tmp, Mpad, (T)c->ybar, ss, dot));
            c->acc_cols = fin;
            c->acc_M = M;
            c->uw_stable = N;
        } else {
            c->acc_cols = 0;
            if (pl.rowstat_done) ALGP_TRY(rowstat_combine_launch<T>(c, p(c->rowstat), Mpad, (int)(Npad / NB), M, ss, dot));
            else ALGP_TRY(rows_reduce_launch<T>(c, p(c->Vt), M, ldc, Npad, p(c->z), ss, dot));
        }
        const T prior = (T)(c->hyp.outputscale + (c->prior_noise ? c->hyp.noise : 0.0));
        ALGP_TRY(cand_finalize_launch<T>(c, M, (const int*)c->ckind.p, (const int64_t*)c->Cidx.p,
                                         c->pool_is_cov ? (const T*)c->Cp.p : nullptr, c->n_pool, prior,
                                         c->cextra.p ? (const T*)c->cextra.p : nullptr, ss, dot, (T)c->ybar, p(c->dstat),
                                         p(c->mu), (unsigned char*)c->alive.p));
        if (alive_host) ALGP_HIP(hipMemcpyAsync(c->alive.p, alive_host, M, hipMemcpyHostToDevice, c->stream));
        ALGP_TRY(sync_checked(c, "solve_candidates"));
        c->ncols = Npad;
        c->picks.clear();
        c->mi_valid = false;
        ALGP_TRY(reset_lazy(c));
        c->solved = true;
        c->vt_fact_idx = c->fact_idx;
        c->vt_fact_var = c->fact_var;
        c->vt_cand_idx = c->cand_idx;
        c->vt_kind.assign(pl.kind.begin(), pl.kind.begin() + M);
        c->vt_hyp_stamp = c->hyp_stamp;
        c->vt_prior_noise = c->prior_noise;
        c->vt_has_extra = c->cextra.p != nullptr;
        c->kept_cols_last = keep;
        return ALGP_OK;
    }

    static int solve_candidates(algp_ctx* c, int incremental, const unsigned char* alive_host) {
        if (!c->factored || c->train_dirty) return fail(c, ALGP_ERR_STATE, "solve_candidates: call algp_factorize first");
        SolvePlan pl;
        ALGP_TRY(solve_prepare(c, incremental, pl));
        ALGP_TRY(solve_run(c, pl));
        return solve_finish(c, incremental, alive_host, pl);
    }

    // GP-fit + candidate solve of one planning step (bench.py's step).  Up to 400 x 128 candidate rows (a rank's share on
    // 2-8 GPUs) the two are ONE launch: the rows of B^T are extra block rows of the factorisation's task list (TRSM / UPD
    // tasks without a diagonal), so V^T = B^T L^-T comes out of the launch that factors S -- the candidates' tile products
    // fill the machine while the diagonal chain alone would leave it idle, and the 140 short launches of a separate
    // mid-sized solve disappear ($ALGP_FOLD=0: the two phases back to back).  Larger candidate sets keep the two phases:
    // the factorisation, then the three-stream sweep of potrf.hip, which wins from ~55 000 rows on.  (Overlapping the two
    // as separate launch sequences on streams was measured in round 1 -- 207 vs 193 ms/step -- and removed.)
    static int fit_and_solve(algp_ctx* c) {
        const bool fold_on = env_switch("ALGP_FOLD", true);                                          // read per call: tests flip it
        if (!fold_on || c->M == 0 || !panel_fits(c->Npad, c->Mpad)) {
            ALGP_TRY(factorize(c, 0));
            return solve_candidates(c, 0, nullptr);
        }
        c->factored = false;
        SolvePlan pl;
        ALGP_TRY(solve_prepare(c, 0, pl));                           // B^T is in place before the launch that consumes it
        Panel pn{p(c->Vt), c->ldv, c->Mpad, 1, false};
        // a spare (padding) row of the candidates' last tile carries y - ybar through the launch: z = L^-1 (y - ybar) comes out
        // as that row of P L^-T, and the forward substitution behind the launch (0.41 ms at N = 10 000, with the machine
        // idle) falls away (a candidate count that fills its last tile keeps the substitution)
        if (c->M < c->Mpad) {
            pn.z_row = c->M;
            ALGP_HIP(hipMemcpyAsync(p(c->Vt) + c->M * c->ldv, c->y0.p, sizeof(T) * c->Npad, hipMemcpyDeviceToDevice, c->stream));
        }
        ALGP_TRY(factorize(c, 0, &pn));
        if (!pn.done) {
            if (pn.z_row >= 0) ALGP_HIP(hipMemsetAsync(p(c->Vt) + pn.z_row * c->ldv, 0, sizeof(T) * c->Npad, c->stream));
            ALGP_TRY(solve_run(c, pl));
        }
        return solve_finish(c, 0, nullptr, pl);
    }

    static int get_posterior(algp_ctx* c, void* mu, void* var) {
        if (!c->solved) return fail(c, ALGP_ERR_STATE, "get_posterior: call algp_solve_candidates first");
        ALGP_TRY(flush_lazy(c));
        if (mu) ALGP_HIP(hipMemcpyAsync(mu, c->mu.p, sizeof(T) * c->M, hipMemcpyDeviceToHost, c->stream));
        if (var) ALGP_HIP(hipMemcpyAsync(var, c->dstat.p, sizeof(T) * c->M, hipMemcpyDeviceToHost, c->stream));
        return sync(c);
    }

    static int get_posterior_cov(algp_ctx* c, void* cov_out, double* mi_out) {
        if (!c->solved) return fail(c, ALGP_ERR_STATE, "get_posterior_cov: call algp_solve_candidates first");
        if (c->pool_is_cov) return fail(c, ALGP_ERR_BAD_ARG, "get_posterior_cov needs a coordinate pool");
        const int64_t M = c->M, Mpad = c->Mpad;
        ALGP_TRY(ensure(c, c->auxA, sizeof(T) * Mpad * Mpad));
        ALGP_TRY(ensure(c, c->auxW, sizeof(T) * Mpad * Mpad));
        ALGP_TRY(ensure(c, c->auxInv, sizeof(T) * Mpad * NB));
        KmatSrc s = make_src(c);
        const T* extra = c->cextra.p ? (const T*)c->cextra.p : nullptr;
        c->last_jitter = 0.0;
        // mi = H(cov_xx) - H(cov) (utils.py:314) takes the log-determinant of cov_xx = K_xx WITHOUT noise, which is
        // singular to working precision on dense grids or with long lengthscales: the reference's slogdet then returns
        // rounding noise (its sign is dropped, utils.py:193) where a Cholesky stops at a non-positive pivot.  Instead of
        // aborting the caller's run, the two matrices are rebuilt with a growing jitter on BOTH diagonals (64 eps * prior
        // variance, x100 per retry) and the jitter that was needed is reported (algp_last_jitter): a deliberate,
        // visible divergence in a regime where the reference's own number carries no information.
        const double eps = sizeof(T) == 8 ? 2.220446049250313e-16 : 1.1920928955078125e-07;
        for (int attempt = 0;; ++attempt) {
            const double jitter = attempt == 0 ? 0.0 : 64.0 * eps * c->hyp.outputscale * pow(100.0, attempt - 1);
            // cov_xx = K_xx + diag(test_var)   (utils.py:297; no likelihood noise)
            ALGP_TRY(kmat_launch<T>(c, s, (const int64_t*)c->Cidx.p, M, Mpad, (const int64_t*)c->Cidx.p, M, Mpad, extra, 0,
                                    nullptr, 1, p(c->auxA), Mpad));
            // cov = cov_xx - V^T V  (utils.py:305)
            ALGP_TRY(gemm_nt_launch<T>(c, ALGP_PROF_GEMM_OTHER, Mpad, Mpad, c->Npad, (T)-1, p(c->Vt), c->ldv, p(c->Vt),
                                       c->ldv, (T)1, p(c->auxA), Mpad, p(c->auxW), Mpad, 0));
            if (attempt == 0 && cov_out)
                ALGP_HIP(hipMemcpy2DAsync(cov_out, sizeof(T) * M, c->auxW.p, sizeof(T) * Mpad, sizeof(T) * M, M,
                                          hipMemcpyDeviceToHost, c->stream));
            ALGP_TRY(sync(c));
            if (!mi_out) break;
            if (jitter > 0.0) {
                ALGP_TRY(add_diag_launch<T>(c, p(c->auxA), M, Mpad, (T)jitter));
                ALGP_TRY(add_diag_launch<T>(c, p(c->auxW), M, Mpad, (T)jitter));
            }
            double ld_xx = 0, ld_cov = 0;
            int rc = factor_resident(c, p(c->auxA), M, Mpad, p(c->auxInv), SC_AUXLOGDET, SC_AUXINFO, &ld_xx);
            if (rc == ALGP_OK) rc = factor_resident(c, p(c->auxW), M, Mpad, p(c->auxInv), SC_AUXLOGDET, SC_AUXINFO, &ld_cov);
            if (rc == ALGP_OK) {
                *mi_out = 0.5 * (ld_xx - ld_cov);     // the k*CONST terms cancel (utils.py:314)
                c->last_jitter = jitter;
                break;
            }
            if (rc != ALGP_ERR_NOT_PD || attempt >= 5) return rc;
            c->err.clear();
        }
        return ALGP_OK;
    }

    static int posterior_mean(algp_ctx* c, const int64_t* idx, int64_t M, void* mu_out) {
        if (!c->factored || c->train_dirty) return fail(c, ALGP_ERR_STATE, "posterior_mean: call algp_factorize first");
        if (c->pool_is_cov) return fail(c, ALGP_ERR_BAD_ARG, "posterior_mean needs a coordinate pool");
        if (M == 0) return ALGP_OK;
        ALGP_TRY(ensure(c, c->auxIdx, sizeof(int64_t) * M));
        ALGP_TRY(ensure(c, c->auxD, sizeof(T) * M));
        ALGP_HIP(hipMemcpyAsync(c->auxIdx.p, idx, sizeof(int64_t) * M, hipMemcpyHostToDevice, c->stream));
        ALGP_TRY(need_alpha(c));
        ALGP_TRY(kgemv_launch<T>(c, M, (const int64_t*)c->auxIdx.p, (const T*)c->Xs.p, c->hyp.DP, c->N,
                                 (const int64_t*)c->Aidx.p, (const T*)c->alpha.p, c->hyp.kernel, (T)c->hyp.outputscale,
                                 (T)c->ybar, p(c->auxD)));
        ALGP_HIP(hipMemcpyAsync(mu_out, c->auxD.p, sizeof(T) * M, hipMemcpyDeviceToHost, c->stream));
        const int rc = sync_checked(c, "posterior_mean");
        if (rc != ALGP_OK) c->alpha_valid = false;
        return rc;
    }

    // ------------------------------------------------------------------ set entropies / inverse diagonals
    static int build_set_matrix(algp_ctx* c, const int64_t* idx, int64_t m, const void* var, int64_t* mpad_out, T* dst = nullptr) {
        const int64_t mpad = round_up(std::max<int64_t>(m, 1), NB);
        *mpad_out = mpad;
        if (!dst) ALGP_TRY(ensure(c, c->auxA, sizeof(T) * mpad * mpad));
        ALGP_TRY(ensure(c, c->auxInv, sizeof(T) * mpad * NB));
        ALGP_TRY(ensure(c, c->auxIdx, sizeof(int64_t) * mpad));
        ALGP_TRY(ensure(c, c->auxVar, sizeof(T) * mpad));
        if (m > 0) ALGP_HIP(hipMemcpyAsync(c->auxIdx.p, idx, sizeof(int64_t) * m, hipMemcpyHostToDevice, c->stream));
        if (var && m > 0)
            ALGP_HIP(hipMemcpyAsync(c->auxVar.p, var, sizeof(T) * m, hipMemcpyHostToDevice, c->stream));
        KmatSrc s = make_src(c);
        return kmat_launch<T>(c, s, (const int64_t*)c->auxIdx.p, m, mpad, (const int64_t*)c->auxIdx.p, m, mpad,
                              var ? (const T*)c->auxVar.p : nullptr, c->pool_is_cov ? 0 : 1, nullptr, 1, dst ? dst : p(c->auxA),
                              mpad);
    }

    static int set_entropy(algp_ctx* c, const int64_t* idx, int64_t m, const void* var, double* H) {
        if (m == 0) { *H = 0.0; return ALGP_OK; }
        int64_t mpad;
        ALGP_TRY(build_set_matrix(c, idx, m, var, &mpad));
        double ld = 0;
        ALGP_TRY(factor_resident(c, p(c->auxA), m, mpad, p(c->auxInv), SC_AUXLOGDET, SC_AUXINFO, &ld));
        *H = (double)m * ENT_CONST + 0.5 * ld;
        return ALGP_OK;
    }

    // diag(S^-1) = row sums of squares of L^-T (the triangular inverse on the MFMA GEMM)
    static int inverse_diag_resident(algp_ctx* c, int64_t m, int64_t mpad, void* diag_out) {
        ALGP_TRY(ensure(c, c->auxW, sizeof(T) * mpad * mpad));
        ALGP_TRY(ensure(c, c->auxD, sizeof(T) * mpad));
        ALGP_TRY(set_identity_launch<T>(c, p(c->auxW), mpad, mpad));
        ALGP_TRY(trinv_upper<T>(c, ALGP_PROF_GEMM_OTHER, p(c->auxW), mpad, mpad, p(c->auxA), mpad, p(c->auxInv)));
        ALGP_TRY(rows_reduce_launch<T>(c, p(c->auxW), m, mpad, mpad, (const T*)nullptr, p(c->auxD), (T*)nullptr));
        ALGP_HIP(hipMemcpyAsync(diag_out, c->auxD.p, sizeof(T) * m, hipMemcpyDeviceToHost, c->stream));
        return sync(c);
    }

    static int set_inverse_diag(algp_ctx* c, const int64_t* idx, int64_t m, const void* var, void* diag_out, double* H) {
        if (m == 0) { if (H) *H = 0.0; return ALGP_OK; }
        int64_t mpad;
        ALGP_TRY(build_set_matrix(c, idx, m, var, &mpad));
        double ld = 0;
        ALGP_TRY(factor_resident(c, p(c->auxA), m, mpad, p(c->auxInv), SC_AUXLOGDET, SC_AUXINFO, &ld));
        if (H) *H = (double)m * ENT_CONST + 0.5 * ld;
        return inverse_diag_resident(c, m, mpad, diag_out);
    }

    static int upload_padded(algp_ctx* c, DevBuf& b, const void* A, int64_t rows, int64_t cols, int64_t rpad,
                             int64_t cpad) {
        ALGP_TRY(ensure(c, b, sizeof(T) * rpad * cpad));
        ALGP_HIP(hipMemsetAsync(b.p, 0, sizeof(T) * rpad * cpad, c->stream));
        if (rows > 0 && cols > 0)
            ALGP_HIP(hipMemcpy2DAsync(b.p, sizeof(T) * cpad, A, sizeof(T) * cols, sizeof(T) * cols, rows,
                                      hipMemcpyHostToDevice, c->stream));
        return ALGP_OK;
    }

    static int entropy_from_cov(algp_ctx* c, const void* cov, int64_t k, double* H, void* L_out, double* logdet) {
        if (k == 0) { if (H) *H = 0.0; if (logdet) *logdet = 0.0; return ALGP_OK; }
        const int64_t kpad = round_up(k, NB);
        ALGP_TRY(upload_padded(c, c->auxA, cov, k, k, kpad, kpad));
        ALGP_TRY(pad_identity_launch<T>(c, p(c->auxA), k, kpad, kpad));
        ALGP_TRY(ensure(c, c->auxInv, sizeof(T) * kpad * NB));
        double ld = 0;
        ALGP_TRY(factor_resident(c, p(c->auxA), k, kpad, p(c->auxInv), SC_AUXLOGDET, SC_AUXINFO, &ld));
        if (H) *H = (double)k * ENT_CONST + 0.5 * ld;
        if (logdet) *logdet = ld;
        if (L_out) {
            ALGP_HIP(hipMemcpy2DAsync(L_out, sizeof(T) * k, c->auxA.p, sizeof(T) * kpad, sizeof(T) * k, k,
                                      hipMemcpyDeviceToHost, c->stream));
            ALGP_TRY(sync(c));
            T* Lh = (T*)L_out;
            for (int64_t i = 0; i < k; ++i)
                for (int64_t j = i + 1; j < k; ++j) Lh[i * k + j] = (T)0;
        }
        return ALGP_OK;
    }

    static int gemm_host(algp_ctx* c, int64_t m, int64_t n, int64_t k, double alpha, const void* A, const void* B,
                         double beta, const void* C, void* D) {
        const int64_t mp = round_up(std::max<int64_t>(m, 1), NB), np = round_up(std::max<int64_t>(n, 1), NB),
                      kp = round_up(std::max<int64_t>(k, 1), NB);
        DevBuf a, b, cc;
        int rc = upload_padded(c, a, A, m, k, mp, kp);
        if (rc == ALGP_OK) rc = upload_padded(c, b, B, n, k, np, kp);
        if (rc == ALGP_OK) rc = upload_padded(c, cc, (beta != 0.0 && C) ? C : nullptr, (beta != 0.0 && C) ? m : 0, n, mp, np);
        if (rc == ALGP_OK)
            rc = gemm_nt_launch<T>(c, ALGP_PROF_GEMM_OTHER, mp, np, kp, (T)alpha, (const T*)a.p, kp, (const T*)b.p, kp,
                                   (T)beta, (const T*)cc.p, np, (T*)cc.p, np, 0);
        if (rc == ALGP_OK) {
            hipError_t e = hipMemcpy2DAsync(D, sizeof(T) * n, cc.p, sizeof(T) * np, sizeof(T) * n, m,
                                            hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) rc = fail(c, ALGP_ERR_HIP, hipGetErrorString(e));
        }
        hipStreamSynchronize(c->stream);
        release(c, a); release(c, b); release(c, cc);
        return rc;
    }

    static int trsm_host(algp_ctx* c, const void* L, int64_t n, const void* B, int64_t m, void* X) {
        const int64_t np = round_up(std::max<int64_t>(n, 1), NB), mp = round_up(std::max<int64_t>(m, 1), NB);
        DevBuf l, b, inv;
        int rc = upload_padded(c, l, L, n, n, np, np);
        if (rc == ALGP_OK) rc = pad_identity_launch<T>(c, (T*)l.p, n, np, np);
        if (rc == ALGP_OK) rc = upload_padded(c, b, B, m, n, mp, np);
        if (rc == ALGP_OK) rc = ensure(c, inv, sizeof(T) * np * NB);
        for (int64_t kb = 0; rc == ALGP_OK && kb < np / NB; ++kb)
            rc = trinv_diag_launch<T>(c, (const T*)l.p + kb * NB * np + kb * NB, np, (T*)inv.p + kb * NB * NB);
        if (rc == ALGP_OK) rc = trsm_blocked<T>(c, ALGP_PROF_GEMM_OTHER, (T*)b.p, mp, np, (const T*)l.p, np, np, (const T*)inv.p);
        if (rc == ALGP_OK) {
            hipError_t e = hipMemcpy2DAsync(X, sizeof(T) * n, b.p, sizeof(T) * np, sizeof(T) * n, m,
                                            hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) rc = fail(c, ALGP_ERR_HIP, hipGetErrorString(e));
        }
        hipStreamSynchronize(c->stream);
        release(c, l); release(c, b); release(c, inv);
        return rc;
    }

    // ------------------------------------------------------------------ greedy
    // ---- MI criterion (agent.py:330-339): H(A u i) + H(Abar \ i) - H(all_i) per candidate --------------------------------
    // The last two terms need the diagonals of P = C_AbarAbar^-1 and Q = (C + D_all)^-1 over the WHOLE pool (see
    // mi_rank1_kernel in vecops.hip).  mi_build factors both matrices once per candidate solve and leaves the triangular
    // inverses X (P = X X^T) resident; mi_apply_pick folds a committed pick into both diagonals with one pass over each X
    // (O(n^2)) where the reference -- and round 2 of this library -- refactorised both matrices for every pick.
    static int mi_build(algp_ctx* c, double ss, double sm) {
        const int64_t n = c->n_pool;
        if (c->train_has_repeats)
            return fail(c, ALGP_ERR_STATE, "mutual_information: the train set lists a site more than once; fuse its readings first");
        const double vf = 1.0 / (1.0 / ss + 1.0 / sm);
        // current state: train set (with its noise) + committed picks
        std::vector<char> sampled(n, 0);
        std::vector<double> noise(n, 0.0);
        std::vector<T> trvar(c->Npad);
        ALGP_HIP(hipMemcpyAsync(trvar.data(), c->varA.p, sizeof(T) * c->Npad, hipMemcpyDeviceToHost, c->stream));
        ALGP_TRY(sync(c));
        for (int64_t a = 0; a < c->N; ++a) { sampled[c->train_idx[a]] = 1; noise[c->train_idx[a]] = (double)trvar[a]; }
        for (auto& pk : c->picks) {
            noise[pk.pool_idx] = sampled[pk.pool_idx] ? vf : ss;
            sampled[pk.pool_idx] = 1;
        }
        std::vector<int64_t> A, Abar, all(n);
        std::vector<T> vA, vall(n);
        c->mi_posbar.assign(n, -1);
        for (int64_t i = 0; i < n; ++i) {
            all[i] = i;
            vall[i] = (T)noise[i];
            if (sampled[i]) { A.push_back(i); vA.push_back((T)noise[i]); }
            else { c->mi_posbar[i] = (int64_t)Abar.size(); Abar.push_back(i); }
        }
        const int64_t mb = (int64_t)Abar.size();
        const int64_t npad = round_up(std::max<int64_t>(n, 1), NB), mbpad = round_up(std::max<int64_t>(mb, 1), NB);
        {
            // Two pool-wide matrices stay resident -- each is built, factored and inverted IN its buffer (L in the strictly
            // lower tiles, X = L^-T on and above the diagonal: trinv_upper_inplace) -- say so with the byte count instead of
            // failing half-way through the allocations.  At config 4's own pool (110 000 sites, fp64) that is 2 x 96.8 GB
            // (round 5 held a third matrix, the factor being inverted: 290 GB) and 4 n^3 / 3 = 1.8e15 flop for the first pick.
            const size_t need = sizeof(T) * ((size_t)npad * npad + (size_t)mbpad * mbpad + (size_t)npad * NB +
                                             (size_t)MAX_APPEND * (npad + mbpad));
            const size_t held = c->auxInv.cap + c->miXbar.cap + c->miXall.cap + c->miU.cap + c->miW.cap;
            size_t free_b = 0, total_b = 0;
            ALGP_HIP(hipMemGetInfo(&free_b, &total_b));
            if (need > held + free_b)
                return fail(c, ALGP_ERR_OOM,
                            "mutual_information: the criterion keeps the triangular inverses of two pool-wide matrices resident: " +
                                std::to_string(need) + " bytes for n_pool = " + std::to_string(n) + ", " +
                                std::to_string(held + free_b) + " available; score this pool with the entropy criterion "
                                "(it needs the candidates' rows only) or a smaller pool");
        }
        double H_A = 0, H_bar = 0, H_all = 0;
        ALGP_TRY(set_entropy(c, A.data(), (int64_t)A.size(), vA.data(), &H_A));
        ALGP_TRY(ensure(c, c->miXbar, sizeof(T) * mbpad * mbpad));
        ALGP_TRY(ensure(c, c->miXall, sizeof(T) * npad * npad));
        ALGP_TRY(ensure(c, c->miDP, sizeof(T) * mbpad));
        ALGP_TRY(ensure(c, c->miDQ, sizeof(T) * npad));
        ALGP_TRY(ensure(c, c->miU, sizeof(T) * (size_t)MAX_APPEND * mbpad));
        ALGP_TRY(ensure(c, c->miW, sizeof(T) * (size_t)MAX_APPEND * npad));
        ALGP_TRY(ensure(c, c->miCol, sizeof(T) * npad));
        ALGP_TRY(ensure(c, c->miPos, sizeof(int64_t) * n));
        ALGP_TRY(ensure(c, c->miH, sizeof(double) * (3 + 2 * MAX_APPEND)));
        // C_AbarAbar carries no measurement noise (agent.py:331)
        if (mb > 0) {
            int64_t mp;
            ALGP_TRY(build_set_matrix(c, Abar.data(), mb, nullptr, &mp, p(c->miXbar)));
            double ld = 0;
            ALGP_TRY(factor_resident(c, p(c->miXbar), mb, mbpad, p(c->auxInv), SC_AUXLOGDET, SC_AUXINFO, &ld));
            H_bar = (double)mb * ENT_CONST + 0.5 * ld;
            ALGP_TRY(trinv_upper_inplace<T>(c, ALGP_PROF_GEMM_OTHER, p(c->miXbar), mbpad, mbpad, p(c->auxInv)));
            ALGP_TRY(rows_reduce_launch<T>(c, p(c->miXbar), mb, mbpad, mbpad, (const T*)nullptr, p(c->miDP), (T*)nullptr, 0));
        }
        {
            int64_t np2;
            ALGP_TRY(build_set_matrix(c, all.data(), n, vall.data(), &np2, p(c->miXall)));
            double ld = 0;
            ALGP_TRY(factor_resident(c, p(c->miXall), n, npad, p(c->auxInv), SC_AUXLOGDET, SC_AUXINFO, &ld));
            H_all = (double)n * ENT_CONST + 0.5 * ld;
            ALGP_TRY(trinv_upper_inplace<T>(c, ALGP_PROF_GEMM_OTHER, p(c->miXall), npad, npad, p(c->auxInv)));
            ALGP_TRY(rows_reduce_launch<T>(c, p(c->miXall), n, npad, npad, (const T*)nullptr, p(c->miDQ), (T*)nullptr, 0));
        }
        const double Hs[3] = {H_A, H_bar, H_all};
        ALGP_HIP(hipMemcpyAsync(c->miH.p, Hs, sizeof(Hs), hipMemcpyHostToDevice, c->stream));
        ALGP_HIP(hipMemcpyAsync(c->miPos.p, c->mi_posbar.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice, c->stream));
        ALGP_TRY(sync(c));                                       // Hs / mi_posbar (a member, but be plain about it) are host memory
        c->mi_mb = mb;
        c->mi_mbpad = mbpad;
        c->mi_npad = npad;
        c->mi_npicks = (int64_t)c->picks.size();
        c->mi_base = c->mi_npicks;
        c->mi_nbar = 0;
        c->mi_ss = ss;
        c->mi_sm = sm;
        c->mi_valid = true;
        return ALGP_OK;
    }
    // fold pick number q (committed after mi_build) into P, Q and the three entropies: stream-ordered, O(n^2)
    static int mi_apply_pick(algp_ctx* c, int64_t q, double ss, double sm) {
        const PickRec& pk = c->picks[(size_t)q];
        const int r = (int)(q - c->mi_base);                      // its slot in the rank-1 lists
        const double delta = 1.0 / (1.0 / ss + 1.0 / sm) - sm;
        double* Hs = (double*)c->miH.p;
        const LazyPick* lp = (const LazyPick*)c->lazypicks.p + q;
        const int64_t n = c->n_pool, npad = c->mi_npad, mbpad = c->mi_mbpad;
        if (!pk.in_train) {
            // the site leaves the complement: column of P = X X^T at its row, then the rank-1 removal
            const int64_t cb = c->mi_posbar[pk.pool_idx];
            if (cb < 0) return fail(c, ALGP_ERR_STATE, "mutual_information: a picked site is missing from the complement set");
            // column cb of P = X X^T: X's row cb is zero (the buffer holds L there) left of its own diagonal tile
            ALGP_TRY(rows_reduce_launch<T>(c, p(c->miXbar), c->mi_mb, mbpad, mbpad, p(c->miXbar) + cb * mbpad, (T*)nullptr, p(c->miCol),
                                           cb / NB * NB));
            ALGP_TRY(mi_rank1_launch<T>(c, c->mi_mb, p(c->miCol), p(c->miU), mbpad, Hs + 3, c->mi_nbar, cb, 0, 0.0, p(c->miDP), Hs + 1,
                                        (double*)nullptr, lp));
            c->mi_nbar += 1;
        }
        // its noise in C + D_all changes by ss (new site: 0 -> ss) or by v_fused - sm (mobile-sampled site)
        ALGP_TRY(rows_reduce_launch<T>(c, p(c->miXall), n, npad, npad, p(c->miXall) + pk.pool_idx * npad, (T*)nullptr, p(c->miCol),
                                       pk.pool_idx / NB * NB));
        ALGP_TRY(mi_rank1_launch<T>(c, n, p(c->miCol), p(c->miW), npad, Hs + 3 + MAX_APPEND, r, pk.pool_idx, 1, pk.in_train ? delta : ss,
                                    p(c->miDQ), Hs + 2, Hs + 0, lp));
        return ALGP_OK;
    }
    static int mi_scores_enqueue(algp_ctx* c, double ss, double sm, double delta, double* dst) {
        if (!c->mi_valid || c->mi_ss != ss || c->mi_sm != sm || (int64_t)c->picks.size() < c->mi_npicks) {
            c->mi_valid = false;
            ALGP_TRY(mi_build(c, ss, sm));
        }
        for (; c->mi_npicks < (int64_t)c->picks.size(); ++c->mi_npicks) ALGP_TRY(mi_apply_pick(c, c->mi_npicks, ss, sm));
        return mi_score_launch<T>(c, c->M, (const int*)c->ckind.p, (const int64_t*)c->Cidx.p, (const unsigned char*)c->alive.p,
                                  (const T*)c->dstat.p, ss, delta, (const int64_t*)c->miPos.p, (const T*)c->miDP.p,
                                  (const T*)c->miDQ.p, (const double*)c->miH.p, dst);
    }

    // utilities of every row into `dst` (device; null = c->scores), stream-ordered; the entropy criterion never
    // synchronises here, the MI criterion only when it (re)builds its pool-wide inverses (first scoring after a solve)
    static int scores_enqueue(algp_ctx* c, int criterion, double static_std, double mobile_std, double* dst) {
        if (!c->solved) return fail(c, ALGP_ERR_STATE, "scores: call algp_solve_candidates first");
        if (!c->prior_noise) return fail(c, ALGP_ERR_STATE, "scores: candidates were set with predictive semantics");
        ALGP_TRY(flush_lazy(c));
        const double ss = static_std * static_std, sm = mobile_std * mobile_std;
        const double delta = 1.0 / (1.0 / ss + 1.0 / sm) - sm;
        if (!dst) dst = (double*)c->scores.p;
        if (criterion == ALGP_CRIT_MUTUAL_INFORMATION) {
            ALGP_TRY(mi_scores_enqueue(c, ss, sm, delta, dst));
        } else if (criterion == ALGP_CRIT_ENTROPY) {
            ALGP_TRY(score_launch<T>(c, c->M, (const int*)c->ckind.p, (const unsigned char*)c->alive.p, (const T*)c->dstat.p,
                                     ss, delta, (const double*)nullptr, dst));
        } else {
            return fail(c, ALGP_ERR_BAD_ARG, "unknown criterion");
        }
        // entropy utilities of up-to-date rows: from here on c->scores can serve as upper bounds (lazy greedy)
        c->bounds_valid = criterion == ALGP_CRIT_ENTROPY;
        c->lazy_ss = ss;
        c->lazy_delta = delta;
        if (dst != (double*)c->scores.p)
            ALGP_HIP(hipMemcpyAsync(c->scores.p, dst, sizeof(double) * c->M, hipMemcpyDeviceToDevice, c->stream));
        return ALGP_OK;
    }
    static int scores(algp_ctx* c, int criterion, double static_std, double mobile_std, void* out, int out_is_device) {
        ALGP_TRY(scores_enqueue(c, criterion, static_std, mobile_std, out_is_device ? (double*)out : nullptr));
        if (!out_is_device && out)
            ALGP_HIP(hipMemcpyAsync(out, c->scores.p, sizeof(double) * c->M, hipMemcpyDeviceToHost, c->stream));
        return sync(c);
    }

    static int argmax(algp_ctx* c, int64_t* local_pos, int64_t* pool_idx, double* value) {
        if (!c->solved) return fail(c, ALGP_ERR_STATE, "argmax: no scores");
        if (c->M == 0) return fail(c, ALGP_ERR_BAD_ARG, "argmax: empty candidate set");
        double* sc = (double*)c->scal.p;
        ALGP_TRY(argmax_launch(c, (const double*)c->scores.p, c->M, sc + SC_AMAXV, (int64_t*)(sc + SC_AMAXI)));
        double host[2];
        ALGP_HIP(hipMemcpyAsync(host, sc + SC_AMAXV, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        ALGP_TRY(sync(c));
        int64_t pos;
        memcpy(&pos, &host[1], sizeof(int64_t));
        if (local_pos) *local_pos = pos;
        if (pool_idx) *pool_idx = pos >= 0 ? c->cand_idx[pos] : -1;
        if (value) *value = host[0];
        return ALGP_OK;
    }

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
    static RemoteSlots remote_slots(algp_ctx* c) {
        char* b = (char*)c->remote.p;
        RemoteSlots r;
        r.cidx = (int64_t*)(b + 0);
        r.ckind = (int*)(b + 64);
        r.ss = (T*)(b + 128);
        r.dot = (T*)(b + 192);
        r.dstat = (T*)(b + 256);
        r.mu = (T*)(b + 320);
        r.alive = (unsigned char*)(b + 384);
        r.fresh = (int*)(b + 448);
        r.score = (double*)(b + 512);
        return r;
    }
    static int remote_row(algp_ctx* c, int64_t pool_idx, int in_train) {
        const int64_t N = c->N, Npad = c->Npad, ldv = c->ldv;
        T* l = p(c->lrow);
        ALGP_TRY(ensure(c, c->remote, 640));
        RemoteSlots r = remote_slots(c);
        ALGP_HIP(hipMemsetAsync(l, 0, sizeof(T) * ldv, c->stream));
        ALGP_HIP(hipMemsetAsync(c->remote.p, 0, 640, c->stream));
        const int unit_host = in_train ? (int)c->pos_in_train[pool_idx] : -1;
        ALGP_HIP(hipMemcpyAsync(r.cidx, &pool_idx, sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        ALGP_HIP(hipMemcpyAsync(r.ckind, &unit_host, sizeof(int), hipMemcpyHostToDevice, c->stream));
        KmatSrc s = make_src(c);
        ALGP_TRY(kmat_launch<T>(c, s, r.cidx, 1, 1, (const int64_t*)c->Aidx.p, N, Npad, nullptr, 0, r.ckind, 0, l, ldv));
        ALGP_TRY(trsv_forward<T>(c, p(c->L), Npad, c->Lld, p(c->invD), l));
        ALGP_TRY(rows_reduce_launch<T>(c, l, 1, ldv, Npad, (const T*)nullptr, r.ss, (T*)nullptr));
        const T prior = (T)(c->hyp.outputscale + (c->prior_noise ? c->hyp.noise : 0.0));
        ALGP_TRY(cand_finalize_launch<T>(c, 1, r.ckind, r.cidx, c->pool_is_cov ? (const T*)c->Cp.p : nullptr, c->n_pool, prior,
                                         (const T*)nullptr, r.ss, r.dot, (T)0, r.dstat, r.mu, r.alive));
        if (!c->picks.empty())
            ALGP_TRY(lazy_refresh_launch<T>(c, 1, 2, 0, (const LazyPick*)c->lazypicks.p, (int)c->picks.size(), r.ckind, r.cidx,
                                            (const T*)c->Xs.p, c->pool_is_cov ? (const T*)c->Cp.p : nullptr, c->n_pool,
                                            c->hyp.DP, c->hyp.kernel, (T)c->hyp.outputscale, (T)c->hyp.noise, p(c->prevrows),
                                            ldv, l, r.dstat, r.fresh, r.alive, r.score, c->lazy_ss, c->lazy_delta));
        return ALGP_OK;
    }

    // Make `pool_idx` static-sampled.  Only the pick is recorded (its row of V^T, its scale); the other rows
    // of V^T / dstat catch up on demand (lazy_refresh_kernel) -- before anything reads the full state
    // (flush_lazy) or, while the next pick is resolved, only the rows that can still win.
    // commit_enqueue: everything stream-ordered, nothing read back (the winner's statistic d_c and the scale of the
    // appended row stay on the device, in scal[SC_COMMIT..]); the local / remote decision is the host's, from the pool
    // index it already holds.
    // winner_payload (device, or null): the owner's contribution to the pick's all-gather (comm.hip) -- for a winner another
    // rank owns, its statistic and its row of V^T are copied from there instead of being rebuilt from the factor.
    static int commit_enqueue(algp_ctx* c, int64_t pool_idx, double ss, double delta, const char* winner_payload = nullptr) {
        if (c->debug_fail_next_commit) {
            const int code = c->debug_fail_next_commit;
            c->debug_fail_next_commit = 0;
            return fail(c, code, "commit_pick: failure injected by algp_debug_fail_at");
        }
        if (!c->solved) return fail(c, ALGP_ERR_STATE, "commit_pick: call algp_solve_candidates first");
        if (pool_idx < 0 || pool_idx >= c->n_pool) return fail(c, ALGP_ERR_BAD_ARG, "commit_pick: index outside the pool");
        if ((int64_t)c->picks.size() >= MAX_APPEND) return fail(c, ALGP_ERR_STATE, "commit_pick: append capacity exhausted; re-factorize");
        for (auto& pk : c->picks)
            if (pk.pool_idx == pool_idx) return fail(c, ALGP_ERR_BAD_ARG, "commit_pick: site already static-sampled");
        const int in_train = c->pos_in_train[pool_idx] >= 0 ? 1 : 0;
        const int64_t local = c->cand_pos[pool_idx];
        const int64_t ldv = c->ldv, ncols = c->ncols;
        const size_t q = c->picks.size();
        const T* dsrc;
        if (local >= 0) {
            if (c->lazy_stale) ALGP_TRY(lazy_launch(c, 0, local, ss, delta));     // the winner's own row must be current
            ALGP_HIP(hipMemsetAsync(c->lrow.p, 0, sizeof(T) * ldv, c->stream));
            ALGP_HIP(hipMemcpyAsync(c->lrow.p, p(c->Vt) + local * ldv, sizeof(T) * ncols, hipMemcpyDeviceToDevice, c->stream));
            dsrc = p(c->dstat) + local;
        } else if (winner_payload) {
            // the owner's row, bit for bit (it was current when it was packed: a stale best row asks for another round)
            ALGP_HIP(hipMemsetAsync(c->lrow.p, 0, sizeof(T) * ldv, c->stream));
            ALGP_HIP(hipMemcpyAsync(c->lrow.p, winner_payload + 32, sizeof(T) * ncols, hipMemcpyDeviceToDevice, c->stream));
            dsrc = (const T*)(winner_payload + 24);
        } else {
            ALGP_TRY(remote_row(c, pool_idx, in_train));
            dsrc = remote_slots(c).dstat;
        }
        double* sc = (double*)c->scal.p;
        ALGP_HIP(hipMemcpyAsync(p(c->prevrows) + (int64_t)q * ldv, c->lrow.p, sizeof(T) * ldv, hipMemcpyDeviceToDevice, c->stream));
        ALGP_TRY(commit_finalize_launch<T>(c, dsrc, in_train, ss, delta, (LazyPick*)c->lazypicks.p + q, pool_idx, ncols,
                                           local >= 0 ? (unsigned char*)c->alive.p + local : nullptr,
                                           local >= 0 ? (double*)c->scores.p + local : nullptr, sc + SC_COMMIT));
        PickRec pr;
        pr.pool_idx = pool_idx;
        pr.in_train = in_train;
        c->picks.push_back(pr);
        c->ncols = ncols + 1;
        c->lazy_stale = true;
        return ALGP_OK;
    }
    // the ABI's algp_commit_pick: any pool index the caller names, so the scale is read back and checked (a pick the
    // library resolved itself has a finite utility, which already implies a positive variance under the square root)
    static int commit_pick(algp_ctx* c, int64_t pool_idx, double static_std, double mobile_std) {
        const double ss = static_std * static_std, sm = mobile_std * mobile_std;
        const double delta = 1.0 / (1.0 / ss + 1.0 / sm) - sm;
        const bool was_stale = c->lazy_stale;
        ALGP_TRY(commit_enqueue(c, pool_idx, ss, delta));
        double host[2];
        ALGP_HIP(hipMemcpyAsync(host, (double*)c->scal.p + SC_COMMIT, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        const int src = sync_checked(c, "commit_pick");          // a remote row's forward substitution may have given up
        if (src != ALGP_OK) {
            c->picks.pop_back();
            c->ncols -= 1;
            c->lazy_stale = was_stale;
            c->bounds_valid = false;
            return src;
        }
        const double scale = host[1];
        if (!(scale == scale) || isinf(scale)) {
            c->picks.pop_back();                                  // the rows never see the pick: its record is not counted
            c->ncols -= 1;
            c->lazy_stale = was_stale;
            return fail(c, ALGP_ERR_NOT_PD, "commit_pick: posterior variance of the pick is not positive");
        }
        return ALGP_OK;
    }

    // ---- lazy greedy (entropy criterion, picks only): see lazy_refresh_kernel in vecops.hip ----
    static int lazy_launch(algp_ctx* c, int mode, int64_t pos, double ss, double delta, const int64_t* pos_dev = nullptr) {
        return lazy_refresh_launch<T>(c, c->M, mode, pos, (const LazyPick*)c->lazypicks.p, (int)c->picks.size(),
                                      (const int*)c->ckind.p, (const int64_t*)c->Cidx.p, (const T*)c->Xs.p,
                                      c->pool_is_cov ? (const T*)c->Cp.p : nullptr, c->n_pool, c->hyp.DP, c->hyp.kernel,
                                      (T)c->hyp.outputscale, (T)c->hyp.noise, p(c->prevrows), c->ldv, p(c->Vt),
                                      p(c->dstat), (int*)c->fresh.p, (const unsigned char*)c->alive.p,
                                      (double*)c->scores.p, ss, delta, pos_dev);
    }
    // after a candidate solve: no picks, every row current, no bounds
    static int reset_lazy(algp_ctx* c) {
        ALGP_TRY(ensure(c, c->fresh, sizeof(int) * std::max<int64_t>(c->Mpad, 1)));
        ALGP_TRY(ensure(c, c->lazypicks, sizeof(LazyPick) * MAX_APPEND));
        ALGP_HIP(hipMemsetAsync(c->fresh.p, 0, sizeof(int) * std::max<int64_t>(c->Mpad, 1), c->stream));
        c->lazy_stale = false;
        c->bounds_valid = false;
        return ALGP_OK;
    }
    // bring every row of V^T / dstat up to date with the committed picks (stream-ordered, no host sync)
    static int flush_lazy(algp_ctx* c) {
        if (!c->lazy_stale) return ALGP_OK;
        ALGP_TRY(lazy_launch(c, 2, 0, c->lazy_ss, c->lazy_delta));
        c->lazy_stale = false;
        return ALGP_OK;
    }

    // The best local candidate under the current state, left ON THE DEVICE (scal[SC_AMAXV], scal[SC_AMAXI]) by one
    // stream-ordered chain with no host decision inside.  Entropy criterion: c->scores holds, per row, the utility as
    // of the picks applied to that row -- an upper bound of the current one (submodularity).  argmax -> refresh of that
    // row (its now-exact utility is the threshold) -> refresh of every stale row whose bound reaches the threshold ->
    // argmax: every row that is still stale now scores below a fresh one, so the second argmax is a fresh row and the
    // true first maximum.  (Only a NaN utility breaks that argument; the status word of the pick then asks for one more
    // round.)  The kernels take the row from the device, and a refresh of an up-to-date row is a no-op.
    static int enqueue_local_best(algp_ctx* c, double ss, double delta) {
        double* sc = (double*)c->scal.p;
        int64_t* pos_dev = (int64_t*)(sc + SC_AMAXI);
        if (c->lazy_stale) {
            ALGP_TRY(argmax_launch(c, (const double*)c->scores.p, c->M, sc + SC_AMAXV, pos_dev));
            ALGP_TRY(lazy_launch(c, 0, 0, ss, delta, pos_dev));
            ALGP_TRY(lazy_launch(c, 1, 0, ss, delta, pos_dev));
        }
        return argmax_launch(c, (const double*)c->scores.p, c->M, sc + SC_AMAXV, pos_dev);
    }
    // c->scores must hold bounds for (ss, delta): otherwise (first pick after a solve, MI criterion, lazy greedy
    // switched off) every row is scored, which also brings every row up to date
    static int ensure_bounds(algp_ctx* c, int criterion, double static_std, double mobile_std, double ss, double delta) {
        static const bool lazy_on = env_switch("ALGP_LAZY_GREEDY", true);
        if (criterion != ALGP_CRIT_ENTROPY || !lazy_on || !c->bounds_valid || c->lazy_ss != ss || c->lazy_delta != delta)
            return scores_enqueue(c, criterion, static_std, mobile_std, nullptr);
        return ALGP_OK;
    }

    // algp_best_candidate: the local first maximum, one read-back (value, position, how many picks its row has seen)
    static int best_candidate(algp_ctx* c, int criterion, double static_std, double mobile_std, int64_t* local_pos,
                              int64_t* pool_idx, double* value) {
        if (!c->solved) return fail(c, ALGP_ERR_STATE, "best_candidate: call algp_solve_candidates first");
        if (c->M == 0) return fail(c, ALGP_ERR_BAD_ARG, "best_candidate: empty candidate set");
        const double ss = static_std * static_std, sm = mobile_std * mobile_std;
        const double delta = 1.0 / (1.0 / ss + 1.0 / sm) - sm;
        ALGP_TRY(ensure_bounds(c, criterion, static_std, mobile_std, ss, delta));
        double* sc = (double*)c->scal.p;
        int64_t pos = -1;
        double val = -INFINITY;
        for (int round = 0; round < 8; ++round) {
            ALGP_TRY(enqueue_local_best(c, ss, delta));
            ALGP_TRY(fresh_at_launch(c, (const int*)c->fresh.p, (const int64_t*)(sc + SC_AMAXI), sc + SC_AMAXF));
            double host[3];
            ALGP_HIP(hipMemcpyAsync(host, sc + SC_AMAXV, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            ALGP_HIP(hipMemcpyAsync(host + 2, sc + SC_AMAXF, sizeof(double), hipMemcpyDeviceToHost, c->stream));
            ALGP_TRY(sync(c));
            memcpy(&pos, &host[1], sizeof(int64_t));
            val = host[0];
            if (pos < 0 || !c->lazy_stale) break;                 // all NaN, or nothing committed since the full scoring
            if ((int)host[2] >= (int)c->picks.size()) break;      // the maximum is an up-to-date row: it wins
        }
        if (local_pos) *local_pos = pos;
        if (pool_idx) *pool_idx = pos >= 0 ? c->cand_idx[pos] : -1;
        if (value) *value = val;
        return ALGP_OK;
    }

    // k picks of the entropy criterion, on one rank or over the candidate shards of several (agent.py:313-354 with the
    // loop over candidates cut into shards): per pick ONE host round trip -- the 40-byte record (utility, pool index,
    // owner, status, failing rank) that comm_pick_exchange reads back after [local best on the device -> pack ->
    // all-gather of the triples -> first maximum in rank order].  The commit of the winner is enqueued behind it and
    // not waited for (the next pick's kernels, or whatever the caller does next, are stream-ordered after it).
    // Nothing rank-local returns before the exchange: a failure becomes this rank's status word, every rank sees it in
    // the same gather and every rank returns it -- nobody is left waiting in a collective.
    static int greedy_picks(algp_ctx* c, double static_std, double mobile_std, int k, int64_t* picks_out, double* ut_out) {
        const double ss = static_std * static_std, sm = mobile_std * mobile_std;
        const double delta = 1.0 / (1.0 / ss + 1.0 / sm) - sm;
        double* sc = (double*)c->scal.p;
        for (int pck = 0; pck < k; ++pck) {
            double rec[5];
            const char* winner = nullptr;
            for (int round = 0;; ++round) {
                int st = ALGP_OK;
                if (c->debug_fail_next_pick) {
                    st = fail(c, c->debug_fail_next_pick, "greedy: failure injected by algp_debug_fail_next_pick");
                    c->debug_fail_next_pick = 0;
                } else if (!c->solved) {
                    st = fail(c, ALGP_ERR_STATE, "greedy: call algp_solve_candidates first");
                } else if (!c->prior_noise) {
                    st = fail(c, ALGP_ERR_STATE, "greedy: candidates were set with predictive semantics");
                }
                if (st == ALGP_OK && c->pending_pick_error) {
                    // the commit of an earlier winner failed on this rank after the exchange that chose it: reported here,
                    // in the next gather this rank takes part in, so that every rank returns it from the same call
                    st = fail(c, c->pending_pick_error, c->pending_pick_msg);
                    c->pending_pick_error = 0;
                }
                if (st == ALGP_OK && c->M > 0) st = ensure_bounds(c, ALGP_CRIT_ENTROPY, static_std, mobile_std, ss, delta);
                if (st == ALGP_OK && c->M > 0) st = enqueue_local_best(c, ss, delta);
                const bool have = st == ALGP_OK && c->M > 0;             // an empty shard offers nothing; that is not an error
                const std::string local_err = c->err;
                ALGP_TRY(comm_pick_exchange(c, have ? sc + SC_AMAXV : nullptr, have ? (const int64_t*)(sc + SC_AMAXI) : nullptr,
                                            (const int64_t*)c->Cidx.p, c->lazy_stale ? (const int*)c->fresh.p : nullptr,
                                            (int)c->picks.size(), st, rec, &winner));
                if (rec[3] >= 2.0) {
                    const int code = (int)rec[3];
                    if (st != ALGP_OK) return fail(c, st, local_err);
                    if ((int)rec[4] == c->comm_rank || !(c->comm || c->host_gather)) {
                        // this rank's own status word, raised on the device: the sticky stall word (sync_checked clears it)
                        const int rc2 = sync_checked(c, "greedy");
                        if (rc2 != ALGP_OK) return rc2;
                    }
                    return fail(c, code, "greedy_sharded: rank " + std::to_string((int)rec[4]) + " failed with error " +
                                             std::to_string(code) + " while resolving its best candidate; no rank committed pick " +
                                             std::to_string(pck));
                }
                if (rec[3] == 0.0) break;
                if (round == 8) return fail(c, ALGP_ERR_STATE, "greedy: the best candidate could not be resolved (NaN utilities)");
            }
            if (rec[1] < 0) return fail(c, ALGP_ERR_STATE, "greedy: no candidate left on any rank");
            if (!(rec[0] > -INFINITY))
                return fail(c, ALGP_ERR_STATE, "greedy: every remaining candidate is already static-sampled (a further pick would "
                                               "re-sample a static site)");
            const int64_t pool_idx = (int64_t)rec[1];
            if (picks_out) picks_out[pck] = pool_idx;
            if (ut_out) ut_out[pck] = rec[0];
            const int crc = commit_enqueue(c, pool_idx, ss, delta, winner);
            if (crc != ALGP_OK) {
                // after the exchange: the other ranks have committed.  With a collective still ahead in this call the failure
                // travels in the next pick's status word (every rank then returns it); after the last pick it is returned here
                // AND kept for the first gather of this rank's next call.
                if (!(c->comm || c->host_gather)) return crc;             // one rank: nobody else to tell
                c->pending_pick_error = crc;
                c->pending_pick_msg = "greedy: committing pick " + std::to_string(pck) + " failed on this rank: " + c->err;
                if (pck + 1 == k) return crc;
            }
        }
        return ALGP_OK;
    }

    static int greedy(algp_ctx* c, int criterion, double static_std, double mobile_std, int k, const int64_t* forced,
                      int64_t* picks_out, double* ut_out) {
        if (!ut_out && !forced && criterion == ALGP_CRIT_ENTROPY && !c->comm && !c->host_gather)
            return greedy_picks(c, static_std, mobile_std, k, picks_out, nullptr);
        for (int pck = 0; pck < k; ++pck) {
            int64_t pool_idx;
            if (ut_out || forced) {
                ALGP_TRY(scores(c, criterion, static_std, mobile_std, ut_out ? ut_out + (int64_t)pck * c->M : nullptr, 0));
                if (forced) pool_idx = forced[pck];
                else ALGP_TRY(argmax(c, nullptr, &pool_idx, nullptr));
            } else {
                ALGP_TRY(best_candidate(c, criterion, static_std, mobile_std, nullptr, &pool_idx, nullptr));
            }
            if (picks_out) picks_out[pck] = pool_idx;
            ALGP_TRY(commit_pick(c, pool_idx, static_std, mobile_std));
        }
        return ALGP_OK;
    }

    // Paths of 65 .. 256 distinct sites (cpos / lpos: candidate row and train row per site, packed to the front of each path's
    // maxlen entries).  Per batch of paths: the paths' rows of V^T gathered into a scratch (a site that is a train row
    // already: L[lpos, :] - var_lpos * its unit row, as in the LDS kernel), the Gram matrices as ONE batched lower-tile MFMA
    // product, G = C_PP + sigma_m^2 I - Gram, then the ppad x ppad blocks factored as 2 x 2 tiles of 128: diagonal-block
    // kernel, L21 = G21 inv(L11)^T, G22 -= L21 L21^T, diagonal-block kernel -- every step one launch for the whole batch.
    static int score_paths_big(algp_ctx* c, const std::vector<int64_t>& cpos, const std::vector<int64_t>& lpos, int npaths, int maxlen,
                               int maxused, double mobile_std, double* dH) {
        const int64_t Npad = c->Npad;
        const int ppad = maxused <= NB ? NB : 2 * NB;
        // rows scratch: batch * ppad * Npad elements, at most ~4 GB
        const int64_t per_path = (int64_t)ppad * Npad * (int64_t)sizeof(T);
        const int bmax = (int)std::max<int64_t>(1, std::min<int64_t>(npaths, (int64_t)4e9 / per_path));
        ALGP_TRY(ensure(c, c->auxW, (size_t)bmax * per_path));
        ALGP_TRY(ensure(c, c->auxA, sizeof(T) * (size_t)bmax * ppad * ppad));
        ALGP_TRY(ensure(c, c->auxInv, sizeof(T) * (size_t)bmax * 2 * NB * NB));
        ALGP_TRY(ensure(c, c->auxD, sizeof(T) * (size_t)bmax * NB * NB));
        ALGP_TRY(ensure(c, c->auxIdx, sizeof(int64_t) * 2 * (size_t)bmax * ppad));
        ALGP_TRY(ensure(c, c->auxVar, sizeof(T) * (size_t)bmax * ppad + 256));
        ALGP_TRY(ensure(c, c->hostStage, sizeof(double) * (size_t)(npaths + bmax) + sizeof(int) * (size_t)bmax + 64));
        double* d_out = (double*)c->hostStage.p;
        double* d_ld = d_out + npaths;
        int* d_info = (int*)(d_ld + bmax);
        T* rows = p(c->auxW);
        T* G = p(c->auxA);
        T* inv = p(c->auxInv);
        T* L21 = p(c->auxD);
        int64_t* d_src = (int64_t*)c->auxIdx.p;
        int64_t* d_lrow = d_src + (size_t)bmax * ppad;
        std::vector<int64_t> src((size_t)bmax * ppad), lr((size_t)bmax * ppad);
        std::vector<T> lsc((size_t)bmax * ppad);
        for (int p0 = 0; p0 < npaths; p0 += bmax) {
            const int B = std::min(bmax, npaths - p0);
            bool second = false;
            for (int b = 0; b < B; ++b)
                for (int a = 0; a < ppad; ++a) {
                    const size_t e = (size_t)b * ppad + a;
                    const int64_t cp = a < maxlen ? cpos[(size_t)(p0 + b) * maxlen + a] : -1;
                    const int64_t lp = a < maxlen ? lpos[(size_t)(p0 + b) * maxlen + a] : -1;
                    src[e] = cp;
                    lr[e] = cp >= 0 ? lp : -1;
                    lsc[e] = (cp >= 0 && lp >= 0) ? (T)c->train_var_host[(size_t)lp] : (T)0;
                    second |= cp >= 0 && lp >= 0;
                }
            const size_t nrow = (size_t)B * ppad;
            ALGP_HIP(hipMemcpyAsync(d_src, src.data(), sizeof(int64_t) * nrow, hipMemcpyHostToDevice, c->stream));
            ALGP_HIP(hipMemcpyAsync(d_lrow, lr.data(), sizeof(int64_t) * nrow, hipMemcpyHostToDevice, c->stream));
            ALGP_HIP(hipMemcpyAsync(c->auxVar.p, lsc.data(), sizeof(T) * nrow, hipMemcpyHostToDevice, c->stream));
            ALGP_HIP(hipMemsetAsync(d_ld, 0, sizeof(double) * B, c->stream));
            ALGP_HIP(hipMemsetAsync(d_info, 0, sizeof(int) * B, c->stream));
            ALGP_TRY(gather_rows_launch<T>(c, p(c->Vt), c->ldv, d_src, rows, Npad, (int64_t)nrow, Npad, second ? d_lrow : nullptr,
                                           second ? (const T*)c->auxVar.p : nullptr, p(c->L), c->Lld));
            ALGP_TRY(gemm_nt_launch_batched<T>(c, ALGP_PROF_GEMM_OTHER, ppad, ppad, Npad, (T)1, rows, Npad, (int64_t)ppad * Npad, rows, Npad,
                                               (int64_t)ppad * Npad, (T)0, nullptr, 0, 0, G, ppad, (int64_t)ppad * ppad, 1, B));
            ALGP_TRY(path_assemble_launch<T>(c, d_src, B, ppad, (const int64_t*)c->Cidx.p, (const T*)c->Xs.p,
                                             c->pool_is_cov ? (const T*)c->Cp.p : nullptr, c->n_pool, c->hyp.DP, c->hyp.kernel,
                                             c->hyp.outputscale, c->hyp.noise, mobile_std * mobile_std, G));
            ALGP_TRY(potrf_diag_batched_launch<T>(c, G, (int64_t)ppad * ppad, ppad, inv, 2 * NB * NB, d_ld, d_info, B));
            if (ppad > NB) {
                T* G21 = G + (int64_t)NB * ppad;
                T* G22 = G21 + NB;
                ALGP_TRY(gemm_nt_launch_batched<T>(c, ALGP_PROF_GEMM_OTHER, NB, NB, NB, (T)1, G21, ppad, (int64_t)ppad * ppad, inv, NB, 2 * NB * NB,
                                                   (T)0, nullptr, 0, 0, L21, NB, NB * NB, 0, B));
                ALGP_TRY(gemm_nt_launch_batched<T>(c, ALGP_PROF_GEMM_OTHER, NB, NB, NB, (T)-1, L21, NB, NB * NB, L21, NB, NB * NB, (T)1, G22, ppad,
                                                   (int64_t)ppad * ppad, G22, ppad, (int64_t)ppad * ppad, 0, B));
                ALGP_TRY(potrf_diag_batched_launch<T>(c, G22, (int64_t)ppad * ppad, ppad, inv + NB * NB, 2 * NB * NB, d_ld, d_info, B));
            }
            ALGP_TRY(path_finish_launch(c, d_src, ppad, B, d_ld, d_info, d_out + p0));
            ALGP_TRY(sync(c));                                   // the index vectors are reused by the next batch
        }
        ALGP_HIP(hipMemcpyAsync(dH, d_out, sizeof(double) * npaths, hipMemcpyDeviceToHost, c->stream));
        return sync(c);
    }

    // a8 / f3: the entropy gain of every enumerated path (agent.py:374-400 computes one slogdet per path) from ONE
    // resident factor and candidate solve: sites[p][a] are pool indices (-1 = none); a site that already is a train
    // row receives a second (mobile) row, a new site a first one; dH[p] = H(A u path_p) - H(A)
    static int score_paths(algp_ctx* c, const int64_t* sites, int npaths, int maxlen, double mobile_std, double* dH) {
        if (!c->solved) return fail(c, ALGP_ERR_STATE, "score_paths: call algp_solve_candidates first");
        if (!c->prior_noise) return fail(c, ALGP_ERR_STATE, "score_paths: candidates were set with predictive semantics");
        if (!c->picks.empty()) return fail(c, ALGP_ERR_STATE, "score_paths: picks were committed since the candidate solve; solve again");
        const size_t tot = (size_t)npaths * maxlen;
        std::vector<int64_t> cpos(tot, -1), lpos(tot, -1);
        int maxused = 0;
        for (int pth = 0; pth < npaths; ++pth) {
            int used = 0;
            for (int a = 0; a < maxlen; ++a) {
                const int64_t j = sites[(size_t)pth * maxlen + a];
                if (j < 0) continue;
                if (j >= c->n_pool) return fail(c, ALGP_ERR_BAD_ARG, "score_paths: index outside the pool");
                const int64_t cp = c->cand_pos[j];
                if (cp < 0) return fail(c, ALGP_ERR_BAD_ARG, "score_paths: site " + std::to_string(j) + " is not a resident candidate");
                bool dup = false;
                for (int b = 0; b < used; ++b) dup |= cpos[(size_t)pth * maxlen + b] == cp;
                if (dup) continue;                                  // a site crossed twice is measured once (mobile mask)
                cpos[(size_t)pth * maxlen + used] = cp;
                lpos[(size_t)pth * maxlen + used] = c->pos_in_train[j];
                ++used;
            }
            if (used > 256) return fail(c, ALGP_ERR_BAD_ARG, "score_paths: more than 256 distinct sites in a path");
            maxused = std::max(maxused, used);
        }
        if (maxused > 64) return score_paths_big(c, cpos, lpos, npaths, maxlen, maxused, mobile_std, dH);
        ALGP_TRY(ensure(c, c->auxIdx, sizeof(int64_t) * 2 * tot));
        ALGP_TRY(ensure(c, c->hostStage, sizeof(double) * std::max<size_t>(npaths, 1)));
        int64_t* d_c = (int64_t*)c->auxIdx.p;
        int64_t* d_l = d_c + tot;
        ALGP_HIP(hipMemcpyAsync(d_c, cpos.data(), sizeof(int64_t) * tot, hipMemcpyHostToDevice, c->stream));
        ALGP_HIP(hipMemcpyAsync(d_l, lpos.data(), sizeof(int64_t) * tot, hipMemcpyHostToDevice, c->stream));
        ALGP_TRY(path_score_launch<T>(c, d_c, d_l, npaths, maxlen, (const int64_t*)c->Cidx.p, p(c->Vt), c->ldv, c->ncols, p(c->L),
                                      c->Lld, (const T*)c->varA.p, (const T*)c->Xs.p, c->pool_is_cov ? (const T*)c->Cp.p : nullptr,
                                      c->n_pool, c->hyp.DP, c->hyp.kernel, c->hyp.outputscale, c->hyp.noise,
                                      mobile_std * mobile_std, (double*)c->hostStage.p));
        ALGP_HIP(hipMemcpyAsync(dH, c->hostStage.p, sizeof(double) * npaths, hipMemcpyDeviceToHost, c->stream));
        return sync(c);
    }

    // have_X: c->auxW already holds X = L^-T (it rode along with the factorisation as an identity panel); inv_enqueued: and
    // S^-1 = X X^T is already running on the helper stream (event 21 marks its end)
    static int mll_grad(algp_ctx* c, double* grad_out, bool have_X = false, bool inv_enqueued = false) {
        if (!c->factored || c->train_dirty) return fail(c, ALGP_ERR_STATE, "get_mll_grad: call algp_factorize first");
        if (c->pool_is_cov) return fail(c, ALGP_ERR_BAD_ARG, "get_mll_grad needs a coordinate pool");
        const int64_t N = c->N, Npad = c->Npad;
        const int D = c->hyp.D, DP = c->hyp.DP;
        if (have_X && !c->alpha_valid) {
            // alpha = L^-T z = X z with the X the launch left in auxW: one pass over its upper triangle (0.4 GB at N = 10 000)
            // instead of the backward substitution's chain of 79 hand-offs
            ALGP_TRY(upper_gemv_launch<T>(c, p(c->auxW), Npad, Npad, (const T*)c->z.p, p(c->alpha)));
            c->alpha_valid = true;
        }
        ALGP_TRY(need_alpha(c));
        ALGP_TRY(ensure(c, c->auxA, sizeof(T) * Npad * Npad));
        // X = I L^-T = L^-T ;  S^-1 = X X^T (lower tiles)
        if (!have_X) {
            ALGP_TRY(ensure(c, c->auxW, sizeof(T) * Npad * Npad));
            ALGP_TRY(set_identity_launch<T>(c, p(c->auxW), Npad, Npad));
            ALGP_TRY(trinv_upper<T>(c, ALGP_PROF_GEMM_OTHER, p(c->auxW), Npad, Npad, p(c->L), c->Lld, p(c->invD)));
        }
        if (inv_enqueued) ALGP_HIP(hipStreamWaitEvent(c->stream, sync_event_api(c, 21), 0));
        else ALGP_TRY(syrk_upper<T>(c, ALGP_PROF_GEMM_OTHER, p(c->auxW), Npad, Npad, p(c->auxA), Npad));
        double* sc = (double*)c->scal.p + SC_GRAD;        // slots 16..27: os, trace, ls[0..8)
        ALGP_HIP(hipMemsetAsync(sc, 0, sizeof(double) * 12, c->stream));
        ALGP_TRY(mll_grad_launch<T>(c, p(c->auxA), Npad, N, (const T*)c->Xs.p, DP, (const int64_t*)c->Aidx.p,
                                    (const T*)c->alpha.p, c->hyp.kernel, (T)c->hyp.outputscale, sc,
                                    (double*)c->auxW.p /* X = L^-T is spent: room for the per-workgroup partials */));
        double h[12];
        ALGP_HIP(hipMemcpyAsync(h, sc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        {
            const int rc = sync_checked(c, "get_mll_grad");
            if (rc != ALGP_OK) { c->alpha_valid = false; return rc; }
        }
        for (int d = 0; d < D; ++d) grad_out[d] = 0.5 * h[2 + d];
        grad_out[D] = 0.5 * h[0];
        grad_out[D + 1] = 0.5 * c->hyp.noise * h[1];
        return ALGP_OK;
    }

    // f2: the device work of ONE iteration of GPR.fit (models.py:145-158: loss = -mll(model(train_x), train_y); backward) in
    // one ABI call: S, its factor AND X = L^-T out of the same task-list launch (the identity rides along as a panel
    // whose zero tiles are never touched), alpha by the two one-launch substitutions, S^-1 = X X^T as one
    // triangular-aware launch, the pairwise gradient reduction.  Same values as algp_factorize + algp_get_mll +
    // algp_get_mll_grad (tested); N^3 flop in all (N^3/3 each for the factor, the inverse of the factor and the product).
    static int fit_step(algp_ctx* c, double* mll_out, double* grad_out) {
        if (c->pool_is_cov) return fail(c, ALGP_ERR_BAD_ARG, "fit_step needs a coordinate pool");
        const int64_t Npad = c->Npad;
        bool have_X = false, inv_enq = false;
        // Round 5: z rides along too (a dense tile row behind the identity that carries y - ybar), and alpha = X z is one pass
        // over the X the launch leaves -- no substitution chain runs beside S^-1 = X X^T any more (the two took 5.5 ms there,
        // starved by the GEMM).  (X X^T as tasks of the same launch as well was built
        // and measured in round 5 -- the launch grew by what the separate 5.1-ms GEMM launch costs, 12.9 -> 18.9 ms at N = 10 000
        // fp64: the list leaves nothing idle to fill -- and removed again: EXPERIMENTS.md.)
        const int64_t prow = grad_out ? Npad + NB : Npad;
        if (c->N > 0 && panel_fits(Npad, prow)) {
            ALGP_TRY(ensure(c, c->auxW, sizeof(T) * prow * Npad));
            ALGP_TRY(ensure(c, c->auxA, sizeof(T) * Npad * Npad));
            ALGP_TRY(set_identity_launch<T>(c, p(c->auxW), Npad, Npad));
            Panel pn{p(c->auxW), Npad, prow, 2, false};
            pn.inv_out = grad_out ? p(c->auxA) : nullptr;
            if (prow > Npad) {
                ALGP_HIP(hipMemsetAsync(p(c->auxW) + Npad * Npad, 0, sizeof(T) * NB * Npad, c->stream));
                ALGP_HIP(hipMemcpyAsync(p(c->auxW) + Npad * Npad, c->y0.p, sizeof(T) * Npad, hipMemcpyDeviceToDevice, c->stream));
                pn.z_row = Npad;
            }
            const int frc = factorize(c, 0, &pn);
            if (pn.inv_enqueued && (frc != ALGP_OK || !grad_out)) hipStreamSynchronize(c->stream2);   // nothing outlives the call
            ALGP_TRY(frc);
            have_X = pn.done;
            inv_enq = pn.inv_enqueued;
        } else {
            ALGP_TRY(factorize(c, 0));
        }
        if (mll_out) *mll_out = -0.5 * c->yalpha - 0.5 * c->logdet - 0.5 * (double)c->N * 1.8378770664093453;
        if (!grad_out) return ALGP_OK;
        const int grc = mll_grad(c, grad_out, have_X, inv_enq);
        if (grc != ALGP_OK && inv_enq) hipStreamSynchronize(c->stream2);
        return grc;
    }

    static int get_alpha(algp_ctx* c, void* out) {
        if (!c->factored || c->train_dirty) return fail(c, ALGP_ERR_STATE, "get_alpha: call algp_factorize first");
        ALGP_TRY(need_alpha(c));
        ALGP_HIP(hipMemcpyAsync(out, c->alpha.p, sizeof(T) * c->N, hipMemcpyDeviceToHost, c->stream));
        const int rc = sync_checked(c, "get_alpha");
        if (rc != ALGP_OK) c->alpha_valid = false;
        return rc;
    }
    static int get_factor(algp_ctx* c, void* out) {
        if (!c->factored || c->train_dirty) return fail(c, ALGP_ERR_STATE, "get_factor: call algp_factorize first");
        const int64_t N = c->N;
        ALGP_HIP(hipMemcpy2DAsync(out, sizeof(T) * N, c->L.p, sizeof(T) * c->Lld, sizeof(T) * N, N, hipMemcpyDeviceToHost, c->stream));
        ALGP_TRY(sync(c));
        T* Lh = (T*)out;
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = i + 1; j < N; ++j) Lh[i * N + j] = (T)0;
        return ALGP_OK;
    }
    static int selftest(algp_ctx* c, int* mism) {
        double* sc = (double*)c->scal.p;
        int* d = (int*)(sc + SC_PROBE);
        ALGP_HIP(hipMemsetAsync(d, 0, sizeof(double), c->stream));
        ALGP_TRY(test_mfma_launch<double>(c, d));
        ALGP_TRY(test_mfma_launch<float>(c, d));
        ALGP_HIP(hipMemcpyAsync(mism, d, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        return sync(c);
    }
};

}  // namespace

#define DISPATCH(c, call) ((c)->dtype == ALGP_F64 ? Impl<double>::call : Impl<float>::call)
#define CHECK_CTX(c) do { if (!(c)) return ALGP_ERR_BAD_ARG; (c)->err.clear(); hipSetDevice((c)->device); } while (0)
#define NEED_HYPERS(c) do { if (!(c)->hyp.set) return fail(c, ALGP_ERR_STATE, "call algp_set_hypers first"); } while (0)
#define FINISH(c, expr) do { int rc__ = (expr); if ((c)->prof_on) prof_collect(c); return rc__; } while (0)

extern "C" {

int algp_version(void) { return 100; }

int algp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int algp_create(int device_id, int dtype, algp_ctx** out) {
    if (!out || (dtype != ALGP_F32 && dtype != ALGP_F64)) return ALGP_ERR_BAD_ARG;
    *out = nullptr;
    // more hardware queues than the default 4 so the helper streams do not share one with RCCL / torch
    // streams of the same process (no effect if the HIP runtime is already initialised)
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return ALGP_ERR_NO_DEVICE; }
    if (device_id < 0 || device_id >= n) return ALGP_ERR_BAD_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return ALGP_ERR_HIP;
    algp_ctx* c = new algp_ctx();
    c->device = device_id;
    c->dtype = dtype;
    c->es = dtype == ALGP_F64 ? 8 : 4;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return ALGP_ERR_HIP; }
    if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) { hipStreamDestroy(c->stream); delete c; return ALGP_ERR_HIP; }
    if (hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) != hipSuccess) c->stream3 = nullptr;
    if (hipStreamCreateWithFlags(&c->stream4, hipStreamNonBlocking) != hipSuccess) c->stream4 = nullptr;
    c->trsm_chunks = env_int("ALGP_TRSM_CHUNKS", c->trsm_chunks);
    c->cur = c->stream;
    if (ensure(c, c->scal, sizeof(double) * SC_COUNT) != ALGP_OK) { hipStreamDestroy(c->stream); delete c; return ALGP_ERR_OOM; }
    hipMemsetAsync(c->scal.p, 0, sizeof(double) * SC_COUNT, c->stream);
    hipStreamSynchronize(c->stream);
    *out = c;
    return ALGP_OK;
}

void algp_destroy(algp_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    DevBuf* bufs[] = {&c->Xs, &c->Xraw, &c->Cp, &c->Aidx, &c->yA, &c->varA, &c->y0, &c->L, &c->invD, &c->z, &c->alpha,
                      &c->scal, &c->Cidx, &c->ckind, &c->cextra, &c->Vt, &c->dstat, &c->mu, &c->alive, &c->scores,
                      &c->lrow, &c->remote, &c->commbuf, &c->tvec, &c->amax, &c->prevrows, &c->fresh, &c->lazypicks, &c->yraw, &c->uvec, &c->wvec, &c->acc3, &c->rowstat, &c->inv512, &c->inv512_scr, &c->trsm_tmp, &c->splitk, &c->dag_state, &c->dag_stats, &c->trsv_ctrl, &c->miXbar, &c->miXall, &c->miDP, &c->miDQ, &c->miPos, &c->miU, &c->miW, &c->miCol, &c->miH, &c->auxA, &c->auxInv, &c->auxW, &c->auxIdx,
                      &c->auxVar, &c->auxD, &c->hostStage, &c->rowx, &c->tailE, &c->tailPart, &c->ldpart};
    for (DevBuf* b : bufs) release(c, *b);
    dag_release(c);
    comm_destroy(c);
    for (hipEvent_t e : c->event_pool) hipEventDestroy(e);
    for (hipEvent_t e : c->sync_events) hipEventDestroy(e);
    hipStreamSynchronize(c->stream2);
    hipStreamDestroy(c->stream2);
    if (c->stream3) { hipStreamSynchronize(c->stream3); hipStreamDestroy(c->stream3); }
    if (c->stream4) { hipStreamSynchronize(c->stream4); hipStreamDestroy(c->stream4); }
    hipStreamDestroy(c->stream);
    delete c;
}

const char* algp_last_error(const algp_ctx* c) { return c ? c->err.c_str() : "null context"; }
int64_t algp_last_pivot(const algp_ctx* c) { return c ? c->pivot : 0; }
double algp_last_jitter(const algp_ctx* c) { return c ? c->last_jitter : 0.0; }
int algp_dtype(const algp_ctx* c) { return c ? c->dtype : -1; }

int algp_set_hypers(algp_ctx* c, int kernel, int D, const double* log_ls, double log_os, double log_noise) {
    CHECK_CTX(c);
    if (D < 1 || D > MAXD || !log_ls) return fail(c, ALGP_ERR_BAD_ARG, "set_hypers: 1 <= D <= 8 required");
    if (kernel != ALGP_KERNEL_RBF && kernel != ALGP_KERNEL_MATERN15) return fail(c, ALGP_ERR_BAD_ARG, "set_hypers: unknown kernel");
    if (c->hyp.set && c->hyp.D != D && c->n_pool > 0 && !c->pool_is_cov) {
        // the resident coordinates have another width: drop the pool, the caller sets a new one
        c->n_pool = 0;
        c->N = 0;
        c->pos_in_train.clear();
    }
    c->hyp.kernel = kernel;
    c->hyp.D = D;
    c->hyp.DP = D <= 2 ? 2 : (D <= 4 ? 4 : 8);
    for (int d = 0; d < MAXD; ++d) c->hyp.inv_ls[d] = d < D ? exp(-log_ls[d]) : 0.0;
    c->hyp.outputscale = exp(log_os);
    c->hyp.noise = exp(log_noise);
    c->hyp.set = true;
    c->hyp_stamp++;
    c->factored = false;
    c->solved = false;
    FINISH(c, DISPATCH(c, rescale_pool(c)));
}

int algp_kernel_matrix(algp_ctx* c, const void* x1, int64_t n1, const void* x2, int64_t n2, const void* diag_add,
                       int add_lik, void* out) {
    CHECK_CTX(c);
    NEED_HYPERS(c);
    if (!x1 || n1 < 0 || n2 < 0 || !out) return fail(c, ALGP_ERR_BAD_ARG, "kernel_matrix: bad arguments");
    FINISH(c, DISPATCH(c, kernel_matrix(c, x1, n1, x2, n2, diag_add, add_lik, out)));
}

int algp_set_pool(algp_ctx* c, const void* x, int64_t n) {
    CHECK_CTX(c);
    NEED_HYPERS(c);
    if (!x || n <= 0) return fail(c, ALGP_ERR_BAD_ARG, "set_pool: bad arguments");
    c->factored = c->solved = false;
    c->hyp_stamp++;
    c->pos_in_train.clear();
    c->N = 0;
    c->site_owner.clear();                                   // the owner map belongs to the pool it was given for (re-attach it: algp_comm_set_owners)
    c->site_owner_hash = 0;
    FINISH(c, DISPATCH(c, set_pool(c, x, n)));
}

int algp_set_pool_cov(algp_ctx* c, const void* cov, int64_t n) {
    CHECK_CTX(c);
    NEED_HYPERS(c);
    if (!cov || n <= 0) return fail(c, ALGP_ERR_BAD_ARG, "set_pool_cov: bad arguments");
    c->factored = c->solved = false;
    c->hyp_stamp++;
    c->pos_in_train.clear();
    c->N = 0;
    c->site_owner.clear();                                   // the owner map belongs to the pool it was given for (re-attach it: algp_comm_set_owners)
    c->site_owner_hash = 0;
    FINISH(c, DISPATCH(c, set_pool_cov(c, cov, n)));
}

int algp_set_train(algp_ctx* c, const int64_t* idx, int64_t N, const void* y, const void* var) {
    CHECK_CTX(c);
    if (c->n_pool <= 0) return fail(c, ALGP_ERR_STATE, "set_train: set a pool first");
    if (N < 0 || (N > 0 && (!idx || !y))) return fail(c, ALGP_ERR_BAD_ARG, "set_train: bad arguments");
    for (int64_t i = 0; i < N; ++i)
        if (idx[i] < 0 || idx[i] >= c->n_pool) return fail(c, ALGP_ERR_BAD_ARG, "set_train: index outside the pool");
    FINISH(c, DISPATCH(c, set_train(c, idx, N, y, var)));
}

int algp_set_constant_mean(algp_ctx* c, int enable, double value) {
    CHECK_CTX(c);
    if (enable && !(value == value)) return fail(c, ALGP_ERR_BAD_ARG, "set_constant_mean: NaN");
    c->mean_override = enable != 0;
    c->mean_value = value;
    return ALGP_OK;
}

int algp_factorize(algp_ctx* c) {
    CHECK_CTX(c);
    NEED_HYPERS(c);
    if (c->n_pool <= 0) return fail(c, ALGP_ERR_STATE, "factorize: set a pool first");
    // an empty train set is legal (greedy from an empty field: agent.py:308 with a 0 x 0 slogdet = 0),
    // but it has to be declared through algp_set_train(ctx, NULL, 0, NULL, NULL)
    if (!c->y0.p || (int64_t)c->pos_in_train.size() != c->n_pool)
        return fail(c, ALGP_ERR_STATE, "factorize: call algp_set_train first");
    FINISH(c, DISPATCH(c, factorize(c, 0)));
}

int algp_factorize_update(algp_ctx* c, int64_t* kept_rows) {
    CHECK_CTX(c);
    NEED_HYPERS(c);
    if (c->n_pool <= 0) return fail(c, ALGP_ERR_STATE, "factorize_update: set a pool first");
    if (!c->y0.p || (int64_t)c->pos_in_train.size() != c->n_pool)
        return fail(c, ALGP_ERR_STATE, "factorize_update: call algp_set_train first");
    int rc = DISPATCH(c, factorize(c, 1));
    if (kept_rows) *kept_rows = rc == ALGP_OK ? c->kept_rows_last : 0;
    if (c->prof_on) prof_collect(c);
    return rc;
}

int algp_get_logdet(algp_ctx* c, double* logdet) {
    CHECK_CTX(c);
    if (!c->factored || c->train_dirty || !logdet) return fail(c, ALGP_ERR_STATE, "get_logdet: call algp_factorize first");
    *logdet = c->logdet;
    return ALGP_OK;
}
int algp_get_entropy(algp_ctx* c, double* H) {
    CHECK_CTX(c);
    if (!c->factored || c->train_dirty || !H) return fail(c, ALGP_ERR_STATE, "get_entropy: call algp_factorize first");
    *H = (double)c->N * ENT_CONST + 0.5 * c->logdet;
    return ALGP_OK;
}
int algp_get_mll(algp_ctx* c, double* mll) {
    CHECK_CTX(c);
    if (!c->factored || c->train_dirty || !mll) return fail(c, ALGP_ERR_STATE, "get_mll: call algp_factorize first");
    *mll = -0.5 * c->yalpha - 0.5 * c->logdet - 0.5 * (double)c->N * 1.8378770664093453;
    return ALGP_OK;
}
int algp_get_mll_grad(algp_ctx* c, double* grad) {
    CHECK_CTX(c);
    if (!grad) return fail(c, ALGP_ERR_BAD_ARG, "get_mll_grad: bad arguments");
    FINISH(c, DISPATCH(c, mll_grad(c, grad)));
}
int algp_fit_step(algp_ctx* c, double* mll, double* grad) {
    CHECK_CTX(c);
    NEED_HYPERS(c);
    if (c->n_pool <= 0) return fail(c, ALGP_ERR_STATE, "fit_step: set a pool first");
    if (!c->y0.p || (int64_t)c->pos_in_train.size() != c->n_pool) return fail(c, ALGP_ERR_STATE, "fit_step: call algp_set_train first");
    FINISH(c, DISPATCH(c, fit_step(c, mll, grad)));
}
int algp_get_alpha(algp_ctx* c, void* out) { CHECK_CTX(c); FINISH(c, DISPATCH(c, get_alpha(c, out))); }
int algp_get_factor(algp_ctx* c, void* out) { CHECK_CTX(c); FINISH(c, DISPATCH(c, get_factor(c, out))); }

int algp_set_candidates(algp_ctx* c, const int64_t* idx, int64_t M, int prior_noise, const void* extra) {
    CHECK_CTX(c);
    if (c->n_pool <= 0) return fail(c, ALGP_ERR_STATE, "set_candidates: set a pool first");
    if (M < 0 || (M > 0 && !idx)) return fail(c, ALGP_ERR_BAD_ARG, "set_candidates: bad arguments");
    for (int64_t i = 0; i < M; ++i)
        if (idx[i] < 0 || idx[i] >= c->n_pool) return fail(c, ALGP_ERR_BAD_ARG, "set_candidates: index outside the pool");
    FINISH(c, DISPATCH(c, set_candidates(c, idx, M, prior_noise, extra)));
}
int algp_factorize_from(algp_ctx* c, algp_ctx* src, int64_t* kept_rows) {
    CHECK_CTX(c);
    if (!src) return fail(c, ALGP_ERR_BAD_ARG, "factorize_from: null source");
    NEED_HYPERS(c);
    if (c->n_pool <= 0) return fail(c, ALGP_ERR_STATE, "factorize_from: set a pool first");
    if (!c->y0.p || (int64_t)c->pos_in_train.size() != c->n_pool)
        return fail(c, ALGP_ERR_STATE, "factorize_from: call algp_set_train first");
    int rc = DISPATCH(c, factorize_from(c, src));
    if (kept_rows) *kept_rows = rc == ALGP_OK ? c->kept_rows_last : 0;
    if (c->prof_on) prof_collect(c);
    return rc;
}
int algp_fit_and_solve(algp_ctx* c) {
    CHECK_CTX(c);
    NEED_HYPERS(c);
    if (c->n_pool <= 0) return fail(c, ALGP_ERR_STATE, "fit_and_solve: set a pool first");
    if (!c->y0.p || (int64_t)c->pos_in_train.size() != c->n_pool)
        return fail(c, ALGP_ERR_STATE, "fit_and_solve: call algp_set_train first");
    if ((int64_t)c->cand_pos.size() != c->n_pool) return fail(c, ALGP_ERR_STATE, "fit_and_solve: call algp_set_candidates first");
    FINISH(c, DISPATCH(c, fit_and_solve(c)));
}
int algp_solve_candidates(algp_ctx* c) { CHECK_CTX(c); FINISH(c, DISPATCH(c, solve_candidates(c, 0, nullptr))); }
int algp_solve_candidates_update(algp_ctx* c, const uint8_t* alive, int64_t* kept_cols) {
    CHECK_CTX(c);
    int rc = DISPATCH(c, solve_candidates(c, 1, alive));
    if (kept_cols) *kept_cols = rc == ALGP_OK ? c->kept_cols_last : 0;
    if (c->prof_on) prof_collect(c);
    return rc;
}
int algp_set_candidate_alive(algp_ctx* c, const uint8_t* alive) {
    CHECK_CTX(c);
    if (!c->solved || !alive) return fail(c, ALGP_ERR_STATE, "set_candidate_alive: solve the candidates first");
    if (hipMemcpyAsync(c->alive.p, alive, c->M, hipMemcpyHostToDevice, c->stream) != hipSuccess)
        return fail(c, ALGP_ERR_HIP, "set_candidate_alive: copy failed");
    c->bounds_valid = false;            // a re-enabled row has no bound in c->scores
    return sync(c);
}
int algp_get_posterior(algp_ctx* c, void* mu, void* var) { CHECK_CTX(c); FINISH(c, DISPATCH(c, get_posterior(c, mu, var))); }
int algp_get_posterior_cov(algp_ctx* c, void* cov, double* mi) { CHECK_CTX(c); FINISH(c, DISPATCH(c, get_posterior_cov(c, cov, mi))); }
int algp_posterior_mean(algp_ctx* c, const int64_t* idx, int64_t M, void* mu) {
    CHECK_CTX(c);
    if (M < 0 || (M > 0 && (!idx || !mu))) return fail(c, ALGP_ERR_BAD_ARG, "posterior_mean: bad arguments");
    for (int64_t i = 0; i < M; ++i)
        if (idx[i] < 0 || idx[i] >= c->n_pool) return fail(c, ALGP_ERR_BAD_ARG, "posterior_mean: index outside the pool");
    FINISH(c, DISPATCH(c, posterior_mean(c, idx, M, mu)));
}

int algp_scores(algp_ctx* c, int criterion, double static_std, double mobile_std, void* out, int out_is_device) {
    CHECK_CTX(c);
    FINISH(c, DISPATCH(c, scores(c, criterion, static_std, mobile_std, out, out_is_device)));
}
int algp_argmax(algp_ctx* c, int64_t* local_pos, int64_t* pool_idx, double* value) {
    CHECK_CTX(c);
    FINISH(c, DISPATCH(c, argmax(c, local_pos, pool_idx, value)));
}
int algp_best_candidate(algp_ctx* c, int criterion, double static_std, double mobile_std, int64_t* local_pos,
                        int64_t* pool_idx, double* value) {
    CHECK_CTX(c);
    FINISH(c, DISPATCH(c, best_candidate(c, criterion, static_std, mobile_std, local_pos, pool_idx, value)));
}
int algp_commit_pick(algp_ctx* c, int64_t pool_idx, double static_std, double mobile_std) {
    CHECK_CTX(c);
    FINISH(c, DISPATCH(c, commit_pick(c, pool_idx, static_std, mobile_std)));
}
int algp_greedy(algp_ctx* c, int criterion, double static_std, double mobile_std, int k, const int64_t* forced,
                int64_t* picks_out, double* ut_out) {
    CHECK_CTX(c);
    if (k < 0 || k > MAX_APPEND) return fail(c, ALGP_ERR_BAD_ARG, "greedy: 0 <= k <= 128");
    FINISH(c, DISPATCH(c, greedy(c, criterion, static_std, mobile_std, k, forced, picks_out, ut_out)));
}

int algp_score_paths(algp_ctx* c, const int64_t* sites, int npaths, int maxlen, double mobile_std, double* dH_out) {
    CHECK_CTX(c);
    if (npaths < 0 || maxlen < 1 || (npaths > 0 && (!sites || !dH_out))) return fail(c, ALGP_ERR_BAD_ARG, "score_paths: bad arguments");
    if (npaths == 0) return ALGP_OK;
    FINISH(c, DISPATCH(c, score_paths(c, sites, npaths, maxlen, mobile_std, dH_out)));
}
int algp_comm_unique_id(void* out128) {
    if (!out128) return ALGP_ERR_BAD_ARG;
    return comm_unique_id(out128, nullptr);
}
int algp_comm_init(algp_ctx* c, int nranks, int rank, const void* unique_id128) {
    CHECK_CTX(c);
    if (nranks < 1 || rank < 0 || rank >= nranks || !unique_id128) return fail(c, ALGP_ERR_BAD_ARG, "comm_init: bad arguments");
    return comm_init(c, nranks, rank, unique_id128);
}
int algp_comm_init_host(algp_ctx* c, int nranks, int rank, algp_allgather_fn fn, void* user) {
    CHECK_CTX(c);
    if (nranks < 1 || rank < 0 || rank >= nranks || !fn) return fail(c, ALGP_ERR_BAD_ARG, "comm_init_host: bad arguments");
    hipStreamSynchronize(c->stream);
    return comm_init_host(c, nranks, rank, fn, user);
}
#if ALGP_TEST_HOOKS
int algp_debug_first_max(algp_ctx* c, const double* triples, int nranks, double out5[5]) {
    CHECK_CTX(c);
    if (!triples || nranks < 1 || nranks > 4096 || !out5) return fail(c, ALGP_ERR_BAD_ARG, "debug_first_max: bad arguments");
    return comm_debug_first_max(c, triples, nranks, out5);
}
int64_t algp_debug_counter(algp_ctx* c, int which) {
    if (!c) return -1;
    switch (which) {
        case 0: return c->n_syncs;
        case 1: return c->factor_rows_from_vt;
        case 2: return c->rows_from_peers;
        case 3: return c->row_exchanges;
        case 4: return c->row_fallbacks;
        default: return -1;
    }
}
#endif
int algp_comm_set_owners(algp_ctx* c, const int32_t* owner, int64_t n_pool) {
    CHECK_CTX(c);
    if (!owner) { c->site_owner.clear(); c->site_owner_hash = 0; return ALGP_OK; }
    if (!c->comm && !c->host_gather) return fail(c, ALGP_ERR_STATE, "comm_set_owners: call algp_comm_init (or algp_comm_init_host) first");
    if (n_pool != c->n_pool || n_pool <= 0) return fail(c, ALGP_ERR_BAD_ARG, "comm_set_owners: one entry per pool site (set the pool first)");
    for (int64_t i = 0; i < n_pool; ++i)
        if (owner[i] < -1 || owner[i] >= c->comm_nranks) return fail(c, ALGP_ERR_BAD_ARG, "comm_set_owners: rank outside the communicator");
    c->site_owner.assign(owner, owner + n_pool);
    uint64_t h = 1469598103934665603ull;
    for (int64_t i = 0; i < n_pool; ++i) h = (h ^ (uint64_t)(int64_t)owner[i]) * 1099511628211ull;
    c->site_owner_hash = h;
    return ALGP_OK;
}
#if ALGP_TEST_HOOKS
int algp_debug_set_trsm_chunks(algp_ctx* c, int chunks) {
    CHECK_CTX(c);
    if (chunks < 0 || chunks > 4) return fail(c, ALGP_ERR_BAD_ARG, "debug_set_trsm_chunks: 0 (default) .. 4");
    hipStreamSynchronize(c->stream);
    if (chunks == 0) chunks = env_int("ALGP_TRSM_CHUNKS", 3);
    c->trsm_chunks = chunks;
    return ALGP_OK;
}
int algp_debug_trsv_stall(algp_ctx* c, int block) {
    CHECK_CTX(c);
    if (block < -1) return fail(c, ALGP_ERR_BAD_ARG, "debug_trsv_stall: a block >= 0, or -1 to disarm");
    c->debug_trsv_stall_block = block;
    return ALGP_OK;
}
int algp_debug_get_pick(algp_ctx* c, int q, void* row_out, int64_t row_capacity, int64_t* ncols_out, double* d_out) {
    CHECK_CTX(c);
    if (!c->solved || q < 0 || q >= (int)c->picks.size()) return fail(c, ALGP_ERR_BAD_ARG, "debug_get_pick: no such pick since the last solve");
    LazyPick lp;
    ALGP_HIP(hipMemcpyAsync(&lp, (const LazyPick*)c->lazypicks.p + q, sizeof(lp), hipMemcpyDeviceToHost, c->stream));
    ALGP_HIP(hipStreamSynchronize(c->stream));
    if (ncols_out) *ncols_out = lp.ncols;
    if (d_out) *d_out = lp.d;
    if (row_out) {
        if (row_capacity < lp.ncols) return fail(c, ALGP_ERR_BAD_ARG, "debug_get_pick: row buffer too small");
        ALGP_HIP(hipMemcpyAsync(row_out, (const char*)c->prevrows.p + (size_t)q * c->ldv * c->es, (size_t)lp.ncols * c->es,
                                hipMemcpyDeviceToHost, c->stream));
        ALGP_HIP(hipStreamSynchronize(c->stream));
    }
    return ALGP_OK;
}
int algp_debug_get_factor_rows(algp_ctx* c, int64_t row0, int64_t nrows, int64_t ncols, void* out) {
    CHECK_CTX(c);
    if (!c->factored || c->train_dirty) return fail(c, ALGP_ERR_STATE, "debug_get_factor_rows: call algp_factorize first");
    if (row0 < 0 || nrows < 0 || ncols < 0 || row0 + nrows > c->Npad || ncols > c->Npad || (nrows > 0 && ncols > 0 && !out))
        return fail(c, ALGP_ERR_BAD_ARG, "debug_get_factor_rows: rows / columns outside the factor");
    if (nrows == 0 || ncols == 0) return ALGP_OK;
    ALGP_HIP(hipMemcpy2DAsync(out, c->es * (size_t)ncols, (const char*)c->L.p + (size_t)row0 * c->Lld * c->es, c->es * (size_t)c->Lld,
                              c->es * (size_t)ncols, (size_t)nrows, hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}
int algp_debug_dag_stall(algp_ctx* c, int ticket) {
    CHECK_CTX(c);
    if (ticket < -1) return fail(c, ALGP_ERR_BAD_ARG, "debug_dag_stall: a ticket >= 0, or -1 to disarm");
    c->debug_dag_stall_ticket = ticket;
    return ALGP_OK;
}
int algp_debug_fail_next_pick(algp_ctx* c, int code) {
    CHECK_CTX(c);
    if (code != 0 && (code < 2 || code > ALGP_ERR_NO_DEVICE)) return fail(c, ALGP_ERR_BAD_ARG, "debug_fail_next_pick: an ALGP_ERR_* code >= 2, or 0");
    c->debug_fail_next_pick = code;
    return ALGP_OK;
}
int algp_debug_fail_at(algp_ctx* c, int where, int code) {
    CHECK_CTX(c);
    if (code != 0 && (code < 2 || code > ALGP_ERR_NO_DEVICE)) return fail(c, ALGP_ERR_BAD_ARG, "debug_fail_at: an ALGP_ERR_* code >= 2, or 0");
    if (where == 0) c->debug_fail_next_pick = code;
    else if (where == 1) c->debug_fail_next_commit = code;
    else if (where == 2) c->debug_fail_next_pack = code;
    else if (where == 3) c->debug_fail_next_rowx = code;
    else return fail(c, ALGP_ERR_BAD_ARG, "debug_fail_at: where = 0 (pick), 1 (commit), 2 (pack), 3 (row exchange)");
    return ALGP_OK;
}
#endif
int algp_comm_destroy(algp_ctx* c) {
    CHECK_CTX(c);
    hipStreamSynchronize(c->stream);
    comm_destroy(c);
    return ALGP_OK;
}
int algp_greedy_sharded(algp_ctx* c, int criterion, double static_std, double mobile_std, int k, int64_t* picks_out,
                        double* utilities_out) {
    CHECK_CTX(c);
    if (k < 0 || k > MAX_APPEND) return fail(c, ALGP_ERR_BAD_ARG, "greedy_sharded: 0 <= k <= 128");
    if (criterion != ALGP_CRIT_ENTROPY)
        return fail(c, ALGP_ERR_BAD_ARG, "greedy_sharded: only the entropy criterion shards (the MI criterion needs the "
                                         "pool-wide complement on one GPU: use algp_greedy)");
    if (!c->comm && !c->host_gather) return fail(c, ALGP_ERR_STATE, "greedy_sharded: call algp_comm_init (or algp_comm_init_host) first");
    FINISH(c, DISPATCH(c, greedy_picks(c, static_std, mobile_std, k, picks_out, utilities_out)));
}

int algp_entropy_from_cov(algp_ctx* c, const void* cov, int64_t k, double* H) {
    CHECK_CTX(c);
    if (k < 0 || !H || (k > 0 && !cov)) return fail(c, ALGP_ERR_BAD_ARG, "entropy_from_cov: bad arguments");
    FINISH(c, DISPATCH(c, entropy_from_cov(c, cov, k, H, nullptr, nullptr)));
}
int algp_set_entropy(algp_ctx* c, const int64_t* idx, int64_t m, const void* var, double* H) {
    CHECK_CTX(c);
    NEED_HYPERS(c);
    if (m < 0 || !H || (m > 0 && !idx)) return fail(c, ALGP_ERR_BAD_ARG, "set_entropy: bad arguments");
    for (int64_t i = 0; i < m; ++i)
        if (idx[i] < 0 || idx[i] >= c->n_pool) return fail(c, ALGP_ERR_BAD_ARG, "set_entropy: index outside the pool");
    FINISH(c, DISPATCH(c, set_entropy(c, idx, m, var, H)));
}
int algp_set_inverse_diag(algp_ctx* c, const int64_t* idx, int64_t m, const void* var, void* diag_out, double* H) {
    CHECK_CTX(c);
    NEED_HYPERS(c);
    if (m < 0 || (m > 0 && (!idx || !diag_out))) return fail(c, ALGP_ERR_BAD_ARG, "set_inverse_diag: bad arguments");
    for (int64_t i = 0; i < m; ++i)
        if (idx[i] < 0 || idx[i] >= c->n_pool) return fail(c, ALGP_ERR_BAD_ARG, "set_inverse_diag: index outside the pool");
    FINISH(c, DISPATCH(c, set_inverse_diag(c, idx, m, var, diag_out, H)));
}

int algp_cholesky(algp_ctx* c, const void* A, int64_t n, void* L_out, double* logdet) {
    CHECK_CTX(c);
    if (n < 0 || (n > 0 && !A)) return fail(c, ALGP_ERR_BAD_ARG, "cholesky: bad arguments");
    FINISH(c, DISPATCH(c, entropy_from_cov(c, A, n, nullptr, L_out, logdet)));
}
int algp_gemm_nt(algp_ctx* c, int64_t m, int64_t n, int64_t k, double alpha, const void* A, const void* B, double beta,
                 const void* C, void* D) {
    CHECK_CTX(c);
    if (m <= 0 || n <= 0 || k <= 0 || !A || !B || !D) return fail(c, ALGP_ERR_BAD_ARG, "gemm_nt: bad arguments");
    FINISH(c, DISPATCH(c, gemm_host(c, m, n, k, alpha, A, B, beta, C, D)));
}
int algp_trsm_right_lt(algp_ctx* c, const void* L, int64_t n, const void* B, int64_t m, void* X) {
    CHECK_CTX(c);
    if (m <= 0 || n <= 0 || !L || !B || !X) return fail(c, ALGP_ERR_BAD_ARG, "trsm: bad arguments");
    FINISH(c, DISPATCH(c, trsm_host(c, L, n, B, m, X)));
}
int algp_selftest_mfma(algp_ctx* c, int* mismatches) {
    CHECK_CTX(c);
    if (!mismatches) return fail(c, ALGP_ERR_BAD_ARG, "selftest: bad arguments");
    return Impl<double>::selftest(c, mismatches);
}

int algp_bench_gemm(algp_ctx* c, int64_t m, int64_t n, int64_t k, int lower_only, int beta_one, int reps,
                    double* ms) {
    CHECK_CTX(c);
    if (!ms || m <= 0 || n <= 0 || k <= 0 || reps <= 0) return fail(c, ALGP_ERR_BAD_ARG, "bench_gemm: bad arguments");
    const bool was = c->prof_on;
    c->prof_on = false;
    int rc = c->dtype == ALGP_F64 ? bench_gemm<double>(c, m, n, k, lower_only, beta_one, reps, ms)
                                  : bench_gemm<float>(c, m, n, k, lower_only, beta_one, reps, ms);
    c->prof_on = was;
    return rc;
}

int algp_sync(algp_ctx* c) { CHECK_CTX(c); return sync(c); }
int64_t algp_device_bytes(const algp_ctx* c) { return c ? c->dev_bytes : 0; }

int algp_prof_enable(algp_ctx* c, int on) {
    CHECK_CTX(c);
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    c->prof_on = on != 0;
    return ALGP_OK;
}
int algp_prof_reset(algp_ctx* c) {
    CHECK_CTX(c);
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    for (int i = 0; i < ALGP_PROF_COUNT; ++i) c->prof[i] = ProfSlot();
    if (c->dag_stats.p) ALGP_HIP(hipMemset(c->dag_stats.p, 0, 64));
    return ALGP_OK;
}
int algp_cholesky_task_stats(algp_ctx* c, double out[4]) {
    CHECK_CTX(c);
    if (!out) return fail(c, ALGP_ERR_BAD_ARG, "cholesky_task_stats: null output");
    unsigned long long h[4] = {0, 0, 0, 0};
    if (c->dag_stats.p) {
        ALGP_HIP(hipStreamSynchronize(c->stream));
        ALGP_HIP(hipMemcpy(h, c->dag_stats.p, sizeof(h), hipMemcpyDeviceToHost));
    }
    out[0] = (double)h[0] * 0.01;                              // 100 MHz ticks -> microseconds
    out[1] = (double)h[1];
    out[2] = (double)h[2] * 0.01;
    out[3] = (double)h[3];
    return ALGP_OK;
}
int algp_prof_get(algp_ctx* c, int klass, double* ms, double* flops, double* bytes, int64_t* launches) {
    CHECK_CTX(c);
    if (klass < 0 || klass >= ALGP_PROF_COUNT) return fail(c, ALGP_ERR_BAD_ARG, "prof_get: unknown class");
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    if (ms) *ms = c->prof[klass].ms;
    if (flops) *flops = c->prof[klass].flops;
    if (bytes) *bytes = c->prof[klass].bytes;
    if (launches) *launches = c->prof[klass].launches;
    return ALGP_OK;
}

}  // extern "C"
