// api_factor.hip -- the factor of the train set: S = C_AA + D -> L (one-launch task list or launch sequence), incremental updates
// (kept leading blocks, new rows gathered from V^T or exchanged between ranks), alpha, the factor read-back.
#include "api_impl.h"

using namespace algp;

namespace algp {

template <typename T>
bool Impl<T>::panel_fits(int64_t npad, int64_t mpad) {
    const int64_t nt = npad / NB, mt = mpad / NB;
    return dag_enabled() && nt >= DAG_MIN_TILES && nt <= DAG_MAX_TILES && mt >= 1 && mt <= DAG_MAX_PANEL_TILES;
}


// factor an npad x npad matrix already resident in A; returns logdet; NOT_PD -> error with pivot
template <typename T>
int Impl<T>::factor_resident(algp_ctx* c, T* A, int64_t n, int64_t npad, T* invD, int slot_logdet, int slot_info,
                               double* logdet, int64_t ld, int64_t pivot_offset, typename Impl<T>::Panel* panel) {
    if (ld == 0) ld = npad;
    double* sc = (double*)c->scal.p;
    ALGP_HIP(hipMemsetAsync(sc + slot_logdet, 0, 2 * sizeof(double), c->stream));
    if (panel && panel_fits(npad, panel->mpad)) {
        ALGP_TRY(cholesky_dag_panel<T>(c, A, npad, ld, invD, sc + slot_logdet, (int*)(sc + slot_info), panel->P, panel->ldp,
                                       panel->mpad, panel->mode, (int)(panel->short_rows / NB)));
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
template <typename T>
int Impl<T>::reserve_factor(algp_ctx* c, int64_t npad_need, int64_t keep_rows, int64_t keep_height,
                              bool headroom) {
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
template <typename T>
bool Impl<T>::vt_rows_for_new_sites(algp_ctx* c, int64_t Nb, int64_t p0, std::vector<int64_t>& src_row,
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
template <typename T>
int Impl<T>::exchange_new_rows(algp_ctx* c, int64_t Nb, int64_t p0, int* placed, int st_in) {
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
    // (a rank that takes part only to say "I keep nothing" reserves nothing: its plan would cover the whole train set)
    if (st <= 1 && cap > 0 && Nb > 0) {
        const int rc = comm_rows_reserve(c, sizeof(T) * (size_t)c->Lld * (size_t)std::max<int64_t>(16, round_up(cap, 8)));
        if (rc != ALGP_OK) st = rc;
    }
    if (st <= 1 && Nb > 0) {
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


template <typename T>
int Impl<T>::factorize(algp_ctx* c, int incremental, typename Impl<T>::Panel* panel) {
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
template <typename T>
int Impl<T>::finish_factor(algp_ctx* c, int64_t keep, int64_t p0, double ld_total, T* z_src) {
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
template <typename T>
int Impl<T>::factorize_from(algp_ctx* c, algp_ctx* src) {
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


template <typename T>
int Impl<T>::need_alpha(algp_ctx* c) {
    if (c->alpha_valid) return ALGP_OK;
    ALGP_HIP(hipMemcpyAsync(c->alpha.p, c->z.p, sizeof(T) * c->Npad, hipMemcpyDeviceToDevice, c->stream));
    ALGP_TRY(trsv_backward<T>(c, p(c->L), c->Npad, c->Lld, p(c->invD), p(c->alpha)));
    c->alpha_valid = true;
    return ALGP_OK;
}


template <typename T>
int Impl<T>::get_alpha(algp_ctx* c, void* out) {
    if (!c->factored || c->train_dirty) return fail(c, ALGP_ERR_STATE, "get_alpha: call algp_factorize first");
    ALGP_TRY(need_alpha(c));
    ALGP_HIP(hipMemcpyAsync(out, c->alpha.p, sizeof(T) * c->N, hipMemcpyDeviceToHost, c->stream));
    const int rc = sync_checked(c, "get_alpha");
    if (rc != ALGP_OK) c->alpha_valid = false;
    return rc;
}

template <typename T>
int Impl<T>::get_factor(algp_ctx* c, void* out) {
    if (!c->factored || c->train_dirty) return fail(c, ALGP_ERR_STATE, "get_factor: call algp_factorize first");
    const int64_t N = c->N;
    ALGP_HIP(hipMemcpy2DAsync(out, sizeof(T) * N, c->L.p, sizeof(T) * c->Lld, sizeof(T) * N, N, hipMemcpyDeviceToHost, c->stream));
    ALGP_TRY(sync(c));
    T* Lh = (T*)out;
    for (int64_t i = 0; i < N; ++i)
        for (int64_t j = i + 1; j < N; ++j) Lh[i * N + j] = (T)0;
    return ALGP_OK;
}

template struct Impl<float>;
template struct Impl<double>;

}  // namespace algp

extern "C" {

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

int algp_get_alpha(algp_ctx* c, void* out) { CHECK_CTX(c); FINISH(c, DISPATCH(c, get_alpha(c, out))); }

int algp_get_factor(algp_ctx* c, void* out) { CHECK_CTX(c); FINISH(c, DISPATCH(c, get_factor(c, out))); }

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

#if ALGP_TEST_HOOKS
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
#endif

}  // extern "C"
