// api_greedy.hip -- information-gain scoring and the greedy picks: utilities, lazy resolution of the argmax, commits, the
// sharded pick exchange and its communicator entry points, the MI criterion's pool-wide inverses.
#include "api_impl.h"

using namespace algp;

namespace algp {


// ------------------------------------------------------------------ greedy
// ---- MI criterion (agent.py:330-339): H(A u i) + H(Abar \ i) - H(all_i) per candidate --------------------------------
// The last two terms need the diagonals of P = C_AbarAbar^-1 and Q = (C + D_all)^-1 over the WHOLE pool (see
// mi_rank1_kernel in vecops.hip).  mi_build factors both matrices once per candidate solve and leaves the triangular
// inverses X (P = X X^T) resident; mi_apply_pick folds a committed pick into both diagonals with one pass over each X
// (O(n^2)) where the reference -- and round 2 of this library -- refactorised both matrices for every pick.
template <typename T>
int Impl<T>::mi_build(algp_ctx* c, double ss, double sm) {
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
template <typename T>
int Impl<T>::mi_apply_pick(algp_ctx* c, int64_t q, double ss, double sm) {
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

template <typename T>
int Impl<T>::mi_scores_enqueue(algp_ctx* c, double ss, double sm, double delta, double* dst) {
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
template <typename T>
int Impl<T>::scores_enqueue(algp_ctx* c, int criterion, double static_std, double mobile_std, double* dst) {
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

template <typename T>
int Impl<T>::scores(algp_ctx* c, int criterion, double static_std, double mobile_std, void* out, int out_is_device) {
    ALGP_TRY(scores_enqueue(c, criterion, static_std, mobile_std, out_is_device ? (double*)out : nullptr));
    if (!out_is_device && out)
        ALGP_HIP(hipMemcpyAsync(out, c->scores.p, sizeof(double) * c->M, hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}


template <typename T>
int Impl<T>::argmax(algp_ctx* c, int64_t* local_pos, int64_t* pool_idx, double* value) {
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

template <typename T>
typename Impl<T>::RemoteSlots Impl<T>::remote_slots(algp_ctx* c) {
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

template <typename T>
int Impl<T>::remote_row(algp_ctx* c, int64_t pool_idx, int in_train) {
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
template <typename T>
int Impl<T>::commit_enqueue(algp_ctx* c, int64_t pool_idx, double ss, double delta, const char* winner_payload) {
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
template <typename T>
int Impl<T>::commit_pick(algp_ctx* c, int64_t pool_idx, double static_std, double mobile_std) {
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
template <typename T>
int Impl<T>::lazy_launch(algp_ctx* c, int mode, int64_t pos, double ss, double delta, const int64_t* pos_dev) {
    return lazy_refresh_launch<T>(c, c->M, mode, pos, (const LazyPick*)c->lazypicks.p, (int)c->picks.size(),
                                  (const int*)c->ckind.p, (const int64_t*)c->Cidx.p, (const T*)c->Xs.p,
                                  c->pool_is_cov ? (const T*)c->Cp.p : nullptr, c->n_pool, c->hyp.DP, c->hyp.kernel,
                                  (T)c->hyp.outputscale, (T)c->hyp.noise, p(c->prevrows), c->ldv, p(c->Vt),
                                  p(c->dstat), (int*)c->fresh.p, (const unsigned char*)c->alive.p,
                                  (double*)c->scores.p, ss, delta, pos_dev);
}

// after a candidate solve: no picks, every row current, no bounds
template <typename T>
int Impl<T>::reset_lazy(algp_ctx* c) {
    ALGP_TRY(ensure(c, c->fresh, sizeof(int) * std::max<int64_t>(c->Mpad, 1)));
    ALGP_TRY(ensure(c, c->lazypicks, sizeof(LazyPick) * MAX_APPEND));
    ALGP_HIP(hipMemsetAsync(c->fresh.p, 0, sizeof(int) * std::max<int64_t>(c->Mpad, 1), c->stream));
    c->lazy_stale = false;
    c->bounds_valid = false;
    return ALGP_OK;
}

// bring every row of V^T / dstat up to date with the committed picks (stream-ordered, no host sync)
template <typename T>
int Impl<T>::flush_lazy(algp_ctx* c) {
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
template <typename T>
int Impl<T>::enqueue_local_best(algp_ctx* c, double ss, double delta) {
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
template <typename T>
int Impl<T>::ensure_bounds(algp_ctx* c, int criterion, double static_std, double mobile_std, double ss, double delta) {
    static const bool lazy_on = env_switch("ALGP_LAZY_GREEDY", true);
    if (criterion != ALGP_CRIT_ENTROPY || !lazy_on || !c->bounds_valid || c->lazy_ss != ss || c->lazy_delta != delta)
        return scores_enqueue(c, criterion, static_std, mobile_std, nullptr);
    return ALGP_OK;
}


// algp_best_candidate: the local first maximum, one read-back (value, position, how many picks its row has seen)
template <typename T>
int Impl<T>::best_candidate(algp_ctx* c, int criterion, double static_std, double mobile_std, int64_t* local_pos,
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
template <typename T>
int Impl<T>::greedy_picks(algp_ctx* c, double static_std, double mobile_std, int k, int64_t* picks_out, double* ut_out) {
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


template <typename T>
int Impl<T>::greedy(algp_ctx* c, int criterion, double static_std, double mobile_std, int k, const int64_t* forced,
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

template struct Impl<float>;
template struct Impl<double>;

}  // namespace algp

extern "C" {

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
#endif

#if ALGP_TEST_HOOKS
int algp_debug_fail_next_pick(algp_ctx* c, int code) {
    CHECK_CTX(c);
    if (code != 0 && (code < 2 || code > ALGP_ERR_NO_DEVICE)) return fail(c, ALGP_ERR_BAD_ARG, "debug_fail_next_pick: an ALGP_ERR_* code >= 2, or 0");
    c->debug_fail_next_pick = code;
    return ALGP_OK;
}
#endif

#if ALGP_TEST_HOOKS
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

}  // extern "C"
