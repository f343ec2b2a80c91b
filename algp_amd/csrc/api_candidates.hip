// api_candidates.hip -- the candidate solve V^T = B^T L^-T (all orders, incremental columns, the fold with the factorisation)
// and the posterior read-backs (mean / variance / full covariance / mi).
#include "api_impl.h"

using namespace algp;

namespace algp {


template <typename T>
int Impl<T>::set_candidates(algp_ctx* c, const int64_t* idx, int64_t M, int prior_noise, const void* extra) {
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

template <typename T>
int Impl<T>::solve_prepare(algp_ctx* c, int incremental, typename Impl<T>::SolvePlan& pl) {
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
template <typename T>
int Impl<T>::solve_run(algp_ctx* c, typename Impl<T>::SolvePlan& pl) {
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
    } else if (solve_dag_on && keep == 0 && Mpad / NB > 32 && panel_fits(Npad, Mpad) && c->cur == c->stream) {
        // (a narrow last tile: every tile row leaves it out of the list, the tail kernel solves its columns -- as in
        // fit_and_solve, so that the two forms stay the same arithmetic)
        const int64_t r = c->N % NB, N1 = Npad - NB;
        const bool narrow = r > 0 && r <= 64 && N1 >= 2048 && Mpad >= 2048 && env_switch("ALGP_TAIL_COLS", true);
        // the tile row in which fit_and_solve carries y - ybar (a last tile with padding rows) keeps every column here too
        const int64_t short_rows = !narrow ? 0 : (c->M < Mpad ? Mpad - NB : Mpad);
        trc = solve_dag_panel<T>(c, p(c->L), Npad, c->Lld, p(c->invD), (int*)((double*)c->scal.p + SC_STALL), p(c->Vt), ldc, Mpad, 1,
                                 (int)(short_rows / NB));
        if (short_rows > 0 && trc == ALGP_OK)
            trc = tail_cols_launch<T>(c, ALGP_PROF_TAIL_COLS, p(c->Vt), short_rows, ldc, p(c->L), c->Lld, Npad, p(c->invD) + (N1 / NB) * NB * NB,
                                      N1, (int)round_up(r, 16));
    } else {
        // a from-scratch solve of more than 320 tile rows: its launches leave the rows' sums of v^2 and v z per column tile
        // (utils.py:301-304 needs nothing else of V^T), the 8 GB pass over V^T at config 4 falls away
        T* stat = nullptr;
        const bool stats_on = env_switch("ALGP_ROW_STATS", true);                                  // read per call: tests flip it
        if (stats_on && keep == 0 && !pl.carried_sums && ensure(c, c->rowstat, sizeof(T) * 2 * (size_t)(Npad / NB) * (size_t)Mpad) == ALGP_OK)
            stat = p(c->rowstat);
        // A train set that ends a few columns into its last 128-column tile (N = 10 000: 16 of 128) pays the sweep's full price
        // for that tile -- 2 M 128 N flop, 2.1 % of config 4's solve, for 16 columns.  Up to 64 such columns are instead what an
        // APPEND would add to a factor of N - r rows: the sweep solves the full tiles, the tail kernel (tail.hip, HBM-bound:
        // one pass over V^T) the rest, and one 128-column reduction leaves the last tile's row statistics where the sweep's
        // launches would have (round 6; $ALGP_TAIL_COLS=0: the sweep alone).
        const int64_t r = c->N % NB, N1 = Npad - NB;
        const bool narrow = keep == 0 && r > 0 && r <= 64 && N1 >= 2048 && Mpad >= 2048 && c->cur == c->stream &&
                            Mpad / NB > 320 && env_switch("ALGP_TAIL_COLS", true);      // (320 tile rows: where the chunked sweep begins, potrf.hip)
        trc = trsm_blocked<T>(c, ALGP_PROF_GEMM_TRSM, p(c->Vt), Mpad, ldc, p(c->L), narrow ? N1 : Npad, c->Lld, p(c->invD), keep, p(c->z),
                              stat, Mpad, &pl.rowstat_done);
        if (narrow && trc == ALGP_OK)
            trc = tail_cols_launch<T>(c, ALGP_PROF_TAIL_COLS, p(c->Vt), Mpad, ldc, p(c->L), c->Lld, Npad, p(c->invD) + (N1 / NB) * NB * NB, N1,
                                      (int)round_up(r, 16));
        if (narrow && trc == ALGP_OK && pl.rowstat_done)
            trc = rows_reduce_launch<T>(c, p(c->Vt) + N1, Mpad, ldc, NB, p(c->z) + N1, stat + 2 * (N1 / NB) * Mpad, stat + (2 * (N1 / NB) + 1) * Mpad);
    }
    prof_span_end(c);
    return trc;
}


template <typename T>
int Impl<T>::solve_finish(algp_ctx* c, int incremental, const unsigned char* alive_host, const typename Impl<T>::SolvePlan& pl) {
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


template <typename T>
int Impl<T>::solve_candidates(algp_ctx* c, int incremental, const unsigned char* alive_host) {
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
template <typename T>
int Impl<T>::fit_and_solve(algp_ctx* c) {
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
    // A narrow last tile (N = 10 000: 16 of 128 columns) costs every panel tile row 79 K-steps of the list for 16 columns --
    // 2.5 % of the panel's work.  The tile rows that do not carry z leave that column tile out (DagShape::pshort) and the
    // tail kernel solves the r columns behind the launch (one pass over the rank's V^T: 0.2 ms for 12 500 rows).
    const int64_t r = c->N % NB, N1 = c->Npad - NB;
    if (r > 0 && r <= 64 && N1 >= 2048 && c->Mpad >= 2048 && env_switch("ALGP_TAIL_COLS", true))
        pn.short_rows = pn.z_row >= 0 ? c->Mpad - NB : c->Mpad;
    ALGP_TRY(factorize(c, 0, &pn));
    if (!pn.done) {
        if (pn.z_row >= 0) ALGP_HIP(hipMemsetAsync(p(c->Vt) + pn.z_row * c->ldv, 0, sizeof(T) * c->Npad, c->stream));
        ALGP_TRY(solve_run(c, pl));
    } else if (pn.short_rows > 0) {
        ALGP_TRY(tail_cols_launch<T>(c, ALGP_PROF_TAIL_COLS, p(c->Vt), pn.short_rows, c->ldv, p(c->L), c->Lld, c->Npad,
                                     p(c->invD) + (N1 / NB) * NB * NB, N1, (int)round_up(r, 16)));
    }
    return solve_finish(c, 0, nullptr, pl);
}


template <typename T>
int Impl<T>::get_posterior(algp_ctx* c, void* mu, void* var) {
    if (!c->solved) return fail(c, ALGP_ERR_STATE, "get_posterior: call algp_solve_candidates first");
    ALGP_TRY(flush_lazy(c));
    if (mu) ALGP_HIP(hipMemcpyAsync(mu, c->mu.p, sizeof(T) * c->M, hipMemcpyDeviceToHost, c->stream));
    if (var) ALGP_HIP(hipMemcpyAsync(var, c->dstat.p, sizeof(T) * c->M, hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}


template <typename T>
int Impl<T>::get_posterior_cov(algp_ctx* c, void* cov_out, double* mi_out) {
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


template <typename T>
int Impl<T>::posterior_mean(algp_ctx* c, const int64_t* idx, int64_t M, void* mu_out) {
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

template struct Impl<float>;
template struct Impl<double>;

}  // namespace algp

extern "C" {

int algp_set_candidates(algp_ctx* c, const int64_t* idx, int64_t M, int prior_noise, const void* extra) {
    CHECK_CTX(c);
    if (c->n_pool <= 0) return fail(c, ALGP_ERR_STATE, "set_candidates: set a pool first");
    if (M < 0 || (M > 0 && !idx)) return fail(c, ALGP_ERR_BAD_ARG, "set_candidates: bad arguments");
    for (int64_t i = 0; i < M; ++i)
        if (idx[i] < 0 || idx[i] >= c->n_pool) return fail(c, ALGP_ERR_BAD_ARG, "set_candidates: index outside the pool");
    FINISH(c, DISPATCH(c, set_candidates(c, idx, M, prior_noise, extra)));
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

}  // extern "C"
