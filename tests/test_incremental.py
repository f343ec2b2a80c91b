"""f1: incremental factor maintenance.  algp_factorize_update must give the same
factor / alpha / log-det as a from-scratch factorisation while reusing the unchanged leading rows."""
import os

import numpy as np
import pytest

from algp_amd import _hip
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

HYP = O.Hypers(np.log([2.5, 2.0]), np.log(1.2), np.log(0.02))


def setup(dt=np.float64, n=2500, seed=0):
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 40, (n, 2))
    y = np.sin(X[:, 0] / 4) + np.cos(X[:, 1] / 5) + 0.1 * rng.standard_normal(n)
    var = rng.choice([0.01, 1.0], n)
    c = _hip.Context(dt)
    c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
    c.set_pool(X)
    return c, X, y, var, rng


def fresh(X, idx, y, var, dt=np.float64):
    c = _hip.Context(dt)
    c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
    c.set_pool(X)
    c.set_train(idx, y, var)
    c.factorize()
    out = (c.factor(), c.alpha(), c.logdet(), c.mll())
    c.close()
    return out


def same(c, ref, tol=1e-10):
    L, a, ld, mll = ref
    assert np.max(np.abs(c.factor() - L)) < tol
    assert np.max(np.abs(c.alpha() - a)) / np.max(np.abs(a)) < 1e-7
    assert c.logdet() == pytest.approx(ld, rel=1e-11, abs=1e-9)
    assert c.mll() == pytest.approx(mll, rel=1e-10)


def test_append_reuses_leading_rows_and_matches_scratch():
    c, X, y, var, rng = setup()
    perm = rng.permutation(len(X))
    idx = perm[:1000]
    c.set_train(idx, y[idx], var[idx])
    assert c.factorize(incremental=True) == 0               # nothing resident yet
    # append in several steps, including crossing 128-boundaries and the allocated capacity
    for add in (1, 50, 77, 128, 300, 700):
        idx = np.r_[idx, perm[len(idx):len(idx) + add]]
        c.set_train(idx, y[idx], var[idx])
        kept = c.factorize(incremental=True)
        assert kept == (len(idx) - add) // 128 * 128, (kept, len(idx), add)
        same(c, fresh(X, idx, y[idx], var[idx]))
    # the posterior built on the updated factor is right as well
    test = perm[-200:]
    c.set_candidates(test, prior_includes_noise=False)
    c.solve_candidates()
    mu, pv = c.posterior()
    ref = O.posterior_chol(HYP, X[idx], y[idx], X[test], var[idx])
    assert np.max(np.abs(mu - ref['mu'])) < 1e-8 and np.max(np.abs(pv - ref['var'])) < 1e-9
    c.close()


def test_noise_change_and_reorder_rebuild_from_the_first_changed_block():
    c, X, y, var, rng = setup(n=1200)
    idx = np.arange(1000)
    c.set_train(idx, y[idx], var[idx])
    c.factorize()
    v2 = var[idx].copy()
    v2[700] = 0.123                                         # re-measured site: its noise changes
    c.set_train(idx, y[idx], v2)
    assert c.factorize(incremental=True) == 640              # 700 // 128 * 128
    same(c, fresh(X, idx, y[idx], v2))
    idx2 = idx.copy()
    idx2[[300, 301]] = idx2[[301, 300]]                      # order matters: rows are positions
    c.set_train(idx2, y[idx2], v2[[*range(300), 301, 300, *range(302, 1000)]])
    assert c.factorize(incremental=True) == 256
    same(c, fresh(X, idx2, y[idx2], v2[[*range(300), 301, 300, *range(302, 1000)]]))
    # only the targets change: everything is kept, alpha follows the new y
    y2 = y[idx2] + 1.0
    c.set_train(idx2, y2, v2[[*range(300), 301, 300, *range(302, 1000)]])
    assert c.factorize(incremental=True) == 896
    same(c, fresh(X, idx2, y2, v2[[*range(300), 301, 300, *range(302, 1000)]]))
    # shrinking the set keeps the common prefix
    c.set_train(idx2[:500], y2[:500], v2[[*range(300), 301, 300, *range(302, 1000)]][:500])
    assert c.factorize(incremental=True) == 384
    same(c, fresh(X, idx2[:500], y2[:500], v2[[*range(300), 301, 300, *range(302, 1000)]][:500]))
    c.close()


def test_hyper_or_pool_change_invalidates_the_factor():
    c, X, y, var, rng = setup(n=600)
    idx = np.arange(400)
    c.set_train(idx, y[idx], var[idx])
    c.factorize()
    c.set_hypers(HYP.log_lengthscale + 0.1, HYP.log_outputscale, HYP.log_noise)
    c.set_train(idx, y[idx], var[idx])
    assert c.factorize(incremental=True) == 0
    c.set_pool(X + 0.5)
    c.set_train(idx, y[idx], var[idx])
    assert c.factorize(incremental=True) == 0
    c.close()


def test_not_pd_in_the_appended_part_reports_global_pivot():
    c, X, y, var, rng = setup(n=600)
    idx = np.arange(300)
    c.set_train(idx, y[idx], var[idx])
    c.factorize()
    idx2 = np.r_[idx, 300, 301]
    v = np.r_[var[idx], 0.01, -50.0]                         # a negative "variance" breaks positivity
    c.set_train(idx2, y[idx2], v)
    with pytest.raises(np.linalg.LinAlgError) as ei:
        c.factorize(incremental=True)
    assert ei.value.pivot == 302
    c.close()


def test_fp32_append_within_tolerance():
    c, X, y, var, rng = setup(np.float32, n=1500)
    idx = np.arange(900)
    c.set_train(idx, y[idx], var[idx])
    c.factorize()
    idx = np.arange(1100)
    c.set_train(idx, y[idx], var[idx])
    assert c.factorize(incremental=True) == 896
    L, a, ld, mll = fresh(X, idx, y[idx], var[idx], np.float32)
    assert np.max(np.abs(c.factor() - L)) < 1e-4
    assert c.logdet() == pytest.approx(ld, rel=1e-5)
    c.close()


def test_candidate_solve_reuses_columns_and_matches_scratch():
    c, X, y, var, rng = setup(n=3000)
    perm = rng.permutation(len(X))
    idx = perm[:900]
    cand = np.sort(perm[900:])                               # fixed candidate list over the steps
    static = np.zeros(len(X), bool)
    v = np.full(len(idx), 0.01)
    c.set_train(idx, y[idx], v)
    c.factorize()
    c.set_candidates(cand, prior_includes_noise=True)
    assert c.solve_candidates(incremental=True) == 0
    for step in range(4):
        picks = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4)
        static[picks] = True
        # the picked sites get static readings and join the train set at its END (insertion order);
        # a few more candidates get MOBILE readings: they stay candidates but turn into unit rows
        mob = [int(m) for m in cand[rng.permutation(len(cand))[:5]] if m not in idx and m not in picks]
        idx = np.r_[idx, picks, mob].astype(np.int64)
        v = np.r_[v if step else np.full(900, 0.01), np.full(4, 0.01), np.full(len(mob), 1.0)]
        c.set_train(idx, y[idx], v)
        kept_rows = c.factorize(incremental=True)
        alive = ~static[cand]
        kept_cols = c.solve_candidates(incremental=True, alive=alive)
        assert kept_rows == (len(idx) - 4 - len(mob)) // 128 * 128 and kept_cols == kept_rows
        s = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
        # from scratch in a new context with the static sites removed from the candidate list
        f = _hip.Context(np.float64)
        f.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
        f.set_pool(X)
        f.set_train(idx, y[idx], v)
        f.factorize()
        f.set_candidates(cand[alive], prior_includes_noise=True)
        f.solve_candidates()
        want = f.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
        assert np.all(np.isneginf(s[~alive])) and not np.any(np.isnan(s))
        assert np.max(np.abs(s[alive] - want)) < 1e-10
        mu, pv = c.posterior()
        mu2, pv2 = f.posterior()
        assert np.max(np.abs(mu[alive] - mu2)) < 1e-9 and np.max(np.abs(pv[alive] - pv2)) < 1e-10
        f.close()
    # a different candidate list cannot reuse anything
    c.set_candidates(cand[:-1], prior_includes_noise=True)
    assert c.solve_candidates(incremental=True) == 0
    c.close()


@pytest.mark.parametrize('dtname', ['f64', 'f32'])
def test_only_the_new_columns_are_solved_after_an_append(dtname, monkeypatch):
    """After rows were appended behind >= 2 048 unchanged ones, algp_solve_candidates_update solves only the columns the
    append added (tail.hip: exactly the appended columns when there are at most 64 -- any first column, block boundaries
    inside the range included) instead of the whole open 128-block.  Appends chosen to hit: one range whose first column is 16 (mod 32) -- the fp32
    kernel's half k-tile --, two ranges across a block boundary with the first 64 wide, an append too wide for it (the
    128-blocks again), a range ending on a block boundary, and a range of <= 64 columns that STRADDLES a block boundary
    (one pass over V^T with the inverse of the window of L at its first column; two passes before round 5).  Every step
    against a from-scratch context, against the 128-block order ($ALGP_TAIL_COLS=0) on a copy of the state, and -- a tail
    step's own anchor -- against the NumPy oracle's from-scratch posterior (O.posterior_chol, utils.py:293-319) on 200
    sampled candidates."""
    dt = np.float64 if dtname == 'f64' else np.float32
    tol, loose = (1e-9, 1e-10) if dt == np.float64 else (3e-3, 3e-4)
    rng = np.random.RandomState(4)
    n, M = 2600, 4300
    X = rng.uniform(0, 60, (n + M, 2))
    y = np.sin(X[:, 0] / 4) + np.cos(X[:, 1] / 5) + 0.1 * rng.standard_normal(n + M)
    cand = np.arange(n, n + M)

    def ctx():
        c = _hip.Context(dt)
        c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
        c.set_pool(X)
        return c

    c, b = ctx(), ctx()                                        # b: the same steps with the 128-block order
    idx = np.arange(2100)
    v = rng.choice([0.01, 1.0], n)
    for k, c_ in enumerate((c, b)):
        c_.set_train(idx, y[idx], v[idx])
        c_.factorize(incremental=True)
        c_.set_candidates(cand, prior_includes_noise=False)
        assert c_.solve_candidates(incremental=True) == 0
    expect = [(20, 1, 2100), (60, 1, 2120), (70, 0, 2176), (50, 1, 2250), (14, 1, 2300)]      # rows added, tail launches, kept columns
    for add, launches, kept_want in expect:
        idx = np.arange(len(idx) + add)
        for which, c_ in (('tail', c), ('blocks', b)):
            if which == 'blocks':
                monkeypatch.setenv('ALGP_TAIL_COLS', '0')
            c_.set_train(idx, y[idx], v[idx])
            c_.factorize(incremental=True)
            c_.set_candidates(cand, prior_includes_noise=False)
            c_.prof_enable(True)
            c_.prof_reset()
            kept = c_.solve_candidates(incremental=True)
            got, gemms = c_.prof_get('tail_cols')['launches'], c_.prof_get('gemm_trsm')['launches']
            c_.prof_enable(False)
            if which == 'blocks':
                monkeypatch.delenv('ALGP_TAIL_COLS')
                assert kept == (len(idx) - add) // 128 * 128
            else:
                assert kept == kept_want, (add, kept)
                assert got == launches and (gemms == 0) == (launches > 0), (add, got, gemms)     # 0: too wide, the 128-blocks
        f = ctx()
        f.set_train(idx, y[idx], v[idx])
        f.factorize()
        f.set_candidates(cand, prior_includes_noise=False)
        f.solve_candidates()
        mu_f, pv_f = f.posterior()
        f.close()
        (mu, pv), (mu_b, pv_b) = c.posterior(), b.posterior()
        scale = max(1.0, np.max(np.abs(mu_f)))
        assert np.max(np.abs(mu - mu_f)) < tol * scale and np.max(np.abs(pv - pv_f)) < tol, (add, 'tail vs scratch')
        assert np.max(np.abs(mu - mu_b)) < loose * scale and np.max(np.abs(pv - pv_b)) < loose, (add, 'tail vs blocks')
        samp = rng.permutation(M)[:200]
        o = O.posterior_chol(HYP, X[idx], y[idx], X[cand[samp]], v[idx])
        otol = 1e-8 if dt == np.float64 else 3e-3
        assert np.max(np.abs(mu[samp] - o['mu'])) < otol * scale and np.max(np.abs(pv[samp] - o['var'])) < otol, (add, 'tail vs oracle')
    c.close()
    b.close()


@pytest.mark.parametrize('dtname', ['f64', 'f32'])
def test_factor_update_takes_new_rows_from_resident_candidates(dtname):
    """New train sites that are resident (ordinary) candidates: their rows of L left of the tail block are the
    leading parts of their rows of V^T, so the update gathers them instead of solving against the kept
    factor.  Same factor as from scratch; the triangular solve's GEMM launches must be gone.  The factor
    grows across a 128 boundary AND its buffer is re-allocated (rows of the partial last block survive)."""
    dt = np.float64 if dtname == 'f64' else np.float32
    tol = 1e-10 if dtname == 'f64' else 2e-3
    rng = np.random.RandomState(2)
    N0, M = 1000, 3000
    X = rng.uniform(0, 60, (N0 + M, 2))
    hyp = (np.log([3.0, 2.0]), 0.0, np.log(1e-2))
    c = _hip.Context(dt)
    c.set_hypers(*hyp)
    c.set_pool(X)
    idx, var = np.arange(N0), rng.choice([0.01, 1.0], N0)
    cand = np.arange(N0 - 20, N0 + M)                       # 20 train sites are candidates too (unit rows)
    c.set_train(idx, np.zeros(N0), var)
    c.factorize(incremental=True)
    c.set_candidates(cand, prior_includes_noise=True)
    c.solve_candidates(incremental=True)
    c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 3)                # appended columns / stale rows must not matter
    for step in range(4):
        new = np.setdiff1d(cand[rng.permutation(len(cand))[:60]], idx)[:40]
        nv = rng.choice([0.01, 1.0], len(new))
        if step >= 1:
            # second measurements: sites that already are train rows (and unit-row candidates) get another row --
            # old ones (rows < kept) and one from the previous step's appends
            again = np.r_[np.arange(N0 - 20, N0 - 20 + 3 * step), idx[-5]]
            again = np.array([a for a in again if np.sum(idx == a) == 1])
            new = np.r_[new[:10], again, new[10:]]
            nv = np.r_[nv[:10], np.full(len(again), 0.01), nv[10:]]
        idx = np.r_[idx, new]
        var = np.r_[var, nv]
        c.set_train(idx, np.zeros(len(idx)), var)
        c.prof_enable(True)
        c.prof_reset()
        kept = c.factorize(incremental=True)
        launches = c.prof_get('gemm_chol')['launches']
        c.prof_enable(False)
        assert kept == (len(idx) - len(new)) // 128 * 128
        if os.environ.get('ALGP_FACTOR_FROM_VT') != '0':     # (the switch that disables the shortcut, for A/B timing)
            assert launches <= 8, launches                  # Schur product + the small tail factorisation only
        ref = _hip.Context(dt)
        ref.set_hypers(*hyp)
        ref.set_pool(X)
        ref.set_train(idx, np.zeros(len(idx)), var)
        ref.factorize()
        L1, L2 = np.tril(c.factor()), np.tril(ref.factor())
        assert np.max(np.abs(L1 - L2)) < tol * np.max(np.abs(L2))
        assert abs(c.logdet() - ref.logdet()) < tol * abs(ref.logdet())
        if dtname == 'f64':                                 # and against NumPy on the explicit matrix (repeated sites: the
            h = O.Hypers(*hyp)                              # cross entry of two rows of one site is C(i,i) = k + sigma_n^2)
            S = O.kernel_matrix(h, X[idx]) + h.noise * (idx[:, None] == idx[None, :]) + np.diag(var)
            assert np.max(np.abs(L1 - np.linalg.cholesky(S))) < 1e-11
        ref.close()
        c.set_candidates(cand, prior_includes_noise=True)
        c.solve_candidates(incremental=True)
    # a new site that is NOT a candidate falls back to the triangular solve (and is still right)
    c.set_candidates(cand[:-50], prior_includes_noise=True)
    c.solve_candidates(incremental=True)
    idx = np.r_[idx, cand[-1]]
    var = np.r_[var, 0.01]
    c.set_train(idx, np.zeros(len(idx)), var)
    c.factorize(incremental=True)
    ref = _hip.Context(dt)
    ref.set_hypers(*hyp)
    ref.set_pool(X)
    ref.set_train(idx, np.zeros(len(idx)), var)
    ref.factorize()
    assert np.max(np.abs(np.tril(c.factor()) - np.tril(ref.factor()))) < tol
    ref.close()
    c.close()


@pytest.mark.parametrize('dtname', ['f64', 'f32'])
def test_incremental_targets_mean_and_y_only_changes(dtname):
    """z = L^-1 (y - ybar) is maintained as u - ybar w (u = L^-1 y, w = L^-1 1) and only extended on a factor
    update: appended sites shift ybar (every entry of z changes), a changed target at an old site restarts the
    substitution at its block.  alpha, MLL and posterior means must equal a fresh context's."""
    dt = np.float64 if dtname == 'f64' else np.float32
    tol = 1e-9 if dtname == 'f64' else 2e-3
    rng = np.random.RandomState(9)
    N0, M = 700, 900
    X = rng.uniform(0, 40, (N0 + M, 2))
    hyp = (np.log([3.0, 2.0]), 0.0, np.log(1e-2))
    truth = 5.0 + np.sin(X[:, 0] / 4) + np.cos(X[:, 1] / 5)           # non-zero mean: ybar matters
    c = _hip.Context(dt)
    c.set_hypers(*hyp)
    c.set_pool(X)
    idx, var = np.arange(N0), rng.choice([0.01, 1.0], N0)
    y = truth[idx] + 0.1 * rng.standard_normal(N0)
    test_idx = np.arange(N0 + 600, N0 + M)

    def check():
        ref = _hip.Context(dt)
        ref.set_hypers(*hyp)
        ref.set_pool(X)
        ref.set_train(idx, y, var)
        ref.factorize()
        a1, a2 = c.alpha(), ref.alpha()
        assert np.max(np.abs(a1 - a2)) < tol * np.max(np.abs(a2))
        assert abs(c.mll() - ref.mll()) < tol * abs(ref.mll())
        m1, m2 = c.posterior_mean(test_idx), ref.posterior_mean(test_idx)
        assert np.max(np.abs(m1 - m2)) < tol * np.max(np.abs(m2))
        ref.close()

    c.set_train(idx, y, var)
    c.factorize(incremental=True)
    check()
    for step in range(3):                                   # appends (cross a 128 boundary)
        new = np.arange(N0 + 50 * step, N0 + 50 * step + 50)
        idx = np.r_[idx, new]
        var = np.r_[var, rng.choice([0.01, 1.0], 50)]
        y = np.r_[y, truth[new] + 0.1 * rng.standard_normal(50)]
        c.set_train(idx, y, var)
        assert c.factorize(incremental=True) > 0
        check()
    y = y.copy()
    y[300] += 0.7                                           # a re-measured old site: same noise, new fused target
    c.set_train(idx, y, var)
    assert c.factorize(incremental=True) == len(idx) // 128 * 128
    check()
    c.close()


def test_factorize_from_another_context():
    """algp_factorize_from: a second context (other candidates, other targets) adopts the factor of the same
    train set; a mismatch in train set or hyper-parameters is refused."""
    rng = np.random.RandomState(12)
    N, M = 600, 300
    X = rng.uniform(0, 40, (N + M, 2))
    hyp = (np.log([3.0, 2.0]), 0.0, np.log(1e-2))
    idx, var = np.arange(N), rng.choice([0.01, 1.0], N)
    y = 3.0 + np.sin(X[:N, 0] / 4) + 0.1 * rng.standard_normal(N)
    a, b, ref = (_hip.Context(np.float64) for _ in range(3))
    for c in (a, b, ref):
        c.set_hypers(*hyp)
        c.set_pool(X)
    a.set_train(idx, np.zeros(N), var)                      # the greedy-style context: targets are irrelevant there
    a.factorize()
    b.set_train(idx, y, var)
    assert b.factorize_from(a) == 0
    ref.set_train(idx, y, var)
    ref.factorize()
    assert np.array_equal(np.tril(b.factor()), np.tril(ref.factor()))
    assert b.logdet() == ref.logdet() and np.array_equal(b.alpha(), ref.alpha()) and b.mll() == ref.mll()
    test_idx = np.arange(N, N + M)
    for c in (b, ref):
        c.set_candidates(test_idx, prior_includes_noise=False)
        c.solve_candidates()
    assert all(np.array_equal(u, v) for u, v in zip(b.posterior(), ref.posterior()))
    # grow the train set in the source, adopt again: the leading rows b already has are kept
    idx2, var2 = np.r_[idx, N + np.arange(40)], np.r_[var, np.full(40, 1.0)]
    y2 = np.r_[y, 3.0 + 0.1 * rng.standard_normal(40)]
    a.set_train(idx2, np.zeros(N + 40), var2)
    a.factorize(incremental=True)
    b.set_train(idx2, y2, var2)
    assert b.factorize_from(a) == N // 128 * 128
    ref.set_train(idx2, y2, var2)
    ref.factorize()
    assert np.max(np.abs(np.tril(b.factor()) - np.tril(ref.factor()))) < 1e-11
    assert np.max(np.abs(b.alpha() - ref.alpha())) < 1e-9 * np.max(np.abs(ref.alpha()))
    # refusals
    b.set_train(idx2[:-1], y2[:-1], var2[:-1])
    with pytest.raises(ValueError):
        b.factorize_from(a)
    b.set_train(idx2, y2, var2)
    b.set_hypers(np.log([3.0, 2.5]), 0.0, np.log(1e-2))
    with pytest.raises(ValueError):
        b.factorize_from(a)
    with pytest.raises(ValueError):
        a.factorize_from(a)
    # same indices, other coordinates: refused (the pool may be larger -- only the train sites' coordinates count)
    b.set_hypers(*hyp)
    X2 = X.copy()
    X2[5] += 0.25
    b.set_pool(X2)
    b.set_train(idx2, y2, var2)
    with pytest.raises(ValueError):
        b.factorize_from(a)
    b.set_pool(np.vstack([X, rng.uniform(0, 40, (10, 2))]))
    b.set_train(idx2, y2, var2)
    assert b.factorize_from(a) >= 0
    for c in (a, b, ref):
        c.close()
