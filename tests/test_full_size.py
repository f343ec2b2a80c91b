"""GPU tests at BASELINE.json's sizes through size-independent properties (the oracle cannot
finish these in seconds), plus sampled direct checks against the oracle's formulas.

  C2  2 000-point field, fp64: kernel build + Cholesky + posterior
  C3 10 000-point field, fp32: blocked MFMA Cholesky
  C4 10 000 train x 100 000 candidates (fp64), greedy scoring (also what bench.py times)
"""
import numpy as np
import pytest

from algp_amd import _hip
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def field(n_side_r, n_side_c, seed=1):
    rng = np.random.RandomState(seed)
    grid, y = O.generate_gaussian_data(n_side_r, n_side_c, k=5, rng=rng)
    return grid.astype(np.float64), y, rng


HYP = O.Hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))


def test_c2_2000_points_fp64_against_oracle():
    X, f, rng = field(50, 40)
    n = len(X)
    perm = rng.permutation(n)
    A, T = np.sort(perm[:1600]), np.sort(perm[1600:])
    var = rng.choice([0.01, 1.0], len(A))
    y = np.maximum(f[A] + rng.standard_normal(len(A)) * np.sqrt(var), 0)
    c = _hip.Context(np.float64)
    c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
    K = c.kernel_matrix(X)                                    # a1 at full size: 2000 x 2000
    assert np.max(np.abs(K - O.kernel_matrix(HYP, X))) < 1e-13
    c.set_pool(X)
    c.set_train(A, y, var)
    c.factorize()
    c.set_candidates(T, prior_includes_noise=False)
    c.solve_candidates()
    mu, pv = c.posterior()
    ref = O.posterior_chol(HYP, X[A], y, X[T], var)           # 1600^3: still seconds on the CPU
    assert np.max(np.abs(mu - ref['mu'])) / np.max(np.abs(ref['mu'])) < 1e-9     # north_star: 1e-5
    assert np.max(np.abs(pv - ref['var'])) / np.max(np.abs(ref['var'])) < 1e-9
    assert c.logdet() == pytest.approx(ref['logdet'], rel=1e-11)
    L = c.factor()
    S = O.kernel_matrix(HYP, X[A]) + np.diag(var) + HYP.noise * np.eye(len(A))
    assert np.max(np.abs(L @ L.T - S)) < 1e-12
    c.close()


def test_c3_10000_points_fp32_cholesky_properties():
    """fp32 Cholesky at N = 10 000: residual |L L^T v - S v| on random probes, log-det against the
    fp64 context, posterior within 1e-3 of the fp64 context AND of the oracle's posterior (north_star fp32 tolerance)."""
    X, f, rng = field(100, 100)
    n = len(X)
    var = rng.choice([0.01, 1.0], n)
    y = np.maximum(f + rng.standard_normal(n) * np.sqrt(var), 0)
    T = np.arange(0, n, 97)
    Xt = X[T] + 0.37
    pool = np.vstack([X, Xt])
    out = {}
    for dt in (np.float32, np.float64):
        c = _hip.Context(dt)
        c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
        c.set_pool(pool)
        c.set_train(np.arange(n), y, var)
        c.factorize()
        c.set_candidates(np.arange(n, n + len(T)), prior_includes_noise=False)
        c.solve_candidates()
        mu, pv = c.posterior()
        out[dt] = (c.logdet(), mu.astype(np.float64), pv.astype(np.float64), c.factor() if dt == np.float32 else None)
        c.close()
    ld32, mu32, pv32, L32 = out[np.float32]
    ld64, mu64, pv64, _ = out[np.float64]
    assert ld32 == pytest.approx(ld64, rel=1e-4)
    assert np.max(np.abs(mu32 - mu64)) / np.max(np.abs(mu64)) < 1e-3
    assert np.max(np.abs(pv32 - pv64)) / np.max(np.abs(pv64)) < 1e-3
    # and both against the oracle's posterior at the full size (one 10 000^3 / 3 Cholesky on the CPU: seconds with a
    # threaded BLAS): north_star's tolerances, 1e-3 for fp32 and 1e-5 for fp64 (held at 1e-8 here)
    ref = O.posterior_chol(HYP, X, y, Xt, var)
    assert np.max(np.abs(mu32 - ref['mu'])) / np.max(np.abs(ref['mu'])) < 1e-3
    assert np.max(np.abs(pv32 - ref['var'])) / np.max(np.abs(ref['var'])) < 1e-3
    assert np.max(np.abs(mu64 - ref['mu'])) / np.max(np.abs(ref['mu'])) < 1e-8
    assert np.max(np.abs(pv64 - ref['var'])) / np.max(np.abs(ref['var'])) < 1e-8
    assert ld64 == pytest.approx(ref['logdet'], rel=1e-10)
    # residual through random probes (no 10k^3 product on the CPU): S v vs L (L^T v)
    S = O.kernel_matrix(HYP, X) + np.diag(var) + HYP.noise * np.eye(n)
    V = rng.standard_normal((n, 4))
    L = L32.astype(np.float64)
    res = L @ (L.T @ V) - S @ V
    assert np.max(np.abs(res)) / np.max(np.abs(S @ V)) < 5e-6
    assert np.all(np.diag(L32) > 0) and np.array_equal(np.triu(L32, 1), np.zeros_like(L32))


def test_c4_10000_x_100000_scoring_properties():
    """The bench workload.  Properties: (i) a sample of candidates against the direct formula
    pv = C_jj - b' S^-1 b computed from the fp64 factor on the host, (ii) utilities are
    monotone: every committed pick lowers every other candidate's utility, (iii) each pick is
    the argmax of its utility vector and is never picked twice, (iv) the appended-row updates
    equal a from-scratch refactorisation with the picks added to the train set."""
    X, f, rng = field(100, 100)
    N = len(X)
    M = 100000
    ii, jj = np.meshgrid(np.arange(400), np.arange(250), indexing='ij')
    Xc = np.vstack([(ii.ravel() + 0.37) * 0.25, (jj.ravel() + 0.41) * 0.4]).T[:M]
    pool = np.vstack([X, Xc])
    is_static = rng.uniform(size=N) < 0.5
    var = np.where(is_static, 0.01, 1.0)
    y = np.maximum(f + rng.standard_normal(N) * np.sqrt(var), 0)
    c = _hip.Context(np.float64)
    c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
    c.set_pool(pool)
    c.set_train(np.arange(N), y, var)
    c.factorize()
    cand = np.arange(N, N + M)
    c.set_candidates(cand, prior_includes_noise=True)
    c.solve_candidates()
    mu, pv = c.posterior()
    # (i) sampled direct check
    from scipy.linalg import solve_triangular
    L = c.factor()
    samp = rng.permutation(M)[:64]
    B = O.kernel_matrix(HYP, X, Xc[samp])
    V = solve_triangular(L, B, lower=True)
    pv_want = HYP.outputscale + HYP.noise - np.sum(V * V, axis=0)
    z = solve_triangular(L, y - y.mean(), lower=True)
    assert np.max(np.abs(pv[samp] - pv_want)) < 1e-10
    assert np.max(np.abs(mu[samp] - (y.mean() + V.T @ z))) < 1e-9
    assert np.all(pv > 0) and np.all(pv <= HYP.outputscale + HYP.noise + 1e-12)
    # (ii)-(iii)
    picks, ut = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4, want_utilities=True)
    assert len(set(picks.tolist())) == 4
    for p in range(4):
        assert cand[int(np.argmax(ut[p]))] == picks[p]
        if p:
            alive = np.isfinite(ut[p])
            assert np.all(ut[p][alive] <= ut[p - 1][alive] + 1e-12)
            assert np.isneginf(ut[p][picks[p - 1] - N])
    assert np.allclose(ut[0], O.CONST + 0.5 * np.log(pv + 0.01), rtol=0, atol=1e-12)
    # (iv) refactorise with the first 3 picks as static samples; utilities must match pick 4's
    A2 = np.r_[np.arange(N), picks[:3]]
    c.set_train(A2, np.r_[y, np.zeros(3)], np.r_[var, np.full(3, 0.01)])
    c.factorize()
    keep = np.setdiff1d(cand, picks[:3])
    sub = keep[rng.permutation(len(keep))[:20000]]
    c.set_candidates(np.sort(sub), prior_includes_noise=True)
    c.solve_candidates()
    s = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
    assert np.max(np.abs(s - ut[3][np.sort(sub) - N])) < 1e-9
    c.close()


def test_c4_one_rank_of_two_is_one_launch_at_the_task_lists_upper_end():
    """The share of a rank of TWO: 50 000 of config 4's candidates = 391 tile rows, close to the 400 the task list carries
    (beyond it: the three-stream sweep).  algp_fit_and_solve is one launch; against SciPy's triangular solves from the
    factor on sampled candidates, and bit for bit against algp_factorize + algp_solve_candidates (the same tile
    arithmetic in the same order, as two task lists)."""
    from scipy.linalg import solve_triangular
    X, f, rng = field(100, 100)
    N, M = len(X), 50000
    ii, jj = np.meshgrid(np.arange(400), np.arange(250), indexing='ij')
    Xc = np.vstack([(ii.ravel() + 0.37) * 0.25, (jj.ravel() + 0.41) * 0.4]).T[50000:50000 + M]
    var = np.where(rng.uniform(size=N) < 0.5, 0.01, 1.0)
    y = np.maximum(f + rng.standard_normal(N) * np.sqrt(var), 0)
    c = _hip.Context(np.float64)
    c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
    c.set_pool(np.vstack([X, Xc]))
    c.set_train(np.arange(N), y, var)
    c.set_candidates(np.arange(N, N + M), prior_includes_noise=True)
    c.prof_enable(True)
    c.prof_reset()
    c.fit_and_solve()
    assert c.prof_get('dag_panel')['launches'] == 1 and c.prof_get('gemm_trsm')['launches'] == 0
    c.prof_enable(False)
    mu, pv = c.posterior()
    L = c.factor()
    samp = rng.permutation(M)[:64]
    V = solve_triangular(L, O.kernel_matrix(HYP, X, Xc[samp]), lower=True)
    z = solve_triangular(L, y - y.mean(), lower=True)
    assert np.max(np.abs(pv[samp] - (HYP.outputscale + HYP.noise - np.sum(V * V, axis=0)))) < 1e-10
    assert np.max(np.abs(mu[samp] - (y.mean() + V.T @ z))) < 1e-9
    assert np.all(pv > 0) and np.all(pv <= HYP.outputscale + HYP.noise + 1e-12)
    c.factorize()
    c.solve_candidates()
    mu2, pv2 = c.posterior()
    # the same tile arithmetic for V^T either way; z = L^-1 (y - ybar) rides along as a row of the panel in the one launch
    # (tile products) where the two-call form substitutes: the mean agrees to rounding
    assert np.array_equal(pv2, pv) and np.max(np.abs(mu2 - mu)) < 1e-12
    c.close()


def test_c4_one_rank_of_eight_fit_and_solve_in_one_launch_and_remote_commits():
    """BASELINE config 4 as ONE of its eight ranks sees it: N = 10 000 train points, the rank's 12 500 of the 100 000
    candidates.  (i) algp_fit_and_solve -- factor and V^T out of one task-list launch -- against SciPy's triangular solves
    from the factor on sampled candidates (the 1e-5 bar of north_star with five digits to spare) and, to rounding, against
    the two-phase path; (ii) algp_greedy_sharded as rank 5 of 8 over a host gather that fabricates the seven absent ranks
    (their true winners: utility, statistic and row of V^T taken from a one-rank run over all candidates, agent.py:313-354
    being one loop over them): the same four picks, and after the four commits -- two of them remote, rows copied out of the
    gather -- every utility of the shard equals the one-rank run's to rounding."""
    import struct
    from scipy.linalg import solve_triangular
    from algp_amd.sharded import partition
    X, f, rng = field(100, 100)
    N, M = len(X), 100000
    ii, jj = np.meshgrid(np.arange(400), np.arange(250), indexing='ij')
    Xc = np.vstack([(ii.ravel() + 0.37) * 0.25, (jj.ravel() + 0.41) * 0.4]).T[:M]
    pool = np.vstack([X, Xc])
    is_static = rng.uniform(size=N) < 0.5
    var = np.where(is_static, 0.01, 1.0)
    y = np.maximum(f + rng.standard_normal(N) * np.sqrt(var), 0)
    cand = np.arange(N, N + M)
    c = _hip.Context(np.float64)
    c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
    c.set_pool(pool)
    c.set_train(np.arange(N), y, var)
    # the one-rank run: picks, utilities, and what each winner's owner would put into the gather
    c.set_candidates(cand, prior_includes_noise=True)
    c.fit_and_solve()
    picks, ut = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4, want_utilities=True)
    picks = [int(q) for q in picks]
    util = [float(np.nanmax(ut[q])) for q in range(4)]
    rows = [c.debug_get_pick(q) for q in range(4)]
    u_after = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
    del ut
    world, r = 8, 5
    parts = partition(M, world)
    lo, hi = parts[r]
    owners = [next(s for s, (a, b) in enumerate(parts) if a <= (q - N) < b) for q in picks]
    assert any(o != r for o in owners)
    c.set_candidates(cand[lo:hi], prior_includes_noise=True)
    c.prof_enable(True)
    c.prof_reset()
    c.fit_and_solve()
    assert c.prof_get('dag_panel')['launches'] == 1 and c.prof_get('gemm_trsm')['launches'] == 0
    c.prof_enable(False)
    mu, pv = c.posterior()
    L = c.factor()
    samp = rng.permutation(hi - lo)[:64]
    B = O.kernel_matrix(HYP, X, Xc[lo:hi][samp])
    V = solve_triangular(L, B, lower=True)
    z = solve_triangular(L, y - y.mean(), lower=True)
    assert np.max(np.abs(pv[samp] - (HYP.outputscale + HYP.noise - np.sum(V * V, axis=0)))) < 1e-10
    assert np.max(np.abs(mu[samp] - (y.mean() + V.T @ z))) < 1e-9
    assert np.all(pv > 0) and np.all(pv <= HYP.outputscale + HYP.noise + 1e-12)
    c.factorize()
    c.solve_candidates()
    mu2, pv2 = c.posterior()
    assert np.max(np.abs(mu2 - mu)) < 1e-10 and np.max(np.abs(pv2 - pv)) < 1e-10
    # (ii) rank 5 of 8
    state = {'q': 0}

    def gather(send):
        pb = len(send)
        q = min(state['q'], 3)
        buf = bytearray(world * pb)
        for s_ in range(world):
            buf[s_ * pb:s_ * pb + 32] = struct.pack('<4d', float('-inf'), -1.0, 0.0, 0.0)
        if owners[q] != r:
            row, d = rows[q]
            o = owners[q] * pb
            buf[o:o + 32] = struct.pack('<4d', util[q], float(picks[q]), 0.0, d)
            buf[o + 32:o + 32 + row.nbytes] = row.tobytes()
        buf[r * pb:(r + 1) * pb] = send
        if struct.unpack_from('<d', send, 16)[0] == 0.0:
            state['q'] += 1
        return bytes(buf)

    c.comm_init_host(world, r, gather)
    try:
        c.fit_and_solve()
        got, gut = c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 4, want_utilities=True)
        assert [int(q) for q in got] == picks
        assert np.max(np.abs(np.asarray(gut) - np.asarray(util))) < 1e-11
        u_sh = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
    finally:
        c.comm_destroy()
    ref = u_after[lo:hi]
    fin = np.isfinite(ref)
    assert np.array_equal(fin, np.isfinite(u_sh)) and np.max(np.abs(u_sh[fin] - ref[fin])) < 1e-10
    c.close()


def test_c5_50000_points_fp64_consistency():
    """C5 size on one GPU (L = 20 GB): two independent routes to the posterior mean agree
    (V^T z after the blocked TRSM vs the fused kernel-GEMV with alpha = L^-T z), variances are
    inside (0, prior], and appending sites through algp_factorize_update reproduces the log-det
    increment given by the Schur complement of the appended block."""
    X, f, rng = field(250, 200)
    n = len(X)
    assert n == 50000
    var = rng.choice([0.01, 1.0], n)
    y = np.maximum(f + rng.standard_normal(n) * np.sqrt(var), 0)
    Xt = X[rng.permutation(n)[:700]] + 0.41
    Xnew = X[rng.permutation(n)[:40]] + np.array([0.23, 0.61])
    pool = np.vstack([X, Xt, Xnew])
    c = _hip.Context(np.float64)
    c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
    c.set_pool(pool)
    c.set_train(np.arange(n), y, var)
    c.factorize()
    ld0 = c.logdet()
    tidx = np.arange(n, n + 700)
    c.set_candidates(tidx, prior_includes_noise=False)
    c.solve_candidates()
    mu, pv = c.posterior()
    mu2 = c.posterior_mean(tidx)
    assert np.max(np.abs(mu - mu2)) < 1e-7 * max(1.0, np.max(np.abs(mu)))
    assert np.all(pv > 0) and np.all(pv <= HYP.outputscale + 1e-12)
    # append 40 sites: log det grows by log det of their posterior covariance + noise
    new = np.arange(n + 700, n + 740)
    c.set_candidates(new, prior_includes_noise=False)
    c.solve_candidates()
    cov, _ = c.posterior_cov()
    want = np.linalg.slogdet(cov + (HYP.noise + 0.01) * np.eye(40))[1]
    c.set_train(np.r_[np.arange(n), new], np.r_[y, np.zeros(40)], np.r_[var, np.full(40, 0.01)])
    kept = c.factorize(incremental=True)
    assert kept == 49920
    assert c.logdet() - ld0 == pytest.approx(want, rel=1e-8, abs=1e-8)
    c.close()


def test_c5_one_rank_of_eight_of_the_loop_at_full_size():
    """BASELINE config 5 in its stated form as ONE of its eight ranks sees it, at full size: N0 = 50 000 train rows (replicated
    factor), the rank's 12 500 of the 100 000 candidates, three planning steps of the active-learning loop (reference
    agent.py:125-229 with the candidate loop of :313-354 sharded) through algp_factorize_update (collective: agreement word +
    the row exchange) -> algp_solve_candidates_update -> algp_greedy_sharded, the seven absent ranks fabricated from a
    one-rank context that runs the same loop in lockstep (bench.c5_rank_of_8: the rows of L the other ranks own are the
    one-rank factor's own new rows, the winners' rows its picks').  bench.c5_rank_of_8 itself asserts, at every step, that the
    rank's picks are the one-rank loop's and that no factor update fell back to the triangular solve; here also: every
    update went through the exchange (the step's time is printed, not asserted)."""
    import bench
    out = bench.c5_rank_of_8(_hip, 0, 4, steps=3, ranks=(5,))
    r = out['ranks']['5']
    assert r['row_exchanges'] == 3 and r['fallbacks'] == 0 and r['candidates'] == 12500
    assert out['new_train_rows_per_step_median'] >= 28 and 1 <= out['rows_per_rank_in_the_exchange_median'] <= 16
    print('c5 rank 5 of 8: %.2f ms per step (median of 3; reported, not asserted: a parity suite must not turn red on a slow box)' % r['ms_per_step_median'])
