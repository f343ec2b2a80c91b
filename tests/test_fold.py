"""The row panel carried by the one-launch task list (algp_amd/csrc/chol_dag.hip, DagShape): algp_fit_and_solve puts the
rows of B^T below the factorisation as extra block rows, so V^T = B^T L^-T (reference utils.py:300-301: cov_xa @ inv(cov_aa))
comes out of the launch that factors S; a from-scratch algp_solve_candidates of 33 .. 400 tile rows runs the same list
without the factorisation's own tasks.  Both against the oracle's posterior (utils.py:293-319 as O.posterior_chol), against
the launch sequences of potrf.hip they replace, across repetitions (bit-identical: every tile receives its updates in
ascending k whatever the timing), with train-site candidates (unit right-hand sides), in fp64 and fp32."""
import time

import numpy as np
import pytest

from algp_amd import _hip
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

HYP = O.Hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))


def _field(N, M, rng, side=40):
    xx, yy = np.meshgrid(np.arange(side), np.arange(side))
    grid = np.vstack([yy.ravel(), xx.ravel()]).T.astype(np.float64)
    A = np.sort(rng.permutation(len(grid))[:N])
    cand = rng.uniform(0, side, (M, 2))
    pool = np.vstack([grid, cand])
    var = rng.choice([0.01, 1.0], N)
    y = rng.uniform(0, 1, N)
    return pool, A, y, var, np.arange(len(grid), len(grid) + M)


def _ctx(dtype, pool, A, y, var):
    c = _hip.Context(dtype)
    c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
    c.set_pool(pool)
    c.set_train(A, y, var)
    return c


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-9), (np.float32, 2e-3)], ids=['f64', 'f32'])
@pytest.mark.parametrize('N,M', [(1400, 100), (1400, 4500), (1100, 12500), (2500, 300)])
def test_fit_and_solve_in_one_launch_matches_oracle_and_the_two_phase_path(dtype, tol, N, M):
    rng = np.random.RandomState(N + M)
    pool, A, y, var, cidx = _field(N, M, rng, side=60 if N > 1600 else 40)
    samp = np.sort(rng.permutation(M)[:min(M, 160)])
    ref = O.posterior_chol(HYP, pool[A], y, pool[cidx[samp]], var)
    c = _ctx(dtype, pool, A, y, var)
    c.set_candidates(cidx, prior_includes_noise=False)
    c.prof_enable(True)
    c.prof_reset()
    c.fit_and_solve()
    assert c.prof_get('dag_panel')['launches'] == 1, 'the folded launch did not run'
    assert c.prof_get('gemm_trsm')['launches'] == 0
    c.prof_enable(False)
    mu1, pv1 = c.posterior()
    ld1 = c.logdet()
    scale = max(1.0, np.max(np.abs(ref['mu'])))
    assert np.max(np.abs(mu1[samp] - ref['mu'])) <= tol * scale
    assert np.max(np.abs(pv1[samp] - ref['var'])) <= tol * max(1.0, np.max(np.abs(ref['var'])))
    assert np.all(np.isfinite(mu1)) and np.all(np.isfinite(pv1)) and np.min(pv1) > -tol
    S = O.kernel_matrix(HYP, pool[A]) + np.diag(var + HYP.noise)
    want_ld = np.linalg.slogdet(S)[1]
    assert abs(ld1 - want_ld) <= (1e-9 if dtype == np.float64 else 2e-3) * abs(want_ld)
    # the same call again: bit-identical (deterministic update order), and the factor is the one algp_factorize computes
    L1 = c.factor()
    c.fit_and_solve()
    mu2, pv2 = c.posterior()
    assert np.array_equal(mu1, mu2) and np.array_equal(pv1, pv2) and ld1 == c.logdet()
    c.factorize()
    assert np.array_equal(L1, c.factor()), 'the panel must not change the factor'
    # the two phases apart (the solve: task list without the factorisation beyond 32 tile rows, launch sequence below)
    c.solve_candidates()
    mu3, pv3 = c.posterior()
    loose = 2e-11 if dtype == np.float64 else 2e-4           # two summation orders of the same products: rounding only
    assert np.max(np.abs(mu3 - mu1)) <= loose * scale and np.max(np.abs(pv3 - pv1)) <= loose
    c.close()


def test_solve_only_task_list_against_the_launch_sequence_and_across_runs():
    """33+ tile rows from scratch: algp_solve_candidates runs the task list (one launch); same posterior as the oracle and,
    to rounding, as a solve of fewer rows (which takes the launch sequence); repeated: the same bits."""
    rng = np.random.RandomState(11)
    N, M = 1400, 4300
    pool, A, y, var, cidx = _field(N, M, rng)
    c = _ctx(np.float64, pool, A, y, var)
    c.factorize()
    c.set_candidates(cidx, prior_includes_noise=False)
    c.prof_enable(True)
    c.prof_reset()
    c.solve_candidates()
    assert c.prof_get('dag_panel')['launches'] == 1 and c.prof_get('gemm_trsm')['launches'] == 0
    c.prof_enable(False)
    mu, pv = c.posterior()
    c.solve_candidates()
    mu_b, pv_b = c.posterior()
    assert np.array_equal(mu, mu_b) and np.array_equal(pv, pv_b)
    ref = O.posterior_chol(HYP, pool[A], y, pool[cidx[:200]], var)
    assert np.max(np.abs(mu[:200] - ref['mu'])) <= 1e-9 * max(1.0, np.max(np.abs(ref['mu'])))
    assert np.max(np.abs(pv[:200] - ref['var'])) <= 1e-9
    c.set_candidates(cidx[:4000], prior_includes_noise=False)      # 32 tile rows: the launch sequence
    c.prof_enable(True)
    c.prof_reset()
    c.solve_candidates()
    assert c.prof_get('dag_panel')['launches'] == 0 and c.prof_get('gemm_trsm')['launches'] > 0
    c.prof_enable(False)
    mu_s, pv_s = c.posterior()
    assert np.max(np.abs(mu_s - mu[:4000])) <= 1e-11 and np.max(np.abs(pv_s - pv[:4000])) <= 1e-11
    c.close()


@pytest.mark.parametrize('dtype', [np.float64, np.float32], ids=['f64', 'f32'])
def test_folded_launch_with_train_site_candidates_and_greedy_picks(dtype):
    """Greedy semantics: candidates that are train sites enter B^T as unit rows (agent.py:318 skips only static sites).
    The folded launch and the two-phase path give the same utilities to rounding and the same four picks; picks and
    utilities equal the oracle's greedy (agent.py:295-356 as O.greedy_fast)."""
    rng = np.random.RandomState(3)
    side = 36
    xx, yy = np.meshgrid(np.arange(side), np.arange(side))
    X = np.vstack([yy.ravel(), xx.ravel()]).T.astype(np.float64)
    n = len(X)
    perm = rng.permutation(n)
    A = np.sort(perm[:1100])                             # sampled: half static, half mobile
    is_static = rng.uniform(size=len(A)) < 0.5
    var = np.where(is_static, 0.01, 1.0)
    y = rng.uniform(0, 1, len(A))
    static = np.zeros(n, bool)
    static[A[is_static]] = True
    cand = np.where(~static)[0]                          # includes the mobile-sampled train sites
    c = _ctx(dtype, X, A, y, var)
    c.set_candidates(cand, prior_includes_noise=True)
    c.fit_and_solve()
    u1 = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
    p1 = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4)
    c.factorize()
    c.solve_candidates()
    u2 = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
    p2 = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4)
    t = 1e-10 if dtype == np.float64 else 2e-3
    assert np.max(np.abs(u1 - u2)) <= t
    if dtype == np.float64:
        assert list(p1) == list(p2)
        C = O.kernel_matrix(HYP, X) + HYP.noise * np.eye(n)
        mobile = np.zeros(n, bool)
        mobile[A[~is_static]] = True
        want, _ = O.greedy_fast(C, static, mobile, 0.1, 1.0, 4, 'entropy')
        assert [int(w) for w in want] == [int(q) for q in p1]
    c.close()


def test_a_stall_in_the_folded_launch_is_reported_and_survivable():
    rng = np.random.RandomState(9)
    pool, A, y, var, cidx = _field(1400, 2000, rng)
    c = _ctx(np.float64, pool, A, y, var)
    c.set_candidates(cidx, prior_includes_noise=False)
    c.fit_and_solve()
    mu0, pv0 = c.posterior()
    c.debug_dag_stall(40)
    t0 = time.time()
    with pytest.raises(_hip.AlgpError) as ei:
        c.fit_and_solve()
    assert 'stalled' in str(ei.value) and ei.value.code == _hip.ERR_HIP and time.time() - t0 < 30
    c.fit_and_solve()
    mu1, pv1 = c.posterior()
    assert np.array_equal(mu0, mu1) and np.array_equal(pv0, pv1)
    c.close()


def test_a_stall_in_the_solve_only_launch_is_reported_and_survivable():
    rng = np.random.RandomState(10)
    pool, A, y, var, cidx = _field(1400, 4300, rng)
    c = _ctx(np.float64, pool, A, y, var)
    c.factorize()
    c.set_candidates(cidx, prior_includes_noise=False)
    c.solve_candidates()
    mu0, pv0 = c.posterior()
    c.debug_dag_stall(25)
    with pytest.raises(_hip.AlgpError) as ei:
        c.solve_candidates()
    assert 'stalled' in str(ei.value) and ei.value.code == _hip.ERR_HIP
    c.solve_candidates()
    mu1, pv1 = c.posterior()
    assert np.array_equal(mu0, mu1) and np.array_equal(pv0, pv1)
    c.close()


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-8), (np.float32, 2e-2)], ids=['f64', 'f32'])
@pytest.mark.parametrize('kernel', [_hip.KERNEL_RBF, _hip.KERNEL_MATERN15], ids=['rbf', 'matern'])
def test_fit_step_with_the_identity_panel_against_the_three_calls_and_the_oracle(dtype, tol, kernel):
    """algp_fit_step: X = L^-T rides along with the factorisation as an identity panel (mode 2: tile row e exists from
    column e on), S^-1 = X X^T is one triangular-aware launch.  Same MLL and gradient as algp_factorize + algp_get_mll +
    algp_get_mll_grad (which take the launch sequence for X) to rounding, as the oracle's closed form (models.py:147-148
    through autograd in the reference; O.mll_and_grad), and the same bits in every run."""
    rng = np.random.RandomState(21)
    N = 1400
    X = rng.uniform(0, 30, (N, 3))
    y = np.sin(X[:, 0]) + 0.1 * rng.standard_normal(N)
    var = rng.uniform(0.005, 0.05, N)
    hyp = O.Hypers(np.log([1.3, 2.1, 0.8]), np.log(0.9), np.log(0.05), O.KERNEL_RBF if kernel == _hip.KERNEL_RBF else O.KERNEL_MATERN15)
    c = _hip.Context(dtype)
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise, kernel)
    c.set_pool(X)
    c.set_train(np.arange(N), y, var)
    c.prof_enable(True)
    c.prof_reset()
    mll, g = c.fit_step()
    assert c.prof_get('dag_panel')['launches'] == 1
    c.prof_enable(False)
    mll_b, g_b = c.fit_step()
    assert mll == mll_b and np.array_equal(g, g_b)
    ld = c.logdet()
    c.factorize()
    mll3, g3 = c.mll(), c.mll_grad()
    assert c.logdet() == ld                              # the factor does not depend on the panel
    # z = L^-1 (y - ybar) rides along as a row of the panel (tile products) where the three calls substitute: rounding only
    assert abs(mll3 - mll) <= (1e-12 if dtype == np.float64 else 1e-6) * abs(mll)
    assert np.max(np.abs(g3 - g) / np.maximum(1.0, np.abs(g))) <= (1e-10 if dtype == np.float64 else 2e-3)
    if kernel == _hip.KERNEL_RBF:
        f0, go = O.mll_and_grad(hyp, X, y, var)          # the oracle's are per train point (the reference's loss is -MLL/N)
        want = np.r_[go['log_lengthscale'], go['log_outputscale'], go['log_noise']] * N
        assert abs(mll - f0 * N) <= tol * abs(f0 * N)
        assert np.max(np.abs(g - want) / np.maximum(1.0, np.abs(want))) <= tol
    c.close()


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-9), (np.float32, 2e-3)], ids=['f64', 'f32'])
@pytest.mark.parametrize('N,M', [(2100, 4001), (2064, 4096), (2112, 2100), (2113, 4001)],
                         ids=['r52_z_in_the_last_tile', 'r16_no_padding_row', 'r64', 'r65_every_column_in_the_list'])
def test_fit_and_solve_leaves_a_narrow_last_tile_to_the_tail_kernel(monkeypatch, dtype, tol, N, M):
    """Round 6: with a train set that ends r <= 64 columns into its last 128-column tile, the panel's tile rows -- all but the
    one that carries y - ybar, or all of them when the candidates fill their last tile -- leave that column tile out of the
    task list (DagShape::pshort) and the tail kernel solves the r columns behind the launch.  Posterior against the oracle
    (utils.py:293-319), against the list with every column ($ALGP_TAIL_COLS=0), z / log-determinant unchanged; r = 65: no
    tail launch."""
    rng = np.random.RandomState(N * 7 + M)
    pool, A, y, var, cidx = _field(N, M, rng, side=50)
    samp = np.sort(rng.permutation(M)[:160])
    ref = O.posterior_chol(HYP, pool[A], y, pool[cidx[samp]], var)
    c = _ctx(dtype, pool, A, y, var)
    c.set_candidates(cidx, prior_includes_noise=False)
    c.prof_enable(True)
    c.prof_reset()
    c.fit_and_solve()
    assert c.prof_get('dag_panel')['launches'] == 1
    assert c.prof_get('tail_cols')['launches'] == (1 if N % 128 <= 64 else 0)
    c.prof_enable(False)
    mu1, pv1 = c.posterior()
    ld1, a1 = c.logdet(), c.alpha()
    scale = max(1.0, np.max(np.abs(ref['mu'])))
    assert np.max(np.abs(mu1[samp] - ref['mu'])) <= tol * scale
    assert np.max(np.abs(pv1[samp] - ref['var'])) <= tol * max(1.0, np.max(np.abs(ref['var'])))
    assert np.all(np.isfinite(mu1)) and np.all(np.isfinite(pv1))
    monkeypatch.setenv('ALGP_TAIL_COLS', '0')
    c.fit_and_solve()
    mu0, pv0 = c.posterior()
    loose = 2e-11 if dtype == np.float64 else 2e-4
    assert np.max(np.abs(mu1 - mu0)) <= loose * scale and np.max(np.abs(pv1 - pv0)) <= loose
    assert c.logdet() == ld1 and np.array_equal(c.alpha(), a1)              # the factor and z do not depend on the panel's shape
    c.close()
