"""Randomised parity sweep through the C-ABI against the fp64 oracle: sizes, input dimension, kernel, dtype,
noise patterns, candidate semantics, picks (pick-only route vs every-utility route vs oracle) and random
sequences of incremental updates (appends, re-measured targets, changed noise) -- each case seeded."""
import numpy as np
import pytest

from algp_amd import _hip
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.RandomState(seed)
    D = int(rng.choice([1, 2, 3, 6]))
    kernel = O.KERNEL_RBF if rng.rand() < 0.6 else O.KERNEL_MATERN15
    N = int(rng.choice([1, 3, 40, 127, 128, 129, 300, 650]))
    M = int(rng.choice([1, 5, 128, 200, 500]))
    dt = np.float64 if rng.rand() < 0.7 else np.float32
    X = rng.uniform(0, 6.0 * (N + M) ** (1.0 / D), (N + M, D)) if D > 1 else rng.uniform(0, 2.0 * (N + M), (N + M, 1))
    hyp = O.Hypers(np.log(rng.uniform(1.0, 3.0, D)), float(rng.uniform(-0.5, 0.5)), float(np.log(rng.uniform(5e-3, 5e-2))),
                   kernel=kernel)
    return rng, D, N, M, dt, X, hyp


@pytest.mark.parametrize('seed', range(40))
def test_posterior_random(seed):
    rng, D, N, M, dt, X, hyp = _case(seed)
    tol = 1e-8 if dt == np.float64 else 2e-3
    y = 2.0 + np.sin(X[:N].sum(1)) + 0.1 * rng.standard_normal(N)
    var = rng.choice([1e-5, 0.01, 1.0], N)
    tvar = rng.choice([0.0, 0.01, 1.0], M) if rng.rand() < 0.5 else None
    c = _hip.Context(dt)
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise, kernel=hyp.kernel)
    c.set_pool(X)
    c.set_train(np.arange(N), y, var)
    c.factorize()
    c.set_candidates(np.arange(N, N + M), prior_includes_noise=False, extra_var=tvar)
    c.solve_candidates()
    mu, pv = c.posterior()
    ref = O.posterior_chol(hyp, X[:N], y, X[N:], var, tvar)
    scale = max(1.0, np.max(np.abs(ref['mu'])))
    assert np.max(np.abs(mu - ref['mu'])) < tol * scale
    assert np.max(np.abs(pv - ref['var'])) < tol * max(1.0, np.max(np.abs(ref['var'])))
    assert abs(c.logdet() - ref['logdet']) < tol * max(1.0, abs(ref['logdet'])) * (1 if dt == np.float64 else 5)
    assert np.max(np.abs(c.posterior_mean(np.arange(N, N + M)) - ref['mu'])) < tol * scale
    c.close()


@pytest.mark.parametrize('seed', range(300, 320))
def test_posterior_random_at_task_list_sizes(seed):
    """The same parity statement where the one-launch task list serves (N/128 >= 8): random train-set and test-set sizes
    around tile edges, the factorisation and the solve folded into one launch (algp_fit_and_solve: what
    predictive_distribution calls, utils.py:293-319), as two task lists, and a solve too short for a list of its own."""
    rng = np.random.RandomState(seed)
    D = int(rng.choice([2, 3, 6]))
    N = int(rng.choice([1024, 1025, 1151, 1152, 1300, 1793, 2060, 2200, 2304, 2900]))   # 2060 / 2200: a narrow last tile (tail kernel)
    M = int(rng.choice([1, 127, 129, 900, 4096, 4097, 5000]))
    dt = np.float64 if rng.rand() < 0.6 else np.float32
    tol = 1e-8 if dt == np.float64 else 3e-3
    kernel = O.KERNEL_RBF if rng.rand() < 0.6 else O.KERNEL_MATERN15
    X = rng.uniform(0, 5.0 * (N + M) ** (1.0 / D), (N + M, D))
    hyp = O.Hypers(np.log(rng.uniform(1.5, 3.0, D)), float(rng.uniform(-0.3, 0.3)), float(np.log(rng.uniform(5e-3, 5e-2))), kernel=kernel)
    y = 2.0 + np.sin(X[:N].sum(1)) + 0.1 * rng.standard_normal(N)
    var = rng.choice([0.01, 1.0], N)
    tvar = rng.choice([0.0, 0.01, 1.0], M) if rng.rand() < 0.5 else None
    ref = O.posterior_chol(hyp, X[:N], y, X[N:], var, tvar)
    c = _hip.Context(dt)
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise, kernel=hyp.kernel)
    c.set_pool(X)
    c.set_train(np.arange(N), y, var)
    c.set_candidates(np.arange(N, N + M), prior_includes_noise=False, extra_var=tvar)
    out = []
    for folded in (True, False):
        if folded:
            c.fit_and_solve()
        else:
            c.factorize()
            c.solve_candidates()
        mu, pv = c.posterior()
        scale = max(1.0, np.max(np.abs(ref['mu'])))
        assert np.max(np.abs(mu - ref['mu'])) < tol * scale, (folded, N, M)
        assert np.max(np.abs(pv - ref['var'])) < tol * max(1.0, np.max(np.abs(ref['var']))), (folded, N, M)
        assert abs(c.logdet() - ref['logdet']) < tol * max(1.0, abs(ref['logdet'])) * (1 if dt == np.float64 else 5)
        out.append((mu, pv))
    loose = 1e-10 if dt == np.float64 else 5e-4                  # the two routes against each other: rounding only
    assert np.max(np.abs(out[0][0] - out[1][0])) <= loose * scale and np.max(np.abs(out[0][1] - out[1][1])) <= loose
    c.close()


@pytest.mark.parametrize('seed', range(100, 124))
def test_greedy_random(seed):
    rng, D, N, M, dt, X, hyp = _case(seed)
    if D == 1 or hyp.kernel != O.KERNEL_RBF:
        D = 2
        X = rng.uniform(0, 4.0 * np.sqrt(N + M), (N + M, 2))
        hyp = O.Hypers(np.log([2.0, 2.5]), 0.0, np.log(1e-2))
    n = N + M
    static = np.zeros(n, bool)
    mobile = np.zeros(n, bool)
    perm = rng.permutation(n)
    ns = int(rng.randint(0, N + 1))
    static[perm[:ns]] = True
    mobile[perm[max(0, ns - N // 4):N]] = True                # some sites carry both kinds of reading
    k = int(min(6, np.sum(~static)))
    C = O.kernel_matrix(hyp, X) + hyp.noise * np.eye(n)
    want, ut = O.greedy_fast(C, static, mobile, 0.1, 1.0, k, 'entropy')
    c = _hip.Context(np.float64)
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise)
    c.set_pool(X)
    A = np.where(static | mobile)[0]
    vf = 1.0 / (1.0 / 0.01 + 1.0)
    var = np.where(static[A] & mobile[A], vf, np.where(static[A], 0.01, 1.0))
    cand = np.where(~static)[0]

    def setup():
        c.set_train(A, np.zeros(len(A)), var)
        c.factorize()
        c.set_candidates(cand, prior_includes_noise=True)
        c.solve_candidates()

    setup()
    picks, u = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, k, forced_picks=np.array(want), want_utilities=True)
    full = np.full((k, n), -np.inf)
    full[:, cand] = u
    fin = np.isfinite(ut)
    assert np.array_equal(np.isfinite(full), fin)
    if fin.any():
        assert np.max(np.abs(full[fin] - ut[fin])) < 1e-8 * max(1.0, np.max(np.abs(ut[fin])))
    setup()
    a = [int(p) for p in c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, k, want_utilities=True)[0]]
    setup()
    b = [int(p) for p in c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, k)]
    assert a == b                                              # lazily resolved picks == rescoring every row
    gaps = [np.sort(ut[p][fin[p]])[-1] - np.sort(ut[p][fin[p]])[-2] if fin[p].sum() > 1 else 1.0 for p in range(k)]
    if min(gaps) > 1e-9:                                       # no rounding-determined ties: the oracle's picks
        assert a == [int(p) for p in want]
    c.close()


@pytest.mark.parametrize('seed', range(200, 216))
def test_incremental_sequences_random(seed):
    rng = np.random.RandomState(seed)
    dt = np.float64 if seed % 4 else np.float32
    tol = 1e-8 if dt == np.float64 else 3e-3
    n_pool = 900
    X = rng.uniform(0, 50, (n_pool, 2))
    hyp = (np.log([3.0, 2.0]), 0.0, np.log(1e-2))
    truth = 4.0 + np.sin(X[:, 0] / 5) * np.cos(X[:, 1] / 6)
    order = list(rng.permutation(n_pool)[:int(rng.choice([5, 120, 260]))])
    var = {int(i): float(rng.choice([0.01, 1.0])) for i in order}
    yv = {int(i): float(truth[i] + 0.1 * rng.standard_normal()) for i in order}
    cand = np.arange(n_pool)                                    # greedy-style: every site is a candidate
    c = _hip.Context(dt)
    c.set_hypers(*hyp)
    c.set_pool(X)
    for step in range(6):
        action = rng.choice(['append', 'append', 'append', 'retarget', 'renoise'])
        if action == 'append' or step == 0:
            new = [int(i) for i in rng.permutation(n_pool)[:int(rng.choice([1, 30, 140]))] if int(i) not in var]
            for i in new:
                order.append(i)
                var[i] = float(rng.choice([0.01, 1.0]))
                yv[i] = float(truth[i] + 0.1 * rng.standard_normal())
        elif action == 'retarget':                              # a re-measured site: new fused target, same noise
            i = int(order[rng.randint(len(order))])
            yv[i] += 0.3
        else:                                                    # a site that received the other kind of reading
            i = int(order[rng.randint(len(order))])
            var[i] = 1.0 / (1.0 / 0.01 + 1.0)
        A = np.array(order, dtype=np.int64)
        y = np.array([yv[int(i)] for i in A])
        v = np.array([var[int(i)] for i in A])
        c.set_train(A, y, v)
        c.factorize(incremental=True)
        c.set_candidates(cand, prior_includes_noise=True)
        c.solve_candidates(incremental=True)
        ref = _hip.Context(np.float64)
        ref.set_hypers(*hyp)
        ref.set_pool(X)
        ref.set_train(A, y, v)
        ref.factorize()
        ref.set_candidates(cand, prior_includes_noise=True)
        ref.solve_candidates()
        (m1, p1), (m2, p2) = c.posterior(), ref.posterior()
        assert np.max(np.abs(m1 - m2)) < tol * max(1.0, np.max(np.abs(m2))), (step, action)
        assert np.max(np.abs(p1 - p2)) < tol * max(1.0, np.max(np.abs(p2))), (step, action)
        assert abs(c.logdet() - ref.logdet()) < tol * max(1.0, abs(ref.logdet())) * (1 if dt == np.float64 else 5)
        assert abs(c.mll() - ref.mll()) < tol * max(1.0, abs(ref.mll())) * (1 if dt == np.float64 else 5)
        k = 3
        s1 = [int(p) for p in c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, k)]
        if dt == np.float64:
            u = ref.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, k, want_utilities=True)[1]
            srt = np.sort(u[:, np.isfinite(u[0])], axis=1)
            if np.min(srt[:, -1] - srt[:, -2]) > 1e-7:
                ref.factorize(); ref.solve_candidates()
                assert s1 == [int(p) for p in ref.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, k)], (step, action)
        ref.close()
    c.close()
