"""Edge cases and error behaviour of the C-ABI (through ctypes): call order, bad arguments, tiny and
ragged sizes, empty sets -- the cases the reference handles implicitly (0 x 0 slogdet = 0,
LinAlgError from inv) or not at all."""
import numpy as np
import pytest

from algp_amd import _hip
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu
HYP = O.Hypers(np.log([2.0, 2.0]), 0.0, np.log(1e-2))


@pytest.fixture()
def ctx():
    c = _hip.Context(np.float64)
    yield c
    c.close()


def test_call_order_errors(ctx):
    with pytest.raises(ValueError):
        ctx.set_pool(np.zeros((4, 2)))                       # hypers first
    ctx.set_hypers(HYP.log_lengthscale, 0.0, HYP.log_noise)
    with pytest.raises(ValueError):
        ctx.set_train([0], [1.0])                            # pool first
    ctx.set_pool(np.random.RandomState(0).uniform(0, 5, (20, 2)))
    with pytest.raises(ValueError):
        ctx.factorize()                                      # train set first
    ctx.set_train(np.arange(10), np.ones(10), np.full(10, 0.01))
    ctx.set_candidates(np.arange(10, 20), prior_includes_noise=True)
    with pytest.raises(ValueError):
        ctx.solve_candidates()                               # factorize first
    ctx.factorize()
    with pytest.raises(ValueError):
        ctx.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)              # solve first
    ctx.solve_candidates()
    ctx.set_train(np.arange(9), np.ones(9), np.full(9, 0.01))
    with pytest.raises(ValueError):
        ctx.logdet()                                         # the factor is stale after set_train


def test_bad_arguments(ctx):
    ctx.set_hypers(HYP.log_lengthscale, 0.0, HYP.log_noise)
    ctx.set_pool(np.random.RandomState(0).uniform(0, 5, (20, 2)))
    with pytest.raises(ValueError):
        ctx.set_train([0, 25], [1.0, 2.0])                   # index outside the pool
    ctx.set_train([3, 3], [1.0, 2.0], [0.01, 1.0])           # one site measured twice is legal (two rows)
    with pytest.raises(ValueError):
        ctx.set_candidates([-1])
    with pytest.raises(ValueError):
        ctx.set_hypers(np.zeros(9), 0.0, 0.0)                # D > 8
    ctx.set_train(np.arange(5), np.ones(5))
    ctx.factorize()
    ctx.set_candidates(np.arange(5, 20))
    ctx.solve_candidates()
    ctx.commit_pick(7, 0.1, 1.0)
    with pytest.raises(ValueError):
        ctx.commit_pick(7, 0.1, 1.0)                         # already static
    with pytest.raises(ValueError):
        ctx.commit_pick(99, 0.1, 1.0)


@pytest.mark.parametrize('N,M', [(0, 7), (1, 1), (2, 129), (127, 1), (128, 128), (129, 127)])
def test_tiny_and_ragged_sizes(ctx, N, M):
    rng = np.random.RandomState(N * 7 + M)
    X = rng.uniform(0, 8, (N + M, 2))
    y = rng.uniform(0, 1, N)
    var = rng.choice([0.01, 1.0], N)
    ctx.set_hypers(HYP.log_lengthscale, 0.0, HYP.log_noise)
    ctx.set_pool(X)
    ctx.set_train(np.arange(N), y, var)
    ctx.factorize()
    ctx.set_candidates(np.arange(N, N + M), prior_includes_noise=False)
    ctx.solve_candidates()
    mu, pv = ctx.posterior()
    if N == 0:
        assert ctx.logdet() == 0.0 and ctx.entropy() == 0.0           # 0 x 0 slogdet (agent.py:308 on an empty field)
        assert np.allclose(pv, HYP.outputscale) and np.allclose(mu, 0.0)
    else:
        ref = O.posterior_chol(HYP, X[:N], y, X[N:], var)
        assert np.max(np.abs(mu - ref['mu'])) < 1e-10 and np.max(np.abs(pv - ref['var'])) < 1e-11
        assert ctx.logdet() == pytest.approx(ref['logdet'], rel=1e-12, abs=1e-12)
    cov, mi = ctx.posterior_cov(want_cov=True, want_mi=False)
    assert cov.shape == (M, M) and np.allclose(np.diag(cov), pv, atol=1e-12)


def test_greedy_from_an_empty_field_and_more_picks_than_a_block(ctx):
    """Empty sampled set (the reference's first batch after reset) and k = 40 picks in one run."""
    rng = np.random.RandomState(2)
    X = rng.uniform(0, 15, (150, 2))
    C = O.kernel_matrix(HYP, X) + HYP.noise * np.eye(150)
    none = np.zeros(150, bool)
    want, ut = O.greedy_fast(C, none, none, 0.1, 1.0, 40, 'entropy')
    ctx.set_hypers(HYP.log_lengthscale, 0.0, HYP.log_noise)
    ctx.set_pool(X)
    ctx.set_train(np.zeros(0, np.int64), np.zeros(0))
    ctx.factorize()
    ctx.set_candidates(np.arange(150), prior_includes_noise=True)
    ctx.solve_candidates()
    picks, u = ctx.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 40, forced_picks=np.array(want), want_utilities=True)
    fin = np.isfinite(ut)
    assert np.array_equal(np.isfinite(u), fin) and np.max(np.abs(u[fin] - ut[fin])) < 1e-9
    assert u[0] == pytest.approx(O.CONST + 0.5 * np.log(HYP.outputscale + HYP.noise + 0.01))


def test_two_contexts_are_independent():
    a, b = _hip.Context(np.float64), _hip.Context(np.float32)
    x = np.random.RandomState(0).uniform(0, 5, (30, 2))
    a.set_hypers(HYP.log_lengthscale, 0.0, HYP.log_noise)
    b.set_hypers(HYP.log_lengthscale + 1.0, 0.5, HYP.log_noise)
    Ka, Kb = a.kernel_matrix(x), b.kernel_matrix(x)
    assert Ka.dtype == np.float64 and Kb.dtype == np.float32
    assert np.allclose(np.diag(Ka), 1.0) and np.allclose(np.diag(Kb), np.exp(0.5), rtol=1e-6)
    assert a.device_bytes() > 0
    a.close()
    b.close()


def test_greedy_exhausts_candidates_and_free_run_from_empty_field(ctx):
    """More picks than candidates: the pick-only (lazily resolved) route and the route that returns every
    utility fail the same way, after the same picks; 40 free-running picks from an empty field follow the oracle."""
    rng = np.random.RandomState(6)
    X = rng.uniform(0, 12, (40, 2))
    ctx.set_hypers(HYP.log_lengthscale, 0.0, HYP.log_noise)
    ctx.set_pool(X)

    def setup(cand):
        ctx.set_train(np.arange(30), np.zeros(30), np.full(30, 1.0))
        ctx.factorize()
        ctx.set_candidates(cand, prior_includes_noise=True)
        ctx.solve_candidates()

    cand = np.array([33, 5, 38])                             # two new sites and one mobile-sampled train site
    setup(cand)
    full = [int(p) for p in ctx.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 3, want_utilities=True)[0]]
    setup(cand)
    lazy = [int(p) for p in ctx.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 3)]
    assert sorted(full) == [5, 33, 38] and lazy == full
    for kw in (dict(want_utilities=True), dict()):
        setup(cand)
        with pytest.raises(ValueError):
            ctx.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4, **kw)       # a 4th pick would re-sample a static site
    # free run from an empty field, picks only
    Xf = rng.uniform(0, 15, (150, 2))
    C = O.kernel_matrix(HYP, Xf) + HYP.noise * np.eye(150)
    none = np.zeros(150, bool)
    want, _ = O.greedy_fast(C, none, none, 0.1, 1.0, 40, 'entropy')
    ctx.set_pool(Xf)
    ctx.set_train(np.zeros(0, np.int64), np.zeros(0))
    ctx.factorize()
    ctx.set_candidates(np.arange(150), prior_includes_noise=True)
    ctx.solve_candidates()
    assert [int(p) for p in ctx.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 40)] == [int(p) for p in want]


@pytest.mark.parametrize('mode', ['coords', 'cov'])
def test_repeated_measurements_equal_the_fused_row(ctx, mode):
    """A site with a static and a mobile reading kept as TWO train rows (ss and sm) gives the posterior of the
    reference's fused row (agent.py:100-109: precision-weighted target, 1/(1/ss+1/sm)); log det S is larger by
    log(ss+sm) per such site; greedy utilities and picks are the same."""
    rng = np.random.RandomState(8)
    n, ss, sm = 260, 0.01, 1.0
    X = rng.uniform(0, 30, (n, 2))
    sites = rng.permutation(n)[:120]
    kind = rng.choice(['s', 'm', 'b'], 120, p=[0.3, 0.4, 0.3])
    truth = 2.0 + np.sin(X[:, 0] / 4)
    ys = truth + 0.1 * rng.standard_normal(n)
    ym = truth + 1.0 * rng.standard_normal(n)
    vf = 1.0 / (1.0 / ss + 1.0 / sm)
    fused_y = np.where(kind == 'b', (sm * ys[sites] + ss * ym[sites]) / (ss + sm), np.where(kind == 's', ys[sites], ym[sites]))
    fused_v = np.where(kind == 'b', vf, np.where(kind == 's', ss, sm))
    rows_i, rows_y, rows_v = [], [], []
    for s_, k in zip(sites, kind):                           # first readings in order, the second kind appended at the end
        rows_i.append(s_)
        rows_y.append(ys[s_] if k in 'sb' else ym[s_])
        rows_v.append(ss if k in 'sb' else sm)
    for s_, k in zip(sites, kind):
        if k == 'b':
            rows_i.append(s_)
            rows_y.append(ym[s_])
            rows_v.append(sm)
    ndup = int(np.sum(kind == 'b'))
    ctx.set_hypers(HYP.log_lengthscale, 0.0, HYP.log_noise)
    if mode == 'cov':
        ctx.set_pool_cov(O.kernel_matrix(HYP, X) + HYP.noise * np.eye(n))
    else:
        ctx.set_pool(X)
    static = np.zeros(n, bool)
    static[sites[kind != 'm']] = True
    cand = np.arange(n)
    out = []
    for idx, y, v in ((sites, fused_y, fused_v), (np.array(rows_i), np.array(rows_y), np.array(rows_v))):
        ctx.set_constant_mean(float(np.mean(fused_y)) if len(idx) != len(sites) else None)   # the reference's mean
        ctx.set_train(idx, y, v)
        ctx.factorize()
        ld = ctx.logdet()
        if mode == 'coords':
            mu = ctx.posterior_mean(cand)
        ctx.set_candidates(cand, prior_includes_noise=True)
        ctx.solve_candidates(alive=~static)
        m2, pv = ctx.posterior()
        picks, ut = ctx.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 5, want_utilities=True)
        out.append((ld, m2 if mode == 'cov' else mu, pv, [int(p) for p in picks], ut))
    (ld1, mu1, pv1, p1, u1), (ld2, mu2, pv2, p2, u2) = out
    assert abs(ld2 - (ld1 + ndup * np.log(ss + sm))) < 1e-9 * abs(ld1)
    ordinary = np.ones(n, bool)
    ordinary[sites] = False                                    # rows of train-site candidates hold [S^-1]_jj instead
    assert np.max(np.abs(mu1 - mu2)[ordinary]) < 1e-9 and np.max(np.abs(pv1 - pv2)[ordinary]) < 1e-10
    fin = np.isfinite(u1)
    assert np.array_equal(fin, np.isfinite(u2)) and np.max(np.abs(u1[fin] - u2[fin])) < 1e-9
    assert p1 == p2
    ctx.set_constant_mean(None)


@pytest.mark.parametrize('dt', [np.float64, np.float32])
def test_mi_of_a_numerically_singular_kxx_is_regularised_and_reported(dt):
    """utils.py:314 takes slogdet of the noise-free K_xx; on a dense grid with a long lengthscale that matrix is
    singular to working precision: the reference returns rounding noise (sign dropped), a Cholesky stops.  The
    library retries with a reported diagonal jitter instead of aborting the caller's run; the covariance itself
    and the train factorisation are untouched (the latter still raises)."""
    xx, yy = np.meshgrid(np.arange(12), np.arange(12))
    X = np.vstack([yy.ravel(), xx.ravel()]).T.astype(np.float64)
    n = len(X)
    rng = np.random.RandomState(0)
    perm = rng.permutation(n)
    A, test = perm[:40], perm[40:]
    c = _hip.Context(dt)
    c.set_hypers(np.log([9.0, 9.0]), 0.0, np.log(1e-2))            # lengthscale of 9 cells: K_xx has rank ~ 20 numerically
    c.set_pool(X)
    c.set_train(A, np.sin(X[A, 0]), np.full(len(A), 0.01))
    c.factorize()
    c.set_candidates(test, prior_includes_noise=False)
    c.solve_candidates()
    cov, mi = c.posterior_cov(want_cov=True, want_mi=True)
    assert np.isfinite(mi) and c.last_jitter() > 0.0
    hyp = O.Hypers(np.log([9.0, 9.0]), 0.0, np.log(1e-2))
    ref = O.posterior_chol(hyp, X[A], np.sin(X[A, 0]), X[test], np.full(len(A), 0.01), want_cov=True)
    assert np.max(np.abs(cov - ref['cov'])) < (1e-9 if dt == np.float64 else 2e-3)
    # a well-conditioned test set needs none
    c.set_candidates(test[:3], prior_includes_noise=False)
    c.solve_candidates()
    _, mi2 = c.posterior_cov(want_cov=False, want_mi=True)
    assert np.isfinite(mi2) and c.last_jitter() == 0.0
    c.close()
