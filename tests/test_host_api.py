"""GPU tests of the host-side mirror of the reference's Python surface (GPR, predictive_distribution,
entropy_from_cov, Agent) against the golden vectors captured from the reference and the oracle."""
import argparse
import types

import numpy as np
import pytest

from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def make_gp(hyp, x, y, var, dtype=np.float64, kernel='rbf'):
    import torch
    from algp_amd.models import GPR
    gp = GPR(kernel_params={'type': kernel}, max_iterations=0, dtype=dtype)
    gp.reset(x, y, var)
    with torch.no_grad():
        gp.model.kernel_covar_module.base_kernel.log_lengthscale.copy_(torch.tensor(hyp.log_lengthscale).reshape(1, 1, -1))
        gp.model.kernel_covar_module.log_outputscale.fill_(hyp.log_outputscale)
        gp.likelihood.log_noise.fill_(hyp.log_noise)
    return gp


def _hyp(g, pre):
    return O.Hypers(g[pre + 'log_ls'], float(g[pre + 'log_os']), float(g[pre + 'log_noise']))


FLAGS = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1), (1, 0, 1)]


@pytest.mark.parametrize('ci', [0, 1, 2])
def test_predictive_distribution_matches_reference_golden(golden, ci):
    from algp_amd.utils import predictive_distribution
    g = golden('g2_predictive')
    pre = 'g2_c%d_' % ci
    hyp = _hyp(g, pre)
    gp = make_gp(hyp, g[pre + 'train_x'], g[pre + 'train_y'], g[pre + 'train_var'])
    n = 0
    for tv in (0, 1):
        for xv in (0, 1):
            for (rv, rc, rm) in FLAGS:
                tag = pre + 'tv%d_xv%d_f%d%d%d_' % (tv, xv, rv, rc, rm)
                if tag + 'arity' not in g.files:
                    continue
                res = predictive_distribution(gp, g[pre + 'train_x'], g[pre + 'train_y'], g[pre + 'test_x'],
                                              g[pre + 'train_var'] if tv else None, g[pre + 'test_var'] if xv else None,
                                              return_var=bool(rv), return_cov=bool(rc), return_mi=bool(rm))
                if not isinstance(res, tuple):
                    res = (res,)
                assert len(res) == int(g[tag + 'arity'])          # tuple convention utils.py:302-319
                for k, r in enumerate(res):
                    want = g[tag + 'r%d' % k]
                    assert np.shape(r) == want.shape
                    # the reference ran fp32 (kernel + inv): this is an fp32-level statement
                    assert rel(r, want) < 3e-3, (tag, k, rel(r, want))
                n += 1
    assert n >= 16
    # and tightly against the fp64 oracle on identical inputs (north_star: 1e-5 relative)
    p = O.posterior_chol(hyp, g[pre + 'train_x'], g[pre + 'train_y'], g[pre + 'test_x'], g[pre + 'train_var'],
                         g[pre + 'test_var'], want_cov=True)
    mu, cov, mi = predictive_distribution(gp, g[pre + 'train_x'], g[pre + 'train_y'], g[pre + 'test_x'],
                                          g[pre + 'train_var'], g[pre + 'test_var'], return_cov=True, return_mi=True)
    assert rel(mu, p['mu']) < 1e-9 and rel(cov, p['cov']) < 1e-9
    assert mi == pytest.approx(p['mi'], rel=1e-8)


def test_predictive_distribution_fp32_context(golden):
    from algp_amd.utils import predictive_distribution
    g = golden('g2_predictive')
    pre = 'g2_c1_'
    hyp = _hyp(g, pre)
    gp = make_gp(hyp, g[pre + 'train_x'], g[pre + 'train_y'], g[pre + 'train_var'], dtype=np.float32)
    mu, var = predictive_distribution(gp, g[pre + 'train_x'], g[pre + 'train_y'], g[pre + 'test_x'],
                                      g[pre + 'train_var'], g[pre + 'test_var'], return_var=True)
    assert mu.dtype == np.float32 and var.dtype == np.float32
    want_mu, want_var = g[pre + 'tv1_xv1_f100_r0'], g[pre + 'tv1_xv1_f100_r1']
    assert rel(mu, want_mu) < 1e-3 and rel(var, want_var) < 1e-3       # north_star fp32 tolerance


@pytest.mark.parametrize('k', [0, 1, 5, 64])
@pytest.mark.parametrize('dt', ['float32', 'float64'])
def test_entropy_from_cov_golden(golden, k, dt):
    from algp_amd.utils import CONST, entropy_from_cov
    g = golden('g1_entropy')
    cov = g['g1_cov_k%d_%s' % (k, dt)]
    want = float(g['g1_ent_k%d_%s' % (k, dt)])
    assert entropy_from_cov(cov) == pytest.approx(want, rel=2e-5 if dt == 'float32' else 1e-11, abs=1e-12)
    assert entropy_from_cov(cov, constant=0.0) == pytest.approx(want - k * CONST, rel=1e-4, abs=1e-4)
    assert CONST == O.CONST


def test_gpr_cov_mat_and_parameter_names(golden):
    g = golden('g3_greedy')
    pre = 'g3_n64_'
    hyp = _hyp(g, pre)
    X = g[pre + 'X']
    gp = make_gp(hyp, X[:10], np.zeros(10), np.full(10, 0.01))
    names = dict(gp.model.named_parameters())
    assert set(names) == {'kernel_covar_module.log_outputscale', 'kernel_covar_module.base_kernel.log_lengthscale',
                          'likelihood.log_noise'}
    assert np.exp(names['likelihood.log_noise'].item()) == pytest.approx(hyp.noise)     # run.py:36-37 usage
    sd = gp.model.state_dict()
    gp2 = make_gp(O.Hypers([0.0, 0.0]), X[:10], np.zeros(10), np.full(10, 0.01))
    gp2.model.load_state_dict(sd)                                                        # agent.py:41
    C = gp2.cov_mat(x1=X, add_likelihood_var=True)                                       # agent.py:90
    assert rel(C, g[pre + 'cov']) < 1e-6            # golden cov is the fp32 closed form
    Cx = gp.cov_mat(X[:7], x2=X[5:20])
    assert Cx.shape == (7, 15)
    assert rel(Cx, O.kernel_matrix(hyp, X[:7], X[5:20])) < 1e-13
    Cd = gp.cov_mat(X[:7], x2=X[:7].copy(), white_noise_var=np.arange(7.0))             # torch.equal branch
    assert rel(Cd, O.kernel_matrix(hyp, X[:7]) + np.diag(np.arange(7.0))) < 1e-13
    gm = make_gp(O.Hypers(hyp.log_lengthscale, 0.3, -2.0, O.KERNEL_MATERN15), X[:10], np.zeros(10), None,
                 kernel='matern')
    assert rel(gm.cov_mat(X), O.kernel_matrix(O.Hypers(hyp.log_lengthscale, 0.3, -2.0, O.KERNEL_MATERN15), X)) < 1e-13
    from algp_amd.models import GPR
    with pytest.raises(NotImplementedError):
        GPR(kernel_params={'type': 'spectral_mixture'}).reset(X[:4], np.zeros(4), None)
    with pytest.raises(NotImplementedError):
        GPR(latent='linear').reset(X[:4], np.zeros(4), None)


@pytest.mark.parametrize('kernel', ['rbf', 'matern'])
def test_mll_and_gradient_vs_oracle_and_finite_differences(kernel):
    rng = np.random.RandomState(3)
    x = rng.uniform(0, 6, (150, 3))
    y = np.sin(x[:, 0]) + 0.1 * rng.standard_normal(150)
    var = rng.uniform(0.005, 0.05, 150)
    kid = O.KERNEL_RBF if kernel == 'rbf' else O.KERNEL_MATERN15
    hyp = O.Hypers(np.log([1.3, 2.1, 0.8]), np.log(0.9), np.log(0.05), kid)
    gp = make_gp(hyp, x, y, var, kernel=kernel)
    gp._load_train_on_device()
    loss, g = gp.neg_mll_and_grad()
    loss2, g2 = gp.neg_mll_and_grad()                   # fixed-order reductions: the same bits every time
    assert loss2 == loss and np.array_equal(np.asarray(g2), np.asarray(g))
    if kernel == 'rbf':
        f0, go = O.mll_and_grad(hyp, x, y, var)
        assert -loss == pytest.approx(f0, rel=1e-10)
        want = np.r_[go['log_lengthscale'], go['log_outputscale'], go['log_noise']]
        assert rel(-g, want) < 1e-8
    # finite differences of the device MLL itself
    import torch
    eps = 1e-5
    for k in range(5):
        def shift(d):
            with torch.no_grad():
                if k < 3:
                    gp.model.kernel_covar_module.base_kernel.log_lengthscale[0, 0, k] += d
                elif k == 3:
                    gp.model.kernel_covar_module.log_outputscale += d
                else:
                    gp.likelihood.log_noise += d
        shift(eps)
        lp, _ = gp.neg_mll_and_grad()
        shift(-2 * eps)
        lm, _ = gp.neg_mll_and_grad()
        shift(eps)
        assert (lp - lm) / (2 * eps) == pytest.approx(g[k], rel=2e-5, abs=1e-8)


def test_fit_increases_likelihood_and_recovers_scale():
    from algp_amd.models import GPR
    rng = np.random.RandomState(0)
    x = rng.uniform(0, 20, (300, 2))
    hyp = O.Hypers(np.log([2.0, 2.0]), np.log(1.5), np.log(0.01))
    K = O.kernel_matrix(hyp, x) + 0.01 * np.eye(300)
    y = np.linalg.cholesky(K) @ rng.standard_normal(300) + 3.0
    gp = GPR(lr=.1, max_iterations=120, kernel_params={'type': 'rbf'})
    losses = gp.fit(x, y, np.full(300, 1e-4))
    assert losses[-1] < losses[0] - 0.1
    ls, los, ln, _ = gp.hypers()
    assert np.all(np.abs(np.exp(ls) - 2.0) < 0.8)
    mu = gp.predict(x[:20])
    assert np.mean(np.abs(mu - y[:20])) < 0.3
    mu2, v = gp.predict(x[:20], return_std=True)
    assert np.all(v > 0) and np.allclose(mu, mu2, rtol=1e-6, atol=1e-8)


def _agent(cov, static_data, mobile_data, criterion, X=None):
    from algp_amd.agent import Agent
    from algp_amd.models import GPR
    a = Agent.__new__(Agent)
    a.env = types.SimpleNamespace(num_samples=len(static_data), X=X)
    a.static_data, a.mobile_data = static_data, mobile_data
    a.static_std, a.mobile_std, a.criterion = 0.1, 1.0, criterion
    a.gp = GPR(kernel_params={'type': 'rbf'})
    a.gp.reset(np.zeros((1, 2)), np.zeros(1), None)
    a._cov_matrix, a._cov_matrix_user, a._pool_key = None, False, None
    if cov is not None:
        a.cov_matrix = cov
    return a


def _state_lists(s0, m0):
    return [[0.5] if v else [] for v in s0], [[0.4] if v else [] for v in m0]


@pytest.mark.parametrize('kind', ['static', 'mobile', 'both'])
@pytest.mark.parametrize('n', [64, 360])
def test_agent_greedy_reference_semantics(golden, n, kind):
    """Agent.greedy consuming an assigned `cov_matrix`, as the reference's does (agent.py:308)."""
    g = golden('g3_greedy')
    pre = 'g3_n%d_' % n
    sd, md = _state_lists(g[pre + kind + '_static'], g[pre + kind + '_mobile'])
    a = _agent(g[pre + 'cov'].astype(np.float64), sd, md, 'entropy')
    assert a.greedy(4) == [int(v) for v in g[pre + kind + '_entropy_picks']]
    # from coordinates (the scalable route): same picks with the fp64 kernel
    b = _agent(None, sd, md, 'entropy', X=g[pre + 'X'])
    hyp = _hyp(g, pre)
    b.gp = make_gp(hyp, g[pre + 'X'][:2], np.zeros(2), None)
    assert b.greedy(4) == [int(v) for v in g[pre + kind + '_entropy_picks']]
    assert b.cov_matrix.shape == (n, n) and rel(b.cov_matrix, g[pre + 'cov']) < 1e-6


@pytest.mark.parametrize('crit', ['entropy', 'mutual_information'])
def test_agent_best_path_golden(golden, crit):
    g = golden('g4_best_path')
    paths, o = [], 0
    for L in g['g4_paths_len']:
        paths.append([int(v) for v in g['g4_paths_flat'][o:o + L]])
        o += L
    sd, md = _state_lists(g['g4_static'], g['g4_mobile'])
    a = _agent(g['g4_cov'].astype(np.float64), sd, md, crit)
    si = [int(v) for v in g['g4_static_indices']]
    assert a.best_path(paths, si) == int(g['g4_%s_idx' % crit])
    assert a.best_path(paths[:1], si) == 0                       # early out, agent.py:362-363


def test_agent_end_to_end_synthetic_field(capsys):
    """BASELINE config 1 plumbing: 20 x 20 MoG field, pre-train, greedy batches, predict."""
    import run as demo
    from algp_amd.arguments import get_args
    args = get_args(['--eval_only', '--kernel', 'rbf', '--max_iterations', '30', '--num_runs', '3'])
    errors = demo.run_demo(args)
    assert len(errors) == 3 and all(np.isfinite(errors)) and errors[-1] < 0.2


def test_agent_incremental_loop_equals_from_scratch():
    """f1 in the agent: the same planning loop with and without factor reuse gives the same picks
    and the same predictions (greedy + sampling + predict over several batches)."""
    from algp_amd.agent import Agent
    from algp_amd.arguments import get_args
    from algp_amd.field import SyntheticField
    outs = []
    for inc in (True, False):
        np.random.seed(3)
        env = SyntheticField(24, 24, num_test=40)
        args = get_args(['--eval_only', '--kernel', 'rbf', '--max_iterations', '15', '--fraction_pretrain', '0.3'])
        args.incremental = inc
        ag = Agent(env, args)
        ag._setup_ipp('entropy')
        log = []
        for step in range(5):
            picks = ag.greedy(4)
            ag._add_samples(picks, [ag.static_std] * 4)
            mob = [int(i) for i in np.random.permutation(env.num_samples)[:6]]
            ag._add_samples(mob, [ag.mobile_std] * 6)
            mu, var = ag.predict(return_var=True)
            log.append((picks, mu.copy(), var.copy()))
        outs.append(log)
    for (p1, m1, v1), (p2, m2, v2) in zip(*outs):
        assert p1 == p2
        assert np.max(np.abs(m1 - m2)) < 1e-8 and np.max(np.abs(v1 - v2)) < 1e-9


def test_best_path_many_paths_vs_oracle_and_reuse(golden):
    """f3: path utilities through the incremental factor equal one fresh slogdet per path."""
    g = golden('g4_best_path')
    rng = np.random.RandomState(2)
    n = len(g['g4_cov'])
    paths = [[int(i) for i in rng.permutation(n)[:rng.randint(5, 14)]] for _ in range(25)]
    sd, md = _state_lists(g['g4_static'], g['g4_mobile'])
    si = [int(v) for v in g['g4_static_indices']]
    for crit in ('entropy', 'mutual_information'):
        a = _agent(g['g4_cov'].astype(np.float64), sd, md, crit)
        idx, ut = O.best_path_ref(g['g4_cov'], g['g4_static'], g['g4_mobile'], paths, si, 0.1, 1.0, crit)
        assert a.best_path(paths, si) == idx


def test_agent_row_form_equals_fused_form():
    """Incremental agents keep one train row per (site, kind of reading) so that a site re-measured by the other
    sensor appends a row instead of changing an old one; greedy picks, best_path choices and predictions must
    equal the reference's fused form (agent.py:100-109) -- including sites with both readings and paths that
    cross static sites."""
    from algp_amd.agent import Agent
    from algp_amd.arguments import get_args
    from algp_amd.field import SyntheticField
    res = []
    for inc in (True, False):
        np.random.seed(11)
        env = SyntheticField(18, 18, num_test=30)
        args = get_args(['--eval_only', '--kernel', 'rbf', '--max_iterations', '10', '--fraction_pretrain', '0.25'])
        args.incremental = inc
        ag = Agent(env, args)
        ag._setup_ipp('entropy')
        rng = np.random.RandomState(4)
        log = []
        for step in range(5):
            picks = ag.greedy(3)
            ag._add_samples(picks, [ag.static_std] * 3)
            # mobile readings: new sites, already mobile-sampled sites and STATIC sites (their noise becomes the fused one)
            static, mobile = ag._masks()
            pool = np.r_[rng.permutation(env.num_samples)[:6], rng.permutation(np.where(static)[0])[:3],
                         rng.permutation(np.where(mobile)[0])[:2] if mobile.any() else []].astype(int)
            ag._add_samples([int(i) for i in pool], [ag.mobile_std] * len(pool))
            paths = [[int(j) for j in rng.permutation(env.num_samples)[:7]] + [int(np.where(static)[0][k])]
                     for k in range(4)]
            choice = ag.best_path(paths, [int(i) for i in rng.permutation(env.num_samples)[:2]])
            mu, var = ag.predict(return_var=True)
            log.append((picks, choice, mu.copy(), var.copy()))
        res.append(log)
        assert ag._use_rows() == inc
    for (p1, c1, m1, v1), (p2, c2, m2, v2) in zip(*res):
        assert p1 == p2 and c1 == c2
        assert np.max(np.abs(m1 - m2)) < 1e-8 and np.max(np.abs(v1 - v2)) < 1e-9


def test_prediction_vs_distance_prefixes_match_oracle():
    """f3: the prefix sweep (agent.py:497-518) through ONE context with factor / solve updates equals a from-scratch
    posterior per prefix (oracle), including re-measured sites (two rows of the same site) and skipped poses (-1)."""
    from algp_amd.agent import Agent
    rng = np.random.RandomState(5)
    R, Cc = 16, 15
    grid, field = O.generate_gaussian_data(R, Cc, k=5, rng=rng)
    X = grid.astype(np.float64)
    n = len(X)
    perm = rng.permutation(n)
    test = perm[:30]
    hyp = O.Hypers(np.log([2.5, 3.0]), np.log(0.9), np.log(0.02))
    a = Agent.__new__(Agent)
    a.env = types.SimpleNamespace(num_samples=n, X=X, test_X=X[test], test_Y=field[test])
    a.gp = make_gp(hyp, X[:2], np.zeros(2), None)
    a._cov_matrix, a._cov_matrix_user, a._pool_key = None, False, None
    walk = [int(i) for i in rng.permutation(perm[30:])[:150]]
    walk[7] = -1                                                   # a pose off the sampling grid (agent.py:500-501)
    walk[40] = walk[3]                                             # re-measured site: a second row, not a fused one
    walk[90] = -1
    stds = [0.1 if k % 5 == 0 else 1.0 for k in range(len(walk))]
    ys = [float(field[i] + (0.05 if i >= 0 else 0.0)) if i >= 0 else -1.0 for i in walk]
    a.collected = {'ind': walk, 'std': stds, 'y': ys}
    test_every, runs = 37, 4
    res = a.prediction_vs_distance(test_every, runs)
    assert len(res['error']) == runs and len(res['mi']) == runs
    for r in range(1, runs + 1):
        cnt = r * test_every
        idx = np.array(walk[:cnt])
        ok = idx != -1
        ref = O.posterior_chol(hyp, X[idx[ok]], np.array(ys[:cnt])[ok], X[test], np.array(stds[:cnt])[ok] ** 2, want_cov=True)
        assert res['error'][r - 1] == pytest.approx(np.mean(np.abs(field[test] - ref['mu'])), rel=1e-9, abs=1e-11)
        assert res['mean_var'][r - 1] == pytest.approx(np.mean(np.diag(ref['cov'])), rel=1e-8)
        assert res['mi'][r - 1] == pytest.approx(ref['mi'], rel=1e-7, abs=1e-7)
    assert np.max(np.abs(res['mean'] - ref['mu'])) < 1e-9


def test_greedy_after_a_foreign_pool_load_reloads_its_pool(golden):
    """A pool loaded into the shared context by somebody else (GPR.predict, predictive_distribution) must not be
    indexed by the agent's next greedy: the picks equal a fresh agent's."""
    from algp_amd.utils import predictive_distribution
    g = golden('g3_greedy')
    pre = 'g3_n64_'
    sd, md = _state_lists(g[pre + 'both_static'], g[pre + 'both_mobile'])
    hyp = _hyp(g, pre)
    X = g[pre + 'X']

    def agent():
        b = _agent(None, sd, md, 'entropy', X=X)
        b.gp = make_gp(hyp, X[:2], np.zeros(2), None)
        return b
    a = agent()
    first = a.greedy(4)
    rng = np.random.RandomState(0)
    # a foreign pool with MORE rows than the agent's (silently wrong picks before the generation counter)
    predictive_distribution(a.gp, rng.uniform(0, 8, (90, 2)), rng.uniform(size=90), rng.uniform(0, 8, (40, 2)), return_var=True)
    assert a.greedy(4) == first == agent().greedy(4)


def test_wrong_input_width_is_rejected():
    from algp_amd import _hip
    c = _hip.Context(np.float64)
    c.set_hypers(np.log([1.0, 2.0]), 0.0, np.log(0.1))
    with pytest.raises(ValueError):
        c.set_pool(np.zeros((10, 3)))
    with pytest.raises(ValueError):
        c.kernel_matrix(np.zeros((4, 2)), np.zeros((5, 1)))
    c.set_pool(np.zeros((10, 2)))
    c.close()


def test_best_path_batched_block_scoring_equals_per_path_factor_updates():
    """f3: all paths scored at once from one resident factor + candidate solve (algp_score_paths: the log-determinant
    of each path's posterior block) against one factor update + entropy per path, on paths that cross static sites
    (second row), mobile sites (no new reading), repeat a site and contain off-grid poses (-1)."""
    from algp_amd.agent import Agent
    from algp_amd.arguments import get_args
    from algp_amd.field import SyntheticField
    np.random.seed(5)
    env = SyntheticField(26, 24, num_test=40)
    args = get_args(['--eval_only', '--kernel', 'rbf', '--max_iterations', '10', '--fraction_pretrain', '0.2'])
    args.incremental = True
    ag = Agent(env, args)
    ag._setup_ipp('entropy')
    rng = np.random.RandomState(8)
    n = env.num_samples
    for step in range(3):
        picks = ag.greedy(3)
        ag._add_samples(picks, [ag.static_std] * 3)
        mob = [int(i) for i in rng.permutation(n)[:15]]
        ag._add_samples(mob, [ag.mobile_std] * len(mob))
        static, mobile = ag._masks()
        new_static = [int(i) for i in rng.permutation(np.where(~static)[0])[:2]]
        st = static.copy()
        st[new_static] = True
        paths = []
        for k in range(40):
            L = rng.randint(3, 30)
            pth = [int(j) for j in rng.permutation(n)[:L]]
            pth[rng.randint(L)] = int(rng.choice(np.where(st)[0]))            # crosses a static site
            if mobile.any():
                pth[rng.randint(L)] = int(rng.choice(np.where(mobile)[0]))    # and a mobile one
            pth.insert(rng.randint(L), -1)
            pth.append(pth[0] if pth[0] != -1 else pth[1])                    # a site crossed twice
            paths.append(pth)
        c = ag._load_pool()
        ub = ag._path_utilities_rows(c, paths, st, mobile, batched=True)
        ul = ag._path_utilities_rows(c, paths, st, mobile, batched=False)
        assert np.all(np.isfinite(ub))
        assert np.max(np.abs(ub - ul)) < 1e-8 * max(1.0, np.max(np.abs(ul))), np.max(np.abs(ub - ul))
        assert int(np.argmax(ub)) == int(np.argmax(ul)) == ag.best_path(paths, new_static)
