"""Pin oracle/gp_oracle.py to the golden vectors captured from the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import gp_oracle as O


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def test_const():
    assert O.CONST == pytest.approx(0.5 * np.log(2 * np.pi * np.e), abs=1e-15)


@pytest.mark.parametrize('k', [0, 1, 5, 64])
@pytest.mark.parametrize('dt', ['float32', 'float64'])
def test_g1_entropy(golden, k, dt):
    g = golden('g1_entropy')
    cov = g['g1_cov_k%d_%s' % (k, dt)]
    want = float(g['g1_ent_k%d_%s' % (k, dt)])
    assert O.entropy_from_cov_ref(cov) == pytest.approx(want, rel=1e-6 if dt == 'float32' else 1e-12, abs=1e-12)
    tol = 2e-5 if dt == 'float32' else 1e-10
    assert O.entropy_from_cov_chol(cov) == pytest.approx(want, rel=tol, abs=1e-12)


def _hyp(g, pre):
    return O.Hypers(g[pre + 'log_ls'], float(g[pre + 'log_os']), float(g[pre + 'log_noise']))


FLAGS = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1), (1, 0, 1)]


@pytest.mark.parametrize('ci', [0, 1, 2])
def test_g2_predictive_ref(golden, ci):
    g = golden('g2_predictive')
    pre = 'g2_c%d_' % ci
    hyp = _hyp(g, pre)

    def cov_mat(x1, x2=None, white_noise_var=None, add_likelihood_var=False):
        return O.cov_mat_ref(hyp, x1, x2, white_noise_var, add_likelihood_var, dtype=np.float32)

    n_checked = 0
    for tv in (0, 1):
        for xv in (0, 1):
            for (rv, rc, rm) in FLAGS:
                tag = pre + 'tv%d_xv%d_f%d%d%d_' % (tv, xv, rv, rc, rm)
                if tag + 'arity' not in g.files:
                    continue
                res = O.predictive_distribution_ref(
                    cov_mat, g[pre + 'train_x'], g[pre + 'train_y'], g[pre + 'test_x'],
                    g[pre + 'train_var'] if tv else None, g[pre + 'test_var'] if xv else None,
                    return_var=bool(rv), return_cov=bool(rc), return_mi=bool(rm))
                if not isinstance(res, tuple):
                    res = (res,)
                assert len(res) == int(g[tag + 'arity'])
                # tuple convention utils.py:302-319
                want_arity = 1 if not (rv or rc or rm) else (3 if (rc and rm) else 2)
                assert len(res) == want_arity
                for k, r in enumerate(res):
                    want = g[tag + 'r%d' % k]
                    assert np.asarray(r).shape == want.shape
                    assert np.asarray(r).dtype == want.dtype          # fp32 var/cov, fp64 mu
                    # fp32 LAPACK inverse: thread-count dependent summation order
                    assert rel(r, want) < 2e-3, (tag, k, rel(r, want))
                n_checked += 1
    assert n_checked >= 16


@pytest.mark.parametrize('ci', [0, 1, 2])
def test_g2_posterior_chol_vs_reference(golden, ci):
    """fp64 Cholesky form vs the literal (fp32-kernel) reference outputs: an
    fp32-level statement (SURVEY.md section 7 'the oracle is itself mixed-precision')."""
    g = golden('g2_predictive')
    pre = 'g2_c%d_' % ci
    hyp = _hyp(g, pre)
    p = O.posterior_chol(hyp, g[pre + 'train_x'], g[pre + 'train_y'], g[pre + 'test_x'],
                         g[pre + 'train_var'], g[pre + 'test_var'], want_cov=True)
    tag = pre + 'tv1_xv1_f011_'
    mu, cov, mi = g[tag + 'r0'], g[tag + 'r1'], float(g[tag + 'r2'])
    assert rel(p['mu'], mu) < 2e-3
    assert rel(p['cov'], cov) < 2e-3
    assert p['mi'] == pytest.approx(mi, rel=5e-3)
    var = g[pre + 'tv1_xv1_f100_r1']
    assert rel(p['var'], var) < 2e-3


CASES_G3 = [(n, kind, crit) for n in (64, 360) for kind in ('empty', 'static', 'mobile', 'both')
            for crit in ('entropy', 'mutual_information')]


@pytest.mark.parametrize('n,kind,crit', CASES_G3)
def test_g3_greedy(golden, n, kind, crit):
    g = golden('g3_greedy')
    pre = 'g3_n%d_' % n
    tag = pre + kind + '_' + crit
    if tag + '_picks' not in g.files:
        pytest.skip('not generated (bounded generation time)')
    cov = g[pre + 'cov']
    s0, m0 = g[pre + kind + '_static'], g[pre + kind + '_mobile']
    picks_want = list(g[tag + '_picks'])
    ut_want = g[tag + '_ut']
    # efficient fp64 form reproduces picks and per-candidate utilities
    fin = np.isfinite(ut_want)
    # Exact ties (symmetric grid sites) and, for MI, fp32 slogdet noise in the reference
    # (cov_abar is pure fp32, agent.py:331) make the argmax among near-equal candidates
    # rounding-determined: follow the reference's picks, compare utilities pick by pick,
    # and require our own argmax to be optimal within the tolerance.
    picks, ut = O.greedy_fast(cov, s0, m0, 0.1, 1.0, 4, crit, forced_picks=picks_want)
    tol = 1e-7 if crit == 'entropy' else 5e-5
    for p in range(4):
        assert ut_want[p][int(np.argmax(ut[p]))] >= np.max(ut_want[p][fin[p]]) - 2 * tol
    if kind != 'empty' and crit == 'entropy':
        assert O.greedy_fast(cov, s0, m0, 0.1, 1.0, 4, crit)[0] == picks_want
    assert np.array_equal(np.isfinite(ut), fin)
    assert np.max(np.abs(ut[fin] - ut_want[fin])) < tol * max(1.0, np.max(np.abs(ut_want[fin])))
    # literal restatement, on the small field only (O(k n^4))
    if n == 64:
        picks2, ut2 = O.greedy_ref(cov, s0, m0, 0.1, 1.0, 4, crit)
        assert picks2 == picks_want
        assert np.max(np.abs(ut2[fin] - ut_want[fin])) < 1e-10


def test_g3_cov_matches_oracle_kernel(golden):
    g = golden('g3_greedy')
    for n in (64, 360):
        pre = 'g3_n%d_' % n
        hyp = _hyp(g, pre)
        cov = O.cov_mat_ref(hyp, g[pre + 'X'], add_likelihood_var=True, dtype=np.float32)
        assert cov.dtype == np.float32
        assert np.array_equal(cov, g[pre + 'cov'])


@pytest.mark.parametrize('crit', ['entropy', 'mutual_information'])
def test_g4_best_path(golden, crit):
    g = golden('g4_best_path')
    lens = g['g4_paths_len']
    flat = g['g4_paths_flat']
    paths, o = [], 0
    for L in lens:
        paths.append([int(v) for v in flat[o:o + L]])
        o += L
    si = [int(v) for v in g['g4_static_indices']]
    idx, ut = O.best_path_ref(g['g4_cov'], g['g4_static'], g['g4_mobile'], paths, si, 0.1, 1.0, crit)
    assert idx == int(g['g4_%s_idx' % crit])
    assert np.max(np.abs(ut - g['g4_%s_ut' % crit])) < 1e-10
    assert O.best_path_ref(g['g4_cov'], g['g4_static'], g['g4_mobile'], paths[:1], si, 0.1, 1.0, crit)[0] == 0


def test_g5_fusion(golden):
    g = golden('g5_fusion')
    sd, md, os_, om = [], [], 0, 0
    for ls, lm in zip(g['g5_lens_s'], g['g5_lens_m']):
        sd.append(list(g['g5_flat_s'][os_:os_ + ls]))
        md.append(list(g['g5_flat_m'][om:om + lm]))
        os_ += ls
        om += lm
    idx, y, var = O.get_sampled_dataset_ref(sd, md, 0.1, 1.0)
    assert idx == list(g['g5_idx'])
    assert np.allclose(y, g['g5_y'], rtol=1e-14, atol=0)
    assert np.allclose(var, g['g5_var'], rtol=1e-14, atol=0)


@pytest.mark.parametrize('seed', [1, 7])
def test_g6_field(golden, seed):
    g = golden('g6_field')
    R, C = g['g6_s%d_shape' % seed]
    np.random.seed(seed)
    grid, y = O.generate_gaussian_data(int(R), int(C), k=5)
    assert np.array_equal(grid, g['g6_s%d_grid' % seed])
    assert np.allclose(y, g['g6_s%d_y' % seed], rtol=1e-14, atol=0)
    grid2, y2 = O.generate_gaussian_data(int(R), int(C), k=5, rng=np.random.RandomState(seed))
    assert np.allclose(y2, y, rtol=1e-14, atol=0)


# ---- known-answer tests (no reference needed), SURVEY.md section 8c ----
def test_kat_kernel_diag_and_symmetry():
    hyp = O.Hypers(np.log([1.5, 0.7, 2.0]), np.log(2.5), np.log(0.1))
    x = np.random.RandomState(0).uniform(0, 5, (40, 3))
    K = O.kernel_matrix(hyp, x)
    assert np.allclose(np.diag(K), 2.5, rtol=1e-15)
    assert np.array_equal(K, K.T)
    assert np.min(np.linalg.eigvalsh(K)) > -1e-10
    Km = O.kernel_matrix(O.Hypers(hyp.log_lengthscale, hyp.log_outputscale, 0, O.KERNEL_MATERN15), x)
    assert np.allclose(np.diag(Km), 2.5, rtol=1e-15)


def test_kat_one_point_posterior():
    hyp = O.Hypers(np.log([2.0]), np.log(1.7), np.log(0.3))
    xa, ya = np.array([[1.0]]), np.array([0.8])
    xs = np.array([[1.0], [2.5], [40.0]])
    p = O.posterior_chol(hyp, xa, ya, xs)
    k = 1.7 * np.exp(-.5 * ((xs[:, 0] - 1.0) / 2.0) ** 2)
    s = 1.7 + 0.3
    assert np.allclose(p['mu'], 0.8)                       # y - ybar == 0 with one point
    assert np.allclose(p['var'], 1.7 - k * k / s, rtol=1e-14)


def test_kat_entropy_scaled_identity():
    for k, s2 in ((1, 0.5), (7, 2.0), (33, 1e-3)):
        want = k * O.CONST + k / 2 * np.log(s2)
        assert O.entropy_from_cov_ref(s2 * np.eye(k)) == pytest.approx(want, rel=1e-13)
        assert O.entropy_from_cov_chol(s2 * np.eye(k)) == pytest.approx(want, rel=1e-13)


def test_mll_grad_finite_difference():
    rng = np.random.RandomState(3)
    x = rng.uniform(0, 6, (30, 2))
    y = np.sin(x[:, 0]) + 0.1 * rng.standard_normal(30)
    var = np.full(30, 0.01)
    hyp = O.Hypers(np.log([1.3, 2.1]), np.log(0.9), np.log(0.05))
    f0, g = O.mll_and_grad(hyp, x, y, var)
    eps = 1e-6
    h = O.Hypers(hyp.log_lengthscale, hyp.log_outputscale + eps, hyp.log_noise)
    assert (O.mll_and_grad(h, x, y, var)[0] - f0) / eps == pytest.approx(g['log_outputscale'], rel=1e-4)
    h = O.Hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise + eps)
    assert (O.mll_and_grad(h, x, y, var)[0] - f0) / eps == pytest.approx(g['log_noise'], rel=1e-4)
    for d in range(2):
        ls = hyp.log_lengthscale.copy()
        ls[d] += eps
        h = O.Hypers(ls, hyp.log_outputscale, hyp.log_noise)
        assert (O.mll_and_grad(h, x, y, var)[0] - f0) / eps == pytest.approx(g['log_lengthscale'][d], rel=1e-4)
