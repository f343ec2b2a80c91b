"""GPU parity of the greedy / scoring path (reference agent.py:295-403) through the C-ABI,
against the golden vectors captured from the reference and against the fp64 oracle."""
import numpy as np
import pytest

from algp_amd import _hip
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

CRIT = {'entropy': _hip.CRIT_ENTROPY, 'mutual_information': _hip.CRIT_MUTUAL_INFORMATION}


@pytest.fixture(scope='module')
def ctx64():
    c = _hip.Context(np.float64)
    yield c
    c.close()


@pytest.fixture(scope='module')
def ctx32():
    c = _hip.Context(np.float32)
    yield c
    c.close()


def _setup_state(c, static, mobile, ss, sm, y=None):
    sampled = static | mobile
    A = np.where(sampled)[0]
    vf = 1.0 / (1.0 / ss + 1.0 / sm)
    var = np.where(static[A] & mobile[A], vf, np.where(static[A], ss, sm))
    c.set_train(A, np.zeros(len(A)) if y is None else y, var)
    c.factorize()
    cand = np.where(~static)[0]
    c.set_candidates(cand, prior_includes_noise=True)
    c.solve_candidates()
    return A, cand


CASES = [(n, kind, crit) for n in (64, 360) for kind in ('empty', 'static', 'mobile', 'both')
         for crit in ('entropy', 'mutual_information')]


@pytest.mark.parametrize('n,kind,crit', CASES)
@pytest.mark.parametrize('mode', ['cov', 'coords'])
def test_greedy_matches_reference_golden(golden, ctx64, n, kind, crit, mode):
    g = golden('g3_greedy')
    pre = 'g3_n%d_' % n
    tag = pre + kind + '_' + crit
    if tag + '_picks' not in g.files:
        pytest.skip('not generated')
    cov = g[pre + 'cov']                       # fp32 values, as the reference's Agent.cov_matrix
    s0, m0 = g[pre + kind + '_static'], g[pre + kind + '_mobile']
    picks_want = g[tag + '_picks']
    ut_want = g[tag + '_ut']
    c = ctx64
    c.set_hypers(g[pre + 'log_ls'], float(g[pre + 'log_os']), float(g[pre + 'log_noise']))
    if mode == 'cov':
        c.set_pool_cov(cov.astype(np.float64))          # identical inputs to the reference
        tol = 1e-7 if crit == 'entropy' else 5e-5       # MI: fp32 slogdet noise in the reference (agent.py:331)
    else:
        c.set_pool(g[pre + 'X'])                        # fp64 kernel vs the reference's fp32 kernel
        tol = 2e-5 if crit == 'entropy' else 1e-4
    A, cand = _setup_state(c, s0, m0, 0.1 ** 2, 1.0 ** 2)
    picks, ut = c.greedy(CRIT[crit], 0.1, 1.0, 4, forced_picks=picks_want, want_utilities=True)
    assert list(picks) == list(picks_want)
    full = np.full((4, n), -np.inf)
    full[:, cand] = ut
    fin = np.isfinite(ut_want)
    assert np.array_equal(np.isfinite(full), fin)
    err = np.max(np.abs(full[fin] - ut_want[fin]))
    assert err < tol * max(1.0, np.max(np.abs(ut_want[fin]))), err
    for p in range(4):          # our own argmax is optimal within the tolerance
        assert ut_want[p][int(np.argmax(full[p]))] >= np.max(ut_want[p][fin[p]]) - 2 * tol


@pytest.mark.parametrize('kind', ['static', 'mobile', 'both'])
def test_greedy_free_run_picks(golden, ctx64, kind):
    """Without forcing: the device argmax reproduces the reference's picks where they are not
    rounding-determined ties (non-empty states, entropy criterion)."""
    g = golden('g3_greedy')
    for n in (64, 360):
        pre = 'g3_n%d_' % n
        c = ctx64
        c.set_hypers(g[pre + 'log_ls'], float(g[pre + 'log_os']), float(g[pre + 'log_noise']))
        c.set_pool_cov(g[pre + 'cov'].astype(np.float64))
        _setup_state(c, g[pre + kind + '_static'], g[pre + kind + '_mobile'], 0.01, 1.0)
        picks = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4)
        assert list(picks) == list(g[pre + kind + '_entropy_picks'])


@pytest.mark.parametrize('dtname', ['f64', 'f32'])
def test_greedy_vs_oracle_larger_field(ctx64, ctx32, dtname):
    """30 x 30 field (the reference's default size, env.py:20-21): picks and utilities vs the
    fp64 oracle; fp32 context within 1e-3."""
    c = ctx64 if dtname == 'f64' else ctx32
    rng = np.random.RandomState(11)
    xx, yy = np.meshgrid(np.arange(30), np.arange(30))
    X = np.vstack([yy.ravel(), xx.ravel()]).T.astype(np.float64)
    X = X + 0.05 * rng.standard_normal(X.shape)          # break exact lattice ties
    n = len(X)
    hyp = O.Hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
    static = np.zeros(n, bool)
    mobile = np.zeros(n, bool)
    perm = rng.permutation(n)
    static[perm[:200]] = True
    mobile[perm[150:400]] = True
    C = O.kernel_matrix(hyp, X) + hyp.noise * np.eye(n)
    picks_o, ut_o = O.greedy_fast(C, static, mobile, 0.1, 1.0, 6, 'entropy')
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise)
    c.set_pool(X)
    A, cand = _setup_state(c, static, mobile, 0.01, 1.0)
    picks, ut = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 6, forced_picks=np.array(picks_o), want_utilities=True)
    full = np.full((6, n), -np.inf)
    full[:, cand] = ut
    fin = np.isfinite(ut_o)
    tol = 1e-9 if dtname == 'f64' else 1e-3
    assert np.max(np.abs(full[fin] - ut_o[fin])) < tol * np.max(np.abs(ut_o[fin]))
    if dtname == 'f64':
        assert list(c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 0)) == []
        _setup_state(c, static, mobile, 0.01, 1.0)
        assert list(c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 6)) == picks_o


def test_commit_remote_winner_equals_local(ctx64):
    """Sharded scoring: a rank commits a winner that is not among its local candidates; its
    candidates' statistics must match the single-shard run."""
    rng = np.random.RandomState(3)
    X = rng.uniform(0, 20, (500, 2))
    hyp = O.Hypers(np.log([2.5, 2.5]), 0.0, np.log(1e-2))
    n = len(X)
    static = np.zeros(n, bool)
    mobile = np.zeros(n, bool)
    perm = rng.permutation(n)
    static[perm[:100]] = True
    mobile[perm[80:200]] = True
    c = ctx64
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise)
    c.set_pool(X)
    A, cand = _setup_state(c, static, mobile, 0.01, 1.0)
    picks, ut = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 5, want_utilities=True)
    # shard: only the second half of the candidates is local; commit the global winners
    half = cand[len(cand) // 2:]
    vf = 1.0 / (1 / 0.01 + 1 / 1.0)
    var = np.where(static[A] & mobile[A], vf, np.where(static[A], 0.01, 1.0))
    c.set_train(A, np.zeros(len(A)), var)
    c.factorize()
    c.set_candidates(half, prior_includes_noise=True)
    c.solve_candidates()
    for p in range(5):
        s = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
        want = ut[p][len(cand) // 2:]
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(s), fin)
        assert np.max(np.abs(s[fin] - want[fin])) < 1e-10
        c.commit_pick(int(picks[p]), 0.1, 1.0)


@pytest.mark.parametrize('dtname', ['f64', 'f32'])
def test_fit_and_solve_equals_separate_calls(dtname):
    """algp_fit_and_solve (one planning step's fit + candidate solve) must give what algp_factorize +
    algp_solve_candidates give, call after call, and report a non-positive pivot (N = 1500: the one-launch
    dependency-driven factorisation)."""
    c = _hip.Context(np.float64 if dtname == 'f64' else np.float32)
    rng = np.random.RandomState(21)
    N, M = 1500, 9000
    X = rng.uniform(0, 60, (N + M, 2))
    hyp = O.Hypers(np.log([3.0, 2.0]), 0.0, np.log(2e-2))
    y = np.sin(X[:N, 0] / 5) + 0.1 * rng.standard_normal(N)
    var = rng.choice([0.01, 1.0], N)
    var[N - 50:] = 1.0                             # the 50 train sites that are also candidates are mobile-sampled
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise)
    c.set_pool(X)
    c.set_train(np.arange(N), y, var)
    cand = np.arange(N - 50, N + M)               # includes 50 train sites: unit rows
    c.set_candidates(cand, prior_includes_noise=True)
    c.factorize()
    c.solve_candidates()
    mu0, d0 = c.posterior()
    s0 = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
    ld0, a0 = c.logdet(), c.alpha()
    for _ in range(3):                             # repeat: event reuse across calls
        c.fit_and_solve()
        mu1, d1 = c.posterior()
        s1 = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
        # V^T, variance, utilities and the factor: the same bits; the mean and alpha go through z, which rides along as a row
        # of the candidates' panel in the one launch (tile products) where the two calls substitute: rounding
        rt = 1e-12 if dtname == 'f64' else 2e-5
        assert np.array_equal(d0, d1) and np.array_equal(s0, s1)
        assert np.max(np.abs(mu0 - mu1)) <= rt * max(1.0, np.max(np.abs(mu0)))
        assert not np.any(np.isnan(s1))
        assert c.logdet() == ld0 and np.max(np.abs(c.alpha() - a0)) <= rt * max(1.0, np.max(np.abs(a0)))
    picks = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 3)
    c.factorize()
    c.solve_candidates()
    assert list(c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 3)) == list(picks)
    # not positive definite is reported with its pivot, not swallowed
    v2 = var.copy()
    v2[700] = -5.0
    c.set_train(np.arange(N), y, v2)
    with pytest.raises(np.linalg.LinAlgError) as ei:
        c.fit_and_solve()
    assert ei.value.pivot == 701
    c.close()


def test_argmax_over_a_long_score_vector_is_the_first_maximum():
    """np.argmax semantics (agent.py:349) for score vectors of 32 768 entries and more, which algp_argmax and the lazy pick
    chain reduce in two launches (64 workgroups' first maxima, then one wave): every site listed three times as a candidate,
    so the maximum is attained three times -- the first position must win, whatever workgroup holds it; then the chain
    algp_greedy runs (argmax -> refresh -> refresh -> argmax on the device) against pick-by-pick scoring on the host."""
    c = _hip.Context(np.float64)
    rng = np.random.RandomState(11)
    n, ntr = 14000, 300
    X = rng.uniform(0, 120, (n, 2))
    c.set_hypers(np.log([3.0, 2.5]), 0.0, np.log(1e-2))
    c.set_pool(X)
    A = rng.permutation(n)[:ntr]
    c.set_train(A, np.zeros(ntr), np.full(ntr, 0.01))
    c.factorize()
    base = np.setdiff1d(np.arange(n), A)
    cand = np.r_[base, base, base]                              # 41 100 candidates, every utility three times
    assert len(cand) >= 32768
    c.set_candidates(cand, prior_includes_noise=True)
    c.solve_candidates()
    for _ in range(3):
        s = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
        want = int(np.argmax(s))
        assert want < len(base) and s[want] == s[want + len(base)] == s[want + 2 * len(base)]
        pos, pool, val = c.argmax()
        assert (pos, pool, val) == (want, int(cand[want]), s[want])
        c.commit_pick(pool, 0.1, 1.0)
    c.factorize()
    c.solve_candidates()
    picks = [int(p) for p in c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 3)]
    c.solve_candidates()
    for q in range(3):
        s = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
        assert picks[q] == int(cand[int(np.argmax(s))])
        c.commit_pick(picks[q], 0.1, 1.0)
    c.close()


@pytest.mark.parametrize('mode', ['coords', 'cov'])
@pytest.mark.parametrize('dtname', ['f64', 'f32'])
def test_lazy_greedy_equals_full_pass(dtname, mode):
    """algp_greedy without utilities resolves each pick with algp_best_candidate (only rows that can still win
    are brought up to date after a pick); picks, and the state once every row has caught up, must equal the
    loop that scores every row before every pick, bit for bit."""
    dt = np.float64 if dtname == 'f64' else np.float32
    c = _hip.Context(dt)
    rng = np.random.RandomState(5)
    n = 3200
    X = rng.uniform(0, 70, (n, 2))
    hyp = O.Hypers(np.log([3.0, 2.5]), 0.0, np.log(1e-2))
    static = np.zeros(n, bool)
    mobile = np.zeros(n, bool)
    perm = rng.permutation(n)
    static[perm[:300]] = True
    mobile[perm[250:700]] = True                   # 50 fused sites, 400 mobile-only ones stay candidates (unit rows)
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise)
    if mode == 'cov':
        Cm = (O.kernel_matrix(hyp, X) + hyp.noise * np.eye(n)).astype(dt)
        c.set_pool_cov(Cm)
    else:
        c.set_pool(X)
    k = 9
    _setup_state(c, static, mobile, 0.01, 1.0)
    cand = np.where(~static)[0]
    full_picks = []
    for _ in range(k):                              # the full pass: every row updated after every pick
        s = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
        w = int(cand[int(np.argmax(s))])
        full_picks.append(w)
        c.commit_pick(w, 0.1, 1.0)
    s_full = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
    var_full = c.posterior()[1]
    # algp_best_candidate on its own: same winner and value as scores + argmax, at every step
    _setup_state(c, static, mobile, 0.01, 1.0)
    for w_full in full_picks[:5]:
        pos, w, val = c.best_candidate(_hip.CRIT_ENTROPY, 0.1, 1.0)
        assert w == w_full and cand[pos] == w
        c.commit_pick(w, 0.1, 1.0)
    pos, w, val = c.best_candidate(_hip.CRIT_ENTROPY, 0.1, 1.0)
    s = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
    assert w == full_picks[5] and val == s[pos] == np.max(s)
    _setup_state(c, static, mobile, 0.01, 1.0)
    lazy_picks = list(c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4))
    lazy_picks += list(c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, k - 4))      # a second call continues from stale rows
    assert lazy_picks == full_picks
    s_lazy = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)                          # flushes
    var_lazy = c.posterior()[1]
    assert np.array_equal(s_lazy, s_full)
    assert np.array_equal(var_lazy, var_full, equal_nan=True)
    # mixing: lazy picks, then a full-pass commit, then lazy again
    _setup_state(c, static, mobile, 0.01, 1.0)
    mixed = list(c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 3))
    s = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
    w = int(cand[int(np.argmax(s))])
    c.commit_pick(w, 0.1, 1.0)
    mixed += [w] + list(c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, k - 4))
    assert mixed == full_picks
    assert np.array_equal(c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0), s_full)
    c.close()


def test_mi_criterion_at_5000_sites_matches_oracle():
    """MI criterion (agent.py:330-339) at a single-GPU size beyond the goldens: pool n = 5 000, |A| = 1 000 sampled
    (static, mobile and both), every utility of the first pick against the fp64 oracle (two pool-wide inverses)."""
    rng = np.random.RandomState(9)
    R, Cc = 50, 100
    grid, _ = O.generate_gaussian_data(R, Cc, k=5, rng=rng)
    X = grid.astype(np.float64)
    n = len(X)
    hyp = O.Hypers(np.log([1.5, 1.5]), 0.0, np.log(1e-2))
    perm = rng.permutation(n)
    static = np.zeros(n, bool)
    mobile = np.zeros(n, bool)
    static[perm[:500]] = True
    mobile[perm[400:1000]] = True                                  # 100 sites carry both readings
    A = np.where(static | mobile)[0]
    var = np.where(static[A] & mobile[A], 1.0 / (1.0 / 0.01 + 1.0), np.where(static[A], 0.01, 1.0))
    Cm = O.kernel_matrix(hyp, X) + hyp.noise * np.eye(n)
    picks, ut = O.greedy_fast(Cm, static, mobile, 0.1, 1.0, 1, 'mutual_information')
    c = _hip.Context(np.float64)
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise)
    c.set_pool(X)
    c.set_train(A, np.zeros(len(A)), var)
    c.factorize()
    cand = np.where(~static)[0]
    c.set_candidates(cand, prior_includes_noise=True)
    c.solve_candidates()
    s = c.scores(_hip.CRIT_MUTUAL_INFORMATION, 0.1, 1.0)
    want = ut[0][cand]
    scale = np.max(np.abs(want))
    assert np.max(np.abs(s - want)) <= 1e-7 * scale, (np.max(np.abs(s - want)), scale)
    # the argmax is optimal within the tolerance (near-ties on the lattice are decided by rounding)
    assert want[int(np.argmax(s))] >= np.max(want) - 1e-7 * scale
    c.close()


@pytest.mark.parametrize('R,Cc,nA', [(10, 12, 30), (25, 26, 120), (30, 40, 200), (45, 50, 300)],
                         ids=['n120_one_tile', 'n650_one_block_and_a_tile', 'n1200_two_blocks_and_a_remainder', 'n2250_four_blocks_ragged'])
def test_mi_criterion_in_place_inverses_at_ragged_sizes(R, Cc, nA):
    """Round 6: each of the MI criterion's pool-wide inverses is the triangle X = L^-T written over its own factor
    (potrf.hip: trinv_upper_inplace; the row reductions walk the triangle only).  Pools whose padded size is one tile, one
    512-column block + a tile, several blocks + a remainder: the first pick's utilities AND two further picks (each a
    column S^-1 e_c = X (X^T e_c) of both inverses) against the fp64 oracle, agent.py:330-339."""
    rng = np.random.RandomState(R * 100 + Cc)
    grid, _ = O.generate_gaussian_data(R, Cc, k=5, rng=rng)
    X = grid.astype(np.float64)
    n = len(X)
    hyp = O.Hypers(np.log([1.5, 1.5]), 0.0, np.log(1e-2))
    perm = rng.permutation(n)
    static = np.zeros(n, bool)
    mobile = np.zeros(n, bool)
    static[perm[:nA // 2]] = True
    mobile[perm[nA // 3:nA]] = True
    A = np.where(static | mobile)[0]
    var = np.where(static[A] & mobile[A], 1.0 / (1.0 / 0.01 + 1.0), np.where(static[A], 0.01, 1.0))
    Cm = O.kernel_matrix(hyp, X) + hyp.noise * np.eye(n)
    picks, ut = O.greedy_fast(Cm, static, mobile, 0.1, 1.0, 3, 'mutual_information')
    c = _hip.Context(np.float64)
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise)
    c.set_pool(X)
    c.set_train(A, np.zeros(len(A)), var)
    c.factorize()
    cand = np.where(~static)[0]
    c.set_candidates(cand, prior_includes_noise=True)
    c.solve_candidates()
    got, gut = c.greedy(_hip.CRIT_MUTUAL_INFORMATION, 0.1, 1.0, 3, forced_picks=[int(q) for q in picks], want_utilities=True)
    for k in range(3):
        want = ut[k][cand]
        fin = np.isfinite(want)
        assert np.array_equal(fin, np.isfinite(gut[k]))
        scale = np.max(np.abs(want[fin]))
        assert np.max(np.abs(gut[k][fin] - want[fin])) <= 1e-7 * scale, (k, np.max(np.abs(gut[k][fin] - want[fin])), scale)
    c.close()


@pytest.mark.parametrize('dtname', ['f64', 'f32'])
def test_mi_criterion_rank1_updates_follow_the_oracle_pick_by_pick(dtname):
    """Picks 2..k of the MI criterion fold the previous winner into the two resident inverse diagonals (O(n^2) rank-1
    updates, api_greedy.hip mi_apply_pick) where the reference and the oracle refactorise two pool-wide matrices per pick
    (agent.py:330-339): every utility of 6 forced picks against the oracle's from-scratch terms -- new sites AND
    mobile-sampled sites among the picks (the latter change the noise of a site that stays outside the complement)."""
    dt = np.float64 if dtname == 'f64' else np.float32
    rng = np.random.RandomState(21)
    grid, _ = O.generate_gaussian_data(40, 50, k=5, rng=rng)
    X = grid.astype(np.float64)
    n = len(X)
    hyp = O.Hypers(np.log([1.5, 1.5]), 0.0, np.log(1e-2))
    perm = rng.permutation(n)
    static = np.zeros(n, bool)
    mobile = np.zeros(n, bool)
    static[perm[:300]] = True
    mobile[perm[250:600]] = True
    A = np.where(static | mobile)[0]
    var = np.where(static[A] & mobile[A], 1.0 / (1.0 / 0.01 + 1.0), np.where(static[A], 0.01, 1.0))
    Cm = O.kernel_matrix(hyp, X) + hyp.noise * np.eye(n)
    cand = np.where(~static)[0]
    mob_only = np.where(mobile & ~static)[0]
    free, _ = O.greedy_fast(Cm, static, mobile, 0.1, 1.0, 2, 'mutual_information')
    forced = [free[0], int(mob_only[3]), free[1], int(mob_only[77]), int(cand[5]), int(mob_only[100])]
    assert len(set(forced)) == 6
    picks, ut = O.greedy_fast(Cm, static, mobile, 0.1, 1.0, 6, 'mutual_information', forced_picks=forced)
    c = _hip.Context(dt)
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise)
    c.set_pool(X)
    c.set_train(A, np.zeros(len(A)), var)
    c.factorize()
    c.set_candidates(cand, prior_includes_noise=True)
    c.solve_candidates()
    got, gut = c.greedy(_hip.CRIT_MUTUAL_INFORMATION, 0.1, 1.0, 6, forced_picks=forced, want_utilities=True)
    assert [int(p) for p in got] == forced
    rel = 1e-7 if dtname == 'f64' else 3e-3
    for k in range(6):
        want = ut[k][cand]
        live = np.isfinite(want)
        assert np.array_equal(np.isfinite(gut[k]), live), k
        scale = np.max(np.abs(want[live]))
        assert np.max(np.abs(gut[k][live] - want[live])) <= rel * scale, (k, np.max(np.abs(gut[k][live] - want[live])), scale)
    # a second run in the same context rebuilds the inverses after the solve and gives the same bits
    c.solve_candidates()
    got2, gut2 = c.greedy(_hip.CRIT_MUTUAL_INFORMATION, 0.1, 1.0, 6, forced_picks=forced, want_utilities=True)
    assert np.array_equal(gut2, gut)
    # picks committed under the ENTROPY criterion before the first MI scoring: the inverses are then built for the state
    # that already contains them (mi_build with picks), and the next MI pick is folded in on top
    c.solve_candidates()
    c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 2, forced_picks=forced[:2])
    s2 = c.scores(_hip.CRIT_MUTUAL_INFORMATION, 0.1, 1.0)
    want = ut[2][cand]
    live = np.isfinite(want)
    assert np.max(np.abs(s2[live] - want[live])) <= rel * np.max(np.abs(want[live]))
    c.commit_pick(forced[2], 0.1, 1.0)
    s3 = c.scores(_hip.CRIT_MUTUAL_INFORMATION, 0.1, 1.0)
    want = ut[3][cand]
    live = np.isfinite(want)
    assert np.max(np.abs(s3[live] - want[live])) <= rel * np.max(np.abs(want[live]))
    c.close()


def test_mi_criterion_reports_the_scratch_it_needs():
    """At a pool the two pool-wide scratch matrices cannot fit (2 x 200 000^2 x 8 B = 640 GB) the MI criterion must
    fail up front with ALGP_ERR_OOM and the byte count, leaving the context usable."""
    rng = np.random.RandomState(1)
    n = 200000
    X = rng.uniform(0, 500, (n, 2))
    c = _hip.Context(np.float64)
    c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
    c.set_pool(X)
    A = np.arange(200)
    c.set_train(A, np.zeros(200), np.full(200, 0.01))
    c.factorize()
    cand = np.arange(200, 1200)
    c.set_candidates(cand, prior_includes_noise=True)
    c.solve_candidates()
    with pytest.raises(MemoryError) as ei:                          # ALGP_ERR_OOM
        c.scores(_hip.CRIT_MUTUAL_INFORMATION, 0.1, 1.0)
    assert 'bytes for n_pool' in str(ei.value) and str(n) in str(ei.value)
    s = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)                      # the entropy criterion still works
    assert np.all(np.isfinite(s))
    c.close()


_SHARDED_ONE = r"""
import numpy as np, sys
sys.path.insert(0, %r)
from algp_amd import _hip
rng = np.random.RandomState(4)
X = rng.uniform(0, 30, (1500, 2))
n = len(X)
static = np.zeros(n, bool); mobile = np.zeros(n, bool)
perm = rng.permutation(n)
static[perm[:200]] = True; mobile[perm[150:400]] = True
c = _hip.Context(np.float64)
c.set_hypers(np.log([2.5, 2.5]), 0.0, np.log(1e-2))
c.set_pool(X)
A = np.where(static | mobile)[0]
vf = 1.0 / (1.0 / 0.01 + 1.0)
var = np.where(static[A] & mobile[A], vf, np.where(static[A], 0.01, 1.0))
c.set_train(A, np.zeros(len(A)), var)
c.factorize()
cand = np.where(~static)[0]
c.set_candidates(cand, prior_includes_noise=True)
c.solve_candidates()
want, ut = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 6, want_utilities=True)
c.comm_init(1, 0, _hip.Context.comm_unique_id())
try:
    c.greedy_sharded(_hip.CRIT_MUTUAL_INFORMATION, 0.1, 1.0, 2)
    raise SystemExit('the MI criterion must be refused')
except ValueError:
    pass
for rep in range(2):
    c.factorize()
    c.solve_candidates()
    got, gut = c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 6, want_utilities=True)
    assert [int(p) for p in got] == [int(p) for p in want], (got, want)
    for p in range(6):
        assert gut[p] == np.nanmax(ut[p]), (p, gut[p], np.nanmax(ut[p]))
c.comm_destroy()
c.close()
print('SHARDED-ONE-OK')
"""


def test_greedy_sharded_world_of_one_equals_greedy():
    """The collective behind the ABI (algp_comm_init + algp_greedy_sharded) in an RCCL world of one rank: same picks
    and utilities as algp_greedy, call after call.  In a process of its own: RCCL is opened with dlopen and must be
    the only copy in its process (PyTorch, which other tests import, ships a second one).  More ranks need one GPU
    per rank: unmeasured here, see DESIGN.md."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', _SHARDED_ONE % repo], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'SHARDED-ONE-OK' in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize('dtname,D,kern', [('f64', 2, 'rbf'), ('f64', 3, 'rbf'), ('f64', 6, 'matern'), ('f32', 2, 'rbf'), ('f32', 5, 'rbf')])
def test_score_paths_against_explicit_logdets(dtname, D, kern):
    """algp_score_paths (a8 / f3, agent.py:386-399: one slogdet of the enlarged covariance per path) against exactly
    that, in NumPy: dH = H(A u path) - H(A) with a mobile reading at every distinct site of the path; a site that already
    is a train row gets a second row (cross entry k + sigma_n^2).  Coordinate widths 2..6 cover the three padded widths of
    the kernel's coordinate loads (2, 4, 8), both kernels and both precisions."""
    dt = np.float64 if dtname == 'f64' else np.float32
    rng = np.random.RandomState(D * 7 + (dt == np.float32))
    n, nA = 260, 110
    X = rng.uniform(0, 9, (n, D))
    kid = O.KERNEL_RBF if kern == 'rbf' else O.KERNEL_MATERN15
    hyp = O.Hypers(np.log(rng.uniform(2.0, 4.0, D)), np.log(0.8), np.log(2e-2), kid)
    A = np.sort(rng.permutation(n)[:nA])
    varA = rng.choice([0.01, 1.0], nA)
    c = _hip.Context(dt)
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise, kernel=kid)
    c.set_pool(X)
    c.set_train(A, np.zeros(nA), varA)
    c.factorize()
    c.set_candidates(np.arange(n), prior_includes_noise=True)
    c.solve_candidates()
    mobile_std = 0.7
    others = np.setdiff1d(np.arange(n), A)
    paths = np.full((24, 20), -1, dtype=np.int64)
    want = np.zeros(24)
    K = O.kernel_matrix(hyp, X) + hyp.noise * np.eye(n)
    SA = K[np.ix_(A, A)] + np.diag(varA)
    ldA = np.linalg.slogdet(SA)[1]
    for p in range(24):
        L = rng.randint(1, 18)
        sites = list(rng.permutation(others)[:L])
        if p % 3 == 0:
            sites[rng.randint(L)] = int(A[rng.randint(nA)])              # crosses a train site: a second row for it
        if p % 4 == 0:
            sites.append(sites[0])                                       # a site crossed twice counts once
        row = list(sites)
        row.insert(rng.randint(len(row) + 1), -1)                        # an off-field pose in the middle
        paths[p, :len(row)] = row
        uniq = list(dict.fromkeys(int(s) for s in sites))
        idx = np.r_[A, uniq].astype(int)
        S = K[np.ix_(idx, idx)] + np.diag(np.r_[varA, np.full(len(uniq), mobile_std ** 2)])
        want[p] = len(uniq) * O.CONST + 0.5 * (np.linalg.slogdet(S)[1] - ldA)
    got = c.score_paths(paths, mobile_std)
    c.close()
    tol = 1e-9 if dt == np.float64 else 2e-3
    assert np.max(np.abs(got - want)) < tol * max(1.0, np.max(np.abs(want))), np.max(np.abs(got - want))


@pytest.mark.parametrize('dtname,kern', [('f64', 'rbf'), ('f64', 'matern'), ('f32', 'rbf')])
def test_score_paths_of_up_to_256_sites(dtname, kern):
    """Paths as long as config 5's field rows (agent.py:358-403 scores every enumerated path; env.py:197-310: a path runs along
    field rows, up to ~250 sites at 250 x 200): 65 .. 256 distinct sites per path go through the batched form -- rows
    gathered, Gram matrices as one batched MFMA product, blocks factored as 2 x 2 tiles of 128 -- and must equal one NumPy
    slogdet of the enlarged covariance per path, exactly as the <= 64-site LDS kernel does.  Lengths 65, 128, 129, 200, 256
    (both padded sizes, both sides of the tile boundary), sites that are train rows already (second rows), a repeated site,
    an off-field pose; a batch mixing a short path with long ones; 257 distinct sites is an error."""
    dt = np.float64 if dtname == 'f64' else np.float32
    rng = np.random.RandomState(3 + (dt == np.float32))
    n, nA, D = 900, 300, 2
    X = rng.uniform(0, 30, (n, D))
    kid = O.KERNEL_RBF if kern == 'rbf' else O.KERNEL_MATERN15
    hyp = O.Hypers(np.log([2.5, 3.0]), np.log(0.9), np.log(2e-2), kid)
    A = np.sort(rng.permutation(n)[:nA])
    varA = rng.choice([0.01, 1.0], nA)
    c = _hip.Context(dt)
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise, kernel=kid)
    c.set_pool(X)
    c.set_train(A, np.zeros(nA), varA)
    c.factorize()
    c.set_candidates(np.arange(n), prior_includes_noise=True)
    c.solve_candidates()
    mobile_std = 0.7
    others = np.setdiff1d(np.arange(n), A)
    K = O.kernel_matrix(hyp, X) + hyp.noise * np.eye(n)
    SA = K[np.ix_(A, A)] + np.diag(varA)
    ldA = np.linalg.slogdet(SA)[1]

    def build(lengths, width):
        paths = np.full((len(lengths), width), -1, dtype=np.int64)
        want = np.zeros(len(lengths))
        for p, L in enumerate(lengths):
            sites = list(rng.permutation(others)[:L])
            for q in range(min(5, L) if p % 2 == 0 else 0):
                sites[rng.randint(L)] = int(A[rng.randint(nA)])          # crosses train sites: second rows
            uniq = list(dict.fromkeys(int(s) for s in sites))
            row = list(sites)
            if len(row) + 2 <= width:
                row.append(row[0])                                       # a site crossed twice counts once
                row.insert(rng.randint(len(row) + 1), -1)                # an off-field pose in the middle
            paths[p, :len(row)] = row
            idx = np.r_[A, uniq].astype(int)
            S = K[np.ix_(idx, idx)] + np.diag(np.r_[varA, np.full(len(uniq), mobile_std ** 2)])
            want[p] = len(uniq) * O.CONST + 0.5 * (np.linalg.slogdet(S)[1] - ldA)
        return paths, want
    tol = 1e-9 if dt == np.float64 else 3e-3
    for lengths, width in (([65, 128, 100, 7], 140), ([129, 200, 256, 30, 256, 131], 262)):
        paths, want = build(lengths, width)
        got = c.score_paths(paths, mobile_std)
        assert np.max(np.abs(got - want)) < tol * max(1.0, np.max(np.abs(want))), (lengths, np.max(np.abs(got - want)))
    too_long = np.full((1, 300), -1, dtype=np.int64)
    too_long[0, :257] = others[:257]
    with pytest.raises(ValueError):
        c.score_paths(too_long, mobile_std)
    c.close()
