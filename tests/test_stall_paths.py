"""The one-launch kernels that hand data between workgroups through flags -- forward and backward substitution
(potrf.hip: trsv_chain_kernel, trsv_chain_back_kernel) -- bound every wait (2 s; 0.2 s under the test hook) and, when a
wait runs out, abandon the launch and set the context's sticky stall word.  VERDICT r3 / ADVICE r3: the forward kernel's
abort was recorded in a word nothing read -- the caller got ALGP_OK and a garbage z.  Here: every call that hands such a
result to the host returns ALGP_ERR_HIP "stalled", the context stays usable, and the repeated call gives the bits of an
undisturbed one.  Also the backward substitution itself (alpha = L^-T z, reference utils.py:300-301 through inv) against
the oracle at sizes from one block to 21."""
import numpy as np
import pytest

from algp_amd import _hip
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

HYP = O.Hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))


def _ctx(dtype, N, M=300, seed=0):
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 50, (N + M, 2))
    var = rng.choice([0.01, 1.0], N)
    y = rng.uniform(0, 1, N)
    c = _hip.Context(dtype)
    c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
    c.set_pool(X)
    c.set_train(np.arange(N), y, var)
    return c, X, y, var


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-9), (np.float32, 5e-3)], ids=['f64', 'f32'])
@pytest.mark.parametrize('N', [100, 128, 129, 1400, 2600])
def test_backward_substitution_in_one_launch_against_the_oracle(dtype, tol, N):
    c, X, y, var = _ctx(dtype, N)
    c.factorize()
    a = c.alpha()
    S = O.kernel_matrix(HYP, X[:N]) + np.diag(var + HYP.noise)
    want = np.linalg.solve(S, y - y.mean())
    assert np.max(np.abs(a - want)) <= tol * max(1.0, np.max(np.abs(want)))
    mu = c.posterior_mean(np.arange(N, N + 300))
    ref = O.posterior_chol(HYP, X[:N], y, X[N:N + 300], var)
    assert np.max(np.abs(mu - ref['mu'])) <= tol * max(1.0, np.max(np.abs(ref['mu'])))
    c.factorize()
    assert np.array_equal(a, c.alpha()), 'fixed summation order: the same bits in every run'
    c.close()


def test_a_stalled_forward_substitution_fails_the_factorisation():
    c, X, y, var = _ctx(np.float64, 1400)
    c.factorize()
    ld0, a0 = c.logdet(), c.alpha()
    c.debug_trsv_stall(3)
    with pytest.raises(_hip.AlgpError) as ei:
        c.factorize()
    assert 'stalled' in str(ei.value) and ei.value.code == _hip.ERR_HIP
    with pytest.raises(ValueError):
        c.logdet()                                          # no factor is on offer after the failure
    c.factorize()
    assert c.logdet() == ld0 and np.array_equal(c.alpha(), a0)
    c.close()


def test_a_stalled_backward_substitution_fails_alpha_and_the_mean():
    c, X, y, var = _ctx(np.float64, 1400)
    c.factorize()
    a0 = c.alpha()
    c.factorize()
    c.debug_trsv_stall(5)
    with pytest.raises(_hip.AlgpError) as ei:
        c.alpha()
    assert 'stalled' in str(ei.value)
    assert np.array_equal(c.alpha(), a0)                    # recomputed, not the abandoned vector
    c.factorize()
    c.debug_trsv_stall(4)                                   # (block 0 is the last one of the backward sweep: nobody waits for it)
    with pytest.raises(_hip.AlgpError):
        c.posterior_mean(np.arange(1400, 1500))
    mu = c.posterior_mean(np.arange(1400, 1500))
    ref = O.posterior_chol(HYP, X[:1400], y, X[1400:1500], var)
    assert np.max(np.abs(mu - ref['mu'])) <= 1e-9
    c.close()


def test_a_stalled_remote_row_fails_the_commit_and_leaves_the_state_intact():
    """algp_commit_pick of a site that is not a local candidate rebuilds its row with a forward substitution: a stall there
    must fail the commit (ALGP_ERR_HIP), take the pick back, and a repeated commit must give the undisturbed result."""
    c, X, y, var = _ctx(np.float64, 1400, M=400)
    cand = np.arange(1400, 1700)
    outsider = 1750
    c.factorize()
    c.set_candidates(cand, prior_includes_noise=True)
    c.solve_candidates()
    c.commit_pick(outsider, 0.1, 1.0)
    u_ok = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
    c.solve_candidates()
    c.debug_trsv_stall(2)
    with pytest.raises(_hip.AlgpError) as ei:
        c.commit_pick(outsider, 0.1, 1.0)
    assert 'stalled' in str(ei.value)
    c.commit_pick(outsider, 0.1, 1.0)
    assert np.array_equal(c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0), u_ok)
    c.close()


def test_commit_pick_that_is_taken_back_leaves_the_site_selectable():
    """a pick whose posterior variance is not positive is rolled back by algp_commit_pick (ALGP_ERR_NOT_PD); the
    device-side bookkeeping must not have retired the candidate meanwhile.  Provoked with a negative static variance
    term: 1 / sqrt(pv + ss) with ss = (i * std)^2 cannot be made negative, so the square root's argument is forced
    through a NaN standard deviation."""
    c, X, y, var = _ctx(np.float64, 600, M=200)
    cand = np.arange(600, 800)
    c.factorize()
    c.set_candidates(cand, prior_includes_noise=True)
    c.solve_candidates()
    u0 = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
    with pytest.raises(np.linalg.LinAlgError):
        c.commit_pick(int(cand[7]), float('nan'), 1.0)
    u1 = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
    assert np.isfinite(u1[7]) and np.array_equal(u0, u1)
    c.close()
