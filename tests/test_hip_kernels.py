"""GPU parity tests of the HIP building blocks, called through the C-ABI (ctypes), against
NumPy / the oracle on the same seeded inputs.  fp64: 1e-9 or tighter where the algebra allows,
never looser than north_star's 1e-5 relative; fp32: 1e-3 relative (stated per test)."""
import numpy as np
import pytest

from algp_amd import _hip
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

DT = [np.float64, np.float32]


def tol(dt, t64, t32):
    return t64 if np.dtype(dt) == np.float64 else t32


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


@pytest.fixture(scope='module')
def ctxs():
    c = {np.dtype(dt): _hip.Context(dt) for dt in DT}
    yield c
    for v in c.values():
        v.close()


def test_mfma_fragment_layout(ctxs):
    assert ctxs[np.dtype(np.float64)].selftest_mfma() == 0


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('shape', [(128, 128, 128), (200, 72, 333), (384, 256, 640), (1, 1, 1), (130, 129, 17)])
def test_gemm_nt(ctxs, dt, shape):
    m, n, k = shape
    rng = np.random.RandomState(m * 7 + n * 3 + k)
    A = rng.standard_normal((m, k)).astype(dt)
    B = rng.standard_normal((n, k)).astype(dt)
    Cm = rng.standard_normal((m, n)).astype(dt)
    D = ctxs[np.dtype(dt)].gemm_nt(A, B, alpha=-1.5, beta=0.5, Cm=Cm)
    want = -1.5 * A.astype(np.float64) @ B.astype(np.float64).T + 0.5 * Cm
    scale = np.abs(A.astype(np.float64)) @ np.abs(B.astype(np.float64)).T + np.abs(Cm)
    err = np.max(np.abs(D - want) / scale)
    assert err < tol(dt, 1e-14, 2e-6), err
    D0 = ctxs[np.dtype(dt)].gemm_nt(A, B)         # beta = 0 path never reads C
    assert np.max(np.abs(D0 - A.astype(np.float64) @ B.astype(np.float64).T) / scale) < tol(dt, 1e-14, 2e-6)


def test_gemm_exact_integers_asymmetric(ctxs):
    """A = I-like / asymmetric integer data: catches swapped row/col maps exactly."""
    m, n, k = 256, 128, 128
    A = (np.arange(m * k).reshape(m, k) % 13 - 6).astype(np.float64)
    B = (np.arange(n * k).reshape(n, k) % 7 - 3 + (np.arange(n)[:, None] % 5)).astype(np.float64)
    for dt in DT:
        D = ctxs[np.dtype(dt)].gemm_nt(A.astype(dt), B.astype(dt))
        assert np.array_equal(D.astype(np.float64), A @ B.T)


def spd(n, rng, cond_shift=1.0):
    a = rng.standard_normal((n, n))
    return a @ a.T / n + cond_shift * np.eye(n)


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('n', [1, 5, 64, 128, 129, 300, 1000, 1500, 2600])   # from 8 tiles of 128 on: the one-launch task list
def test_cholesky(ctxs, dt, n):
    rng = np.random.RandomState(n)
    A = spd(n, rng).astype(dt)
    L, logdet = ctxs[np.dtype(dt)].cholesky(A)
    Lw = np.linalg.cholesky(A.astype(np.float64))
    assert np.array_equal(np.triu(L, 1), np.zeros_like(L))
    assert relerr(L, Lw) < tol(dt, 1e-12, 5e-5), relerr(L, Lw)
    resid = L.astype(np.float64) @ L.astype(np.float64).T - A
    assert np.max(np.abs(resid)) / np.max(np.abs(A)) < tol(dt, 1e-13, 5e-6)
    assert logdet == pytest.approx(2 * np.sum(np.log(np.diag(Lw))), rel=tol(dt, 1e-11, 1e-4), abs=tol(dt, 1e-11, 1e-4))


def test_cholesky_lehmer_and_pascal_exact(ctxs):
    """Known-answer: Pascal matrix has the integer Cholesky factor of binomials."""
    from math import comb
    n = 12
    P = np.array([[comb(i + j, i) for j in range(n)] for i in range(n)], dtype=np.float64)
    Lw = np.array([[comb(i, j) for j in range(n)] for i in range(n)], dtype=np.float64)
    L, logdet = ctxs[np.dtype(np.float64)].cholesky(P)
    assert np.max(np.abs(L - Lw)) < 1e-6          # cond(P_12) ~ 1e12
    assert abs(logdet) < 1e-5                      # det = 1
    i, j = np.meshgrid(np.arange(1, 41), np.arange(1, 41), indexing='ij')
    Leh = np.minimum(i, j) / np.maximum(i, j)
    L2, _ = ctxs[np.dtype(np.float64)].cholesky(Leh)
    assert np.max(np.abs(L2 @ L2.T - Leh)) < 1e-14


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('n,bad', [(200, 150), (1500, 1301), (1500, 5)])    # launch sequence; task list: late and first tile
def test_cholesky_not_pd_reports_pivot(ctxs, dt, n, bad):
    rng = np.random.RandomState(0)
    A = spd(n, rng)
    A[bad, bad] = -5.0
    with pytest.raises(np.linalg.LinAlgError) as ei:
        ctxs[np.dtype(dt)].cholesky(A.astype(dt))
    assert ei.value.pivot == bad + 1
    # the context is usable afterwards, and a matrix with a NaN is refused as well (no hang in the task list's waits)
    L, _ = ctxs[np.dtype(dt)].cholesky(spd(n, rng).astype(dt))
    assert np.all(np.isfinite(L))
    A = spd(n, rng)
    A[bad, bad // 2] = A[bad // 2, bad] = np.nan
    with pytest.raises(np.linalg.LinAlgError):
        ctxs[np.dtype(dt)].cholesky(A.astype(dt))


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('n,m', [(128, 128), (300, 77), (513, 260)])
def test_trsm_right_lt(ctxs, dt, n, m):
    rng = np.random.RandomState(n + m)
    L = np.linalg.cholesky(spd(n, rng)).astype(dt)
    B = rng.standard_normal((m, n)).astype(dt)
    X = ctxs[np.dtype(dt)].trsm_right_lt(L, B)
    from scipy.linalg import solve_triangular
    Xw = solve_triangular(L.astype(np.float64), B.astype(np.float64).T, lower=True).T
    assert relerr(X, Xw) < tol(dt, 1e-12, 1e-4), relerr(X, Xw)
    assert relerr(X.astype(np.float64) @ L.astype(np.float64).T, B) < tol(dt, 1e-13, 2e-5)


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('D', [1, 2, 3, 6])
@pytest.mark.parametrize('kernel', [O.KERNEL_RBF, O.KERNEL_MATERN15])
def test_kernel_matrix(ctxs, dt, D, kernel):
    rng = np.random.RandomState(D)
    hyp = O.Hypers(np.log(rng.uniform(1.0, 3.0, D)), np.log(1.7), np.log(0.03), kernel)
    c = ctxs[np.dtype(dt)]
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise, kernel)
    x1 = rng.uniform(0, 10, (137, D)).astype(dt)
    x2 = rng.uniform(0, 10, (61, D)).astype(dt)
    var = rng.uniform(0.01, 1, 137).astype(dt)
    K = c.kernel_matrix(x1)
    Kw = O.kernel_matrix(hyp, x1.astype(np.float64))
    t = tol(dt, 1e-13, 2e-5)          # fp32: input rounding of coordinates already moves K by ~1e-6
    assert K.dtype == np.dtype(dt) and K.shape == (137, 137)
    assert relerr(K, Kw) < t
    assert np.array_equal(K, K.T)
    assert np.allclose(np.diag(K), hyp.outputscale, rtol=tol(dt, 1e-15, 1e-7))
    Kx = c.kernel_matrix(x1, x2)
    assert Kx.shape == (137, 61)
    assert relerr(Kx, O.kernel_matrix(hyp, x1.astype(np.float64), x2.astype(np.float64))) < t
    Kd = c.kernel_matrix(x1, diag_add=var, add_likelihood_var=True)
    want = O.cov_mat_ref(hyp, x1.astype(np.float64), None, var.astype(np.float64), True, dtype=np.float64)
    assert relerr(Kd, want) < t
    # empty and single-point edge cases
    assert c.kernel_matrix(x1[:1]).shape == (1, 1)


def test_kernel_value_exp_within_two_ulp_down_to_underflow(ctxs):
    """The fp64 kernel values go through the library's own exp (common.h: kexp -- Cody-Waite reduction, Taylor to r^13,
    ldexp; the matrix build was VALU-bound on the general routine): element by element within 2 ulp of NumPy's exp over the
    whole range an RBF value can take, exp(0) = 1 exactly, a clean underflow to 0 -- for the ScaleKernel(RBF) of
    models.py:218 and the Matern-1.5 of models.py:219-220."""
    c = ctxs[np.dtype(np.float64)]
    rng = np.random.RandomState(1)
    t = np.r_[0.0, np.sort(rng.uniform(0, 745.0, 30000)), np.linspace(700, 760, 2000)]
    x2 = np.sqrt(2.0 * t)[:, None]
    x1 = np.zeros((3, 1))
    c.set_hypers(np.zeros(1), 0.0, np.log(1e-2), _hip.KERNEL_RBF)
    K = c.kernel_matrix(x1, x2)[0]
    ref = np.exp(-0.5 * (0.0 - x2[:, 0]) ** 2)
    big = ref > 1e-290
    ulp = np.abs(K[big] - ref[big]) / np.spacing(ref[big])
    assert ulp.max() <= 2.0, ulp.max()
    assert K[0] == 1.0
    assert np.all(K[~big] <= 1.1e-290) and np.all(K >= 0) and K[-1] == 0.0
    c.set_hypers(np.zeros(1), 0.0, np.log(1e-2), _hip.KERNEL_MATERN15)
    r = np.r_[0.0, np.sort(rng.uniform(0, 400.0, 20000))]
    Km = c.kernel_matrix(x1, r[:, None])[0]
    a = r * 1.7320508075688772
    refm = (1.0 + a) * np.exp(-a)
    ok = refm > 1e-290
    assert np.max(np.abs(Km[ok] - refm[ok]) / np.spacing(refm[ok])) <= 4.0


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('N,M,D', [(5, 3, 2), (50, 40, 2), (200, 100, 6), (300, 513, 2), (129, 1, 3)])
def test_factor_and_posterior(ctxs, dt, N, M, D):
    rng = np.random.RandomState(N * 31 + M)
    x = rng.uniform(0, 12, (N + M, D))
    hyp = O.Hypers(np.log(rng.uniform(2, 4, D)), np.log(1.3), np.log(0.05))
    y = np.sin(x[:N, 0] / 3) + 0.1 * rng.standard_normal(N)
    tv = rng.choice([0.01, 1.0], N)
    xv = rng.uniform(0.05, 0.2, M)
    c = ctxs[np.dtype(dt)]
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise)
    c.set_pool(x)
    c.set_train(np.arange(N), y, tv)
    c.factorize()
    ref = O.posterior_chol(hyp, x[:N].astype(dt), y.astype(dt), x[N:].astype(dt), tv.astype(dt), xv.astype(dt), want_cov=True)
    t = tol(dt, 1e-9, 1e-3)
    assert c.logdet() == pytest.approx(ref['logdet'], rel=t, abs=t)
    assert relerr(c.alpha(), ref['alpha']) < tol(dt, 1e-8, 5e-2)      # alpha amplifies by cond(S)
    c.set_candidates(np.arange(N, N + M), prior_includes_noise=False, extra_var=xv)
    c.solve_candidates()
    mu, var = c.posterior()
    assert relerr(mu, ref['mu']) < t, relerr(mu, ref['mu'])
    assert relerr(var, ref['var']) < t, relerr(var, ref['var'])
    mu2 = c.posterior_mean(np.arange(N, N + M))
    assert relerr(mu2, ref['mu']) < tol(dt, 1e-8, 2e-2)
    cov, mi = c.posterior_cov(want_cov=True, want_mi=True)
    assert relerr(cov, ref['cov']) < t
    assert mi == pytest.approx(ref['mi'], rel=tol(dt, 1e-8, 5e-3), abs=tol(dt, 1e-8, 5e-3))
    mll_want = -.5 * (y - y.mean()) @ ref['alpha'] - .5 * ref['logdet'] - .5 * N * np.log(2 * np.pi)
    assert c.mll() == pytest.approx(mll_want, rel=tol(dt, 1e-9, 2e-3))
    H = c.entropy()
    assert H == pytest.approx(N * O.CONST + .5 * ref['logdet'], rel=t)


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('k', [0, 1, 5, 64, 200])
def test_entropy_from_cov(ctxs, dt, k):
    rng = np.random.RandomState(k)
    cov = spd(k, rng).astype(dt) if k else np.zeros((0, 0), dt)
    H = ctxs[np.dtype(dt)].entropy_from_cov(cov)
    assert H == pytest.approx(O.entropy_from_cov_chol(cov), rel=tol(dt, 1e-12, 1e-5), abs=1e-12)


@pytest.mark.parametrize('dt', DT)
def test_set_entropy_and_inverse_diag(ctxs, dt):
    rng = np.random.RandomState(5)
    x = rng.uniform(0, 10, (150, 2))
    hyp = O.Hypers(np.log([1.5, 2.0]), 0.0, np.log(0.02))
    c = ctxs[np.dtype(dt)]
    c.set_hypers(hyp.log_lengthscale, hyp.log_outputscale, hyp.log_noise)
    c.set_pool(x)
    idx = rng.permutation(150)[:70]
    var = rng.uniform(0.01, 1, 70)
    S = O.kernel_matrix(hyp, x[idx].astype(dt)) + hyp.noise * np.eye(70) + np.diag(var)
    assert c.set_entropy(idx, var) == pytest.approx(O.entropy_from_cov_chol(S), rel=tol(dt, 1e-11, 1e-4))
    d, H = c.set_inverse_diag(idx, var)
    assert relerr(d, np.diag(np.linalg.inv(S))) < tol(dt, 1e-10, 1e-3)
    assert H == pytest.approx(O.entropy_from_cov_chol(S), rel=tol(dt, 1e-11, 1e-4))
    assert c.set_entropy(np.zeros(0, np.int64)) == 0.0


@pytest.mark.parametrize('dt', DT)
def test_cholesky_task_list_stall_is_reported_and_survivable(ctxs, dt):
    """A hand-off that never happens (test hook: the task with ticket 7 does not publish its tile) must end the one-launch
    factorisation by its own spin limit (0.2 s under the hook, 2 s otherwise) with ALGP_ERR_HIP "stalled" -- not hang --
    and the next factorisation in the same context is right again (the launch state is rebuilt per launch)."""
    import time
    c = ctxs[np.dtype(dt)]
    rng = np.random.RandomState(5)
    n = 1700                                               # 14 tiles of 128: the task list
    A = spd(n, rng).astype(dt)
    L0, ld0 = c.cholesky(A)
    c.debug_dag_stall(7)
    t0 = time.time()
    with pytest.raises(_hip.AlgpError) as ei:                # (that it RETURNS is the test; how fast is only printed)
        c.cholesky(A)
    assert 'stalled' in str(ei.value) and ei.value.code == _hip.ERR_HIP
    print('stalled launch gave up after %.2f s' % (time.time() - t0))
    L1, ld1 = c.cholesky(A)
    assert np.array_equal(L0, L1) and ld0 == ld1
