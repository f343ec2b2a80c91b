"""Posterior / entropy helpers with the reference's names and return conventions
(reference utils.py:10, 188-194, 227-228, 293-319), computed by libalgp_hip.so.

`predictive_distribution` replaces the reference's explicit inverse (utils.py:300), its two dense
GEMMs (utils.py:300-305) and its slogdets (utils.py:314) with: one blocked Cholesky of
K_aa + diag(var) + sigma_n^2 I, one blocked triangular solve for all test points, fused row
reductions for mean and variance, and (only when asked) the full covariance and two log-dets.
"""
import numpy as np

from . import _hip

CONST = .5 * np.log(2 * np.pi * np.exp(1))                    # utils.py:10

_default_ctx = {}


def default_context(dtype=np.float64, device=0):
    """Lazily created module-level context for the free functions below."""
    key = (np.dtype(dtype).name, device)
    if key not in _default_ctx:
        _default_ctx[key] = _hip.Context(dtype, device)
    return _default_ctx[key]


def entropy_from_cov(cov, constant=CONST, ctx=None):
    """H = k*constant + 1/2 log det cov (utils.py:188-194) from a Cholesky factor on the GPU.
    A non positive definite `cov` raises LinAlgError (the reference silently drops slogdet's
    sign, utils.py:193)."""
    if constant is None:
        constant = CONST
    cov = np.asarray(cov)
    k = cov.shape[0]
    if k == 0:
        return 0.0
    dt = np.float32 if cov.dtype == np.float32 else np.float64
    c = ctx if ctx is not None else default_context(dt)
    return c.entropy_from_cov(cov) + k * (constant - CONST)


def predictive_distribution(gp, train_x, train_y, test_x, train_var=None, test_var=None, return_var=False,
                            return_cov=False, return_mi=False):
    """Exact GP posterior at `test_x` (utils.py:293-319), same return-tuple convention:
    mu | (mu, var) | (mu, cov) | (mu, mi) | (mu, cov, mi); `return_mi` overrides `return_var`."""
    c = gp.ctx
    gp.sync_hypers()
    train_x = np.asarray(train_x, dtype=np.float64)
    train_x = train_x.reshape(len(train_x), -1)
    test_x = np.asarray(test_x, dtype=np.float64)
    test_x = test_x.reshape(len(test_x), -1)
    N, M = len(train_x), len(test_x)
    c.set_pool(np.vstack([train_x, test_x]))
    c.set_train(np.arange(N), train_y, train_var)              # mean-centring inside (utils.py:294)
    test_idx = np.arange(N, N + M)
    if not (return_var or return_cov or return_mi):
        c.factorize()                                          # replaces inv(cov_aa), utils.py:300
        return c.posterior_mean(test_idx)                      # mu only: no triangular solve needed
    c.set_candidates(test_idx, prior_includes_noise=False, extra_var=test_var)
    c.fit_and_solve()                                          # the factorisation and V^T = B^T L^-T (utils.py:300-301): ONE task-list
                                                               # launch up to 51 200 test sites, two phases beyond
    mu, var = c.posterior()
    res = None
    if return_var:
        res = (mu, var)
    cov = mi = None
    if return_cov or return_mi:
        cov, mi = c.posterior_cov(want_cov=return_cov, want_mi=return_mi)
    if return_cov:
        res = (mu, cov)
    if return_mi:
        res = (mu, mi)
    if return_cov and return_mi:
        res = (mu, cov, mi)
    return res


def compute_mae(true, pred):
    return np.mean(np.abs(true - pred))                        # utils.py:227-228


def generate_gaussian_data(num_rows, num_cols, k=5, min_var=10, max_var=100, algo='sum'):
    """Mixture-of-Gaussians field on an integer grid (utils.py:90-108); draws from the global
    np.random stream in the reference's order (row means, column means, variances)."""
    xx, yy = np.meshgrid(np.arange(num_cols), np.arange(num_rows))
    grid = np.vstack([yy.flatten(), xx.flatten()]).transpose()
    mr = np.random.uniform(0, num_rows, size=k)
    mc = np.random.uniform(0, num_cols, size=k)
    var = np.random.uniform(min_var, max_var, size=k)
    y = np.zeros(num_rows * num_cols)
    for i in range(k):
        bump = np.exp(-((grid[:, 0] - mr[i]) ** 2 + (grid[:, 1] - mc[i]) ** 2) / var[i])
        y = np.maximum(y, bump) if algo == 'max' else y + bump
    return grid, y


def find_shortest_path(paths_cost):
    """utils.py:365-368: random choice among the minimum-cost paths."""
    cost = np.asarray(paths_cost)
    return int(np.random.choice(np.where(cost == cost.min())[0]))


def find_equi_sample_path(paths_indices, idx):
    """utils.py:371-373: random choice among the paths that collect as many samples as path idx."""
    lens = np.array([len(p) for p in paths_indices])
    return int(np.random.choice(np.where(lens == lens[idx])[0]))
