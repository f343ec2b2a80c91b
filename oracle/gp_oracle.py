"""CPU oracle for the GP-inference hot path of sumitsk/algp.

TEST INFRASTRUCTURE ONLY.  Nothing under ``algp_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and there only as the checker / reported baseline.

Two families of functions live here:

* ``*_ref``  -- reference-faithful restatements: the same algorithm, dtype flow
  and call order as the reference (explicit ``inv``, one fresh ``slogdet`` per
  candidate, fp32 kernel matrix with in-place fp32 diagonal adds).  These are
  pinned to the reference by the golden vectors in ``tests/golden/`` that were
  produced by importing the reference's own NumPy code (see
  ``tests/golden/make_golden.py``).
* ``*_chol`` / ``*_fast`` -- fp64 restatements with the efficient algebra
  (Cholesky + rank-1 identities).  They are what the HIP
  path is compared against at 1e-5 relative (fp64) / 1e-3 (fp32).

Parity status: PINNED for the NumPy half of the path (predictive_distribution,
entropy_from_cov, Agent.greedy, Agent.best_path, Agent.get_sampled_dataset,
generate_gaussian_data) by golden vectors generated from the reference itself.
UNPINNED for the kernel-matrix values (GPyTorch is an absent, un-pinned
dependency of the reference: README.md:9, models.py:5-9): ``kernel_matrix``
below restates the published ScaleKernel(RBFKernel(ard_num_dims=D)) formula
  k(x, x') = exp(log_outputscale) * exp(-1/2 * sum_d ((x_d - x'_d) / exp(log_lengthscale_d))^2)
and MaternKernel(nu=1.5); hyper-parameters are always explicit inputs.

Every function cites the reference file:line it follows.
"""
import numpy as np

# reference utils.py:10
CONST = .5 * np.log(2 * np.pi * np.exp(1))

KERNEL_RBF = 0
KERNEL_MATERN15 = 1


class Hypers(object):
    """The D+2 scalars of the reference's ExactGPModel (models.py:206-254).

    Parameter names in the reference's state dict (run.py:35-37, models.py:180):
    ``kernel_covar_module.base_kernel.log_lengthscale`` (D),
    ``kernel_covar_module.log_outputscale``, ``likelihood.log_noise``.
    """

    def __init__(self, log_lengthscale, log_outputscale=0.0, log_noise=0.0, kernel=KERNEL_RBF):
        self.log_lengthscale = np.atleast_1d(np.asarray(log_lengthscale, dtype=np.float64))
        self.log_outputscale = float(log_outputscale)
        self.log_noise = float(log_noise)
        self.kernel = kernel

    @property
    def D(self):
        return len(self.log_lengthscale)

    @property
    def noise(self):
        return float(np.exp(self.log_noise))

    @property
    def outputscale(self):
        return float(np.exp(self.log_outputscale))


# --------------------------------------------------------------------------
# a1: kernel matrix (models.py:161-181, kernel composition models.py:213-220)
# --------------------------------------------------------------------------
def kernel_matrix(hyp, x1, x2=None, dtype=np.float64):
    """Scale(RBF-ARD) / Scale(Matern-1.5) kernel matrix, closed form.

    Differences are formed directly (no |a|^2+|b|^2-2ab expansion) so the
    matrix is exactly symmetric with an exact ``outputscale`` diagonal.
    """
    dtype = np.dtype(dtype)
    x1 = np.asarray(x1, dtype=dtype)
    x2 = x1 if x2 is None else np.asarray(x2, dtype=dtype)
    inv_ls = np.exp(-hyp.log_lengthscale).astype(dtype)
    a = x1 * inv_ls
    b = x2 * inv_ls
    r2 = np.zeros((a.shape[0], b.shape[0]), dtype=dtype)
    for d in range(a.shape[1]):
        diff = a[:, d][:, None] - b[:, d][None, :]
        r2 += diff * diff
    os_ = dtype.type(np.exp(hyp.log_outputscale))
    if hyp.kernel == KERNEL_RBF:
        return os_ * np.exp(dtype.type(-0.5) * r2)
    if hyp.kernel == KERNEL_MATERN15:
        r = np.sqrt(r2) * dtype.type(np.sqrt(3.0))
        return os_ * (dtype.type(1.0) + r) * np.exp(-r)
    raise NotImplementedError(hyp.kernel)


def cov_mat_ref(hyp, x1, x2=None, white_noise_var=None, add_likelihood_var=False, dtype=np.float32):
    """GPR.cov_mat semantics (models.py:161-181).

    The reference casts inputs to fp32 (utils.py:19), evaluates the kernel in
    fp32 and then adds ``np.diag(white_noise_var)`` (models.py:175-176) and
    ``exp(log_noise) * I`` (models.py:179-180) *in place*, so the result stays
    float32.  ``x2 is None or equal(x1, x2)`` takes the symmetric branch
    (models.py:169-170).
    """
    cov = kernel_matrix(hyp, x1, x2, dtype=dtype)
    if white_noise_var is not None:
        cov += np.diag(white_noise_var)
    if add_likelihood_var:
        cov += hyp.noise * np.eye(len(cov))
    return cov


# --------------------------------------------------------------------------
# a5: entropy (utils.py:188-194)
# --------------------------------------------------------------------------
def entropy_from_cov_ref(cov, constant=CONST):
    """H = k*constant + 1/2 * log|det cov| ; slogdet's sign is dropped (utils.py:193)."""
    if constant is None:
        constant = CONST
    return cov.shape[0] * constant + .5 * np.linalg.slogdet(cov)[1].item()


def entropy_from_cov_chol(cov):
    """Same value through a Cholesky factor: k*CONST + sum(log diag L)."""
    cov = np.asarray(cov, dtype=np.float64)
    if cov.shape[0] == 0:
        return 0.0
    L = np.linalg.cholesky(cov)
    return cov.shape[0] * CONST + float(np.sum(np.log(np.diag(L))))


# --------------------------------------------------------------------------
# a4: predictive_distribution (utils.py:293-319)
# --------------------------------------------------------------------------
def predictive_distribution_ref(cov_mat, train_x, train_y, test_x, train_var=None, test_var=None,
                                return_var=False, return_cov=False, return_mi=False):
    """Literal restatement of utils.py:293-319 given a ``cov_mat`` callable with
    GPR.cov_mat's signature.  Return-tuple convention per utils.py:302-319
    (``return_mi`` overrides ``return_var``; ``return_cov and return_mi`` -> 3-tuple).
    """
    train_y_mean = np.mean(train_y)                                                       # :294
    cov_aa = cov_mat(x1=train_x, white_noise_var=train_var, add_likelihood_var=True)      # :296
    cov_xx = cov_mat(x1=test_x, white_noise_var=test_var)                                 # :297
    cov_xa = cov_mat(x1=test_x, x2=train_x)                                               # :298
    mat1 = np.dot(cov_xa, np.linalg.inv(cov_aa))                                          # :300
    mu = np.dot(mat1, (train_y - train_y_mean)) + train_y_mean                            # :301
    if not (return_var or return_cov or return_mi):
        return mu
    cov = cov_xx - np.dot(mat1, cov_xa.T)                                                 # :305
    res = None
    if return_var:
        res = (mu, np.diag(cov))
    if return_cov:
        res = (mu, cov)
    if return_mi:
        mi = entropy_from_cov_ref(cov_xx) - entropy_from_cov_ref(cov)                     # :314
        res = (mu, mi)
    if return_cov and return_mi:
        res = (mu, cov, mi)
    return res


def posterior_chol(hyp, train_x, train_y, test_x, train_var=None, test_var=None, want_cov=False):
    """fp64 Cholesky form of utils.py:293-319 (what the HIP path implements).

    Returns dict(mu, var, cov|None, mi|None, logdet, alpha, z).
    L L^T = K_AA + diag(train_var) + sigma_n^2 I ; V = L^-1 K_AX ;
    mu = ybar + V^T z, z = L^-1 (y - ybar) ; var = diag(K_XX)+test_var - colsum(V^2).
    """
    train_x = np.asarray(train_x, np.float64)
    test_x = np.asarray(test_x, np.float64)
    y = np.asarray(train_y, np.float64)
    ybar = y.mean()
    S = kernel_matrix(hyp, train_x) + hyp.noise * np.eye(len(train_x))
    if train_var is not None:
        S = S + np.diag(np.asarray(train_var, np.float64))
    L = np.linalg.cholesky(S)
    from scipy.linalg import solve_triangular
    z = solve_triangular(L, y - ybar, lower=True)
    alpha = solve_triangular(L.T, z, lower=False)
    Kax = kernel_matrix(hyp, train_x, test_x)
    V = solve_triangular(L, Kax, lower=True)
    mu = ybar + V.T @ z
    prior = np.full(len(test_x), hyp.outputscale)
    if test_var is not None:
        prior = prior + np.asarray(test_var, np.float64)
    var = prior - np.sum(V * V, axis=0)
    out = dict(mu=mu, var=var, cov=None, mi=None, alpha=alpha, z=z,
               logdet=2.0 * float(np.sum(np.log(np.diag(L)))))
    if want_cov:
        Kxx = kernel_matrix(hyp, test_x)
        if test_var is not None:
            Kxx = Kxx + np.diag(np.asarray(test_var, np.float64))
        cov = Kxx - V.T @ V
        out['cov'] = cov
        try:
            out['mi'] = entropy_from_cov_chol(Kxx) - entropy_from_cov_chol(cov)
        except np.linalg.LinAlgError:
            out['mi'] = None
    return out


# --------------------------------------------------------------------------
# a10: sensor fusion (agent.py:92-117)
# --------------------------------------------------------------------------
def get_sampled_dataset_ref(static_data, mobile_data, static_std, mobile_std):
    """Per location: mean static reading, mean mobile reading, precision-weighted
    fusion (agent.py:95-109).  Returns (indices, y, var)."""
    ys, vs, idx = [], [], []
    for i in range(len(static_data)):
        has_m = len(mobile_data[i]) > 0
        has_s = len(static_data[i]) > 0
        if has_m and has_s:
            yc = np.mean(mobile_data[i])
            ys_ = np.mean(static_data[i])
            yeq = (mobile_std ** 2 * ys_ + static_std ** 2 * yc) / (mobile_std ** 2 + static_std ** 2)
            var = 1 / (1 / (static_std ** 2) + 1 / (mobile_std ** 2))
        elif has_s:
            yeq = np.mean(static_data[i])
            var = static_std ** 2
        elif has_m:
            yeq = np.mean(mobile_data[i])
            var = mobile_std ** 2
        else:
            continue
        ys.append(yeq)
        vs.append(var)
        idx.append(i)
    return idx, np.array(ys), np.array(vs)


# --------------------------------------------------------------------------
# a7: greedy (agent.py:295-356) -- naive, reference-faithful
# --------------------------------------------------------------------------
def _fused_var(static_var, mobile_var, sampled):
    with np.errstate(divide='ignore'):
        return 1.0 / (1.0 / static_var[sampled] + 1.0 / mobile_var[sampled])


def greedy_ref(cov_matrix, static_sampled, mobile_sampled, static_std, mobile_std, num_samples,
               criterion='entropy'):
    """Naive greedy of agent.py:295-356: one fresh slogdet per candidate.

    Returns (picks, utilities[num_samples, n]) with -inf for skipped candidates.
    """
    n = cov_matrix.shape[0]
    static_sampled = np.array(static_sampled, dtype=bool)
    mobile_sampled = np.array(mobile_sampled, dtype=bool)
    mobile_var = np.full(n, np.inf)
    mobile_var[mobile_sampled] = mobile_std ** 2
    static_var = np.full(n, np.inf)
    static_var[static_sampled] = static_std ** 2

    sampled = static_sampled | mobile_sampled
    var = _fused_var(static_var, mobile_var, sampled)
    cov_v = cov_matrix[sampled].T[sampled].T + np.diag(var)                   # :308
    ent_v = entropy_from_cov_ref(cov_v)                                       # :309

    cumm, picks, all_ut = [], [], []
    for _ in range(num_samples):                                              # :313
        utilities = np.full(n, -np.inf)
        cond = ent_v + sum(cumm)                                              # :315
        for i in range(n):
            if static_sampled[i]:                                             # :318
                continue
            static_sampled[i] = True
            static_var[i] = static_std ** 2
            sampled = static_sampled | mobile_sampled
            var = _fused_var(static_var, mobile_var, sampled)
            cov_a = cov_matrix[sampled].T[sampled].T + np.diag(var)           # :328
            ent_a = entropy_from_cov_ref(cov_a)
            if criterion == 'mutual_information':                             # :330-339
                cov_abar = cov_matrix[~sampled].T[~sampled].T
                ent_abar = entropy_from_cov_ref(cov_abar)
                with np.errstate(divide='ignore'):
                    precision = 1.0 / static_var + 1.0 / mobile_var
                    precision[precision == 0] = np.inf
                    var_all = 1.0 / precision
                cov_all = cov_matrix + np.diag(var_all)
                ent_all = entropy_from_cov_ref(cov_all)
                ut = ent_a + ent_abar - ent_all
            else:
                ut = ent_a - cond                                             # :341
            utilities[i] = ut
            static_sampled[i] = False
            static_var[i] = np.inf
        best = int(np.argmax(utilities))                                      # :349 (first max)
        cumm.append(utilities[best])
        picks.append(best)
        static_sampled[best] = True
        static_var[best] = static_std ** 2
        all_ut.append(utilities)
    return picks, np.array(all_ut)


# --------------------------------------------------------------------------
# a7 efficient form (SURVEY.md section 7 identities), fp64
# --------------------------------------------------------------------------
def greedy_fast(cov_matrix, static_sampled, mobile_sampled, static_std, mobile_std, num_samples,
                criterion='entropy', forced_picks=None):
    r"""Same picks/utilities as ``greedy_ref`` from one Cholesky + rank-1 updates.

    With S = C_AA + D, G = B^T S^-1 B maintained implicitly through V = L^-1 B where
    column j of B is C[A, j] (j not sampled) or e_j (j mobile-sampled, not static):
      j not in A :  dH = CONST + 1/2 log(pv_j + ss),  pv_j = C_jj - |V_j|^2
      j in A     :  dH = 1/2 log(1 + delta * s_jj),   s_jj = |V_j|^2, delta = v_fused - sm
    Each pick appends one row to V ("greedy update").
    The MI criterion (agent.py:330-339) adds H(Abar \ i) and H(all_i) through the
    diagonals of two more inverses, recomputed per pick.

    ``forced_picks``: commit these indices instead of the argmax (the reference's MI
    utilities carry fp32 slogdet noise of ~1e-5 -- ``cov_abar`` is a pure fp32 matrix,
    agent.py:331 -- so near-ties are broken by noise there; tests follow the
    reference's own picks and compare utilities pick by pick).
    """
    from scipy.linalg import solve_triangular
    C = np.asarray(cov_matrix, np.float64)
    n = C.shape[0]
    static_sampled = np.array(static_sampled, dtype=bool)
    mobile_sampled = np.array(mobile_sampled, dtype=bool)
    ss, sm = static_std ** 2, mobile_std ** 2
    vf = 1.0 / (1.0 / ss + 1.0 / sm)
    delta = vf - sm

    A = np.where(static_sampled | mobile_sampled)[0]
    D = np.where(static_sampled[A] & mobile_sampled[A], vf, np.where(static_sampled[A], ss, sm))
    N = len(A)
    pos_in_A = -np.ones(n, dtype=np.int64)
    pos_in_A[A] = np.arange(N)
    cand = np.where(~static_sampled)[0]
    in_A = mobile_sampled[cand]                      # candidates that are mobile-sampled
    M = len(cand)
    if N > 0:
        L = np.linalg.cholesky(C[np.ix_(A, A)] + np.diag(D))
        B = np.zeros((N, M))
        B[:, ~in_A] = C[np.ix_(A, cand[~in_A])]
        B[pos_in_A[cand[in_A]], np.where(in_A)[0]] = 1.0
        V = solve_triangular(L, B, lower=True)
    else:
        V = np.zeros((0, M))
    d = np.where(in_A, 0.0, C[cand, cand]) + np.where(in_A, 1.0, -1.0) * np.sum(V * V, axis=0)
    # d_j = pv_j (not in A) or s_jj (in A)
    alive = np.ones(M, dtype=bool)
    picks, all_ut = [], []
    cur_static = static_sampled.copy()
    cur_mobile = mobile_sampled.copy()
    for _ in range(num_samples):
        with np.errstate(invalid='ignore', divide='ignore'):
            ut_c = np.where(in_A, .5 * np.log1p(delta * d), CONST + .5 * np.log(d + ss))
        if criterion == 'mutual_information':
            ut_c = ut_c + _mi_extra_terms(C, cur_static, cur_mobile, cand, ss, sm)
        ut = np.full(n, -np.inf)
        ut[cand[alive]] = ut_c[alive]
        best = int(np.argmax(ut)) if forced_picks is None else int(forced_picks[len(picks)])
        picks.append(best)
        all_ut.append(ut)
        c = int(np.where(cand == best)[0][0])
        l = V[:, c].copy()
        t = l @ V
        if in_A[c]:
            gamma = delta / (1.0 + delta * d[c])
            r = np.sqrt(-gamma) * (0.0 - t)
        else:
            lam = np.sqrt(d[c] + ss)
            bprime = np.where(in_A, 0.0, C[best, cand])
            r = (bprime - t) / lam
        d = d + np.where(in_A, 1.0, -1.0) * r * r
        V = np.vstack([V, r[None, :]])
        alive[c] = False
        cur_static[best] = True
    ut_arr = np.array(all_ut)
    if criterion == 'entropy':
        return picks, ut_arr
    return picks, ut_arr


def _mi_extra_terms(C, static_sampled, mobile_sampled, cand, ss, sm):
    r"""For each candidate i: H(Abar \ i) - H(all_i) (agent.py:331-339) + the
    constant pieces, through diag of two inverses.  fp64, O(n^3) per call.

    H(A u i) is added by the caller as dH; the reference's ut = ent_a + ent_abar - ent_all
    is *not* relative to H(A), so H(A) is added here too.
    """
    n = C.shape[0]
    sampled = static_sampled | mobile_sampled
    A = np.where(sampled)[0]
    vf = 1.0 / (1.0 / ss + 1.0 / sm)
    D = np.where(static_sampled[A] & mobile_sampled[A], vf, np.where(static_sampled[A], ss, sm))
    H_A = entropy_from_cov_chol(C[np.ix_(A, A)] + np.diag(D)) if len(A) else 0.0
    Abar = np.where(~sampled)[0]
    if len(Abar):
        Cb = C[np.ix_(Abar, Abar)]
        H_Abar = entropy_from_cov_chol(Cb)
        inv_b = np.diag(np.linalg.inv(Cb))
    else:
        H_Abar, inv_b = 0.0, np.zeros(0)
    pos_b = -np.ones(n, dtype=np.int64)
    pos_b[Abar] = np.arange(len(Abar))
    var_all = np.zeros(n)
    var_all[A] = D
    Call = C + np.diag(var_all)
    H_all = entropy_from_cov_chol(Call)
    inv_all = np.diag(np.linalg.inv(Call))
    out = np.zeros(len(cand))
    delta = vf - sm
    for k, i in enumerate(cand):
        if sampled[i]:          # mobile-sampled: Abar unchanged; all: noise sm -> vf
            with np.errstate(invalid='ignore'):   # picked (now static) sites are masked by the caller
                out[k] = H_A + H_Abar - (H_all + .5 * np.log1p(delta * inv_all[i]))
        else:                   # leaves Abar; all: noise 0 -> ss
            out[k] = H_A + (H_Abar - CONST + .5 * np.log(inv_b[pos_b[i]])) \
                - (H_all + .5 * np.log1p(ss * inv_all[i]))
    return out


# --------------------------------------------------------------------------
# a8: best_path (agent.py:358-403)
# --------------------------------------------------------------------------
def best_path_ref(cov_matrix, static_sampled, mobile_sampled, paths_mobile_indices, static_indices,
                  static_std, mobile_std, criterion='entropy'):
    """Utility of each enumerated path; argmax.  Returns (idx, utilities).
    Early-out 0 when only one path (agent.py:362-363) -> utilities is None."""
    if len(paths_mobile_indices) == 1:
        return 0, None
    n = cov_matrix.shape[0]
    org_mobile = np.array(mobile_sampled, dtype=bool)
    static_sampled = np.array(static_sampled, dtype=bool)
    static_sampled[static_indices] = True
    static_var = np.full(n, np.inf)
    static_var[static_sampled] = static_std ** 2
    all_ut = []
    for path in paths_mobile_indices:
        mob = org_mobile.copy()
        mob[path] = True
        mobile_var = np.full(n, np.inf)
        mobile_var[mob] = mobile_std ** 2
        sampled = static_sampled | mob
        var = _fused_var(static_var, mobile_var, sampled)
        cov_a = cov_matrix[sampled].T[sampled].T + np.diag(var)
        ent_a = entropy_from_cov_ref(cov_a)
        if criterion == 'mutual_information':
            cov_abar = cov_matrix[~sampled].T[~sampled].T
            ent_abar = entropy_from_cov_ref(cov_abar)
            with np.errstate(divide='ignore'):
                precision = 1.0 / static_var + 1.0 / mobile_var
                precision[precision == 0] = np.inf
                var_all = 1.0 / precision
            ent_all = entropy_from_cov_ref(cov_matrix + np.diag(var_all))
            ut = ent_a + ent_abar - ent_all
        else:
            ut = ent_a
        all_ut.append(ut)
    return int(np.argmax(all_ut)), np.array(all_ut)


# --------------------------------------------------------------------------
# synthetic mixture-of-Gaussians field (utils.py:90-108)
# --------------------------------------------------------------------------
def generate_gaussian_data(num_rows, num_cols, k=5, min_var=10, max_var=100, algo='sum', rng=None):
    """RNG call order of the reference: uniform(0,R,k), uniform(0,C,k), uniform(min,max,k)
    (utils.py:94-97).  ``rng=None`` uses the global ``np.random`` like the reference."""
    rng = np.random if rng is None else rng
    xx, yy = np.meshgrid(np.arange(num_cols), np.arange(num_rows))
    grid = np.vstack([yy.flatten(), xx.flatten()]).transpose()
    means_x = rng.uniform(0, num_rows, size=k)
    means_y = rng.uniform(0, num_cols, size=k)
    means = np.vstack([means_x, means_y]).transpose()
    variances = rng.uniform(min_var, max_var, size=k)
    y = np.zeros(num_rows * num_cols)
    for i in range(k):
        dist_sq = np.sum(np.square(grid - means[i].reshape(1, -1)), axis=1)
        tmp = np.exp(-dist_sq / variances[i])
        y = np.maximum(y, tmp) if algo == 'max' else y + tmp
    return grid, y


# --------------------------------------------------------------------------
# exact-GP marginal log likelihood and its gradient (models.py:137-159 loss)
# --------------------------------------------------------------------------
def mll_and_grad(hyp, x, y, var):
    """loss = -MLL/N as gpytorch's ExactMarginalLogLikelihood divides by N
    (models.py:148).  Returns (mll_per_point, dict of d mll_per_point / d log-params).
    RBF kernel only.  Parity unpinned (no reference fixture); used to check the
    HIP fit path by finite differences."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    N = len(y)
    y0 = y - y.mean()
    K = kernel_matrix(hyp, x)
    S = K + hyp.noise * np.eye(N) + (np.diag(var) if var is not None else 0.0)
    L = np.linalg.cholesky(S)
    from scipy.linalg import cho_solve
    alpha = cho_solve((L, True), y0)
    mll = -.5 * y0 @ alpha - np.sum(np.log(np.diag(L))) - .5 * N * np.log(2 * np.pi)
    Sinv = cho_solve((L, True), np.eye(N))
    W = np.outer(alpha, alpha) - Sinv          # dMLL/dtheta = 1/2 tr(W dS/dtheta)
    g = {}
    g['log_outputscale'] = .5 * np.sum(W * K) / N
    g['log_noise'] = .5 * np.trace(W) * hyp.noise / N
    inv_ls = np.exp(-hyp.log_lengthscale)
    gl = np.zeros(hyp.D)
    for d in range(hyp.D):
        diff = (x[:, d][:, None] - x[:, d][None, :]) * inv_ls[d]
        gl[d] = .5 * np.sum(W * K * diff * diff) / N
    g['log_lengthscale'] = gl
    return mll / N, g
