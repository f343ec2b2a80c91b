"""GP model wrapper with the reference's `GPR` surface (reference models.py:86-203), backed by
libalgp_hip.so instead of GPyTorch.

Same constructor, methods, properties and parameter names as the reference so that its callers
(agent.py:34-45, 84-90; utils.py:296-298; run.py:35-37) work unchanged:

    gp = GPR(latent=None, lr=.1, max_iterations=200, kernel_params={'type': 'rbf'})
    gp.fit(x, y, var); gp.cov_mat(x1, x2, white_noise_var, add_likelihood_var)
    dict(gp.model.named_parameters())['kernel_covar_module.log_outputscale']
    gp.model.state_dict() / load_state_dict()

What differs, deliberately:
  * arithmetic runs on the GPU in `dtype` (float64 by default; the reference is float32 through
    utils.py:19) -- pass dtype=np.float32 for the reference's precision;
  * only the identity latent function and the rbf / matern(nu=1.5) kernels exist (north_star scope;
    the reference also has linear / non_linear latents and a spectral mixture kernel,
    models.py:23-44, 221-223): anything else raises NotImplementedError like models.py:227, 248;
  * `fit` uses the analytic MLL gradient computed on the device with the reference's own optimiser
    objects (torch.optim.Adam + ReduceLROnPlateau(patience=50), models.py:121-124).
"""
import numpy as np
import torch
import torch.nn as nn

from . import _hip

_KERNELS = {None: _hip.KERNEL_RBF, 'rbf': _hip.KERNEL_RBF, 'matern': _hip.KERNEL_MATERN15}


class IdentityLatentFunction(nn.Module):
    """models.py:15-21"""

    def __init__(self):
        super(IdentityLatentFunction, self).__init__()
        self.embed_dim = None

    def forward(self, x):
        return x


class _BaseKernel(nn.Module):
    def __init__(self, ard_num_dims):
        super(_BaseKernel, self).__init__()
        self.log_lengthscale = nn.Parameter(torch.zeros(1, 1, ard_num_dims, dtype=torch.float64))


class _ScaleKernel(nn.Module):
    def __init__(self, ard_num_dims):
        super(_ScaleKernel, self).__init__()
        self.base_kernel = _BaseKernel(ard_num_dims)
        self.log_outputscale = nn.Parameter(torch.zeros(1, dtype=torch.float64))


class _GaussianLikelihood(nn.Module):
    def __init__(self):
        super(_GaussianLikelihood, self).__init__()
        self.log_noise = nn.Parameter(torch.zeros(1, dtype=torch.float64))


class ExactGPModel(nn.Module):
    """Parameter container with the reference's names (models.py:206-254): zero mean on
    mean-centred targets, ScaleKernel(RBF-ARD | Matern-1.5), per-point white noise, Gaussian
    likelihood.  It holds no arithmetic: the GPU library reads the parameters."""

    def __init__(self, train_x, train_y, likelihood, var=None, latent=None, kernel_params=None, latent_params=None):
        super(ExactGPModel, self).__init__()
        if latent is not None and latent != 'identity':
            raise NotImplementedError('latent function %r is outside the MI355X hot path (identity only)' % (latent,))
        self.latent_func = IdentityLatentFunction()
        kernel = kernel_params['type'] if kernel_params is not None else 'rbf'
        if kernel not in _KERNELS:
            raise NotImplementedError(kernel)
        self.kernel_type = _KERNELS[kernel]
        self.kernel_covar_module = _ScaleKernel(int(np.shape(train_x)[-1]))
        self.likelihood = likelihood


class GPR(object):
    def __init__(self, latent=None, lr=.01, max_iterations=200, kernel_params=None, latent_params=None,
                 learn_likelihood_noise=True, dtype=np.float64, device=0):
        self._train_x = None
        self._train_y = None
        self._train_y_mean = None
        self._train_var = None
        self.likelihood = None
        self.model = None
        self.optimizer = None
        self.lr_scheduler = None
        self.lr = lr
        self.latent = latent
        self.kernel_params = kernel_params
        self.latent_params = latent_params
        self.max_iter = max_iterations
        self.learn_likelihood_noise = learn_likelihood_noise
        self.dtype = np.dtype(dtype)
        self.device = device
        self._ctx = None
        self._hyp_key = None

    # ---- device context -------------------------------------------------------------------
    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _hip.Context(self.dtype, self.device)      # raises without a GPU: no fallback
        return self._ctx

    def hypers(self):
        """(log_lengthscale[D], log_outputscale, log_noise, kernel id) as plain floats."""
        m = self.model
        return (m.kernel_covar_module.base_kernel.log_lengthscale.detach().double().reshape(-1).numpy().copy(),
                float(m.kernel_covar_module.log_outputscale.item()), float(self.likelihood.log_noise.item()),
                m.kernel_type)

    def sync_hypers(self):
        """Push the current parameters to the device context if they changed."""
        ls, los, ln, kt = self.hypers()
        key = (tuple(ls), los, ln, kt)
        if key != self._hyp_key:
            self.ctx.set_hypers(ls, los, ln, kt)
            self._hyp_key = key
            return True
        return False

    # ---- models.py:103-115 ------------------------------------------------------------------
    @property
    def train_x(self):
        return self._train_x

    @property
    def train_y(self):
        return self._train_y

    @property
    def train_var(self):
        return self._train_var

    # ---- models.py:117-135 ------------------------------------------------------------------
    def reset(self, x, y, var):
        self.set_train_data(x, y, var)
        self.likelihood = _GaussianLikelihood()
        self.model = ExactGPModel(self._train_x, self._zero_mean_train_y, self.likelihood, self._train_var, self.latent,
                                  self.kernel_params, self.latent_params)
        params = [p for n, p in self.model.named_parameters()
                  if self.learn_likelihood_noise or n != 'likelihood.log_noise']
        self.optimizer = torch.optim.Adam([{'params': params}, ], lr=self.lr)
        self.lr_scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode='min', patience=50)
        self._hyp_key = None

    def set_train_data(self, x, y, var=None):
        x = np.asarray(x, dtype=np.float64)
        self._train_x = x.reshape(len(x), -1)
        self._train_y = np.asarray(y, dtype=np.float64).reshape(-1)
        self._train_y_mean = float(np.mean(self._train_y)) if len(self._train_y) else 0.0     # models.py:129
        self._zero_mean_train_y = self._train_y - self._train_y_mean                         # models.py:130
        if var is not None:
            self._train_var = np.asarray(var, dtype=np.float64).reshape(-1)

    # ---- models.py:137-159 ------------------------------------------------------------------
    def _load_train_on_device(self):
        c = self.ctx
        self.sync_hypers()
        c.set_pool(self._train_x)
        c.set_train(np.arange(len(self._train_x)), self._train_y, self._train_var)

    def neg_mll_and_grad(self):
        """loss = -MLL/N and its gradient w.r.t. the model parameters, from the device."""
        c = self.ctx
        if self.sync_hypers():
            pass
        mll, grad = c.fit_step()             # factorisation + MLL + gradient: one ABI call per Adam iteration
        n = max(1, len(self._train_y))
        return -mll / n, -grad / n

    def fit(self, x, y, var=None, disp=False):
        if var is None:
            var = np.full(len(y), 1e-5)                                           # models.py:138-139
        self.reset(x, y, var)
        self._load_train_on_device()
        named = dict(self.model.named_parameters())
        p_ls = named['kernel_covar_module.base_kernel.log_lengthscale']
        p_os = named['kernel_covar_module.log_outputscale']
        p_n = named['likelihood.log_noise']
        D = p_ls.numel()
        initial_ll = final_ll = None
        losses = []
        for i in range(self.max_iter):
            self.optimizer.zero_grad()
            loss, g = self.neg_mll_and_grad()
            p_ls.grad = torch.from_numpy(g[:D].copy()).reshape(p_ls.shape)
            p_os.grad = torch.tensor([g[D]], dtype=torch.float64)
            p_n.grad = torch.tensor([g[D + 1]], dtype=torch.float64)
            self.optimizer.step()
            self.lr_scheduler.step(loss)
            if disp:
                print(i, loss)
            if i == 0:
                initial_ll = -loss
            final_ll = -loss
            losses.append(loss)
        if self.max_iter > 0:
            print('Initial LogLikelihood {:.3f} Final LogLikelihood {:.3f}'.format(initial_ll, final_ll))
        self.sync_hypers()
        return losses

    # ---- models.py:161-181 ------------------------------------------------------------------
    def cov_mat(self, x1, x2=None, white_noise_var=None, add_likelihood_var=False):
        self.sync_hypers()
        x1 = np.asarray(x1, dtype=self.dtype)
        x1 = x1.reshape(len(x1), -1)
        if x2 is not None:
            x2 = np.asarray(x2, dtype=self.dtype)
            x2 = x2.reshape(len(x2), -1)
            if x2.shape == x1.shape and np.array_equal(x1, x2):                   # models.py:169 torch.equal
                x2 = None
        if x2 is None:
            return self.ctx.kernel_matrix(x1, None, white_noise_var, add_likelihood_var)
        cov = self.ctx.kernel_matrix(x1, x2)
        if white_noise_var is not None:                                          # models.py:175-176 (square only)
            cov += np.diag(np.asarray(white_noise_var, dtype=self.dtype))
        if add_likelihood_var:
            cov += self.dtype.type(np.exp(self.likelihood.log_noise.item())) * np.eye(len(cov), dtype=self.dtype)
        return cov

    # ---- models.py:183-197 ------------------------------------------------------------------
    def predict(self, x, return_cov=False, return_std=False):
        """Posterior of the noisy observation y* = f* + eps (the likelihood is applied, as
        models.py:191 does), conditioned on the stored training data."""
        c = self.ctx
        self.sync_hypers()
        x = np.asarray(x, dtype=np.float64)
        x = x.reshape(len(x), -1)
        N, M = len(self._train_x), len(x)
        c.set_pool(np.vstack([self._train_x, x]))
        c.set_train(np.arange(N), self._train_y, self._train_var)
        noise = float(np.exp(self.likelihood.log_noise.item()))
        if not (return_std or return_cov):
            c.factorize()
            return c.posterior_mean(np.arange(N, N + M))
        c.set_candidates(np.arange(N, N + M), prior_includes_noise=False)
        c.fit_and_solve()                                                        # factorisation + V^T in one launch where it fits
        mu, var = c.posterior()
        if return_std:                                                           # models.py:193 returns the variance
            return mu, var + self.dtype.type(noise)
        cov, _ = c.posterior_cov()
        return mu, cov + self.dtype.type(noise) * np.eye(M, dtype=self.dtype)

    def get_embeddings(self, x):
        return np.asarray(x)                                                     # identity latent (models.py:199-203)
