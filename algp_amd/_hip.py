"""ctypes binding of libalgp_hip.so (include/algp_hip.h).

This is the only place the package touches native code.  There is no CPU
fallback: if the library is missing, or no MI355X is visible, the calls raise.
"""
import ctypes as C
import os

import numpy as np

# The library overlaps independent row chunks of the candidate solve on several HIP streams; ROCm maps all
# streams of a process onto GPU_MAX_HW_QUEUES (default 4) hardware queues, and streams that share a queue
# serialise.  With RCCL (or torch) streams in the same process 4 is not enough (measured: 161 -> 183 ms).
# Only effective if set before the HIP runtime initialises, hence at import time.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ALGP_LIB') or os.path.join(_HERE, 'lib', 'libalgp_hip.so')   # $ALGP_LIB: another build, for A/B timing

F32, F64 = 0, 1
KERNEL_RBF, KERNEL_MATERN15 = 0, 1
CRIT_ENTROPY, CRIT_MUTUAL_INFORMATION = 0, 1
OK, ERR_BAD_ARG, ERR_HIP, ERR_NOT_PD, ERR_OOM, ERR_STATE, ERR_NO_DEVICE = range(7)
PROF = dict(kmat=0, gemm_chol=1, gemm_trsm=2, potrf_diag=3, trsv=4, rows=5, score=6, gemm_other=7, cholesky=8, trsm=9,
            gemm_chol_update=10, chol_dag=11, dag_panel=12, tail_cols=13)

_c_ctx = C.c_void_p
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)   # algp_allgather_fn
_i64p = C.POINTER(C.c_int64)
_dblp = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/algp_hip.h declares
SIGNATURES = {
    'algp_version': (C.c_int, []),
    'algp_device_count': (C.c_int, []),
    'algp_create': (C.c_int, [C.c_int, C.c_int, C.POINTER(_c_ctx)]),
    'algp_destroy': (None, [_c_ctx]),
    'algp_last_error': (C.c_char_p, [_c_ctx]),
    'algp_last_pivot': (C.c_int64, [_c_ctx]),
    'algp_last_jitter': (C.c_double, [_c_ctx]),
    'algp_dtype': (C.c_int, [_c_ctx]),
    'algp_set_hypers': (C.c_int, [_c_ctx, C.c_int, C.c_int, _dblp, C.c_double, C.c_double]),
    'algp_kernel_matrix': (C.c_int, [_c_ctx, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int,
                                     C.c_void_p]),
    'algp_set_pool': (C.c_int, [_c_ctx, C.c_void_p, C.c_int64]),
    'algp_set_pool_cov': (C.c_int, [_c_ctx, C.c_void_p, C.c_int64]),
    'algp_set_train': (C.c_int, [_c_ctx, _i64p, C.c_int64, C.c_void_p, C.c_void_p]),
    'algp_factorize': (C.c_int, [_c_ctx]),
    'algp_factorize_update': (C.c_int, [_c_ctx, _i64p]),
    'algp_set_constant_mean': (C.c_int, [_c_ctx, C.c_int, C.c_double]),
    'algp_factorize_from': (C.c_int, [_c_ctx, _c_ctx, _i64p]),
    'algp_fit_and_solve': (C.c_int, [_c_ctx]),
    'algp_get_logdet': (C.c_int, [_c_ctx, _dblp]),
    'algp_get_entropy': (C.c_int, [_c_ctx, _dblp]),
    'algp_get_alpha': (C.c_int, [_c_ctx, C.c_void_p]),
    'algp_get_factor': (C.c_int, [_c_ctx, C.c_void_p]),
    'algp_get_mll': (C.c_int, [_c_ctx, _dblp]),
    'algp_get_mll_grad': (C.c_int, [_c_ctx, _dblp]),
    'algp_fit_step': (C.c_int, [_c_ctx, _dblp, _dblp]),
    'algp_set_candidates': (C.c_int, [_c_ctx, _i64p, C.c_int64, C.c_int, C.c_void_p]),
    'algp_solve_candidates': (C.c_int, [_c_ctx]),
    'algp_solve_candidates_update': (C.c_int, [_c_ctx, C.c_void_p, _i64p]),
    'algp_set_candidate_alive': (C.c_int, [_c_ctx, C.c_void_p]),
    'algp_get_posterior': (C.c_int, [_c_ctx, C.c_void_p, C.c_void_p]),
    'algp_get_posterior_cov': (C.c_int, [_c_ctx, C.c_void_p, _dblp]),
    'algp_posterior_mean': (C.c_int, [_c_ctx, _i64p, C.c_int64, C.c_void_p]),
    'algp_scores': (C.c_int, [_c_ctx, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int]),
    'algp_argmax': (C.c_int, [_c_ctx, _i64p, _i64p, _dblp]),
    'algp_best_candidate': (C.c_int, [_c_ctx, C.c_int, C.c_double, C.c_double, _i64p, _i64p, _dblp]),
    'algp_commit_pick': (C.c_int, [_c_ctx, C.c_int64, C.c_double, C.c_double]),
    'algp_greedy': (C.c_int, [_c_ctx, C.c_int, C.c_double, C.c_double, C.c_int, _i64p, _i64p, _dblp]),
    'algp_entropy_from_cov': (C.c_int, [_c_ctx, C.c_void_p, C.c_int64, _dblp]),
    'algp_set_entropy': (C.c_int, [_c_ctx, _i64p, C.c_int64, C.c_void_p, _dblp]),
    'algp_set_inverse_diag': (C.c_int, [_c_ctx, _i64p, C.c_int64, C.c_void_p, C.c_void_p, _dblp]),
    'algp_cholesky': (C.c_int, [_c_ctx, C.c_void_p, C.c_int64, C.c_void_p, _dblp]),
    'algp_gemm_nt': (C.c_int, [_c_ctx, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_void_p, C.c_void_p,
                               C.c_double, C.c_void_p, C.c_void_p]),
    'algp_trsm_right_lt': (C.c_int, [_c_ctx, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    'algp_selftest_mfma': (C.c_int, [_c_ctx, C.POINTER(C.c_int)]),
    'algp_bench_gemm': (C.c_int, [_c_ctx, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, _dblp]),
    'algp_sync': (C.c_int, [_c_ctx]),
    'algp_device_bytes': (C.c_int64, [_c_ctx]),
    'algp_prof_enable': (C.c_int, [_c_ctx, C.c_int]),
    'algp_prof_reset': (C.c_int, [_c_ctx]),
    'algp_prof_get': (C.c_int, [_c_ctx, C.c_int, _dblp, _dblp, _dblp, _i64p]),
    'algp_cholesky_task_stats': (C.c_int, [_c_ctx, _dblp]),
    'algp_score_paths': (C.c_int, [_c_ctx, _i64p, C.c_int, C.c_int, C.c_double, _dblp]),
    'algp_comm_unique_id': (C.c_int, [C.c_void_p]),
    'algp_comm_init': (C.c_int, [_c_ctx, C.c_int, C.c_int, C.c_void_p]),
    'algp_comm_init_host': (C.c_int, [_c_ctx, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'algp_comm_destroy': (C.c_int, [_c_ctx]),
    'algp_comm_set_owners': (C.c_int, [_c_ctx, C.POINTER(C.c_int32), C.c_int64]),
    'algp_debug_first_max': (C.c_int, [_c_ctx, _dblp, C.c_int, _dblp]),
    'algp_debug_fail_next_pick': (C.c_int, [_c_ctx, C.c_int]),
    'algp_debug_set_trsm_chunks': (C.c_int, [_c_ctx, C.c_int]),
    'algp_debug_dag_stall': (C.c_int, [_c_ctx, C.c_int]),
    'algp_debug_trsv_stall': (C.c_int, [_c_ctx, C.c_int]),
    'algp_debug_fail_at': (C.c_int, [_c_ctx, C.c_int, C.c_int]),
    'algp_debug_get_pick': (C.c_int, [_c_ctx, C.c_int, C.c_void_p, C.c_int64, _i64p, _dblp]),
    'algp_debug_counter': (C.c_int64, [_c_ctx, C.c_int]),
    'algp_debug_get_factor_rows': (C.c_int, [_c_ctx, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    'algp_greedy_sharded': (C.c_int, [_c_ctx, C.c_int, C.c_double, C.c_double, C.c_int, _i64p, _dblp]),
}

_lib = None


class AlgpError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, 'algp_hip error %d: %s' % (code, msg))
        self.code = code


def load():
    """dlopen the library and bind every declared symbol.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          '(make -C algp_amd/csrc); there is no CPU fallback' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i64(a):
    return None if a is None else a.ctypes.data_as(_i64p)


class Context(object):
    """One GPU context (algp_ctx).  dtype: np.float32 or np.float64."""

    def __init__(self, dtype=np.float64, device=0):
        self.lib = load()
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError('dtype must be float32 or float64')
        h = _c_ctx()
        rc = self.lib.algp_create(int(device), F64 if self.dtype == np.float64 else F32, C.byref(h))
        if rc == ERR_NO_DEVICE:
            raise AlgpError(rc, 'no HIP device visible: algp_amd needs an MI355X (gfx950); there is no CPU fallback')
        if rc != OK:
            raise AlgpError(rc, 'algp_create failed')
        self.h = h
        self.device = int(device)
        self.pool_generation = 0            # counts pool loads: owners of derived state compare it (Agent._load_pool)
        self._pool_owner = None

    def close(self):
        if getattr(self, 'h', None):
            self.lib.algp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers --------------------------------------------------------
    def _check(self, rc):
        if rc == OK:
            return
        msg = self.lib.algp_last_error(self.h).decode('utf-8', 'replace')
        if rc == ERR_NOT_PD:
            err = np.linalg.LinAlgError(msg)          # reference: LinAlgError from inv (utils.py:300)
            err.pivot = int(self.lib.algp_last_pivot(self.h))
            raise err
        if rc in (ERR_BAD_ARG, ERR_STATE):
            raise ValueError(msg)
        if rc == ERR_OOM:
            raise MemoryError(msg)
        raise AlgpError(rc, msg)

    def _arr(self, a, shape=None):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if shape is not None and a.shape != tuple(shape):
            raise ValueError('expected shape %s, got %s' % (tuple(shape), a.shape))
        return a

    @staticmethod
    def _idx(a):
        return np.ascontiguousarray(a, dtype=np.int64).reshape(-1)

    # -- hypers / kernel matrix -------------------------------------------
    def set_hypers(self, log_lengthscale, log_outputscale, log_noise, kernel=KERNEL_RBF):
        ls = np.ascontiguousarray(np.atleast_1d(log_lengthscale), dtype=np.float64)
        self.D = len(ls)
        self._check(self.lib.algp_set_hypers(self.h, int(kernel), len(ls), ls.ctypes.data_as(_dblp),
                                             float(log_outputscale), float(log_noise)))

    def _check_width(self, x, what):
        """The C side reads n * D values: a wrong-width array would be over-read or silently reinterpreted."""
        D = getattr(self, 'D', None)
        if D is None:
            raise ValueError('%s: call set_hypers first (it fixes the input dimension)' % what)
        if x.ndim != 2 or x.shape[1] != D:
            raise ValueError('%s: expected coordinates of shape (n, %d), got %s' % (what, D, tuple(x.shape)))

    def kernel_matrix(self, x1, x2=None, diag_add=None, add_likelihood_var=False):
        x1 = self._arr(x1)
        x1 = x1.reshape(len(x1), -1)
        self._check_width(x1, 'kernel_matrix x1')
        n1 = x1.shape[0]
        if x2 is None:
            n2, x2p = n1, None
        else:
            x2 = self._arr(x2)
            x2 = x2.reshape(len(x2), -1)
            self._check_width(x2, 'kernel_matrix x2')
            n2, x2p = x2.shape[0], _ptr(x2)
        d = None if diag_add is None else self._arr(diag_add, (n1,))
        out = np.empty((n1, n2), dtype=self.dtype)
        self._check(self.lib.algp_kernel_matrix(self.h, _ptr(x1), n1, x2p, n2, _ptr(d), int(bool(add_likelihood_var)),
                                                _ptr(out)))
        return out

    # -- pool / train / factor -----------------------------------------------
    def set_pool(self, x):
        x = self._arr(x)
        x = x.reshape(len(x), -1)
        self._check_width(x, 'set_pool')
        self.n_pool = x.shape[0]
        self.pool_generation += 1
        self._pool_owner = None
        self._check(self.lib.algp_set_pool(self.h, _ptr(x), x.shape[0]))

    def set_pool_cov(self, cov):
        cov = self._arr(cov)
        if cov.ndim != 2 or cov.shape[0] != cov.shape[1]:
            raise ValueError('cov must be square')
        self.n_pool = cov.shape[0]
        self.pool_generation += 1
        self._pool_owner = None
        self._check(self.lib.algp_set_pool_cov(self.h, _ptr(cov), cov.shape[0]))

    def set_train(self, idx, y, var=None):
        idx = self._idx(idx)
        y = self._arr(y, (len(idx),))
        v = None if var is None else self._arr(var, (len(idx),))
        self.N = len(idx)
        self._check(self.lib.algp_set_train(self.h, _i64(idx), len(idx), _ptr(y), _ptr(v)))

    def factorize(self, incremental=False):
        """incremental=True reuses the resident factor's unchanged leading rows (returns how many)."""
        if not incremental:
            self._check(self.lib.algp_factorize(self.h))
            return 0
        kept = C.c_int64()
        self._check(self.lib.algp_factorize_update(self.h, C.byref(kept)))
        return kept.value

    def set_constant_mean(self, value):
        """Use `value` as the GP's constant mean in the following set_train calls (None: back to the mean of the
        train targets, models.py:129)."""
        self._check(self.lib.algp_set_constant_mean(self.h, int(value is not None), 0.0 if value is None else float(value)))

    def factorize_from(self, other):
        """Adopt the factor `other` holds for the same train set (same device, dtype, hyper-parameters); returns
        how many leading rows of this context's own factor were kept.  Raises if `other`'s factor does not match."""
        kept = C.c_int64()
        self._check(self.lib.algp_factorize_from(self.h, other.h, C.byref(kept)))
        return kept.value

    def fit_and_solve(self):
        """factorize() + solve_candidates() back to back in one ABI call."""
        self._check(self.lib.algp_fit_and_solve(self.h))

    def _get_double(self, fn):
        v = C.c_double()
        self._check(fn(self.h, C.byref(v)))
        return v.value

    def logdet(self):
        return self._get_double(self.lib.algp_get_logdet)

    def entropy(self):
        return self._get_double(self.lib.algp_get_entropy)

    def mll(self):
        return self._get_double(self.lib.algp_get_mll)

    def mll_grad(self):
        """d MLL / d (log_lengthscale[D], log_outputscale, log_noise); not divided by N."""
        g = np.empty(self.D + 2, dtype=np.float64)
        self._check(self.lib.algp_get_mll_grad(self.h, g.ctypes.data_as(_dblp)))
        return g

    def fit_step(self):
        """(mll, grad): factorisation, marginal log likelihood and its gradient for the current hyper-parameters in one
        call -- the device work of one iteration of GPR.fit (algp_fit_step)."""
        g = np.empty(self.D + 2, dtype=np.float64)
        v = C.c_double()
        self._check(self.lib.algp_fit_step(self.h, C.byref(v), g.ctypes.data_as(_dblp)))
        return v.value, g

    def alpha(self):
        out = np.empty(self.N, dtype=self.dtype)
        self._check(self.lib.algp_get_alpha(self.h, _ptr(out)))
        return out

    def factor(self):
        out = np.empty((self.N, self.N), dtype=self.dtype)
        self._check(self.lib.algp_get_factor(self.h, _ptr(out)))
        return out

    # -- candidates ----------------------------------------------------------
    def set_candidates(self, idx, prior_includes_noise=True, extra_var=None):
        idx = self._idx(idx)
        e = None if extra_var is None else self._arr(extra_var, (len(idx),))
        self.M = len(idx)
        self._check(self.lib.algp_set_candidates(self.h, _i64(idx), len(idx), int(bool(prior_includes_noise)), _ptr(e)))

    def solve_candidates(self, incremental=False, alive=None):
        """incremental=True keeps the V^T columns that are still valid (returns how many columns)."""
        a = None if alive is None else np.ascontiguousarray(alive, dtype=np.uint8)
        if a is not None and a.shape != (self.M,):
            raise ValueError('alive must have one entry per candidate')
        if not incremental:
            self._check(self.lib.algp_solve_candidates(self.h))
            if a is not None:
                self._check(self.lib.algp_set_candidate_alive(self.h, _ptr(a)))
            return 0
        kept = C.c_int64()
        self._check(self.lib.algp_solve_candidates_update(self.h, _ptr(a), C.byref(kept)))
        return kept.value

    def posterior(self, want_var=True):
        mu = np.empty(self.M, dtype=self.dtype)
        var = np.empty(self.M, dtype=self.dtype) if want_var else None
        self._check(self.lib.algp_get_posterior(self.h, _ptr(mu), _ptr(var)))
        return (mu, var) if want_var else mu

    def posterior_cov(self, want_cov=True, want_mi=False):
        cov = np.empty((self.M, self.M), dtype=self.dtype) if want_cov else None
        mi = C.c_double()
        self._check(self.lib.algp_get_posterior_cov(self.h, _ptr(cov), C.byref(mi) if want_mi else None))
        return cov, (mi.value if want_mi else None)

    def last_jitter(self):
        """Diagonal jitter the last posterior_cov(want_mi=True) needed (0.0: none); see algp_last_jitter."""
        return float(self.lib.algp_last_jitter(self.h))

    def posterior_mean(self, idx):
        idx = self._idx(idx)
        mu = np.empty(len(idx), dtype=self.dtype)
        self._check(self.lib.algp_posterior_mean(self.h, _i64(idx), len(idx), _ptr(mu)))
        return mu

    # -- greedy ----------------------------------------------------------------
    def scores(self, criterion, static_std, mobile_std, out_device_ptr=None):
        if out_device_ptr is not None:
            self._check(self.lib.algp_scores(self.h, int(criterion), float(static_std), float(mobile_std),
                                             C.c_void_p(int(out_device_ptr)), 1))
            return None
        out = np.empty(self.M, dtype=np.float64)
        self._check(self.lib.algp_scores(self.h, int(criterion), float(static_std), float(mobile_std), _ptr(out), 0))
        return out

    def argmax(self):
        pos, pool, val = C.c_int64(), C.c_int64(), C.c_double()
        self._check(self.lib.algp_argmax(self.h, C.byref(pos), C.byref(pool), C.byref(val)))
        return pos.value, pool.value, val.value

    def best_candidate(self, criterion, static_std, mobile_std):
        """(local position, pool index, utility) of the first maximum; rows are brought up to date with the
        committed picks only as far as needed (algp_best_candidate)."""
        pos, pool, val = C.c_int64(), C.c_int64(), C.c_double()
        self._check(self.lib.algp_best_candidate(self.h, int(criterion), float(static_std), float(mobile_std),
                                                 C.byref(pos), C.byref(pool), C.byref(val)))
        return pos.value, pool.value, val.value

    def commit_pick(self, pool_idx, static_std, mobile_std):
        self._check(self.lib.algp_commit_pick(self.h, int(pool_idx), float(static_std), float(mobile_std)))

    def greedy(self, criterion, static_std, mobile_std, k, forced_picks=None, want_utilities=False):
        picks = np.empty(k, dtype=np.int64)
        ut = np.empty((k, self.M), dtype=np.float64) if want_utilities else None
        f = None if forced_picks is None else self._idx(forced_picks)
        if f is not None and len(f) != k:
            raise ValueError('forced_picks must have k entries')
        self._check(self.lib.algp_greedy(self.h, int(criterion), float(static_std), float(mobile_std), int(k), _i64(f),
                                         _i64(picks), None if ut is None else ut.ctypes.data_as(_dblp)))
        return (picks, ut) if want_utilities else picks

    # -- entropies ---------------------------------------------------------------
    def score_paths(self, sites, mobile_std):
        """dH[p] = H(A u path_p) - H(A) for every row of `sites` (npaths x maxlen pool indices, -1 padded), all paths
        in one launch from the resident candidate solve (see algp_score_paths)."""
        sites = np.ascontiguousarray(sites, dtype=np.int64)
        if sites.ndim != 2:
            raise ValueError('sites must be (npaths, maxlen)')
        out = np.empty(sites.shape[0], dtype=np.float64)
        if sites.shape[0]:
            self._check(self.lib.algp_score_paths(self.h, _i64(sites.ravel()), sites.shape[0], sites.shape[1], float(mobile_std),
                                                  out.ctypes.data_as(_dblp)))
        return out

    # -- multi-GPU: the collective behind the ABI (RCCL) ---------------------------------------
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes identifying a new communicator: one rank creates them, every rank passes them to comm_init."""
        buf = C.create_string_buffer(128)
        rc = load().algp_comm_unique_id(buf)
        if rc != OK:
            raise AlgpError(rc, 'algp_comm_unique_id failed (is librccl.so loadable?)')
        return buf.raw

    def comm_init(self, nranks, rank, unique_id):
        if len(unique_id) != 128:
            raise ValueError('unique_id must be the 128 bytes of comm_unique_id()')
        self._check(self.lib.algp_comm_init(self.h, int(nranks), int(rank), C.create_string_buffer(unique_id, 128)))

    def comm_init_host(self, nranks, rank, all_gather, raw=False):
        """The sharded greedy loop over a transport the caller owns: all_gather(send: bytes) -> bytes must return the
        concatenation, in rank order, of what every rank passed (algp_comm_init_host).  raw=True: the callable gets
        (send_address, recv_address, bytes_per_rank) -- the library's pinned staging itself, no copies through Python
        objects -- and returns 0 for success."""
        def tramp(_user, send, recv, nbytes):
            try:
                if raw:
                    return int(all_gather(int(send or 0), int(recv or 0), int(nbytes)))
                out = all_gather(C.string_at(send, nbytes))
                if len(out) != nbytes * int(nranks):
                    return 2
                C.memmove(recv, out, len(out))
                return 0
            except Exception:                                     # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._gather_cb = ALLGATHER_FN(tramp)                     # keep the trampoline alive as long as the ctx uses it
        self._check(self.lib.algp_comm_init_host(self.h, int(nranks), int(rank), C.cast(self._gather_cb, C.c_void_p), None))

    def comm_destroy(self):
        self._check(self.lib.algp_comm_destroy(self.h))
        self._gather_cb = None
        self._shard_link = None

    def comm_set_owners(self, owner):
        """owner[q] = rank that holds pool site q as a candidate (-1: nobody), the same array on every rank; None clears
        it.  With a map attached factorize(incremental=True) is a collective: the rows of L of the new train sites travel
        from their owners' V^T in one all-gather instead of being solved on every rank (algp_comm_set_owners)."""
        if owner is None:
            self._check(self.lib.algp_comm_set_owners(self.h, None, 0))
            return
        o = np.ascontiguousarray(owner, dtype=np.int32).reshape(-1)
        self._check(self.lib.algp_comm_set_owners(self.h, o.ctypes.data_as(C.POINTER(C.c_int32)), len(o)))

    def counter(self, which):
        """algp_debug_counter: 0 stream synchronisations so far | 1 rows of L the last factor update placed without a
        triangular solve | 2 how many of them arrived from other ranks | 3 row exchanges so far | 4 agreed fall-backs."""
        return int(self.lib.algp_debug_counter(self.h, int(which)))

    def debug_first_max(self, triples):
        """first_max_kernel on a fabricated (nranks, 3) buffer -> (utility, pool index, owner, status, failing rank)."""
        t = np.ascontiguousarray(triples, dtype=np.float64).reshape(-1, 3)
        out = np.empty(5, dtype=np.float64)
        self._check(self.lib.algp_debug_first_max(self.h, t.ctypes.data_as(_dblp), len(t), out.ctypes.data_as(_dblp)))
        return out

    def debug_fail_next_pick(self, code):
        self._check(self.lib.algp_debug_fail_next_pick(self.h, int(code)))

    def debug_dag_stall(self, ticket):
        """The next one-launch factorisation loses the publish of the task with this ticket (spin limit 0.2 s)."""
        self._check(self.lib.algp_debug_dag_stall(self.h, int(ticket)))

    def debug_fail_at(self, where, code):
        """Inject `code` into this rank's next greedy pick: where = 0 resolving its best, 1 committing the winner (after the
        exchange), 2 packing its contribution, 3 the agreement word of its next sharded factor update (algp_debug_fail_at)."""
        self._check(self.lib.algp_debug_fail_at(self.h, int(where), int(code)))

    def debug_trsv_stall(self, block):
        """The next one-launch forward / backward substitution loses the flag of this 128-block (spin limit 0.2 s)."""
        self._check(self.lib.algp_debug_trsv_stall(self.h, int(block)))

    def debug_factor_rows(self, row0, nrows, ncols):
        """Rows [row0, row0 + nrows) x columns [0, ncols) of the resident factor (algp_debug_get_factor_rows)."""
        out = np.empty((int(nrows), int(ncols)), dtype=self.dtype)
        self._check(self.lib.algp_debug_get_factor_rows(self.h, int(row0), int(nrows), int(ncols), _ptr(out)))
        return out

    def debug_get_pick(self, q):
        """(row, d): pick q since the last solve as every rank committed it -- the winner's row of V^T (ncols values) and
        its statistic; what the winner's owner contributes to the pick's all-gather (algp_debug_get_pick)."""
        n, d = C.c_int64(), C.c_double()
        self._check(self.lib.algp_debug_get_pick(self.h, int(q), None, 0, C.byref(n), C.byref(d)))
        row = np.empty(n.value, dtype=self.dtype)
        self._check(self.lib.algp_debug_get_pick(self.h, int(q), _ptr(row), len(row), C.byref(n), C.byref(d)))
        return row, d.value

    def set_trsm_chunks(self, chunks):
        """Row-chunk streams of the candidate solve (1..4; 0 = default).  Same results for every setting."""
        self._check(self.lib.algp_debug_set_trsm_chunks(self.h, int(chunks)))

    def sync_count(self):
        """Stream synchronisations the library has issued for this context so far."""
        return int(self.lib.algp_debug_counter(self.h, 0))

    def greedy_sharded(self, criterion, static_std, mobile_std, k, want_utilities=False):
        """k picks over the candidate shards of all ranks (one RCCL all-gather per pick inside the library)."""
        picks = np.empty(int(k), dtype=np.int64)
        ut = np.empty(int(k), dtype=np.float64) if want_utilities else None
        self._check(self.lib.algp_greedy_sharded(self.h, int(criterion), float(static_std), float(mobile_std), int(k),
                                                 _i64(picks), None if ut is None else ut.ctypes.data_as(_dblp)))
        return (picks, ut) if want_utilities else picks

    def entropy_from_cov(self, cov):
        cov = self._arr(cov)
        k = cov.shape[0]
        v = C.c_double()
        self._check(self.lib.algp_entropy_from_cov(self.h, _ptr(cov), k, C.byref(v)))
        return v.value

    def set_entropy(self, idx, var=None):
        idx = self._idx(idx)
        vv = None if var is None else self._arr(var, (len(idx),))
        v = C.c_double()
        self._check(self.lib.algp_set_entropy(self.h, _i64(idx), len(idx), _ptr(vv), C.byref(v)))
        return v.value

    def set_inverse_diag(self, idx, var=None):
        idx = self._idx(idx)
        vv = None if var is None else self._arr(var, (len(idx),))
        out = np.empty(len(idx), dtype=self.dtype)
        v = C.c_double()
        self._check(self.lib.algp_set_inverse_diag(self.h, _i64(idx), len(idx), _ptr(vv), _ptr(out), C.byref(v)))
        return out, v.value

    # -- dense building blocks ---------------------------------------------------
    def cholesky(self, A):
        A = self._arr(A)
        n = A.shape[0]
        L = np.empty((n, n), dtype=self.dtype)
        ld = C.c_double()
        self._check(self.lib.algp_cholesky(self.h, _ptr(A), n, _ptr(L), C.byref(ld)))
        return L, ld.value

    def gemm_nt(self, A, B, alpha=1.0, beta=0.0, Cm=None):
        A, B = self._arr(A), self._arr(B)
        m, k = A.shape
        n = B.shape[0]
        if B.shape[1] != k:
            raise ValueError('inner dimensions differ')
        Cm = None if Cm is None else self._arr(Cm, (m, n))
        D = np.empty((m, n), dtype=self.dtype)
        self._check(self.lib.algp_gemm_nt(self.h, m, n, k, float(alpha), _ptr(A), _ptr(B), float(beta), _ptr(Cm), _ptr(D)))
        return D

    def trsm_right_lt(self, L, B):
        L, B = self._arr(L), self._arr(B)
        n = L.shape[0]
        m = B.shape[0]
        X = np.empty((m, n), dtype=self.dtype)
        self._check(self.lib.algp_trsm_right_lt(self.h, _ptr(L), n, _ptr(B), m, _ptr(X)))
        return X

    def selftest_mfma(self):
        v = C.c_int()
        self._check(self.lib.algp_selftest_mfma(self.h, C.byref(v)))
        return v.value

    def bench_gemm(self, m, n, k, lower_only=False, beta_one=True, reps=5):
        ms = C.c_double()
        self._check(self.lib.algp_bench_gemm(self.h, m, n, k, int(bool(lower_only)), int(bool(beta_one)),
                                             int(reps), C.byref(ms)))
        return ms.value

    # -- misc ----------------------------------------------------------------------
    def sync(self):
        self._check(self.lib.algp_sync(self.h))

    def device_bytes(self):
        return int(self.lib.algp_device_bytes(self.h))

    def prof_enable(self, on=True):
        self._check(self.lib.algp_prof_enable(self.h, int(bool(on))))

    def prof_reset(self):
        self._check(self.lib.algp_prof_reset(self.h))

    def cholesky_task_stats(self):
        """In-kernel accounting of the one-launch Cholesky since prof_reset (while profiling is enabled):
        dict(update_us, update_steps, trsm_us, trsm_count), see include/algp_hip.h."""
        out = (C.c_double * 4)()
        self._check(self.lib.algp_cholesky_task_stats(self.h, out))
        return dict(update_us=out[0], update_steps=out[1], trsm_us=out[2], trsm_count=out[3])

    def prof_get(self, klass):
        ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        k = PROF[klass] if isinstance(klass, str) else int(klass)
        self._check(self.lib.algp_prof_get(self.h, k, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)))
        return dict(ms=ms.value, flops=fl.value, bytes=by.value, launches=n.value)


def device_count():
    return int(load().algp_device_count())
