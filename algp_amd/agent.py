"""Active-learning agent with the reference's `Agent` surface (reference agent.py:12-518); the
GP arithmetic of every planning step runs in libalgp_hip.so.

Hot-path methods (device):
  _post_update  agent.py:89-90     pool coordinates go to the GPU; `cov_matrix` is built lazily
  greedy        agent.py:295-356   one factorisation + one blocked solve + k rank-1 row appends
                                    instead of k*M fresh slogdets
  best_path     agent.py:358-403   one set entropy per path
  predict       agent.py:289-293   -> utils.predictive_distribution
  update_model  agent.py:84-87     -> GPR.fit (analytic MLL gradient on the device)
Host bookkeeping (sample logs, sensor fusion, the IPP loop) follows the reference's behaviour and
talks to the environment only through the reference's FieldEnv interface (env.X, env.test_X,
env.num_samples, env.collect_samples, env.get_all_paths, ...), so a reference-style environment
object drops in.  Path planning itself (env.py / map.py) is outside this package.
"""
import time
from copy import deepcopy

import numpy as np
import torch

from . import _hip
from .models import GPR
from .utils import CONST, compute_mae, find_equi_sample_path, find_shortest_path, predictive_distribution

_CRIT = {'entropy': _hip.CRIT_ENTROPY, 'mutual_information': _hip.CRIT_MUTUAL_INFORMATION}


def get_heading(prev, cur):
    """Axis-aligned unit heading of the move prev -> cur; None if they coincide; a move with a row
    component reports the row direction (reference graph_utils.py:15-25)."""
    dr, dc = int(cur[0] - prev[0]), int(cur[1] - prev[1])
    if dr == 0 and dc == 0:
        return None
    return (0, 1 if dc > 0 else -1) if dr == 0 else (1 if dr > 0 else -1, 0)


class Agent(object):
    def __init__(self, env, args, parent_agent=None, learn_likelihood_noise=True, mobile_std=None, static_std=None,
                 comm=None):
        """comm (not in the reference): a `sharded.ShardLink` -- this process is one rank of a job that shards the
        candidates of `greedy` over several GPUs; every rank constructs the same agent on the same environment and
        calls the same methods (the library's collectives sit inside greedy and the factor updates)."""
        self.env = env
        self._comm = comm if (comm is not None and comm.world_size > 1) else None
        self.learn_likelihood_noise = learn_likelihood_noise
        self._init_model(args)
        self.static_std = args.static_std if static_std is None else static_std
        self.mobile_std = 10 * self.static_std if mobile_std is None else mobile_std      # agent.py:20
        self.num_samples_per_batch = args.num_samples_per_batch
        self.update_every = args.update_every
        self.criterion = getattr(args, 'criterion_default', 'entropy')
        self._cov_matrix = None
        self._cov_matrix_user = False
        self._pool_key = None
        # f1: keep the factor across planning steps.  The train set is handed to the
        # device in INSERTION order (a site is appended the first time it is sampled), the candidate
        # list is the whole pool with an alive mask, and predict() has its own context over a fixed
        # [env.X; env.test_X] pool -- so each step re-factorises / re-solves only what changed.
        self.incremental = getattr(args, 'incremental', True)
        self._order = []
        self._rows = []                              # (site, 's' | 'm') in order of the first reading of that kind
        self._pctx = None
        self._pctx_key = None
        self.reset()
        if parent_agent is None:
            self._pre_train(num_samples=int(args.fraction_pretrain * self.env.num_samples))
        else:
            self.load_model(parent_agent)
            self.static_data = deepcopy(parent_agent.static_data)
            self.mobile_data = deepcopy(parent_agent.mobile_data)
            self.collected = deepcopy(parent_agent.collected)

    @property
    def comm(self):
        """The job's ShardLink, or None for a one-GPU agent (also for objects built without __init__, as some tests do)."""
        return getattr(self, '_comm', None)

    # ---- model plumbing (agent.py:34-45) --------------------------------------------------------
    def _init_model(self, args):
        self.gp = GPR(latent=args.latent, lr=args.lr, max_iterations=args.max_iterations,
                      kernel_params={'type': args.kernel}, learn_likelihood_noise=self.learn_likelihood_noise,
                      dtype=getattr(args, 'dtype', np.float64), device=getattr(args, 'device', 0))

    def load_model(self, parent_agent):
        self.gp.reset(parent_agent.gp.train_x, parent_agent.gp.train_y, parent_agent.gp.train_var)
        self.gp.model.load_state_dict(parent_agent.gp.model.state_dict())

    def save_model(self, filename):
        torch.save({'state_dict': self.gp.model.state_dict()}, filename)

    # ---- sample bookkeeping (agent.py:47-82) -----------------------------------------------------
    def reset(self):
        self.pose = (0, 0)
        self.heading = (1, 0)
        self.path = np.copy(self.pose).reshape(-1, 2)
        self.collected = {'ind': [], 'std': [], 'y': []}
        self.static_locations = np.empty((0, 2))
        self.static_data = [[] for _ in range(self.env.num_samples)]
        self.mobile_data = [[] for _ in range(self.env.num_samples)]
        self._order = []
        self._rows = []                              # (site, 's' | 'm') in order of the first reading of that kind

    def _pre_train(self, num_samples):
        print('====================================================')
        print('--- Pretraining ---')
        self.pilot_survey(num_samples, self.static_std)
        self.update_model()

    def pilot_survey(self, num_samples, std):
        ind = np.random.permutation(self.env.num_samples)[:num_samples]
        self._add_samples(ind, stds=[std] * num_samples)

    def _add_samples(self, indices, stds):
        ys = [None] * len(indices)
        for k, (idx, sd) in enumerate(zip(indices, stds)):
            if idx == -1:                                   # off-field cell on a path (agent.py:70-71)
                continue
            y = self.env.collect_samples(idx, sd)
            ys[k] = y
            if not self.static_data[idx] and not self.mobile_data[idx]:
                self._order.append(int(idx))                 # first reading at this site
            data, kind = (self.static_data, 's') if sd == self.static_std else (self.mobile_data, 'm')
            if not data[idx]:
                self._rows.append((int(idx), kind))          # first reading of this kind at this site
            data[idx].append(y)
        self.collected['ind'] += list(indices)
        self.collected['std'] += list(stds)
        self.collected['y'] += ys
        self._gen = getattr(self, '_gen', 0) + 1                # readings changed (see _masks)

    # ---- fusion of repeated readings (agent.py:92-117) --------------------------------------------
    def _site_means(self):
        """Counts and np.mean of the static / mobile readings of every site.  The reference walks all n sites
        and calls np.mean twice per sampled one at every planning step (agent.py:97-109); here a mean is
        re-evaluated only for a site whose reading list changed since the last call (same values, O(changes))."""
        n = self.env.num_samples
        cache = getattr(self, '_mean_cache', None)
        if cache is None or cache['n'] != n:
            cache = dict(n=n, cnt=[np.zeros(n, np.int64), np.zeros(n, np.int64)],
                         ids=[np.zeros(n, np.int64), np.zeros(n, np.int64)], mean=[np.zeros(n), np.zeros(n)])
            self._mean_cache = cache
        for k, data in enumerate((self.static_data, self.mobile_data)):
            cnt = np.fromiter(map(len, data), dtype=np.int64, count=n)
            ids = np.fromiter(map(id, data), dtype=np.int64, count=n)
            for i in np.nonzero((cnt != cache['cnt'][k]) | (ids != cache['ids'][k]))[0]:
                cache['mean'][k][i] = np.mean(data[i]) if cnt[i] else 0.0
            cache['cnt'][k], cache['ids'][k] = cnt, ids
        cache['gen'] = (getattr(self, '_gen', 0), id(self.static_data), id(self.mobile_data))
        return cache['cnt'][0], cache['cnt'][1], cache['mean'][0], cache['mean'][1]

    def get_sampled_dataset(self):
        ss, sm = self.static_std ** 2, self.mobile_std ** 2
        cs, cm, ms, mm = self._site_means()
        has_s, has_m = cs > 0, cm > 0
        idx = np.nonzero(has_s | has_m)[0]
        hs, hm = has_s[idx], has_m[idx]
        ys = np.where(hs & hm, (sm * ms[idx] + ss * mm[idx]) / (sm + ss), np.where(hs, ms[idx], mm[idx]))
        vs = np.where(hs & hm, 1 / (1 / ss + 1 / sm), np.where(hs, ss, sm))
        return [int(i) for i in idx], ys, vs

    def _masks(self):
        n = self.env.num_samples
        cache = getattr(self, '_mean_cache', None)
        if (cache is not None and cache['n'] == n and
                cache.get('gen') == (getattr(self, '_gen', 0), id(self.static_data), id(self.mobile_data))):
            return cache['cnt'][0] > 0, cache['cnt'][1] > 0      # nothing was sampled since the last full scan
        static = np.fromiter(map(len, self.static_data), dtype=np.int64, count=n) > 0
        mobile = np.fromiter(map(len, self.mobile_data), dtype=np.int64, count=n) > 0
        return static, mobile

    def _fused_var(self, static, mobile):
        """Noise variance of every sampled site given the masks (agent.py:298-307)."""
        ss, sm = self.static_std ** 2, self.mobile_std ** 2
        both = 1.0 / (1.0 / ss + 1.0 / sm)
        return np.where(static & mobile, both, np.where(static, ss, sm))

    # ---- model update (agent.py:84-90) ---------------------------------------------------------
    def update_model(self):
        indices, y, var = self.get_sampled_dataset()
        self.gp.fit(self.env.X[indices], y, var)
        self._pool_key = None                                     # fit re-used the device pool

    def _post_update(self):
        """agent.py:89-90 computes the dense n x n pool covariance on the host; here the pool
        coordinates are (re)loaded on the device and the dense matrix is only materialised if
        somebody reads `cov_matrix`."""
        self._cov_matrix = None
        self._cov_matrix_user = False
        self._pool_key = None

    @property
    def cov_matrix(self):
        if self._cov_matrix is None:
            self._cov_matrix = self.gp.cov_mat(x1=self.env.X, add_likelihood_var=True)
        return self._cov_matrix

    @cov_matrix.setter
    def cov_matrix(self, value):
        """Assigning a matrix makes greedy / best_path use exactly it (reference semantics)."""
        self._cov_matrix = np.asarray(value)
        self._cov_matrix_user = True
        self._pool_key = None

    def _load_pool(self):
        c = self.gp.ctx
        changed = self.gp.sync_hypers()
        key = ('cov', id(self._cov_matrix)) if self._cov_matrix_user else ('x', id(self.env.X))
        # the device pool is this agent's only while nobody else has loaded one into the context since
        # (GPR.predict, GPR.fit, utils.predictive_distribution all do): Context counts its pool loads
        if (changed or key != self._pool_key or getattr(c, '_pool_owner', None) is not self
                or c.pool_generation != getattr(self, '_pool_generation', -1)):
            if self._cov_matrix_user:
                c.set_pool_cov(self._cov_matrix)
            else:
                c.set_pool(self.env.X)
            self._pool_key = key
            c._pool_owner = self
            self._pool_generation = c.pool_generation
            self._comm_pool = None
        if self.comm is not None and getattr(self, '_comm_pool', None) != (id(c), c.pool_generation):
            self.comm.attach(c, self.env.num_samples)                # transport + owner map follow the pool
            self._comm_pool = (id(c), c.pool_generation)
        return c

    # ---- planning steps ---------------------------------------------------------------------------
    def _setup_ipp(self, criterion, update=False):
        self.criterion = criterion
        self._post_update()

    def predict(self, x=None, return_var=False, return_cov=False, return_mi=False):
        ind, y, var = self.get_sampled_dataset()
        if x is None and getattr(self, 'incremental', False) and not self._cov_matrix_user:
            return self._predict_incremental(ind, y, var, return_var, return_cov, return_mi)
        x = self.env.test_X if x is None else x
        self._pool_key = None                                       # predictive_distribution reloads the pool
        return predictive_distribution(self.gp, self.env.X[ind], y, x, var, return_var=return_var,
                                       return_cov=return_cov, return_mi=return_mi)

    def _predict_incremental(self, ind, y, var, return_var, return_cov, return_mi):
        """utils.predictive_distribution on the held-out set, with the factor kept across calls."""
        n, m = self.env.num_samples, len(self.env.test_X)
        hkey = self.gp.hypers()
        key = (tuple(hkey[0]), hkey[1], hkey[2], hkey[3], id(self.env.X), id(self.env.test_X))
        if self._pctx is None:
            self._pctx = _hip.Context(self.gp.dtype, self.gp.device)
        c = self._pctx
        if key != self._pctx_key:
            c.set_hypers(hkey[0], hkey[1], hkey[2], hkey[3])
            c.set_pool(np.vstack([np.asarray(self.env.X, np.float64).reshape(n, -1),
                                  np.asarray(self.env.test_X, np.float64).reshape(m, -1)]))
            self._pctx_key = key
        rows = self._use_rows()
        if rows:
            # one row per (site, kind): targets are the per-kind means, the constant mean is the reference's
            # (mean of the fused targets, models.py:129 on agent.py:100-109)
            static, mobile = self._masks()
            A, is_static = self._train_rows(static, mobile)
            _, _, ms, mm = self._site_means()
            ty = np.where(is_static, ms[A], mm[A]) if len(A) else np.zeros(0)
            tv = self._rows_noise(is_static)
            c.set_constant_mean(float(np.mean(y)) if len(y) else 0.0)
            c.set_train(A, ty, tv)
        else:
            pos = {int(i): k for k, i in enumerate(ind)}
            sampled = np.zeros(n, bool)
            sampled[ind] = True
            A = self._train_order(sampled)
            sel = np.array([pos[int(i)] for i in A], dtype=np.int64)
            c.set_constant_mean(None)
            c.set_train(A, y[sel] if len(sel) else y, var[sel] if len(sel) else var)
        # The factor of the sampled sites is the one Agent.greedy keeps in the pool context.  When that context is
        # in step (pool loaded, a candidate solve resident), bring IT up to date first -- the new sites are
        # resident candidates there, so their rows come from its V^T -- and copy the factor across instead of
        # solving the new rows against the kept factor a second time.
        g = self.gp.ctx
        shared = False
        if (len(A) and self._pool_key == ('x', id(self.env.X)) and getattr(g, '_pool_owner', None) is self
                and g.pool_generation == getattr(self, '_pool_generation', -1)
                and not self.gp.sync_hypers()
                and getattr(g, 'M', 0) == (n if self.comm is None else len(self._shard()))):
            try:
                if rows:
                    g.set_train(A, np.zeros(len(A)), tv)
                else:
                    static, mobile = self._masks()
                    g.set_train(A, np.zeros(len(A)), self._fused_var(static[A], mobile[A]))
                g.factorize(incremental=True)
                c.factorize_from(g)
                shared = True
            except (RuntimeError, ValueError):
                shared = False
        if not shared:
            c.factorize(incremental=True)
        test_idx = np.arange(n, n + m)
        if not (return_var or return_cov or return_mi):
            return c.posterior_mean(test_idx)
        c.set_candidates(test_idx, prior_includes_noise=False)
        c.solve_candidates(incremental=True)
        mu, pv = c.posterior()
        res = (mu, pv) if return_var else None
        if return_cov or return_mi:
            cov, mi = c.posterior_cov(want_cov=return_cov, want_mi=return_mi)
            if return_cov:
                res = (mu, cov)
            if return_mi:
                res = (mu, mi)
            if return_cov and return_mi:
                res = (mu, cov, mi)
        return res

    def _train_order(self, sampled):
        """Sampled sites in insertion order (falls back to index order for sites whose first
        reading was not seen by _add_samples, e.g. data copied from a parent agent)."""
        seen = set()
        order = [i for i in getattr(self, '_order', []) if sampled[i] and not (i in seen or seen.add(i))]
        if len(order) != int(sampled.sum()):
            order += [int(i) for i in np.where(sampled)[0] if i not in seen]
        return np.array(order, dtype=np.int64)

    # ---- one train row per (site, kind of reading) ---------------------------------------------------------
    # The reference fuses the static and the mobile readings of a site into one row (agent.py:100-109), so a
    # site that receives the other kind of reading CHANGES an old row (its noise) and an incremental factor has
    # to be rebuilt from that row on.  Keeping the site's static mean (noise ss) and mobile mean (noise sm) as
    # two rows is the same GP -- same posterior, log det S larger by log(ss+sm) per such site, the constant mean
    # taken from the fused targets -- and every new reading becomes an append or a change of a target only.
    def _use_rows(self):
        return (getattr(self, 'incremental', False) and getattr(self, 'criterion', 'entropy') == 'entropy'
                and not self._cov_matrix_user)

    def _train_rows(self, static, mobile, extra_static=(), extra_mobile=()):
        """(site per row, is_static per row): logged arrival order first, then whatever the masks add."""
        log = getattr(self, '_rows', [])
        cache = getattr(self, '_rows_cache', None)
        if cache is None or cache[3] is not log or cache[0] > len(log):
            cache = (0, np.zeros(0, np.int64), np.zeros(0, bool), log)
        if cache[0] != len(log):                                             # the log only grows: convert the new tail
            tail = log[cache[0]:]
            site = np.r_[cache[1], np.fromiter((r[0] for r in tail), dtype=np.int64, count=len(tail))]
            is_s = np.r_[cache[2], np.fromiter((r[1] == 's' for r in tail), dtype=bool, count=len(tail))]
            cache = (len(log), site, is_s, log)
        self._rows_cache = cache
        _, site, is_s, _ = cache
        n = len(static)
        ok = np.where(is_s, static[site], mobile[site]) if len(site) else np.zeros(0, bool)
        site, is_s = site[ok], is_s[ok]
        has_s = np.zeros(n, bool)
        has_m = np.zeros(n, bool)
        has_s[site[is_s]] = True
        has_m[site[~is_s]] = True
        parts_i, parts_s = [site], [is_s]
        for kind_is_s, mask, has, extra in ((True, static, has_s, extra_static), (False, mobile, has_m, extra_mobile)):
            miss = np.where(mask & ~has)[0]                         # readings not seen by _add_samples (copied data)
            has[miss] = True
            ex = np.array([i for i in dict.fromkeys(int(j) for j in extra) if not has[i]], dtype=np.int64)
            for arr in (miss, ex):
                parts_i.append(arr.astype(np.int64))
                parts_s.append(np.full(len(arr), kind_is_s))
            has[ex] = True
        # masks first for both kinds, then the extras: keep the documented order (static extras before mobile ones)
        order = [0, 1, 3, 2, 4]
        return np.concatenate([parts_i[k] for k in order]), np.concatenate([parts_s[k] for k in order])

    def _rows_noise(self, is_static):
        return np.where(is_static, self.static_std ** 2, self.mobile_std ** 2)

    def _rows_entropy_offset(self, idx):
        """H of the row form minus H of the fused form: (CONST + log(ss+sm)/2) per site that has two rows."""
        n_two = len(idx) - len(np.unique(idx))
        return n_two * (CONST + 0.5 * np.log(self.static_std ** 2 + self.mobile_std ** 2))

    def greedy(self, num_samples):
        """k most informative static sampling sites, greedily (agent.py:295-356)."""
        c = self._load_pool()
        static, mobile = self._masks()
        sampled = static | mobile
        # sharded: this rank scores its share of the pool (the factor is replicated); one all-gather per pick inside
        # algp_greedy_sharded, and the factor update below takes the new sites' rows from their owners' V^T
        cand = np.arange(self.env.num_samples) if self.comm is None else self._shard()
        if self._use_rows():
            A, is_static = self._train_rows(static, mobile)
            c.set_train(A, np.zeros(len(A)), self._rows_noise(is_static))
            c.factorize(incremental=True)
            c.set_candidates(cand, prior_includes_noise=True)
            c.solve_candidates(incremental=True, alive=~static[cand])
        elif getattr(self, 'incremental', False):
            A = self._train_order(sampled)
            c.set_train(A, np.zeros(len(A)), self._fused_var(static[A], mobile[A]))
            c.factorize(incremental=True)
            c.set_candidates(cand, prior_includes_noise=True)
            c.solve_candidates(incremental=True, alive=~static[cand])
        else:
            A = np.where(sampled)[0]
            c.set_train(A, np.zeros(len(A)), self._fused_var(static[A], mobile[A]))
            c.set_candidates(cand[~static[cand]], prior_includes_noise=True)
            c.fit_and_solve()                                   # one task-list launch up to 51 200 candidates
        if self.comm is not None:
            if self.criterion != 'entropy':
                raise ValueError('only the entropy criterion shards (the MI criterion needs the pool-wide complement on one GPU)')
            picks = c.greedy_sharded(_CRIT[self.criterion], self.static_std, self.mobile_std, int(num_samples))
        else:
            picks = c.greedy(_CRIT[self.criterion], self.static_std, self.mobile_std, int(num_samples))
        return [int(p) for p in picks]

    def _shard(self):
        key = (self.env.num_samples, self.comm.rank, self.comm.world_size, self.comm.layout)
        if getattr(self, '_shard_key', None) != key:
            self._shard_key, self._shard_idx = key, self.comm.mine(self.env.num_samples)
        return self._shard_idx

    def best_path(self, paths_mobile_indices, static_indices):
        """Index of the most informative path (agent.py:358-403)."""
        if len(paths_mobile_indices) == 1:
            return 0
        c = self._load_pool()
        n = self.env.num_samples
        static, mobile0 = self._masks()
        static = static.copy()
        static[static_indices] = True
        if self._use_rows():
            # row form: every path only APPENDS rows (a mobile row for each of its sites that has none yet) behind
            # the common base, even where it re-measures a static site; H is brought back to the fused form
            return int(np.argmax(self._path_utilities_rows(c, paths_mobile_indices, static, mobile0)))
        base = self._train_order(static | mobile0)          # sites sampled whichever path is taken
        in_base = np.zeros(n, bool)
        in_base[base] = True
        utilities = []
        for path in paths_mobile_indices:
            mobile = mobile0.copy()
            mobile[path] = True
            sampled = static | mobile
            # f3: the path's new sites are appended behind the common base, so consecutive paths share
            # the leading rows of the factor and only the appended block is re-factorised (the block
            # determinant lemma, done by algp_factorize_update); a base site whose fused noise changes
            # because the path re-measures it limits the reuse to the rows before it.
            extra = [int(i) for i in dict.fromkeys(int(j) for j in path) if not in_base[i]]
            A = np.r_[base, np.array(extra, dtype=np.int64)] if extra else base
            var = self._fused_var(static[A], mobile[A])
            c.set_train(A, np.zeros(len(A)), var)
            c.factorize(incremental=True)
            ut = c.entropy()
            if self.criterion == 'mutual_information':
                ut += c.set_entropy(np.where(~sampled)[0])            # H(C_AbarAbar), no measurement noise
                var_all = np.zeros(n)
                var_all[A] = var
                ut -= c.set_entropy(np.arange(n), var_all)
            utilities.append(ut)
        return int(np.argmax(utilities))

    def _path_utilities_rows(self, c, paths, static, mobile0, batched=True):
        """Entropy utility of every path relative to the common base (row form).  Batched: ONE factor update + ONE
        candidate-solve update for the base, then the posterior block of every path on the device at once
        (algp_score_paths: paths of up to 256 new sites, config 5's field rows); the per-path loop (one factor update per
        path) remains for longer paths, for sharded agents and as the cross-check of the tests."""
        n = self.env.num_samples
        pen = CONST + 0.5 * np.log(self.static_std ** 2 + self.mobile_std ** 2)     # per site that gets a second row
        clean = [[int(j) for j in dict.fromkeys(int(v) for v in path) if j != -1 and not mobile0[j]] for path in paths]
        if self.comm is not None:
            batched = False          # a shard does not hold every path site's row of V^T: the per-path factor updates run
                                     # on every rank alike (replicated factor; their new rows come through the row exchange)
        if batched and max((len(p) for p in clean), default=0) <= 256:
            A, is_static = self._train_rows(static, mobile0)
            c.set_train(A, np.zeros(len(A)), self._rows_noise(is_static))
            c.factorize(incremental=True)
            c.set_candidates(np.arange(n), prior_includes_noise=True)
            c.solve_candidates(incremental=True, alive=~static)
            width = max(1, max(len(p) for p in clean))
            sites = np.full((len(clean), width), -1, dtype=np.int64)
            for k, pth in enumerate(clean):
                sites[k, :len(pth)] = pth
            dH = c.score_paths(sites, self.mobile_std)
            second = np.array([sum(1 for j in pth if static[j]) for pth in clean], dtype=np.float64)
            return dH - second * pen
        utilities = []
        for path in paths:
            A, is_static = self._train_rows(static, mobile0, extra_mobile=[j for j in path if j != -1])
            c.set_train(A, np.zeros(len(A)), self._rows_noise(is_static))
            c.factorize(incremental=True)
            utilities.append(c.entropy() - self._rows_entropy_offset(A))
        ut = np.array(utilities)
        A0, is_static0 = self._train_rows(static, mobile0)
        c.set_train(A0, np.zeros(len(A0)), self._rows_noise(is_static0))
        c.factorize(incremental=True)
        return ut - (c.entropy() - self._rows_entropy_offset(A0))      # relative to the base, like the batched form

    # ---- the loops (agent.py:125-287, 475-518): host orchestration around the steps above -------------
    def get_samples_sequence_from_path(self, path, waypoints):
        indices, stds = [], []
        taken = [False] * len(waypoints)
        for loc in path:
            loc = tuple(loc)
            gi = self.env.map_pose_to_gp_index_matrix[loc]
            if gi is None:
                indices.append(-1)
                stds.append(-1)
                continue
            indices.append(gi)
            if loc in waypoints and not taken[waypoints.index(loc)]:
                taken[waypoints.index(loc)] = True
                stds.append(self.static_std)
            else:
                stds.append(self.mobile_std)
        return indices, stds

    def _follow(self, checkpoints, waypoints):
        nxt = np.stack(self.env.get_path_from_checkpoints(checkpoints))[1:]
        ind, stds = self.get_samples_sequence_from_path(nxt, waypoints)
        self.path = np.concatenate([self.path, nxt], axis=0).astype(int)
        self.pose = tuple(self.path[-1])
        self.heading = get_heading(self.path[-2], self.path[-1])
        return ind, stds

    def _choose(self, strategy, paths_indices, paths_cost, static_indices):
        if strategy == 'Shortest':
            return find_shortest_path(paths_cost)
        best = self.best_path(paths_indices, static_indices)
        return find_equi_sample_path(paths_indices, best) if strategy == 'Equi-Sample' else best

    def run_ipp(self, render=False, num_runs=10, criterion='entropy', update=False, slack=0, strategy='MaxEnt',
                disp=True):
        assert strategy in ['MaxEnt', 'Shortest', 'Equi-Sample'], 'Unknown strategy!!'
        assert criterion in ['entropy', 'mutual_information'], 'Unknown criterion!!'
        self._setup_ipp(criterion, update)
        test_error, pred, var, error = [], None, None, None
        for i in range(num_runs):
            t_run = time.time()
            new_idx = self.greedy(self.num_samples_per_batch)
            waypoints = [tuple(self.env.gp_index_to_map_pose(g)) for g in new_idx]
            nxt_static = np.stack(waypoints)
            self.static_locations = np.concatenate([self.static_locations, nxt_static]).astype(int)
            ub = self.env.get_heuristic_cost(self.pose, self.heading, waypoints)
            checkpoints, paths_idx, paths_cost = self.env.get_all_paths(self.pose, self.heading, waypoints, ub, slack)
            best = self._choose(strategy, paths_idx, paths_cost, new_idx)
            ind, stds = self._follow(checkpoints[best], waypoints)
            if render:
                self.predict(self.env.all_x)
                self.env.render(checkpoints[best], self.path, nxt_static, self.static_locations)
            self._add_samples(ind, stds)
            if update and (i + 1) % self.update_every == 0:
                self.update_model()
                self._post_update()
            pred, var = self.predict(return_var=True)
            error = compute_mae(self.env.test_Y, pred)
            test_error.append(error)
            if disp:
                print('Run {}/{}: {} feasible paths, test ERROR {:.4f}, predictive variance max {:.3f} min {:.3f} '
                      'mean {:.3f}, {:.3f}s'.format(i + 1, num_runs, len(paths_idx), error, var.max(), var.min(),
                                                    var.mean(), time.time() - t_run))
        print('Strategy: {:s}  final test ERROR: {}'.format(strategy, error))
        return {'mean': pred, 'error': test_error}

    def run_greedy_ipp(self, num_runs=10, criterion='entropy', strategy='MaxEnt', disp=True):
        self._setup_ipp(criterion)
        for i in range(num_runs):
            new_idx = self.greedy(self.num_samples_per_batch)
            waypoints = [tuple(self.env.gp_index_to_map_pose(g)) for g in new_idx]
            self.static_locations = np.concatenate([self.static_locations, np.stack(waypoints)]).astype(int)
            costs, seq = self.env.map.nearest_waypoint_path_cost(self.pose, self.heading, waypoints, return_seq=True)
            for j in range(len(seq)):
                checkpoints, paths_idx, paths_cost = self.env.get_all_paths(self.pose, self.heading, [waypoints[seq[j]]],
                                                                            costs[j], slack=0)
                assert costs[j] == paths_cost[0], 'path costs do not match'
                best = self._choose(strategy, paths_idx, paths_cost, [new_idx[seq[j]]])
                ind, stds = self._follow(checkpoints[best], waypoints)
                self._add_samples(ind, stds)
        pred, var = self.predict(return_var=True)
        error = compute_mae(self.env.test_Y, pred)
        print('Strategy: {:s}  final test ERROR: {:.4f}'.format(strategy, error))
        return {'mean': pred, 'error': [error]}

    def run_naive(self, std, counts, metric='distance'):
        """Lawn-mower baselines 'Naive Static' / 'Naive Mobile' (agent.py:405-473): sweep the rows back
        and forth, sampling every field cell passed with noise `std`; after each leg of `counts`
        ('distance': cells moved, 'samples': readings taken) predict the held-out set with its
        covariance and mutual information."""
        rows = self.env.map.shape[0]
        errors, mis, mean_vars = [], [], []
        mu = cov = None
        for ns in counts:
            inds, moved = [], 0
            while True:
                nxt = (self.pose[0] + self.heading[0], self.pose[1] + self.heading[1])
                gi = self.env.map_pose_to_gp_index_matrix[nxt]
                if gi is not None:
                    inds.append(gi)
                if metric == 'samples':
                    if nxt[0] in (0, rows - 1):
                        # end of a row: step two columns over and turn around
                        back = nxt[0] - 1 if nxt[0] == rows - 1 else nxt[0] + 1
                        hop = [nxt, (nxt[0], nxt[1] + 1), (nxt[0], nxt[1] + 2), (back, nxt[1] + 2)]
                        self.path = np.concatenate([self.path, hop], axis=0).astype(int)
                        self.heading = (-self.heading[0], 0)
                        self.pose = hop[-1]
                    else:
                        self.path = np.concatenate([self.path, [nxt]], axis=0).astype(int)
                        self.pose = nxt
                    if len(inds) == ns:
                        break
                elif metric == 'distance':
                    moved += 1
                    self.path = np.concatenate([self.path, [nxt]], axis=0).astype(int)
                    if nxt[1] % 2 == 0 and nxt[0] == 0:
                        self.heading = (0, 1) if self.heading == (-1, 0) else (1, 0)
                    elif nxt[1] % 2 == 0 and nxt[0] == rows - 1:
                        self.heading = (0, 1) if self.heading == (1, 0) else (-1, 0)
                    self.pose = nxt
                    if moved == ns:
                        break
                else:
                    raise NotImplementedError
            self._add_samples(inds, [std] * len(inds))
            mu, cov, mi = self.predict(return_cov=True, return_mi=True)
            errors.append(compute_mae(self.env.test_Y, mu))
            mis.append(mi)
            mean_vars.append(np.diag(cov).mean())
        var = np.diag(cov)
        print('Strategy: ', 'Naive Static' if std == self.static_std else 'Naive Mobile')
        print('Test ERROR: {:.4f}  predictive variance max {:.3f} min {:.3f} mean {:.3f}'.format(
            errors[-1], var.max(), var.min(), var.mean()))
        return {'mean': mu, 'error': errors, 'mi': mis, 'mean_var': mean_vars}

    def prediction_vs_distance(self, test_every, num_runs):
        """Posterior on the held-out set after the first r * test_every collected samples, r = 1..num_runs
        (agent.py:497-518).  The reference runs a from-scratch predictive_distribution per prefix; here the collected
        samples are the rows of ONE context in collection order, so every prefix extends the previous factor
        (algp_factorize_update) and only the new columns of V^T are solved (algp_solve_candidates_update).  A site
        measured twice is two rows with independent likelihood noise, as in the reference's cov_aa (utils.py:296)."""
        errors, mis, mean_vars, mu = [], [], [], None
        inds = np.array(self.collected['ind'][:num_runs * test_every])
        ok = inds != -1
        rows_before = np.concatenate([[0], np.cumsum(ok)])          # valid rows among the first `count` samples
        sel = inds[ok].astype(int)
        var = np.array(self.collected['std'])[:len(inds)][ok].astype(float) ** 2
        y = np.array(self.collected['y'])[:len(inds)][ok].astype(float)
        x_rows = np.asarray(self.env.X, np.float64).reshape(self.env.num_samples, -1)[sel]
        test_x = np.asarray(self.env.test_X, np.float64).reshape(len(self.env.test_X), -1)
        self.gp.sync_hypers()
        hk = self.gp.hypers()
        c = _hip.Context(self.gp.dtype, self.gp.device)
        try:
            c.set_hypers(hk[0], hk[1], hk[2], hk[3])
            c.set_pool(np.vstack([x_rows, test_x]))                 # every collected sample is a pool entry of its own
            test_idx = np.arange(len(x_rows), len(x_rows) + len(test_x))
            for r in range(1, num_runs + 1):
                nr = int(rows_before[min(r * test_every, len(inds))])
                c.set_train(np.arange(nr), y[:nr], var[:nr])
                c.factorize(incremental=True)
                c.set_candidates(test_idx, prior_includes_noise=False)
                c.solve_candidates(incremental=True)
                mu, _ = c.posterior()
                cov, mi = c.posterior_cov(want_cov=True, want_mi=True)
                errors.append(compute_mae(self.env.test_Y, mu))
                mis.append(mi)
                mean_vars.append(np.diag(cov).mean())
        finally:
            c.close()
        return {'mean': mu, 'error': errors, 'mi': mis, 'mean_var': mean_vars}
