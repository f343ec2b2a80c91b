"""Generate golden vectors G1-G6 from the reference itself.

Run ONLY in the build container (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py

It imports the reference's *NumPy half* of the hot path unmodified from
/root/reference (utils.py: predictive_distribution, entropy_from_cov,
generate_gaussian_data; agent.py: Agent.greedy / best_path / get_sampled_dataset)
after pre-seeding ``sys.modules`` with empty stand-ins for modules the reference
imports but does not use on this path (seaborn, ipdb) and for ``models`` (which
needs the absent gpytorch).  ``gp.cov_mat`` is served by the closed-form fp32
kernel of oracle/gp_oracle.py with the reference's in-place fp32 diagonal adds
(models.py:175-180), because GPyTorch is not installable here: kernel-matrix
values are therefore "parity unpinned", everything downstream of them is pinned.

Only numeric inputs/outputs are written (``*.npz``); no reference source,
bytecode or pickled reference object is copied.
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)

from oracle import gp_oracle as O  # noqa: E402


def _import_reference():
    for name in ('seaborn', 'ipdb'):
        sys.modules.setdefault(name, types.ModuleType(name))
    m = types.ModuleType('models')

    class GPR(object):  # never instantiated on this path
        pass
    m.GPR = GPR
    sys.modules['models'] = m
    import matplotlib
    matplotlib.use('Agg')
    sys.path.insert(0, REF)
    import utils as ref_utils
    import agent as ref_agent
    return ref_utils, ref_agent


class StandInGP(object):
    """gp.cov_mat stand-in: closed-form fp32 kernel + in-place fp32 adds."""

    def __init__(self, hyp):
        self.hyp = hyp

    def cov_mat(self, x1, x2=None, white_noise_var=None, add_likelihood_var=False):
        if x2 is not None and np.array_equal(np.asarray(x1, np.float32), np.asarray(x2, np.float32)):
            x2 = None                                    # models.py:169 torch.equal branch
        return O.cov_mat_ref(self.hyp, x1, x2, white_noise_var, add_likelihood_var, dtype=np.float32)


def spd(k, dtype, rng):
    a = rng.standard_normal((k, k))
    return (a @ a.T + k * np.eye(k)).astype(dtype)


def g1_entropy(ref_utils, rng, out):
    for k in (0, 1, 5, 64):
        for dt in (np.float32, np.float64):
            cov = spd(k, dt, rng) if k else np.zeros((0, 0), dt)
            out['g1_cov_k%d_%s' % (k, np.dtype(dt).name)] = cov
            out['g1_ent_k%d_%s' % (k, np.dtype(dt).name)] = np.float64(ref_utils.entropy_from_cov(cov))


def grid_points(R, C):
    xx, yy = np.meshgrid(np.arange(C), np.arange(R))
    return np.vstack([yy.flatten(), xx.flatten()]).transpose().astype(np.float64)


def g2_predictive(ref_utils, rng, out):
    cases = [(5, 3, 2), (50, 40, 2), (200, 100, 6)]
    flags = [(False, False, False), (True, False, False), (False, True, False),
             (False, False, True), (False, True, True), (True, False, True)]
    for ci, (N, M, D) in enumerate(cases):
        x = rng.uniform(0, 12, size=(N + M, D))
        train_x, test_x = x[:N], x[N:]
        log_ls = np.log(rng.uniform(2.0, 4.0, size=D))
        hyp = O.Hypers(log_ls, log_outputscale=np.log(1.3), log_noise=np.log(0.05))
        f = np.sin(train_x[:, 0] / 3.0) + np.cos(train_x[:, 1] / 2.0)
        train_var = rng.choice([0.01, 1.0, 1.0 / (1 / 0.01 + 1 / 1.0)], size=N)
        train_y = f + rng.standard_normal(N) * np.sqrt(train_var)
        test_var = rng.uniform(0.05, 0.2, size=M)
        gp = StandInGP(hyp)
        pre = 'g2_c%d_' % ci
        out[pre + 'train_x'] = train_x
        out[pre + 'test_x'] = test_x
        out[pre + 'train_y'] = train_y
        out[pre + 'train_var'] = train_var
        out[pre + 'test_var'] = test_var
        out[pre + 'log_ls'] = log_ls
        out[pre + 'log_os'] = np.float64(hyp.log_outputscale)
        out[pre + 'log_noise'] = np.float64(hyp.log_noise)
        for use_tv in (0, 1):
            for use_xv in (0, 1):
                tv = train_var if use_tv else None
                xv = test_var if use_xv else None
                for (rv, rc, rm) in flags:
                    if rm and not use_xv:
                        continue   # slogdet of un-jittered K_xx is noise in the reference too
                    res = ref_utils.predictive_distribution(gp, train_x, train_y, test_x, tv, xv,
                                                            return_var=rv, return_cov=rc, return_mi=rm)
                    tag = pre + 'tv%d_xv%d_f%d%d%d_' % (use_tv, use_xv, rv, rc, rm)
                    if not isinstance(res, tuple):
                        res = (res,)
                    out[tag + 'arity'] = np.int64(len(res))
                    for k, r in enumerate(res):
                        out[tag + 'r%d' % k] = np.asarray(r)


class _Log(object):
    def __init__(self, fn):
        self.fn = fn
        self.vals = []

    def __call__(self, cov, constant=None):
        v = self.fn(cov) if constant is None else self.fn(cov, constant)
        self.vals.append(v)
        return v


def _fake_agent(cov, static_data, mobile_data, static_std, mobile_std, criterion):
    env = types.SimpleNamespace(num_samples=cov.shape[0])
    return types.SimpleNamespace(env=env, static_data=static_data, mobile_data=mobile_data,
                                 static_std=static_std, mobile_std=mobile_std, criterion=criterion,
                                 cov_matrix=cov)


def _greedy_utilities_from_log(vals, n, static0, k, criterion, picks):
    """Rebuild utilities[k, n] from the logged entropy_from_cov values using the
    call order of agent.py:309 (ent_v) and :329-338 (ent_a [, ent_abar, ent_all])."""
    vals = list(vals)
    ent_v = vals.pop(0)
    static = np.array(static0, dtype=bool)
    uts = np.full((k, n), -np.inf)
    cumm = []
    for p in range(k):
        cond = ent_v + sum(cumm)
        for i in range(n):
            if static[i]:
                continue
            ent_a = vals.pop(0)
            if criterion == 'mutual_information':
                ent_abar = vals.pop(0)
                ent_all = vals.pop(0)
                uts[p, i] = ent_a + ent_abar - ent_all
            else:
                uts[p, i] = ent_a - cond
        best = int(np.argmax(uts[p]))
        assert best == picks[p], (best, picks[p])
        cumm.append(uts[p, best])
        static[best] = True
    assert not vals
    return uts


def make_state(n, kind, rng):
    static_data = [[] for _ in range(n)]
    mobile_data = [[] for _ in range(n)]
    perm = rng.permutation(n)
    if kind in ('static', 'both'):
        for i in perm[:max(3, n // 8)]:
            static_data[i].append(float(rng.uniform()))
    if kind in ('mobile', 'both'):
        lo = 0 if kind == 'mobile' else max(3, n // 8) - 2      # 'both': two sites carry both kinds
        for i in perm[lo:lo + max(4, n // 6)]:
            mobile_data[i].append(float(rng.uniform()))
            if rng.uniform() < 0.3:
                mobile_data[i].append(float(rng.uniform()))
    return static_data, mobile_data


def g3_greedy(ref_utils, ref_agent, rng, out):
    static_std, mobile_std = 0.1, 1.0
    for (R, Cc, ntest, ls) in ((8, 8, 0, 1.5), (20, 20, 40, 3.0)):
        X = grid_points(R, Cc)
        keep = np.sort(rng.permutation(len(X))[:len(X) - ntest])
        X = X[keep]
        n = len(X)
        hyp = O.Hypers(np.log([ls, ls]), 0.0, np.log(1e-2))
        gp = StandInGP(hyp)
        cov = gp.cov_mat(X, add_likelihood_var=True)            # agent.py:90
        pre = 'g3_n%d_' % n
        out[pre + 'X'] = X
        out[pre + 'cov'] = cov
        out[pre + 'log_ls'] = hyp.log_lengthscale
        out[pre + 'log_os'] = np.float64(0.0)
        out[pre + 'log_noise'] = np.float64(hyp.log_noise)
        for kind in ('empty', 'static', 'mobile', 'both'):
            sd, md = make_state(n, kind, rng)
            s0 = np.array([len(v) > 0 for v in sd])
            m0 = np.array([len(v) > 0 for v in md])
            out[pre + kind + '_static'] = s0
            out[pre + kind + '_mobile'] = m0
            for crit in ('entropy', 'mutual_information'):
                log = _Log(ref_utils.entropy_from_cov)
                ref_agent.entropy_from_cov = log
                try:
                    fake = _fake_agent(cov, sd, md, static_std, mobile_std, crit)
                    picks = ref_agent.Agent.greedy(fake, 4)
                finally:
                    ref_agent.entropy_from_cov = ref_utils.entropy_from_cov
                uts = _greedy_utilities_from_log(log.vals, n, s0, 4, crit, picks)
                tag = pre + kind + '_' + crit
                out[tag + '_picks'] = np.array(picks, dtype=np.int64)
                out[tag + '_ut'] = uts
                print(tag, picks)


def g4_best_path(ref_utils, ref_agent, rng, out):
    static_std, mobile_std = 0.1, 1.0
    X = grid_points(10, 10)
    n = len(X)
    hyp = O.Hypers(np.log([2.0, 2.0]), 0.0, np.log(1e-2))
    cov = StandInGP(hyp).cov_mat(X, add_likelihood_var=True)
    sd, md = make_state(n, 'both', rng)
    s0 = np.array([len(v) > 0 for v in sd])
    m0 = np.array([len(v) > 0 for v in md])
    out['g4_X'] = X
    out['g4_cov'] = cov
    out['g4_static'] = s0
    out['g4_mobile'] = m0
    static_indices = [int(i) for i in rng.permutation(n)[:3]]
    out['g4_static_indices'] = np.array(static_indices, dtype=np.int64)
    paths = [[int(i) for i in rng.permutation(n)[:L]] for L in (6, 9, 9, 12, 7)]
    out['g4_paths_len'] = np.array([len(p) for p in paths], dtype=np.int64)
    out['g4_paths_flat'] = np.array(sum(paths, []), dtype=np.int64)
    for crit in ('entropy', 'mutual_information'):
        fake = _fake_agent(cov, sd, md, static_std, mobile_std, crit)
        assert ref_agent.Agent.best_path(fake, paths[:1], static_indices) == 0     # agent.py:362
        log = _Log(ref_utils.entropy_from_cov)
        ref_agent.entropy_from_cov = log
        try:
            idx = ref_agent.Agent.best_path(fake, paths, static_indices)
        finally:
            ref_agent.entropy_from_cov = ref_utils.entropy_from_cov
        v = log.vals
        if crit == 'mutual_information':
            ut = np.array([v[3 * i] + v[3 * i + 1] - v[3 * i + 2] for i in range(len(paths))])
        else:
            ut = np.array(v)
        assert int(np.argmax(ut)) == idx
        out['g4_%s_idx' % crit] = np.int64(idx)
        out['g4_%s_ut' % crit] = ut
        print('g4', crit, idx)


def g5_fusion(ref_agent, rng, out):
    n = 12
    sd = [[] for _ in range(n)]
    md = [[] for _ in range(n)]
    lens_s = [0, 1, 3, 0, 2, 0, 1, 0, 0, 4, 1, 0]
    lens_m = [0, 0, 2, 1, 0, 3, 1, 0, 5, 0, 2, 0]
    for i in range(n):
        sd[i] = [float(v) for v in rng.uniform(0, 2, size=lens_s[i])]
        md[i] = [float(v) for v in rng.uniform(0, 2, size=lens_m[i])]
    fake = types.SimpleNamespace(env=types.SimpleNamespace(num_samples=n), static_data=sd, mobile_data=md,
                                 static_std=0.1, mobile_std=1.0)
    idx, y, var = ref_agent.Agent.get_sampled_dataset(fake)
    out['g5_lens_s'] = np.array(lens_s, dtype=np.int64)
    out['g5_lens_m'] = np.array(lens_m, dtype=np.int64)
    out['g5_flat_s'] = np.array(sum(sd, []))
    out['g5_flat_m'] = np.array(sum(md, []))
    out['g5_idx'] = np.array(idx, dtype=np.int64)
    out['g5_y'] = y
    out['g5_var'] = var


def g6_field(ref_utils, out):
    for seed, (R, Cc) in ((1, (20, 20)), (7, (9, 13))):
        np.random.seed(seed)
        grid, y = ref_utils.generate_gaussian_data(R, Cc, k=5)
        out['g6_s%d_grid' % seed] = grid
        out['g6_s%d_y' % seed] = y
        out['g6_s%d_shape' % seed] = np.array([R, Cc], dtype=np.int64)


def main():
    ref_utils, ref_agent = _import_reference()
    assert abs(ref_utils.CONST - O.CONST) < 1e-15
    rng = np.random.RandomState(20261004)
    for name, fn in (('g1_entropy', lambda o: g1_entropy(ref_utils, rng, o)),
                     ('g2_predictive', lambda o: g2_predictive(ref_utils, rng, o)),
                     ('g3_greedy', lambda o: g3_greedy(ref_utils, ref_agent, rng, o)),
                     ('g4_best_path', lambda o: g4_best_path(ref_utils, ref_agent, rng, o)),
                     ('g5_fusion', lambda o: g5_fusion(ref_agent, rng, o)),
                     ('g6_field', lambda o: g6_field(ref_utils, o))):
        out = {}
        fn(out)
        path = os.path.join(os.environ.get('ALGP_GOLDEN_OUT', HERE), name + '.npz')
        np.savez_compressed(path, **out)
        print('wrote', path, len(out), 'arrays', os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
