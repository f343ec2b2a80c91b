"""The library's remaining diagnostic switches (README.md) are read once per process, so each setting runs the same small
planning step in a process of its own; every setting has to reproduce the default's results:

  ALGP_CHOL_DAG=0        launch-sequence factorisation instead of the one-launch task list (same factor up to rounding:
                         the order in which a tile's rank-k updates are batched differs)
  ALGP_LAZY_GREEDY=0     every row of V^T is scored before every pick (bit-identical picks and utilities)
  ALGP_FACTOR_FROM_VT=0  new rows of an updated factor are solved against the kept blocks instead of gathered from V^T
  ALGP_TRSM_CHUNKS=1     the candidate solve on one stream instead of three row chunks (bit-identical: rows are independent)

(reference path: agent.py:295-356 on top of utils.py:293-319; the oracle is not involved, this is a self-consistency test
of code paths that only exist for A/B timing)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_STEP = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
from algp_amd import _hip
rng = np.random.RandomState(3)
N, M = 1500, 3000                                  # 12 tiles of 128: the one-launch factorisation is eligible
X = rng.uniform(0, 40, size=(N + M + 40, 2))
c = _hip.Context(np.float64)
c.set_hypers(np.log([3.0, 2.0]), 0.1, np.log(1e-2))
c.set_pool(X)
idx = np.arange(N)
var = np.where(rng.uniform(size=N) < 0.5, 0.01, 1.0)
y = np.sin(X[:N, 0] / 5) + 0.1 * rng.standard_normal(N)
cand = np.arange(N, N + M)
c.set_train(idx, y, var)
c.factorize(incremental=True)
ld0 = c.logdet()
c.set_candidates(cand, prior_includes_noise=True)
c.solve_candidates(incremental=True)
mu, pv = c.posterior()
picks, util = [], []
for _ in range(4):                                 # the lazily resolved argmax, pick by pick
    pos, site, val = c.best_candidate(_hip.CRIT_ENTROPY, 0.1, 1.0)
    c.commit_pick(site, 0.1, 1.0)
    picks.append(int(site)); util.append(float(val))
# one incremental step: the picks + 20 further candidate sites join the train set
new = np.r_[np.asarray(picks, dtype=np.int64), cand[rng.permutation(M)[:20]]]
new = new[np.sort(np.unique(new, return_index=True)[1])]
idx2 = np.r_[idx, new]
c.set_train(idx2, np.r_[y, np.zeros(len(new))], np.r_[var, np.full(len(new), 0.01)])
kept = c.factorize(incremental=True)
ld1 = c.logdet()
c.set_candidates(cand, prior_includes_noise=True)
c.solve_candidates(incremental=True)
picks2 = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 3)
util2 = [float(v) for v in c.posterior()[1][:64]]  # the state after the picks (every row caught up)
print(json.dumps({'ld0': ld0, 'ld1': ld1, 'kept': int(kept), 'picks': [int(p) for p in picks], 'util': [float(u) for u in util],
                  'picks2': [int(p) for p in picks2], 'util2': [float(u) for u in util2],
                  'mu': [float(v) for v in mu[:64]], 'pv': [float(v) for v in pv[:64]]}))
""" % REPO


def _run(env_extra):
    env = dict(os.environ)
    for k in ('ALGP_CHOL_DAG', 'ALGP_LAZY_GREEDY', 'ALGP_FACTOR_FROM_VT', 'ALGP_TRSM_CHUNKS'):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, '-c', _STEP], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.fixture(scope='module')
def default_run():
    return _run({})


@pytest.mark.parametrize('switch,exact', [('ALGP_CHOL_DAG=0', False), ('ALGP_LAZY_GREEDY=0', True),
                                          ('ALGP_FACTOR_FROM_VT=0', False), ('ALGP_TRSM_CHUNKS=1', True)])
def test_switch_reproduces_default(default_run, switch, exact):
    k, v = switch.split('=')
    got, ref = _run({k: v}), default_run
    assert got['picks'] == ref['picks'] and got['picks2'] == ref['picks2'], (switch, got['picks'], ref['picks'])
    assert got['kept'] == ref['kept']
    for name in ('ld0', 'ld1'):
        assert abs(got[name] - ref[name]) <= (0.0 if exact else 1e-10 * abs(ref[name])), (switch, name, got[name], ref[name])
    for name in ('util', 'util2', 'mu', 'pv'):
        for a, b in zip(got[name], ref[name]):
            assert abs(a - b) <= (0.0 if exact else 1e-9 * max(1.0, abs(b))), (switch, name, a, b)


# ---- the fall-back routes of the incremental step and of fit + solve ---------------------------------------------------------
_STEP5 = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
from algp_amd import _hip
rng = np.random.RandomState(8)
N, M = 2300, 4300                                  # >= 2 048 unchanged rows and >= 2 048 candidate rows: the tail kernel is eligible
X = rng.uniform(0, 60, size=(N + M + 100, 2))
y = np.sin(X[:, 0] / 5) + 0.1 * rng.standard_normal(len(X))
var = np.where(rng.uniform(size=len(X)) < 0.5, 0.01, 1.0)
cand = np.arange(N + 100, N + 100 + M)
c = _hip.Context(np.float64)
c.set_hypers(np.log([3.0, 2.0]), 0.1, np.log(1e-2))
c.set_pool(X)
out = {}
idx = np.arange(N - 40)
for tag, add in (('a', 0), ('b', 23), ('c', 30)):   # 2260 -> 2283 (unaligned start) -> 2313 (straddles column 2304)
    idx = np.arange(len(idx) + add)
    c.set_train(idx, y[idx], var[idx])
    c.factorize(incremental=True)
    c.set_candidates(cand, prior_includes_noise=False)
    kept = c.solve_candidates(incremental=True)
    mu, pv = c.posterior()
    out[tag] = {'kept': int(kept), 'mu': [float(v) for v in mu[::67]], 'pv': [float(v) for v in pv[::67]], 'ld': c.logdet()}
# fit + solve in one launch with padding rows in the candidates' last tile (z rides in one of them), and a fit iteration
c.set_train(np.arange(N), y[:N], var[:N])
c.set_candidates(cand[:4001], prior_includes_noise=True)
c.fit_and_solve()
mu, pv = c.posterior()
mll, g = c.fit_step()
out['fold'] = {'mu': [float(v) for v in mu[::61]], 'pv': [float(v) for v in pv[::61]], 'mll': mll, 'grad': [float(v) for v in g],
               'alpha': [float(v) for v in c.alpha()[::97]]}
print(json.dumps(out))
""" % REPO


def _run5(env_extra):
    env = dict(os.environ)
    for k in ('ALGP_TAIL_SPLIT', 'ALGP_TAIL_COLS', 'ALGP_FOLD', 'ALGP_SOLVE_DAG'):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, '-c', _STEP5], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.fixture(scope='module')
def default_run5():
    return _run5({})


@pytest.mark.parametrize('switch', ['ALGP_TAIL_SPLIT=0', 'ALGP_TAIL_COLS=0', 'ALGP_FOLD=0', 'ALGP_SOLVE_DAG=0'])
def test_fallback_route_reproduces_default(default_run5, switch):
    """The size-range fall-backs that stay in the library (csrc/common.h lists every switch): the tail kernel without its
    k-split, appended columns re-solved as whole 128-column blocks (what inputs below 2 048 rows take), fit and solve as two
    steps, the mid-sized solve as launches instead of the task list -- the same posterior, log-determinant, MLL and gradient
    to rounding.  (Round 5's A/B-only switches -- ALGP_TAIL_EXACT / _STRADDLE / _ROWS, ALGP_Z_IN_PANEL, ALGP_FIT_ONE_LAUNCH,
    ALGP_SYRK_PANELS, ALGP_TRSM_PUSH_STREAMS -- and their branches are gone.)  Reference: agent.py:210 refits from scratch at
    every step; models.py:145-158."""
    got, ref = _run5(dict(kv.split('=') for kv in switch.split())), default_run5
    assert ref['b']['kept'] == 2260 and ref['c']['kept'] == 2283                      # exactly the appended columns by default
    for tag in ('a', 'b', 'c'):
        assert abs(got[tag]['ld'] - ref[tag]['ld']) <= 1e-10 * abs(ref[tag]['ld'])
        for name in ('mu', 'pv'):
            assert max(abs(a - b) for a, b in zip(got[tag][name], ref[tag][name])) <= 1e-9, (switch, tag, name)
    for name in ('mu', 'pv', 'grad', 'alpha'):
        assert max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(got['fold'][name], ref['fold'][name])) <= 1e-9, (switch, name)
    assert abs(got['fold']['mll'] - ref['fold']['mll']) <= 1e-11 * abs(ref['fold']['mll'])
