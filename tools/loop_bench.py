"""f1 measurement: planning steps on the bench workload (N0 = 10 000 train, M = 100 000 candidates,
fp64) where each step appends the 4 greedy picks (static) + 28 path sites (mobile) to the train set.
Compares from-scratch (what the reference does each step: agent.py:210, 295) with prefix reuse."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algp_amd import _hip

rng = np.random.RandomState(1)
R0, C0 = (int(v) for v in os.environ.get('LOOP_GRID', '100x100').split('x'))
N0, M, steps = R0 * C0, int(os.environ.get('LOOP_M', '100000')), int(os.environ.get('LOOP_STEPS', '6'))
xx, yy = np.meshgrid(np.arange(C0), np.arange(R0))
Xa = np.vstack([yy.ravel(), xx.ravel()]).T.astype(np.float64)
ii, jj = np.meshgrid(np.arange(400), np.arange(250), indexing='ij')
Xc = np.vstack([(ii.ravel() + 0.37) * (R0 / 400.0), (jj.ravel() + 0.41) * (C0 / 250.0)]).T[:M]
Xc = Xc + 0.03 * rng.standard_normal(Xc.shape)        # generic positions: no exact lattice ties
pool = np.vstack([Xa, Xc])
y0 = rng.uniform(0, 1, N0)
var0 = np.where(rng.uniform(size=N0) < 0.5, 0.01, 1.0)
res = {}
for mode in ('scratch', 'incremental'):
    c = _hip.Context(np.float64)
    c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
    c.set_pool(pool)
    idx = np.arange(N0)
    var = var0.copy()
    static = np.zeros(len(pool), bool)
    static[:N0] = var == 0.01
    cand = np.arange(N0, N0 + M)
    r2 = np.random.RandomState(7)
    times = []
    for s in range(steps + 1):
        inc = mode == 'incremental'
        phases = {}
        prof = bool(os.environ.get('LOOP_PROF')) and s == int(os.environ.get('LOOP_PROF_STEP', steps))   # phase split of one step
        if prof:
            c.prof_enable(True)
            c.prof_reset()

        def phase(name, t_prev):
            if prof:
                c.sync()
                phases[name] = (time.perf_counter() - t_prev) * 1e3
            return time.perf_counter()
        t0 = tp = time.perf_counter()
        c.set_train(idx, np.zeros(len(idx)), var)
        tp = phase('set_train', tp)
        kr = c.factorize(incremental=inc)
        tp = phase('factorize', tp)
        c.set_candidates(cand, prior_includes_noise=True)
        tp = phase('set_candidates', tp)
        kc = c.solve_candidates(incremental=inc, alive=~static[cand])
        tp = phase('solve_candidates', tp)
        if s == steps:                             # the last step also returns every utility (full pass) for the comparison
            picks, ut = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4, want_utilities=True)
            tp = phase('greedy4_with_utilities', tp)
        else:                                      # what Agent.greedy asks for: the picks
            picks = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4)
            tp = phase('greedy4', tp)
        c.sync()
        times.append((time.perf_counter() - t0) * 1e3)
        if prof:
            phases['classes'] = {k: c.prof_get(k) for k in _hip.PROF if c.prof_get(k)['launches']}
            c.prof_enable(False)
            last_phases = phases
        static[picks] = True
        mob = cand[r2.permutation(M)[:28]]
        mob = mob[~np.isin(mob, idx) & ~np.isin(mob, picks)]
        idx = np.r_[idx, picks, mob]
        var = np.r_[var, np.full(4, 0.01), np.full(len(mob), 1.0)]
    res[mode] = dict(best_utilities=[float(np.nanmax(u)) for u in ut], nan_count=int(np.isnan(ut).sum()), ms_per_step=times, kept_rows_last=int(kr), kept_cols_last=int(kc), picks_last=[int(p) for p in picks])
    if os.environ.get('LOOP_PROF'):
        res[mode]['last_step_phases_ms'] = last_phases
    c.close()
assert res['scratch']['picks_last'] == res['incremental']['picks_last'], (res['scratch']['picks_last'], res['incremental']['picks_last'])
res['max_abs_utility_diff'] = float(np.max(np.abs(np.array(res['scratch']['best_utilities']) - np.array(res['incremental']['best_utilities']))))
print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk != 'ms_per_step'}) for k, v in res.items()}), file=sys.stderr)
assert res['max_abs_utility_diff'] < 1e-8, res['max_abs_utility_diff']
res['speedup_steady_state'] = float(np.median(res['scratch']['ms_per_step'][1:]) / np.median(res['incremental']['ms_per_step'][1:]))
print(json.dumps(res))
