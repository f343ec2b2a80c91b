"""Times one rank's share of BASELINE config 4 -- N = 10 000 train points, M candidates -- as algp_fit_and_solve (fit + solve
in ONE task-list launch up to 51 200 rows) and as algp_factorize + algp_solve_candidates, in the current environment
($ALGP_FOLD, $ALGP_SOLVE_DAG select the paths).  python tools/fold_time.py [f64|f32] [M ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from algp_amd import _hip

dt = np.float32 if len(sys.argv) > 1 and sys.argv[1] == 'f32' else np.float64
Ms = [int(a) for a in sys.argv[2:]] or [12500, 25000, 5000]
rng = np.random.RandomState(1)
grid, field = bench.mog_field(100, 100, rng)
N = len(grid)
var = np.where(rng.uniform(size=N) < 0.5, 0.01, 1.0)
y = np.maximum(field + rng.standard_normal(N) * np.sqrt(var), 0.0)
for M in Ms:
    cand = bench.candidate_lattice(M, 100, 100, 0)
    c = _hip.Context(dt)
    c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
    c.set_pool(np.vstack([grid, cand]))
    c.set_train(np.arange(N), y, var)
    c.set_candidates(np.arange(N, N + M), prior_includes_noise=True)
    out = {}
    for name, fn in (('fit_and_solve', lambda: c.fit_and_solve()), ('factorize+solve', lambda: (c.factorize(), c.solve_candidates()))):
        fn()
        c.sync()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            fn()
            c.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        c.prof_enable(True)
        c.prof_reset()
        fn()
        pr = {k: c.prof_get(k) for k in ('cholesky', 'trsm', 'dag_panel', 'chol_dag', 'kmat', 'rows', 'trsv')}
        c.prof_enable(False)
        mu, pv = c.posterior()
        out[name] = (float(np.median(ts)), {k: round(v['ms'], 3) for k, v in pr.items()}, mu, pv)
    a, b = out['fit_and_solve'], out['factorize+solve']
    print('%s M=%d  fit_and_solve %.2f ms %s | factorize+solve %.2f ms %s | max |dmu| %.2e |dvar| %.2e' %
          (np.dtype(dt).name, M, a[0], a[1], b[0], b[1], float(np.max(np.abs(a[2] - b[2]))), float(np.max(np.abs(a[3] - b[3])))), flush=True)
    c.close()
