"""Where one hyper-parameter fit iteration (factorize + MLL + gradient) spends its time, per kernel class."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algp_amd import _hip

out = {}
for N, dt in ((1600, np.float64), (9000, np.float64), (9000, np.float32)):
    rng = np.random.RandomState(1)
    X = rng.uniform(0, 100, (N, 2))
    c = _hip.Context(dt)
    c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
    c.set_pool(X)
    c.set_train(np.arange(N), rng.uniform(0, 1, N), rng.choice([0.01, 1.0], N))
    c.factorize(); c.mll(); c.mll_grad()
    c.prof_enable(True)
    rec = {}
    for what in ('factorize', 'mll', 'mll_grad'):
        c.prof_reset()
        t0 = time.perf_counter()
        getattr(c, what)()
        c.sync()
        rec[what] = dict(wall_ms=(time.perf_counter() - t0) * 1e3,
                         classes={k: c.prof_get(k) for k in _hip.PROF if c.prof_get(k)['launches']})
    c.prof_enable(False)
    out['N=%d %s' % (N, np.dtype(dt).name)] = rec
    c.close()
print(json.dumps(out, indent=1))
