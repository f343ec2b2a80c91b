"""One iteration of GPR.fit (models.py:145-158) on the device at N = 10 000: algp_fit_step (factor + L^-T in one task-list
launch, S^-1 = X X^T as one launch) against algp_factorize + algp_get_mll + algp_get_mll_grad (launch sequence for L^-T),
with the stage split from the library's HIP events.  python tools/fit_time.py [N]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from algp_amd import _hip

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for dt in (np.float64, np.float32):
    rng = np.random.RandomState(1)
    R = int(round(np.sqrt(N)))
    grid, field = bench.mog_field(R, N // R, rng)
    n = len(grid)
    c = _hip.Context(dt)
    c.set_pool_hyp = None
    c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
    c.set_pool(grid)
    c.set_train(np.arange(n), field + 0.1 * rng.standard_normal(n), np.full(n, 0.01))
    for name, fn in (('fit_step', lambda: c.fit_step()), ('factorize+mll+mll_grad', lambda: (c.factorize(), c.mll(), c.mll_grad()))):
        fn()
        ts = []
        for it in range(5):
            c.set_hypers(np.log([3.0 + 0.01 * it, 3.0]), 0.0, np.log(1e-2))      # an Adam step changes the hyper-parameters
            c.sync()
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        c.prof_enable(True)
        c.prof_reset()
        fn()
        pr = {k: (round(c.prof_get(k)['ms'], 3), c.prof_get(k)['launches']) for k in _hip.PROF if c.prof_get(k)['launches']}
        c.prof_enable(False)
        ms = float(np.median(ts))
        print('%s N=%d %s: %.2f ms per iteration = %.1f TFLOP/s of N^3  %s' % (np.dtype(dt).name, n, name, ms, float(n) ** 3 / (ms * 1e-3) / 1e12, pr), flush=True)
    c.close()
