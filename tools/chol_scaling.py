"""fp64 / fp32 Cholesky throughput of algp_factorize vs N (HIP-event span around the factorisation)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algp_amd import _hip

out = {}
for dt, name in ((np.float64, 'f64'), (np.float32, 'f32')):
    for R, C in ((50, 40), (100, 100), (160, 125), (250, 200)):
        rng = np.random.RandomState(1)
        xx, yy = np.meshgrid(np.arange(C), np.arange(R))
        X = np.vstack([yy.ravel(), xx.ravel()]).T.astype(np.float64)
        n = len(X)
        c = _hip.Context(dt)
        c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
        c.set_pool(X)
        c.set_train(np.arange(n), rng.uniform(0, 1, n), rng.choice([0.01, 1.0], n))
        c.factorize()
        c.prof_enable(True)
        c.prof_reset()
        reps = 3
        for _ in range(reps):
            c.factorize()
        p = c.prof_get('cholesky')
        g = c.prof_get('gemm_chol')
        d = c.prof_get('potrf_diag')
        ms = p['ms'] / reps
        out['%s N=%d' % (name, n)] = dict(ms=ms, tflops=n ** 3 / 3.0 / (ms * 1e-3) / 1e12, gemm_ms=g['ms'] / reps,
                                          gemm_tflops=g['flops'] / (g['ms'] * 1e-3) / 1e12, diag_ms=d['ms'] / reps)
        c.close()
print(json.dumps(out, indent=1))
