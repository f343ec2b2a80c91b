"""Dependency-driven Cholesky (chol_dag.hip) vs the launch sequence and vs SciPy: residual, log-determinant,
bitwise repeatability (twice in one context, and against a second context), HIP-event time.
  python tools/chol_dag_check.py [f64|f32] [N ...]        ALGP_CHOL_DAG=0 selects the launch sequence."""
import json
import os
import sys
import time

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algp_amd import _hip

dt = np.float32 if (len(sys.argv) > 1 and sys.argv[1] == 'f32') else np.float64
sizes = [int(a) for a in sys.argv[2:]] or [1000, 2000, 5000, 10000]
out = {}
for n in sizes:
    rng = np.random.RandomState(1)
    C_ = int(np.ceil(np.sqrt(n)))
    R = (n + C_ - 1) // C_
    xx, yy = np.meshgrid(np.arange(C_), np.arange(R))
    X = np.vstack([yy.ravel(), xx.ravel()]).T.astype(np.float64)[:n]
    var = rng.choice([0.01, 1.0], n)
    y = rng.uniform(0, 1, n)

    def make():
        c = _hip.Context(dt)
        c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
        c.set_pool(X)
        c.set_train(np.arange(n), y, var)
        return c

    c = make()
    t0 = time.time()
    c.factorize()
    first_ms = (time.time() - t0) * 1e3
    L1 = c.factor()
    ld1 = c.logdet()
    c.prof_enable(True)
    c.prof_reset()
    reps = 5
    for _ in range(reps):
        c.factorize()
    p = c.prof_get('cholesky')
    ms = p['ms'] / reps
    L2 = c.factor()
    same_ctx = bool(np.array_equal(np.tril(L1), np.tril(L2)))
    c2 = make()
    c2.factorize()
    L3 = c2.factor()
    same_other = bool(np.array_equal(np.tril(L1), np.tril(L3)))
    c2.close()
    rec = dict(ms=ms, tflops=n ** 3 / 3.0 / (ms * 1e-3) / 1e12, first_call_ms=first_ms, bitwise_same_ctx=same_ctx,
               bitwise_other_ctx=same_other)
    if n <= 12000:
        S = c.kernel_matrix(X, None, var, True).astype(np.float64)
        Lr = sla.cholesky(S, lower=True)
        L = np.tril(L1).astype(np.float64)
        rec['max_rel_L_err'] = float(np.max(np.abs(L - Lr)) / np.max(np.abs(Lr)))
        rec['logdet_err'] = float(abs(ld1 - 2 * np.sum(np.log(np.diag(Lr)))))
        rec['resid'] = float(np.max(np.abs(L @ L.T - S)) / np.max(np.abs(S)))
    c.close()
    out['N=%d' % n] = rec
    print('N=%d' % n, json.dumps(rec), flush=True)
print(json.dumps(out))
