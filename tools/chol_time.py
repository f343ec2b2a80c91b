"""Kernel time of the one-launch Cholesky (chol_dag_kernel, HIP events around the launch) at N = 10 000, fp64 and fp32:
per-launch times of 12 factorisations after 2 warm-ups (min / median / max), for A/B runs of kernel changes.
  python tools/chol_time.py [N]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algp_amd import _hip

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
R = int(round(np.sqrt(N)))
C = N // R
out = {}
for dt, name in ((np.float64, 'f64'), (np.float32, 'f32')):
    rng = np.random.RandomState(1)
    xx, yy = np.meshgrid(np.arange(C), np.arange(R))
    X = np.vstack([yy.ravel(), xx.ravel()]).T.astype(np.float64)
    n = len(X)
    c = _hip.Context(dt)
    c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
    c.set_pool(X)
    c.set_train(np.arange(n), rng.uniform(0, 1, n), rng.choice([0.01, 1.0], n))
    c.factorize()
    c.factorize()
    ms = []
    c.prof_enable(True)
    for _ in range(12):
        c.prof_reset()
        c.factorize()
        ms.append(c.prof_get('chol_dag')['ms'])
    c.close()
    ms = np.array(ms)
    peak = 78.6 if name == 'f64' else 157.3
    tf = n ** 3 / 3.0 / (np.median(ms) * 1e-3) / 1e12
    out[name] = dict(n=n, min_ms=float(ms.min()), median_ms=float(np.median(ms)), max_ms=float(ms.max()), tflops_median=tf,
                     frac_of_peak=tf / peak)
print(json.dumps(out))
