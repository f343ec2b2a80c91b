import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from algp_amd import _hip
rng = np.random.RandomState(1)
xx, yy = np.meshgrid(np.arange(100), np.arange(100))
X = np.vstack([yy.ravel(), xx.ravel()]).T.astype(np.float64)
T = rng.uniform(0, 100, (16384, 2))
c = _hip.Context(np.float64)
c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
c.set_pool(np.vstack([X, T]))
c.set_train(np.arange(10000), rng.uniform(0, 1, 10000), rng.choice([0.01, 1.0], 10000))
c.factorize()
for M in (40, 1000, 2048, 4096, 4097, 8192, 16384):
    c.set_candidates(np.arange(10000, 10000 + M), prior_includes_noise=False)
    ts = []
    for r in range(4):
        t0 = time.perf_counter(); c.solve_candidates(); c.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print(M, round(min(ts[1:]), 2), 'ms')
