"""One factorize + two solve_candidates at N = 10 000 train rows and M candidates (argv[1], default 12 500): for
rocprofv3 --kernel-trace of the mid-sized ("push") order of the candidate solve."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algp_amd import _hip

N, M = 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 12500
rng = np.random.RandomState(1)
xx, yy = np.meshgrid(np.arange(100), np.arange(100))
Xa = np.vstack([yy.ravel(), xx.ravel()]).T.astype(np.float64)
Xc = rng.uniform(0, 100, (M, 2))
c = _hip.Context(np.float64)
c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
c.set_pool(np.vstack([Xa, Xc]))
c.set_train(np.arange(N), rng.uniform(0, 1, N), rng.choice([0.01, 1.0], N))
c.set_candidates(np.arange(N, N + M), prior_includes_noise=True)
c.factorize()
for _ in range(2):
    c.solve_candidates()
c.sync()
c.close()
