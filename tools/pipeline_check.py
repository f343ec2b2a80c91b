"""ALGP_PIPELINE=1 python tools/pipeline_check.py : the overlapped fit_and_solve variant gives the same
bits as the serial composition (run in its own process: the switch is read once per process)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algp_amd import _hip

rng = np.random.RandomState(21)
N, M = 1500, 9000
X = rng.uniform(0, 60, (N + M, 2))
y = np.sin(X[:N, 0] / 5) + 0.1 * rng.standard_normal(N)
var = rng.choice([0.01, 1.0], N)
c = _hip.Context(np.float64)
c.set_hypers(np.log([3.0, 2.0]), 0.0, np.log(2e-2))
c.set_pool(X)
c.set_train(np.arange(N), y, var)
c.set_candidates(np.arange(N, N + M), prior_includes_noise=True)
c.factorize()
c.solve_candidates()
ref = (c.posterior(), c.logdet(), c.alpha())
for _ in range(3):
    c.fit_and_solve()
    got = (c.posterior(), c.logdet(), c.alpha())
    assert np.array_equal(ref[0][0], got[0][0]) and np.array_equal(ref[0][1], got[0][1])
    assert ref[1] == got[1] and np.array_equal(ref[2], got[2])
print('pipeline=%s: identical' % os.environ.get('ALGP_PIPELINE', '0'))
