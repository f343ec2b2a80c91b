"""One algp_get_mll_grad at N=9000 fp64 (for rocprofv3 --kernel-trace)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algp_amd import _hip

N = int(os.environ.get('GRAD_N', '9000'))
rng = np.random.RandomState(1)
c = _hip.Context(np.float64)
c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
c.set_pool(rng.uniform(0, 100, (N, 2)))
c.set_train(np.arange(N), rng.uniform(0, 1, N), rng.choice([0.01, 1.0], N))
c.factorize()
c.mll_grad()
c.sync()
print('MARK')
c.mll_grad()
c.sync()
c.close()
