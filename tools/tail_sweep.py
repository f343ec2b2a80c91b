"""tail_cols_kernel at config 5's size (N0 = 50 000 train rows x M candidates, fp64) against the number of new columns:
appends of 10 / 30 / 40 / 60 rows give 16 / 32 / 48 / 64 new columns = 1..4 MFMA tiles of L rows per k-step, over the
same M x N0 x 8 bytes of V^T -- separates the kernel's HBM side (flat in the width) from its MFMA side (linear in it).
$TAIL_M (default 100000) candidates; prints one JSON line."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from algp_amd import _hip

rng = np.random.RandomState(5)
M = int(os.environ.get('TAIL_M', '100000'))
grid, field, pool = bench._c5_field(rng, M=M)
N0 = len(grid)
c = _hip.Context(np.float64)
c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
c.set_pool(pool)
cidx = np.arange(N0, N0 + M)
var = np.where(rng.uniform(size=N0 + 64) < 0.5, 0.01, 1.0)
y = rng.uniform(0, 1, N0 + 64)
out = {'M': M, 'N0': N0, 'by_append': {}}
base = np.arange(N0)
extra = cidx[rng.permutation(M)[:64]]
for add in (0, 10, 30, 40, 60, 10, 30, 40, 60):
    idx = np.r_[base, extra[:add]]
    c.set_train(idx, y[:len(idx)], var[:len(idx)])
    c.factorize(incremental=True)
    c.set_candidates(cidx, prior_includes_noise=True)
    c.prof_enable(True)
    c.prof_reset()
    kept = c.solve_candidates(incremental=True)
    c.sync()
    t = c.prof_get('tail_cols')
    c.prof_enable(False)
    if add:
        out['by_append'].setdefault(str(add), []).append({'kept': int(kept), 'launches': t['launches'], 'ms': round(t['ms'], 3),
                                                         'GBs': round(t['bytes'] / (t['ms'] * 1e-3) / 1e9, 1) if t['ms'] else None})
    # back to the base: the appended columns are dropped again
    c.set_train(base, y[:N0], var[:N0])
    c.factorize(incremental=True)
    c.set_candidates(cidx, prior_includes_noise=True)
    c.solve_candidates(incremental=True)
print(json.dumps(out))
