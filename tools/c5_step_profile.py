"""Stage split of the active-learning loop's incremental step at BASELINE config 5's size (N0 = 50 000 train points,
100 000 candidates, fp64; each step appends the 4 picks + <= 28 mobile sites): wall time of the five calls of a step and the
library's HIP-event classes, for the steps after the from-scratch one.  Under `rocprofv3 --kernel-trace --stats` it gives the
per-kernel times of profiles/r04_c5_incremental_kernel_stats.csv (tail_cols_kernel: the new columns of V^T).
  python tools/c5_step_profile.py [steps]        $ALGP_TAIL_COLS=0: the 128-column blocks of round 3"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from algp_amd import _hip

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.RandomState(5)
R, C = 250, 200
grid, field = bench.mog_field(R, C, rng)
N0, M = len(grid), 100000
cand = bench.candidate_lattice(M, R, C, 2) + 0.03 * rng.standard_normal((M, 2))
pool = np.vstack([grid, cand])
c = _hip.Context(np.float64)
c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
c.set_pool(pool)
idx = np.arange(N0)
var = np.where(rng.uniform(size=N0) < 0.5, 0.01, 1.0)
y = np.maximum(field + rng.standard_normal(N0) * np.sqrt(var), 0.0)
static = np.zeros(len(pool), bool)
static[:N0] = var == 0.01
cidx = np.arange(N0, N0 + M)
for s in range(steps):
    c.prof_enable(s >= 2)
    c.prof_reset()
    t = [time.perf_counter()]
    c.set_train(idx, y, var)
    t.append(time.perf_counter())
    c.factorize(incremental=True)
    t.append(time.perf_counter())
    c.set_candidates(cidx, prior_includes_noise=True)
    t.append(time.perf_counter())
    c.solve_candidates(incremental=True, alive=~static[cidx])
    t.append(time.perf_counter())
    pk = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4)
    c.sync()
    t.append(time.perf_counter())
    d = [round((b - a) * 1e3, 2) for a, b in zip(t[:-1], t[1:])]
    pr = {k: (round(c.prof_get(k)['ms'], 2), c.prof_get(k)['launches']) for k in _hip.PROF if c.prof_get(k)['launches']} if s >= 2 else {}
    print('step %d, %d train rows: set_train / factorize / set_candidates / solve / 4 picks = %s ms, total %.2f  %s'
          % (s, len(idx), d, sum(d), pr), flush=True)
    static[pk] = True
    mob = cidx[rng.permutation(M)[:28]]
    mob = mob[~np.isin(mob, idx) & ~np.isin(mob, pk)]
    idx = np.r_[idx, pk, mob]
    var = np.r_[var, np.full(len(pk), 0.01), np.full(len(mob), 1.0)]
    y = np.r_[y, rng.uniform(0, 1, len(pk) + len(mob))]
c.close()
