"""One rank (first of 8) of BASELINE config 4's strong-scaling step through the product path -- algp_fit_and_solve on the
rank's 12 500 candidates + algp_greedy_sharded over algp_comm_init_host with the absent ranks fabricated -- a few times in
a row: for a rocprofv3 --kernel-trace of exactly that step (tools/rank_step_trace.sh), or plain timing with phases.
Same fabrication as bench.py's strong_emulation, raw callback (no copies through Python objects)."""
import ctypes
import json
import os
import struct
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from algp_amd import _hip
from algp_amd.sharded import partition


class A(object):
    train, cand, scaling, picks = 10000, 100000, 'strong', 4


n, r, reps = 8, int(os.environ.get('RANK_OF', '0')), int(os.environ.get('REPS', '4'))
w = bench.build_workload(A, 1)
N0, total = w['N'], A.cand
ctx = _hip.Context(np.float64)
ctx.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
ctx.set_pool(w['pool'])
ctx.set_train(np.arange(N0), w['y'], w['var'])
allc = np.arange(N0, N0 + total)
ctx.set_candidates(allc, prior_includes_noise=True)
ctx.fit_and_solve()
picks, ut = ctx.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4, want_utilities=True)
picks = [int(p) for p in picks]
util = [float(np.nanmax(ut[q])) for q in range(4)]
rows = [ctx.debug_get_pick(q) for q in range(4)]
parts = partition(total, n)
owners = [next(s for s, (a, b) in enumerate(parts) if a <= (p - N0) < b) for p in picks]
lo, hi = parts[r]
state = {'q': 0}
absent = struct.pack('<4d', float('-inf'), -1.0, 0.0, 0.0)


def fn(send, recv, nbytes):
    q = min(state['q'], 3)
    for k in range(n):
        if k != r:
            ctypes.memmove(recv + nbytes * k, absent, 32)
    if owners[q] != r:
        o = recv + nbytes * owners[q]
        row, d = rows[q]
        ctypes.memmove(o, struct.pack('<4d', util[q], float(picks[q]), 0.0, d), 32)
        ctypes.memmove(o + 32, row.ctypes.data, row.nbytes)
    ctypes.memmove(recv + nbytes * r, send, nbytes)
    if struct.unpack('<d', ctypes.string_at(send + 16, 8))[0] == 0.0:
        state['q'] += 1
    return 0


ctx.set_candidates(allc[lo:hi], prior_includes_noise=True)
ctx.comm_init_host(n, r, fn, raw=True)
out = []
for rep in range(reps + 1):
    state['q'] = 0
    ctx.sync()
    t0 = time.perf_counter()
    ctx.fit_and_solve()
    t1 = time.perf_counter()
    got = [int(p) for p in ctx.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 4)]
    ctx.sync()
    t2 = time.perf_counter()
    assert got == picks
    out.append((round((t1 - t0) * 1e3, 3), round((t2 - t1) * 1e3, 3)))
print(json.dumps({'rank': r, 'of': n, 'fit_and_solve_ms__picks_ms': out}))
