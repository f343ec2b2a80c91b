"""Latency of the GP path at the smaller BASELINE.json configs (parity-test cases, not bench lines):
C1 20x20 field (n=360), C2 2 000 points fp64, C3 10 000 points fp32 (and fp64 for comparison)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algp_amd import _hip


def run(name, R, C, dt, n_test):
    rng = np.random.RandomState(1)
    xx, yy = np.meshgrid(np.arange(C), np.arange(R))
    X = np.vstack([yy.ravel(), xx.ravel()]).T.astype(np.float64)
    n = len(X)
    perm = rng.permutation(n)
    A, T = np.sort(perm[n_test:]), np.sort(perm[:n_test])
    y = rng.uniform(0, 1, len(A))
    var = rng.choice([0.01, 1.0], len(A))
    c = _hip.Context(dt)
    c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
    c.set_pool(X)
    c.set_train(A, y, var)
    c.set_candidates(T, prior_includes_noise=False)
    out = {}
    for what in ('factorize', 'posterior', 'greedy4'):
        ts = []
        for rep in range(6):
            if what == 'greedy4':
                c.set_candidates(T, prior_includes_noise=True)
                c.solve_candidates()
                c.sync()
            t0 = time.perf_counter()
            if what == 'factorize':
                c.factorize()
            elif what == 'posterior':
                c.solve_candidates()
                c.posterior()
            else:
                c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4)
            c.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        out[what + '_ms'] = float(np.median(ts[1:]))
        if what == 'greedy4':
            c.set_candidates(T, prior_includes_noise=False)
    ts = []
    for rep in range(5):                       # one hyper-parameter fit iteration (f2): factor + MLL + gradient
        t0 = time.perf_counter()
        c.factorize()
        c.mll()
        c.mll_grad()
        ts.append((time.perf_counter() - t0) * 1e3)
    out['fit_iteration_ms'] = float(np.median(ts[1:]))
    N = len(A)
    out['N'] = N
    out['M'] = len(T)
    out['cholesky_tflops_incl_kernel_build_and_solves'] = N ** 3 / 3.0 / (out['factorize_ms'] * 1e-3) / 1e12
    c.close()
    return name, out


if 'best_path' not in sys.argv[1:]:
    res = dict([run('C1 20x20 fp64', 20, 20, np.float64, 40), run('C2 50x40 fp64', 50, 40, np.float64, 400),
                run('C3 100x100 fp32', 100, 100, np.float32, 1000), run('C3 100x100 fp64', 100, 100, np.float64, 1000)])
    print(json.dumps(res, indent=1))


def best_path_time():
    """f3 'done' figure: Agent.best_path's block scoring of 1 000 paths of <= 32 sites at N = 10 000 train rows
    (algp_score_paths on a resident candidate solve over the whole pool)."""
    rng = np.random.RandomState(2)
    R, C = 125, 100
    xx, yy = np.meshgrid(np.arange(C), np.arange(R))
    X = np.vstack([yy.ravel(), xx.ravel()]).T.astype(np.float64)          # 12 500 pool sites
    n = len(X)
    perm = rng.permutation(n)
    A = np.sort(perm[:10000])
    var = rng.choice([0.01, 1.0], len(A))
    c = _hip.Context(np.float64)
    c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
    c.set_pool(X)
    c.set_train(A, np.zeros(len(A)), var)
    c.factorize()
    c.set_candidates(np.arange(n), prior_includes_noise=True)
    c.solve_candidates()
    rest = perm[10000:]
    sites = np.full((1000, 32), -1, dtype=np.int64)
    for p in range(1000):
        L = rng.randint(8, 33)
        sites[p, :L] = rng.permutation(rest)[:L]
    c.score_paths(sites, 1.0)
    ts = []
    for rep in range(5):
        c.sync()
        t0 = time.perf_counter()
        dH = c.score_paths(sites, 1.0)
        ts.append((time.perf_counter() - t0) * 1e3)
    out = {'best_path 1000 paths x <=32 sites, N=10000': dict(score_paths_ms=float(np.median(ts)), finite=bool(np.all(np.isfinite(dH))))}
    # paths as long as config 5's field rows: 1 000 paths x 200 distinct sites (the 65 .. 256-site form: rows gathered, one
    # batched MFMA product for the Gram matrices, 2 x 2 tiles of 128 per block); sites drawn from the unsampled 2 500 and,
    # for every second path, 20 train sites (second rows)
    sites = np.full((1000, 200), -1, dtype=np.int64)
    for p in range(1000):
        row = rng.permutation(rest)[:200]
        if p % 2:
            row[:20] = A[rng.permutation(len(A))[:20]]
        sites[p] = row
    c.score_paths(sites, 1.0)
    ts = []
    for rep in range(5):
        c.sync()
        t0 = time.perf_counter()
        dH = c.score_paths(sites, 1.0)
        ts.append((time.perf_counter() - t0) * 1e3)
    ms = float(np.median(ts))
    flops = 1000 * 3 * 2.0 * 128 * 128 * 10112            # the Gram products: 3 lower tiles of 128 x 128 x Npad per path
    out['best_path 1000 paths x 200 sites, N=10000'] = dict(score_paths_ms=ms, finite=bool(np.all(np.isfinite(dH))),
                                                            gram_tflops_over_whole_call=flops / (ms * 1e-3) / 1e12,
                                                            gathered_bytes=1000 * 256 * 10112 * 8)
    c.close()
    return out


if 'best_path' in sys.argv[1:]:
    print(json.dumps(best_path_time(), indent=1))
