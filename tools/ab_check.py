"""Correctness of an A/B library build ($ALGP_LIB) on GEMM shapes its variant kernels take (m >= 2048): D = alpha A B^T + beta C
against NumPy, fp64 and fp32, K with every remainder of a 96-byte tile, a row count that is an odd multiple of 128."""
import sys

import numpy as np

from algp_amd import _hip

bad = 0
for dt, tol in ((np.float64, 1e-12), (np.float32, 2e-5)):
    c = _hip.Context(dt)
    rng = np.random.RandomState(3)
    for (m, n, k) in ((2304, 512, 1024), (2176, 640, 1152), (4224, 128, 128), (2048, 256, 1280), (3200, 384, 384)):
        A = rng.uniform(-1, 1, (m, k)).astype(dt)
        B = rng.uniform(-1, 1, (n, k)).astype(dt)
        Cm = rng.uniform(-1, 1, (m, n)).astype(dt)
        want = -1.5 * A.astype(np.float64) @ B.astype(np.float64).T + 0.5 * Cm
        D = c.gemm_nt(A, B, alpha=-1.5, beta=0.5, Cm=Cm)
        e1 = np.max(np.abs(D - want)) / k
        D0 = c.gemm_nt(A, B)
        e0 = np.max(np.abs(D0 - A.astype(np.float64) @ B.astype(np.float64).T)) / k
        ok = e1 <= tol and e0 <= tol
        bad += not ok
        print(np.dtype(dt).name, (m, n, k), 'err/k', e1, e0, 'ok' if ok else 'FAIL')
    c.close()
sys.exit(1 if bad else 0)
