"""This library's GEMM on the shapes tools/rocblas_yardstick.cpp gives the vendor's (the same card, the same minute)."""
import sys

import numpy as np

from algp_amd import _hip

for dt, peak in ((np.float64, 78.6), (np.float32, 157.3)):
    c = _hip.Context(dt)
    shapes = [(33408, 512, 2048), (33408, 512, 5120), (33408, 512, 9728), (100096, 512, 5120), (100096, 512, 9728), (4096, 4096, 4096), (8192, 8192, 8192)]
    if dt == np.float32:
        shapes = [(33408, 512, 5120), (8192, 8192, 8192)]
    for (m, n, k) in shapes:
        ms = c.bench_gemm(m, n, k, beta_one=True, reps=5)
        tf = 2.0 * m * n * k / ms / 1e9
        print('algp %s  m %6d n %5d k %5d: %8.3f ms  %6.1f TFLOP/s = %5.1f %% of %.1f' % ('dgemm' if dt == np.float64 else 'sgemm', m, n, k, ms, tf, 100 * tf / peak, peak))
        sys.stdout.flush()
    c.close()
