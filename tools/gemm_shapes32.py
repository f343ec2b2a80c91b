"""fp32 only: this library's SGEMM on the solve's shapes (A/B of kernel variants through $ALGP_LIB)."""
import sys
import numpy as np
from algp_amd import _hip
c = _hip.Context(np.float32)
for (m, n, k) in [(33408, 512, 5120), (100096, 512, 5120), (100096, 512, 9728), (4096, 4096, 4096), (8192, 8192, 8192)]:
    ms = c.bench_gemm(m, n, k, beta_one=True, reps=5)
    tf = 2.0 * m * n * k / ms / 1e9
    print('algp sgemm  m %6d n %5d k %5d: %8.3f ms  %6.1f TFLOP/s = %5.1f %% of 157.3' % (m, n, k, ms, tf, 100 * tf / 157.3))
    sys.stdout.flush()
c.close()
