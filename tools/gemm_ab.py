"""A/B of GEMM kernel variants in one process (interleaved rounds, random operands).
usage: python tools/gemm_ab.py [f64|f32] [variants...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algp_amd import _hip

dt = np.float32 if (len(sys.argv) > 1 and sys.argv[1] == 'f32') else np.float64
variants = [int(v) for v in sys.argv[2:]] or [0, 1]
c = _hip.Context(dt)
shapes = [  # (m, n, k, lower_only, beta_one, label)
    (100096, 128, 128, 0, 0, 'trsm diag   K=128'),
    (100096, 128, 384, 0, 1, 'trsm inblk  K=384'),
    (100096, 512, 2048, 0, 1, 'trsm outer n=512 K=2048'),
    (100096, 512, 8192, 0, 1, 'trsm outer n=512 K=8192'),
    (9984, 128, 128, 0, 0, 'chol panel  m=9984'),
    (9984, 9984, 128, 1, 1, 'chol syrk   m=9984 K=128'),
    (4992, 4992, 128, 1, 1, 'chol syrk   m=4992 K=128'),
    (8192, 8192, 256, 1, 1, 'syrk m=8192 K=256'),
    (4096, 4096, 4096, 0, 1, 'square 4096^3'),
]
res = {}
for rnd in range(3):
    for (m, n, k, lo, b1, label) in shapes:
        for v in variants:
            ms = c.bench_gemm(m, n, k, variant=v, lower_only=lo, beta_one=b1, reps=3)
            tiles = (m // 128) * (m // 128 + 1) // 2 if lo else (m // 128) * (n // 128)
            tf = 2.0 * 128 * 128 * k * tiles / (ms * 1e-3) / 1e12
            res.setdefault((label, v), []).append((ms, tf))
print('%-28s' % 'shape', ''.join('   v%d: ms (TF) med/max ' % v for v in variants))
for (m, n, k, lo, b1, label) in shapes:
    line = '%-28s' % label
    for v in variants:
        r = res[(label, v)]
        tfs = sorted(t for _, t in r)
        line += '   %8.3f ms %6.1f / %6.1f TF' % (sorted(x for x, _ in r)[len(r) // 2], tfs[len(tfs) // 2], tfs[-1])
    print(line)
c.close()
