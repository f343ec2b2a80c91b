#!/bin/bash
python - <<PY
import numpy as np, sys
sys.path.insert(0, '.')
from algp_amd import _hip
c = _hip.Context(np.float64)
for v in (0, 5):
    out = []
    for (m, n, k) in ((4096, 4096, 4096), (33408, 512, 5120), (100096, 512, 5120), (8192, 128, 128), (2048, 512, 512)):
        ms = c.bench_gemm(m, n, k, variant=v, lower_only=False, beta_one=True, reps=5)
        out.append('%dx%dx%d %.1f TF' % (m, n, k, 2.0 * m * n * k / ms / 1e9))
    print('variant', v, ' | '.join(out))
c.close()
c = _hip.Context(np.float32)
for v in (0, 5):
    out = []
    for (m, n, k) in ((4096, 4096, 4096), (100096, 512, 5120)):
        ms = c.bench_gemm(m, n, k, variant=v, lower_only=False, beta_one=True, reps=5)
        out.append('%dx%dx%d %.1f TF' % (m, n, k, 2.0 * m * n * k / ms / 1e9))
    print('f32 variant', v, ' | '.join(out))
PY
for v in 0 5; do ALGP_GEMM_VARIANT=$v python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant $v bench', round(d['ms_per_step'],1), round(d['stage_ms_per_step']['trsm'],1), round(d['roofline']['achieved'],1), round(d['stage_ms_per_step']['cholesky'],2))"; done
