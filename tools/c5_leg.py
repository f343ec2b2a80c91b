"""Runs bench.py's config-5 leg on its own (extra.c5_fp64_50k: the 200-step loop on one GPU, the tail kernel's roofline, one
rank of eight) and prints its JSON -- for profiling and for iterating on the loop without the headline run around it.
$C5_STEPS / $C5_EMU_STEPS shorten it; $C5_ONLY=loop|rank8 runs one half."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from algp_amd import _hip

steps = int(os.environ.get('C5_STEPS', '200'))
emu = int(os.environ.get('C5_EMU_STEPS', '40'))
only = os.environ.get('C5_ONLY', '')
if only == 'rank8':
    out = bench.c5_rank_of_8(_hip, 0, 4, emu)
else:
    if only == 'loop':
        bench.c5_rank_of_8 = lambda *a, **k: {'ms_per_step': float('nan'), 'skipped': True}
    out = bench.extra_c5(_hip, 0, 4, steps, emu)
print(json.dumps(out))
