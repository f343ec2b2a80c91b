"""End-to-end time of the Python Agent planning loop (greedy picks + sampling + predict) per step on a
R x C synthetic field -- finds host-side overhead around the device calls."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algp_amd.agent import Agent
from algp_amd.arguments import get_args
from algp_amd.field import SyntheticField

R, C = (int(v) for v in os.environ.get('FIELD', '100x100').split('x'))
steps = int(os.environ.get('STEPS', '8'))
np.random.seed(3)
env = SyntheticField(R, C, num_test=1000)
args = get_args(['--eval_only', '--kernel', 'rbf', '--max_iterations', '0', '--fraction_pretrain', '0.3'])
args.incremental = True
ag = Agent(env, args)
ag._setup_ipp('entropy')
# start from a third of the field sampled by the mobile sensor
first = [int(i) for i in np.random.permutation(env.num_samples)[:env.num_samples // 3]]
ag._add_samples(first, [ag.mobile_std] * len(first))
times = []
prof = cProfile.Profile()
for step in range(steps):
    t0 = time.perf_counter()
    if step == steps - 1:
        prof.enable()
    picks = ag.greedy(4)
    ag._add_samples(picks, [ag.static_std] * 4)
    mob = [int(i) for i in np.random.permutation(env.num_samples)[:28]]
    ag._add_samples(mob, [ag.mobile_std] * 28)
    mu, var = ag.predict(return_var=True)
    if step == steps - 1:
        prof.disable()
    times.append((time.perf_counter() - t0) * 1e3)
print('field %dx%d: ms per step' % (R, C), [round(t, 1) for t in times])
pstats.Stats(prof).sort_stats('cumulative').print_stats(18)
