#!/usr/bin/env python
"""Demo driver for the GP path (BASELINE config 1: `python run.py --eval_only`, a 20 x 20 synthetic
mixture-of-Gaussians field).  It mirrors the reference's run_demo flow (reference run.py:207-219)
minus the networkx path planner: pre-train the GP on a pilot survey, then per batch greedily pick
the most informative static sites, sample them, (optionally) refit, and predict the held-out set.
Needs an MI355X: there is no CPU path."""
import time

import numpy as np

from algp_amd.agent import Agent
from algp_amd.arguments import get_args
from algp_amd.field import SyntheticField
from algp_amd.utils import compute_mae


def run_demo(args):
    np.random.seed(args.seed)
    env = SyntheticField(args.rows, args.cols, num_test=args.num_test)
    agent = Agent(env, args, static_std=args.static_std, mobile_std=10 * args.static_std)
    agent.reset()
    agent._setup_ipp(args.criterion, args.update)
    errors = []
    for i in range(args.num_runs):
        t0 = time.time()
        picks = agent.greedy(args.num_samples_per_batch)
        agent._add_samples(picks, [agent.static_std] * len(picks))
        if args.update and (i + 1) % args.update_every == 0:
            agent.update_model()
            agent._post_update()
        pred, var = agent.predict(return_var=True)
        err = compute_mae(env.test_Y, pred)
        errors.append(err)
        print('Run {}/{}: picks {} test ERROR {:.4f} predictive variance max {:.3f} min {:.3f} mean {:.3f} ({:.3f}s)'
              .format(i + 1, args.num_runs, picks, err, var.max(), var.min(), var.mean(), time.time() - t0))
    return errors


if __name__ == '__main__':
    run_demo(get_args())
