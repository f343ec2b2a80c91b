"""Command-line flags of the reference (reference arguments.py:7-52), same names and defaults.
Differences: `--kernel` accepts {rbf, matern} only; `--dtype`/`--device` select the GPU precision
and card; no results directory is created and nothing prompts (the reference's interactive
`input()` at arguments.py:41-49 blocks unattended runs)."""
import argparse

import numpy as np


def get_parser():
    p = argparse.ArgumentParser(description='Adaptive Sampling and Informative Planning (MI355X GP path)')
    p.add_argument('--lr', default=.1, type=float)
    p.add_argument('--max_iterations', default=200, type=int)
    p.add_argument('--data_file', default=None)
    p.add_argument('--phenotype', default='plant_height')
    p.add_argument('--kernel', default='matern', choices=['rbf', 'matern'])
    p.add_argument('--latent', default=None)
    p.add_argument('--num_sims', default=10, type=int)
    p.add_argument('--num_runs', default=6, type=int)
    p.add_argument('--fraction_pretrain', default=.75, type=float)
    p.add_argument('--num_samples_per_batch', default=4, type=int)
    p.add_argument('--slack', default=0, type=int)
    p.add_argument('--num_test', default=40, type=int)
    p.add_argument('--update', action='store_true')
    p.add_argument('--update_every', default=1, type=int)
    p.add_argument('--criterion', default='entropy', choices=['entropy', 'mutual_information'])
    p.add_argument('--static_std', default=.1, type=float)
    p.add_argument('--render', action='store_true')
    p.add_argument('--seed', default=1, type=int)
    p.add_argument('--id', default=1, type=int)
    p.add_argument('--save_dir', default='results')
    p.add_argument('--eval_only', action='store_true')
    p.add_argument('--dtype', default='float64', choices=['float32', 'float64'])
    p.add_argument('--device', default=0, type=int)
    p.add_argument('--rows', default=20, type=int, help='synthetic field rows')
    p.add_argument('--cols', default=20, type=int, help='synthetic field columns')
    return p


def get_args(argv=None):
    args = get_parser().parse_args(argv)
    args.dtype = np.dtype(args.dtype)
    return args
