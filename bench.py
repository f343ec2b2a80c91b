#!/usr/bin/env python
"""bench.py -- GP-fit + candidate-scoring step of the algp hot path on MI355X.

One "step" = one planning step of the reference's agent at fixed hyper-parameters:
  GP-fit   : kernel-matrix build S = K_AA + D + sigma_n^2 I, Cholesky, z = L^-1(y-ybar), alpha
  MI-score : V^T = B^T L^-T for all candidates (blocked TRSM on MFMA), posterior mean/variance,
             k = 4 greedy picks (arguments.py:22): utilities -> all-gather -> argmax -> rank-1 commit
Workload (BASELINE.json configs[3] on ONE GPU; it fits: L 0.8 GB + V^T 8.2 GB fp64):
  N = 10 000 train points (100 x 100 mixture-of-Gaussians field, utils.py:90-108),
  M = 100 000 candidates PER GPU (weak scaling: rank r scores its own 100 000), D = 2, fp64,
  entropy criterion (the reference's effective default, agent.py:125), sigma_s = 0.1, sigma_m = 1.
Inputs (coordinates, targets, noise) are resident in HBM before the timed region.

Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 under torch.distributed.run.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# The candidate solve overlaps independent row chunks on 3 HIP streams.  ROCm multiplexes all streams of
# a process onto GPU_MAX_HW_QUEUES (default 4) hardware queues; once RCCL's streams exist the chunk streams
# share a queue and serialise (TRSM 161 -> 183 ms).  Must be set before the HIP runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP64_MATRIX_PEAK_TFLOPS = 78.6      # MI355X fp64 matrix (= vector) peak, SURVEY.md section 8(d)
FP32_MATRIX_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md chip-level parameters
HBM_PEAK_GBS = 8000.0


def mog_field(R, C, rng, k=5, min_var=10, max_var=100):
    """utils.py:90-108 (algo='sum'), seeded."""
    xx, yy = np.meshgrid(np.arange(C), np.arange(R))
    grid = np.vstack([yy.flatten(), xx.flatten()]).T.astype(np.float64)
    mx, my = rng.uniform(0, R, k), rng.uniform(0, C, k)
    var = rng.uniform(min_var, max_var, k)
    y = np.zeros(R * C)
    for i in range(k):
        y += np.exp(-((grid[:, 0] - mx[i]) ** 2 + (grid[:, 1] - my[i]) ** 2) / var[i])
    return grid, y


def build_workload(args, world):
    rng = np.random.RandomState(1)                       # arguments.py:33 default seed
    R = int(round(np.sqrt(args.train)))
    grid, field = mog_field(R, args.train // R, rng)
    N = len(grid)
    static_std, mobile_std = 0.1, 1.0
    # half the sampled sites carry static readings, half mobile ones (fusion rule agent.py:100-109)
    is_static = rng.uniform(size=N) < 0.5
    var = np.where(is_static, static_std ** 2, mobile_std ** 2)
    y = np.maximum(field + rng.standard_normal(N) * np.sqrt(var), 0.0)     # env.py:110-112
    # candidates: per rank an offset lattice over the same field, none coinciding with a train site
    per = args.cand
    cw = int(np.ceil(np.sqrt(per * (args.train // R) / R)))
    ch = int(np.ceil(per / cw))
    cands = []
    for r in range(world):
        ii, jj = np.meshgrid(np.arange(ch), np.arange(cw), indexing='ij')
        c = np.vstack([(ii.ravel() + 0.37 + 0.011 * r) * (R / ch), (jj.ravel() + 0.41 + 0.007 * r) * ((args.train // R) / cw)]).T
        cands.append(c[:per])
    pool = np.vstack([grid] + cands)
    return dict(pool=pool, N=N, y=y, var=var, per=per, static_std=static_std, mobile_std=mobile_std)


def cpu_baseline(w, hyp_vals, args):
    """Reference-faithful CPU path (oracle 'port') on a bounded sample of the same workload:
    full-size fit (fp32 kernel + np.linalg.inv, utils.py:296-300) on a sub-sampled train set when
    the full one would take minutes, plus `ncand` per-candidate slogdets (agent.py:328-329);
    extrapolated linearly in candidates x picks and cubically in N (stated in `sample`)."""
    from oracle import gp_oracle as O
    hyp = O.Hypers(np.log(hyp_vals['ls']), np.log(hyp_vals['os']), np.log(hyp_vals['noise']))
    N = w['N']
    Ns = min(N, args.cpu_train)
    sel = np.sort(np.random.RandomState(0).permutation(N)[:Ns])
    X = w['pool'][:N][sel]
    var = w['var'][sel]
    ncand = 2
    t0 = time.time()
    cov_aa = O.cov_mat_ref(hyp, X, None, var, True, dtype=np.float32)
    inv = np.linalg.inv(cov_aa)
    mat1 = inv @ (w['y'][sel] - w['y'][sel].mean()).astype(np.float32)
    t_fit = time.time() - t0
    t0 = time.time()
    xc = w['pool'][N:N + ncand]
    for i in range(ncand):
        Xa = np.vstack([X, xc[i:i + 1]])
        cov_a = O.cov_mat_ref(hyp, Xa, None, None, True, dtype=np.float32) + np.diag(np.r_[var, 0.01])
        O.entropy_from_cov_ref(cov_a)
    t_cand = (time.time() - t0) / ncand
    scale = (N / Ns) ** 3
    k = 4
    t_step = scale * (t_fit + k * w['per'] * t_cand)
    # efficient CPU form (Cholesky + identities), sampled candidates, for an honest second figure
    from scipy.linalg import solve_triangular
    t0 = time.time()
    S = O.kernel_matrix(hyp, X) + np.diag(var) + hyp.noise * np.eye(Ns)
    L = np.linalg.cholesky(S)
    ms = 2048
    B = O.kernel_matrix(hyp, X, w['pool'][N:N + ms])
    V = solve_triangular(L, B, lower=True)
    pv = hyp.outputscale + hyp.noise - np.sum(V * V, axis=0)
    assert np.all(pv > 0)
    t_eff = time.time() - t0
    # the O(N^2 M) triangular solve dominates at these sizes: scale by (N/Ns)^2 * (M/ms)
    t_eff_step = (N / Ns) ** 2 * (w['per'] / ms) * t_eff
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    return {
        'value': w['per'] / t_step, 'unit': 'candidates/s', 'cores': cores, 'kind': 'port',
        'sample': 'reference-faithful oracle (fp32 kernel, np.linalg.inv, one slogdet per candidate) at '
                  'N=%d of %d train, %d of %d candidates, 1 of %d picks; %.1fs measured; extrapolated x(N/Ns)^3, '
                  'linear in candidates x picks' % (Ns, N, ncand, w['per'], k, t_fit + ncand * t_cand),
        'ms_per_step_extrapolated': 1e3 * t_step,
        'efficient_cpu_candidates_per_s': w['per'] / t_eff_step,
        'efficient_cpu_sample': 'numpy/scipy Cholesky + triangular solve + variance, N=%d, %d candidates, %.1fs; '
                                'extrapolated x(N/Ns)^2 x M/%d' % (Ns, ms, t_eff, ms),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--train', type=int, default=10000)
    ap.add_argument('--cand', type=int, default=100000, help='candidates per GPU')
    ap.add_argument('--dtype', default='f64', choices=['f64', 'f32'])
    ap.add_argument('--picks', type=int, default=4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-dist', action='store_true', help='use the torch.distributed/RCCL path even with one rank')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='gloo only for rehearsing several ranks on ONE card (RCCL refuses duplicate devices)')
    ap.add_argument('--cpu-train', type=int, default=8000)
    ap.add_argument('--traffic-json', default=os.path.join(REPO, 'profiles', 'traffic.json'))
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line: libraries that write to fd 1 (RCCL prints a version banner
    # there when the communicator is created) are sent to stderr until the result is printed.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'ALGP_BENCH_DEVICE' in os.environ:          # rehearsal: several ranks on one card
        local_rank = int(os.environ['ALGP_BENCH_DEVICE'])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ...'
                             % (args.gpus, args.gpus))
    import torch
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo')

    from algp_amd import _hip
    from algp_amd.sharded import LocalComm, ShardedGreedy, TorchComm

    dt = np.float64 if args.dtype == 'f64' else np.float32
    w = build_workload(args, world)
    hyp_vals = dict(ls=[3.0, 3.0], os=1.0, noise=1e-2)
    ctx = _hip.Context(dt, device=local_rank)
    ctx.set_hypers(np.log(hyp_vals['ls']), np.log(hyp_vals['os']), np.log(hyp_vals['noise']))
    ctx.set_pool(w['pool'])
    N, per = w['N'], w['per']
    ctx.set_train(np.arange(N), w['y'], w['var'])
    all_cand = np.arange(N, N + per * world)
    mine = all_cand[rank * per:(rank + 1) * per]
    ctx.set_candidates(mine, prior_includes_noise=True)
    comm = TorchComm(torch.device('cuda', local_rank)) if dist is not None else LocalComm()

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    picks_log = []

    def step():
        ctx.fit_and_solve()                                       # = algp_factorize + algp_solve_candidates
        if dist is None:
            picks = list(ctx.greedy(_hip.CRIT_ENTROPY, w['static_std'], w['mobile_std'], args.picks))
        else:
            sg = ShardedGreedy(ctx, comm, all_cand)
            picks, _ = sg.greedy(_hip.CRIT_ENTROPY, w['static_std'], w['mobile_std'], args.picks)
        picks_log.append([int(p) for p in picks])

    for _ in range(args.warmup):
        step()
    ctx.prof_enable(True)
    ctx.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = {k: ctx.prof_get(k) for k in _hip.PROF}
    ctx.prof_enable(False)

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        total_c = per * world
        peak = FP64_MATRIX_PEAK_TFLOPS if args.dtype == 'f64' else FP32_MATRIX_PEAK_TFLOPS
        g = prof['gemm_trsm']
        span = prof['trsm']                                       # wall time of the solves (two overlapped streams)
        ach = g['flops'] / (span['ms'] * 1e-3) / 1e12 if span['ms'] > 0 else 0.0
        traffic = None
        if os.path.exists(args.traffic_json):
            try:
                traffic = json.load(open(args.traffic_json)).get('gemm_nt_%s_bytes_per_launch' % args.dtype)
            except Exception:
                traffic = None
        gc = prof['gemm_chol']
        gu = prof['gemm_chol_update']
        gu_tf = gu['flops'] / (gu['ms'] * 1e-3) / 1e12 if gu['ms'] > 0 else 0.0
        chol_ms = prof['cholesky']['ms'] / args.steps            # wall time of the factorisation (two overlapped streams)
        chol_tf = (N ** 3 / 3.0) / (chol_ms * 1e-3) / 1e12 if chol_ms > 0 else 0.0
        out = {
            'metric': 'GP-fit+MI-score throughput (N train x M candidates)',
            'value': total_c / (elapsed / args.steps),
            'unit': 'candidates/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': '%d-point MoG field (train) x %d candidates/GPU, %d greedy picks, entropy criterion, D=2'
                                   % (N, per, args.picks),
                       'n_train': N, 'candidates_per_gpu': per, 'candidates_total': total_c,
                       'parallelism': 'candidate shards x%d, one all-gather per pick (each rank\'s best utility + position)' % world},
            'roofline': {'bound': 'mfma', 'kernel': 'gemm_nt_kernel_dma4<%s> (candidate TRSM)' % ('double' if args.dtype == 'f64' else 'float'),
                         'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
                         'traffic': traffic, 'launches': g['launches'], 'avg_launch_ms': g['ms'] / max(1, g['launches']),
                         'wall_ms_all_launches': span['ms'], 'sum_launch_ms': g['ms'],
                         'note': 'launches of the two row halves overlap on two streams: achieved = flops / wall time of the solves',
                         'flops_per_launch': g['flops'] / max(1, g['launches']),
                         # the same kernel symbol also serves the factorisation: this is the figure rocprofv3 --stats
                         # reports for the symbol (all its launches), for the cross-check against profiles/
                         'symbol_launches': g['launches'] + gc['launches'] + gu['launches'],
                         'symbol_avg_launch_ms': (g['ms'] + gc['ms'] + gu['ms']) / max(1, g['launches'] + gc['launches'] + gu['launches'])},
            'cholesky_tflops': chol_tf, 'cholesky_ms': chol_ms,
            'cholesky_gemm_tflops': (gc['flops'] + gu['flops']) / ((gc['ms'] + gu['ms']) * 1e-3) / 1e12 if gc['ms'] + gu['ms'] > 0 else 0.0,
            # the factorisation's dense rank-512 trailing ("panel") updates on MFMA, HIP-event time of those launches
            'cholesky_panel_update': {'achieved': gu_tf, 'peak': peak, 'unit': 'TFLOP/s', 'frac': gu_tf / peak,
                                      'launches': gu['launches'], 'ms_per_step': gu['ms'] / args.steps},
            'stage_ms_per_step': {k: v['ms'] / args.steps for k, v in prof.items()},
            'picks_last_step': picks_log[-1],
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(w, hyp_vals, args)
        else:
            out['cpu_baseline'] = None
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out))
        sys.stdout.flush()
        os.dup2(2, 1)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
