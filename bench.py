#!/usr/bin/env python
"""bench.py -- GP-fit + candidate-scoring step of the algp hot path on MI355X.

One "step" = one planning step of the reference's agent at fixed hyper-parameters:
  GP-fit   : kernel-matrix build S = K_AA + D + sigma_n^2 I, Cholesky, z = L^-1(y-ybar), alpha
  MI-score : V^T = B^T L^-T for all candidates (blocked TRSM on MFMA), posterior mean/variance,
             k = 4 greedy picks (arguments.py:22): utilities -> all-gather -> argmax -> rank-1 commit
Workload (BASELINE.json configs[3] on ONE GPU; it fits: L 0.8 GB + V^T 8.2 GB fp64):
  N = 10 000 train points (100 x 100 mixture-of-Gaussians field, utils.py:90-108), D = 2, fp64,
  M = 100 000 candidates IN TOTAL (--scaling strong, the headline: BASELINE config 4 -- the 100 000 are split over
  the ranks, agent.py:317-347 being independent per candidate), or PER GPU with --scaling weak; --scaling both (the
  default) reports the strong run as the headline and, for N > 1, the weak one under "weak_scaling";
  information-gain criterion = entropy (the reference's effective default, agent.py:125; the MI criterion needs
  diag(C_rest^-1) over the whole pool: 2 x 110 000^2 x 8 B = 194 GB of scratch would fit the 288 GB, the two
  pool-wide O(n^3) factorisations behind it -- 9e14 flop -- are what rules it out at this size; it does not shard
  and is timed at single-GPU sizes in `extra.mi_criterion`), sigma_s = 0.1, sigma_m = 1.
Inputs (coordinates, targets, noise) are resident in HBM before the timed region.

Launch: python bench.py [--gpus N --steps K --warmup W].  For N > 1 either under torch.distributed.run, or plainly:
without WORLD_SIZE in the environment the script starts the N ranks itself as a child `python -m torch.distributed.run`
(before importing torch or touching HIP) and relays rank 0's line.  Prints ONE JSON line.
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
# the host driver of this pool only supports dmabuf IPC: without it RCCL's intra-node transport fails in hipIpcGetMemHandle
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WATCHDOG_EXIT = 3                   # status of every rank when the N > 1 extra leg is abandoned by its watchdog
FP64_MATRIX_PEAK_TFLOPS = 78.6      # MI355X fp64 matrix (= vector) peak, SURVEY.md section 8(d)
FP32_MATRIX_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md chip-level parameters
HBM_PEAK_GBS = 8000.0
CUS = 256


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


def candidate_lattice(n, R, C, shift=0):
    """n candidate sites on an offset lattice over the R x C field, none coinciding with a grid site."""
    cw = int(np.ceil(np.sqrt(n * C / R)))
    ch = int(np.ceil(n / cw))
    ii, jj = np.meshgrid(np.arange(ch), np.arange(cw), indexing='ij')
    c = np.vstack([(ii.ravel() + 0.37 + 0.011 * shift) * (R / ch), (jj.ravel() + 0.41 + 0.007 * shift) * (C / cw)]).T
    return c[:n]


def build_workload(args, world):
    rng = np.random.RandomState(1)                       # arguments.py:33 default seed
    R = int(round(np.sqrt(args.train)))
    C = args.train // R
    grid, field = mog_field(R, C, rng)
    N = len(grid)
    static_std, mobile_std = 0.1, 1.0
    # half the sampled sites carry static readings, half mobile ones (fusion rule agent.py:100-109)
    is_static = rng.uniform(size=N) < 0.5
    var = np.where(is_static, static_std ** 2, mobile_std ** 2)
    y = np.maximum(field + rng.standard_normal(N) * np.sqrt(var), 0.0)     # env.py:110-112
    if args.scaling == 'weak':
        # per rank an offset lattice of args.cand sites over the same field
        cands = [candidate_lattice(args.cand, R, C, r) for r in range(world)]
        counts = [args.cand] * world
    else:
        # ONE list of args.cand sites, cut into contiguous shards (the partition ShardedGreedy uses)
        from algp_amd.sharded import partition
        allc = candidate_lattice(args.cand, R, C, 0)
        parts = partition(args.cand, world)
        cands = [allc[a:b] for a, b in parts]
        counts = [b - a for a, b in parts]
    pool = np.vstack([grid] + cands)
    return dict(pool=pool, N=N, y=y, var=var, counts=counts, static_std=static_std, mobile_std=mobile_std, R=R, C=C)


def blas_info():
    try:
        cfg = np.show_config(mode='dicts')
        b = cfg.get('Build Dependencies', {}).get('blas', {})
        name = '%s %s' % (b.get('name', '?'), b.get('version', ''))
    except Exception:
        name = 'unknown'
    threads = os.environ.get('OPENBLAS_NUM_THREADS') or os.environ.get('OMP_NUM_THREADS') or os.environ.get('MKL_NUM_THREADS')
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    return name.strip(), (int(threads) if threads else cores), cores


def cpu_baseline(w, hyp_vals, args):
    """Reference-faithful CPU path (oracle 'port') on a bounded sample of the same workload:
    fit (fp32 kernel + np.linalg.inv, utils.py:296-300) on a sub-sampled train set plus `ncand` per-candidate
    slogdets (agent.py:328-329); one warm-up, then the median of 3 runs; extrapolated linearly in
    candidates x picks and cubically in N (stated in `sample`)."""
    from oracle import gp_oracle as O
    hyp = O.Hypers(np.log(hyp_vals['ls']), np.log(hyp_vals['os']), np.log(hyp_vals['noise']))
    N, M = w['N'], w['counts'][0]
    Ns = min(N, args.cpu_train)
    sel = np.sort(np.random.RandomState(0).permutation(N)[:Ns])
    X = w['pool'][:N][sel]
    var = w['var'][sel]
    ncand = 8                                                     # timed one by one: the printed rate is the MEDIAN candidate's (round 5: the
    xc = w['pool'][N:N + ncand]                                   # mean of 2, and the figure wandered +-25 % between runs)

    def fit_once():
        t0 = time.time()
        cov_aa = O.cov_mat_ref(hyp, X, None, var, True, dtype=np.float32)
        inv = np.linalg.inv(cov_aa)
        inv @ (w['y'][sel] - w['y'][sel].mean()).astype(np.float32)
        return time.time() - t0

    def candidate_once(i):
        t0 = time.time()
        Xa = np.vstack([X, xc[i:i + 1]])
        cov_a = O.cov_mat_ref(hyp, Xa, None, None, True, dtype=np.float32) + np.diag(np.r_[var, 0.01])
        O.entropy_from_cov_ref(cov_a)
        return time.time() - t0

    from scipy.linalg import solve_triangular
    ms = 2048

    def efficient():
        t0 = time.time()
        S = O.kernel_matrix(hyp, X) + np.diag(var) + hyp.noise * np.eye(Ns)
        L = np.linalg.cholesky(S)
        B = O.kernel_matrix(hyp, X, w['pool'][N:N + ms])
        V = solve_triangular(L, B, lower=True)
        pv = hyp.outputscale + hyp.noise - np.sum(V * V, axis=0)
        assert np.all(pv > 0)
        return time.time() - t0

    fit_once()                                                    # warm-up (BLAS thread pool, page faults)
    candidate_once(0)
    t_fit = float(np.median([fit_once() for _ in range(3)]))
    cand_times = [candidate_once(i) for i in range(ncand)]
    t_cand = float(np.median(cand_times))
    efficient()
    t_eff = float(np.median([efficient() for _ in range(3)]))
    k = args.picks
    t_step = (N / Ns) ** 3 * (t_fit + k * M * t_cand)
    # the O(N^2 M) triangular solve dominates the efficient form at these sizes: scale by (N/Ns)^2 * (M/ms)
    t_eff_step = (N / Ns) ** 2 * (M / ms) * t_eff
    blas, threads, cores = blas_info()
    return {
        'value': M / t_step, 'unit': 'candidates/s', 'cores': cores, 'kind': 'port',
        'blas': blas, 'blas_threads': threads, 'os_cpu_count': os.cpu_count(),
        'sample': 'reference-faithful oracle (fp32 kernel, np.linalg.inv, one slogdet per candidate) at '
                  'N=%d of %d train, %d of %d candidates each timed on its own, 1 of %d picks; 1 warm-up, then the median of 3 fits '
                  '(%.2fs) and the median candidate (%.2fs; min %.2f, max %.2f); extrapolated x(N/Ns)^3, linear in candidates x picks'
                  % (Ns, N, ncand, M, k, t_fit, t_cand, min(cand_times), max(cand_times)),
        'ms_per_step_extrapolated': 1e3 * t_step,
        'efficient_cpu_candidates_per_s': M / t_eff_step,
        'efficient_cpu_sample': 'numpy/scipy Cholesky + triangular solve + variance, N=%d, %d candidates, median of 3: %.2fs; '
                                'extrapolated x(N/Ns)^2 x M/%d' % (Ns, ms, t_eff, ms),
    }


def extra_c3(_hip, device):
    """BASELINE config 3: 10 000-point field, fp32 -- Cholesky + posterior on 10 000 held-out sites."""
    rng = np.random.RandomState(3)
    grid, field = mog_field(100, 100, rng)
    N = len(grid)
    test = candidate_lattice(10000, 100, 100, 1)
    c = _hip.Context(np.float32, device=device)
    c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
    c.set_pool(np.vstack([grid, test]))
    c.set_train(np.arange(N), field + 0.1 * rng.standard_normal(N), np.full(N, 0.01))
    c.set_candidates(np.arange(N, N + len(test)), prior_includes_noise=False)
    reps = 3
    # the step as algp_amd/utils.py:predictive_distribution runs it: the factorisation and the test sites' solve in one
    # task-list launch
    c.fit_and_solve()
    c.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        c.fit_and_solve()
        mu, pv = c.posterior()
    wall = (time.perf_counter() - t0) / reps * 1e3
    # the same as two phases, for the factorisation's own kernel time (the roofline below) and the solve's (HIP events)
    c.factorize()
    c.solve_candidates()
    c.prof_enable(True)
    c.prof_reset()
    for _ in range(reps):
        c.factorize()
        c.solve_candidates()
        mu, pv = c.posterior()
    ch, dag, tr = c.prof_get('cholesky'), c.prof_get('chol_dag'), c.prof_get('trsm')
    st = c.cholesky_task_stats()
    c.prof_enable(False)
    c.close()
    chol_ms = ch['ms'] / reps
    dag_ms = dag['ms'] / reps
    out = {'workload': 'C3: 10 000-point field, fp32: fit (kernel build + Cholesky + alpha) + posterior on 10 000 sites',
           'dtype': 'f32', 'ms_per_step': wall, 'fit_and_posterior_in_one_launch': True,
           'fit_ms': chol_ms,
           'cholesky_kernel_ms': dag_ms, 'cholesky_tflops': N ** 3 / 3.0 / (dag_ms * 1e-3) / 1e12 if dag_ms > 0 else None,
           'posterior_solve_ms': tr['ms'] / reps,
           'roofline': {'bound': 'mfma', 'kernel': 'chol_dag_kernel<float>', 'unit': 'TFLOP/s', 'peak': FP32_MATRIX_PEAK_TFLOPS}}
    if dag_ms > 0:
        out['roofline']['achieved'] = N ** 3 / 3.0 / (dag_ms * 1e-3) / 1e12
        out['roofline']['frac'] = out['roofline']['achieved'] / FP32_MATRIX_PEAK_TFLOPS
    if st['update_us'] > 0:
        out['cholesky_update_tflops_while_computing'] = 2 * 128.0 ** 3 * st['update_steps'] / (st['update_us'] * 1e-6 / (2 * CUS)) / 1e12
    return out


def _c5_field(rng, R=250, C=200, M=100000):
    grid, field = mog_field(R, C, rng)
    cand = candidate_lattice(M, R, C, 2) + 0.03 * rng.standard_normal((M, 2))
    return grid, field, np.vstack([grid, cand])


def _pct(v, q):
    return float(np.percentile(np.asarray(v, dtype=np.float64), q))


def _tail_traffic(alg_bytes):
    """HBM-side bytes per launch of the tail kernel from the PMC passes in profiles/ (2 x FETCH_SIZE + WRITE_SIZE, own --pmc
    passes of tools/collect_c5_profiles.sh), scaled from the sizes they were taken at to this run's: measured / algorithmic
    there x algorithmic here.  None when the file or the kernel's sources (tail.hip) changed since."""
    import hashlib
    try:
        tj = json.load(open(os.path.join(REPO, 'profiles', 'r06_traffic_pmc.json')))
        sha = hashlib.sha256(open(os.path.join(REPO, 'algp_amd', 'csrc', 'tail.hip'), 'rb').read()).hexdigest()[:16]
        if tj.get('tail_sources_sha16') != sha:
            return None, {'note': 'null: tail.hip changed since the PMC passes', 'tail_sources_sha16_now': sha,
                          'tail_sources_sha16_measured': tj.get('tail_sources_sha16')}
        ratio = tj['tail_part_f64_bytes_per_launch'] / tj['tail_part_f64_algorithmic_bytes_per_launch_same_run']
        return ratio * alg_bytes, {'source': 'profiles/r06_traffic_pmc.json', 'measured_over_algorithmic': ratio, 'tail_sources_sha16': sha}
    except Exception as e:
        return None, {'note': 'null: %s' % e}


def extra_c5(_hip, device, picks, steps=200, emu_steps=40):
    """BASELINE config 5: 50 000-point field fp64, 100 000 candidates, the active-learning loop (reference agent.py:125-229:
    greedy :141 -> _add_samples :66-82 -> refit :196-210) -- on ONE GPU: one from-scratch planning step, then `steps`
    incremental ones (each appends the picks + 26 mobile readings: factor update, the new columns of V^T, 4 picks), the
    second half with the library's event pairs on for the roofline of the kernel that dominates the step
    (tail_cols_kernel: s * M * N_old bytes streamed once).  Then `c5_rank_of_8`: the step of one rank of an 8-rank run of
    the same loop (see c5_rank_of_8)."""
    rng = np.random.RandomState(5)
    grid, field, pool = _c5_field(rng)
    N0, M = len(grid), len(pool) - len(grid)
    c = _hip.Context(np.float64, device=device)
    c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
    c.set_pool(pool)
    idx = np.arange(N0)
    var = np.where(rng.uniform(size=N0) < 0.5, 0.01, 1.0)
    y = np.maximum(field + rng.standard_normal(N0) * np.sqrt(var), 0.0)
    static = np.zeros(len(pool), bool)
    static[:N0] = var == 0.01
    cidx = np.arange(N0, N0 + M)
    times, rows, chol_ms, crossing = [], [], None, []
    prof_from = steps // 2 + 1
    tail = None
    t_loop = None
    for s in range(steps + 1):
        if s == 0 or s == prof_from:
            c.prof_enable(True)
            c.prof_reset()
        if s == 1:
            c.sync()
            t_loop = time.perf_counter()
        t0 = time.perf_counter()
        rows.append(int(len(idx)))
        # incremental=True from the first step, as algp_amd/agent.py calls it: with nothing resident it is a from-scratch
        # fit, but L and V^T get their 12.5 % row-stride headroom at once (otherwise the first 128-row growth re-lays
        # out 60 GB: 1.8 s)
        c.set_train(idx, y, var)
        c.factorize(incremental=True)
        c.set_candidates(cidx, prior_includes_noise=True)
        kc = c.solve_candidates(incremental=True, alive=~static[cidx])
        pk = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, picks)
        c.sync()
        times.append((time.perf_counter() - t0) * 1e3)
        crossing.append(bool(s > 0 and rows[-2] // 128 != (rows[-1] - 1) // 128))  # the new columns straddle a 128-column block of the factor
        if s == 0:
            chol_ms = c.prof_get('cholesky')['ms']
            trsm = c.prof_get('trsm')
            c.prof_enable(False)
        static[pk] = True
        mob = cidx[rng.permutation(M)[:26]]
        mob = mob[~np.isin(mob, idx) & ~np.isin(mob, pk)]
        idx = np.r_[idx, pk, mob]
        var = np.r_[var, np.full(len(pk), 0.01), np.full(len(mob), 1.0)]
        y = np.r_[y, rng.uniform(0, 1, len(pk) + len(mob))]
    c.sync()
    loop_s = time.perf_counter() - t_loop
    tail = c.prof_get('tail_cols')
    gt = c.prof_get('gemm_trsm')
    c.prof_enable(False)
    dev_gb = c.device_bytes() / 1e9
    c.close()
    ctf = N0 ** 3 / 3.0 / (chol_ms * 1e-3) / 1e12
    ttf = float(N0) ** 2 * M / (trsm['ms'] * 1e-3) / 1e12
    inc = np.array(times[1:])
    unprof, prof = inc[:prof_from - 1], inc[prof_from - 1:]
    cross = np.array(crossing[1:])
    tail_gbs = tail['bytes'] / (tail['ms'] * 1e-3) / 1e9 if tail['ms'] > 0 else None
    tail_traffic, tail_traffic_info = _tail_traffic(tail['bytes'] / max(1, tail['launches']))
    out = {'workload': 'C5 on one GPU: 50 000-point field, fp64, 100 000 candidates, %d picks + 26 mobile readings per step, %d '
                       'incremental steps after one from-scratch step' % (picks, steps), 'dtype': 'f64',
           'from_scratch_step_ms': times[0], 'incremental_steps': int(steps), 'loop_total_s': loop_s,
           'incremental_step_ms_median': float(np.median(unprof)), 'incremental_step_ms_p95': _pct(unprof, 95),
           'incremental_step_ms_max': float(np.max(unprof)),
           'incremental_step_ms_median_with_event_pairs': float(np.median(prof)),
           'block_crossing_steps': int(cross.sum()),
           'block_crossing_step_ms_median': float(np.median(inc[cross])) if cross.any() else None,
           'plain_step_ms_median': float(np.median(inc[~cross])),
           'step_ms_first_20': [round(t, 2) for t in times[:21]],
           'train_rows_first_last': [rows[0], rows[-1]],
           'fit_ms': chol_ms, 'cholesky_tflops': ctf, 'device_gb': dev_gb,
           'incremental_roofline': {'bound': 'hbm', 'kernel': 'tail_part_kernel<double> + tail_finish_kernel<double> (tail.hip: the new columns of V^T after an append; one launch pair)',
                                    'achieved': tail_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                    'frac': tail_gbs / HBM_PEAK_GBS if tail_gbs else None,
                                    'traffic': tail_traffic, 'traffic_provenance': tail_traffic_info,
                                    'launches': tail['launches'], 'avg_launch_ms': tail['ms'] / max(1, tail['launches']),
                                    'algorithmic_bytes_per_launch': tail['bytes'] / max(1, tail['launches']),
                                    'gemm_launches_in_block_crossing_steps': gt['launches'],
                                    'note': 'achieved = s * (Mpad * N_old + 64 * N_old) bytes (V^T read once + the new rows of L) / the '
                                            'kernel\'s own duration (HIP events on its stream), steps %d..%d' % (prof_from, steps)},
           'roofline': {'bound': 'mfma', 'kernel': 'gemm_nt_kernel_dma4<double> (candidate TRSM, N^2 M / wall time of the solve)',
                        'achieved': ttf, 'peak': FP64_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': ttf / FP64_MATRIX_PEAK_TFLOPS},
           'cholesky_roofline': {'achieved': ctf, 'peak': FP64_MATRIX_PEAK_TFLOPS, 'frac': ctf / FP64_MATRIX_PEAK_TFLOPS,
                                 'note': 'N^3/3 over the whole fit (kernel build + factorisation + z); launch sequence of '
                                         'potrf.hip (the one-launch task list is used up to N = 24 576)'}}
    try:
        out['c5_rank_of_8'] = c5_rank_of_8(_hip, device, picks, emu_steps)
        out['c5_rank_of_8']['speedup_vs_1_gpu_step'] = out['incremental_step_ms_median'] / out['c5_rank_of_8']['ms_per_step']
    except Exception as e:
        import traceback
        traceback.print_exc(file=sys.stderr)
        out['c5_rank_of_8'] = {'error': '%s: %s' % (type(e).__name__, e)}
    return out


def c5_rank_of_8(_hip, device, picks, steps=40, nranks=8, ranks=(0, 7), layout='strided', field=None):
    """What ONE rank of an 8-rank run of config 5's loop does per planning step, measured on this GPU through the product
    path: the replicated factor update with the new train sites' rows arriving in the row exchange
    (algp_comm_set_owners), the new columns of the rank's 12 500 rows of V^T, 4 picks through algp_greedy_sharded --
    over algp_comm_init_host(8, r, fn) with a raw callback that fabricates the 7 absent ranks from a one-rank
    "teacher" context running the same loop in lockstep: the agreement words (copies of this rank's), the rows of L of
    the sites other ranks own (the teacher's own new factor rows: the same numbers to rounding), the winners' rows and
    statistics per pick.  Picks are asserted equal to the teacher's at every step.  In the time: everything the rank's
    process does in a step, the host transport's staging copies (D2H of its own rows, H2D of all ranks') and
    synchronisations included.  Not in it: xGMI wire time and skew.  The absent ranks' rows are written into the
    library's pinned staging BEFORE the step (as a transport's receive would have left them there); the callback
    copies only headers and this rank's own slice."""
    import ctypes
    import struct
    from algp_amd.sharded import ShardLink
    rng = np.random.RandomState(5)
    if field is None:
        grid, fld, pool = _c5_field(rng)
    else:
        grid, fld, pool = field
    N0, M = len(grid), len(pool) - len(grid)
    n = len(pool)
    SS, SM = 0.1, 1.0
    es = 8

    def make():
        c = _hip.Context(np.float64, device=device)
        c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
        c.set_pool(pool)
        return c
    teacher = make()
    owners = ShardLink(0, nranks, all_gather=lambda b: b, layout=layout).owners(n)
    cidx = np.arange(N0, n)
    emus = []
    for r in ranks:
        e = make()
        st = {'r': r, 'q': 0, 'picks': None, 'rows_ptr': None, 'rows_plan': None, 'refills': 0, 'calls': [0, 0, 0]}

        def fn(send, recv, nbytes, st=st):
            r = st['r']
            if nbytes == 32:                                         # the factor update's agreement word: everyone agrees
                st['calls'][0] += 1
                for k in range(nranks):
                    ctypes.memmove(recv + 32 * k, send, 32)
                return 0
            if nbytes == st.get('pb'):                                # one pick's exchange
                st['calls'][1] += 1
                q = min(st['q'], len(st['picks']) - 1)
                pk, owner, util, d, row = st['picks'][q]
                absent = struct.pack('<4d', float('-inf'), -1.0, 0.0, 0.0)
                for k in range(nranks):
                    if k != r:
                        ctypes.memmove(recv + nbytes * k, absent, 32)
                if owner != r:
                    o = recv + nbytes * owner
                    ctypes.memmove(o, struct.pack('<4d', util, float(pk), 0.0, d), 32)
                    ctypes.memmove(o + 32, row.ctypes.data, row.nbytes)
                ctypes.memmove(recv + nbytes * r, send, nbytes)
                if struct.unpack('<d', ctypes.string_at(send + 16, 8))[0] == 0.0:
                    st['q'] += 1
                return 0
            # the row exchange: cap rows of Nb elements per rank
            st['calls'][2] += 1
            plan = st['rows_plan']
            if plan is None or plan['bytes'] != nbytes:
                return 7
            if st['rows_ptr'] != send:                               # staging moved (first use, or it grew): fill it now
                st['rows_ptr'] = send                                # [own rows | every rank's rows]: the base is what stays put
                st['refills'] += 1
                ctypes.memmove(recv, plan['buf'].ctypes.data, plan['buf'].nbytes)
            ctypes.memmove(recv + nbytes * r, send, nbytes)
            return 0
        e.comm_init_host(nranks, r, fn, raw=True)
        e.comm_set_owners(owners)
        emus.append((e, st, np.nonzero(owners[N0:] == r)[0] + N0))
    idx = np.arange(N0)
    var = np.where(rng.uniform(size=N0) < 0.5, SS ** 2, SM ** 2)
    y = np.maximum(fld + rng.standard_normal(N0) * np.sqrt(var), 0.0)
    static = np.zeros(n, bool)
    static[:N0] = var == SS ** 2
    per_rank = {str(r): [] for r in ranks}
    phases = bool(os.environ.get('C5_PHASES'))          # synchronise after each call and report the split (slower steps)
    phase_log = {}
    caps, nnew_log = [], []
    nnew = 0
    try:
        for s in range(steps + 1):
            Npad = (len(idx) + 127) // 128 * 128
            p0 = len(idx) - nnew
            Nb = p0 // 128 * 128
            # the teacher's step (untimed): picks, their utilities, statistics and rows; its new factor rows
            teacher.set_train(idx, y, var)
            teacher.factorize(incremental=True)
            teacher.set_candidates(cidx, prior_includes_noise=True)
            teacher.solve_candidates(incremental=True, alive=~static[cidx])
            new_rows = teacher.debug_factor_rows(p0, nnew, Nb) if (s > 0 and nnew > 0) else None
            pk, ut = teacher.greedy(_hip.CRIT_ENTROPY, SS, SM, picks, want_utilities=True)
            pk = [int(p) for p in pk]
            util = [float(np.nanmax(ut[q])) for q in range(picks)]
            del ut
            prow = [teacher.debug_get_pick(q) for q in range(picks)]
            pb = 32 + ((Npad + 128) * es + 15) // 16 * 16
            for e, st, mine in emus:
                r = st['r']
                st['pb'], st['q'] = pb, 0
                st['picks'] = [(pk[q], int(owners[pk[q]]), util[q], prow[q][1], prow[q][0]) for q in range(picks)]
                if new_rows is not None:
                    own = owners[idx[p0:]]
                    cnt = np.bincount(own, minlength=nranks)
                    cap = int(cnt.max())
                    buf = np.zeros((nranks, cap, Nb), dtype=np.float64)
                    slot = np.zeros(nranks, dtype=np.int64)
                    for i, o in enumerate(own):
                        if o != r:
                            buf[o, slot[o]] = new_rows[i]
                        slot[o] += 1
                    st['rows_plan'] = {'bytes': cap * Nb * es, 'buf': buf}
                    if st['rows_ptr'] is not None:                   # as a transport's receive would have left them
                        ctypes.memmove(st['rows_ptr'] + cap * Nb * es, buf.ctypes.data, buf.nbytes)
                    if r == ranks[0]:
                        caps.append(cap)
                e.sync()
                ph = []
                t0 = tp = time.perf_counter()

                def mark(tp):
                    if phases:
                        e.sync()
                        ph.append((time.perf_counter() - tp) * 1e3)
                    return time.perf_counter()
                e.set_train(idx, y, var)
                tp = mark(tp)
                e.factorize(incremental=True)
                tp = mark(tp)
                e.set_candidates(mine, prior_includes_noise=True)
                tp = mark(tp)
                e.solve_candidates(incremental=True, alive=~static[mine])
                tp = mark(tp)
                got = [int(p) for p in e.greedy_sharded(_hip.CRIT_ENTROPY, SS, SM, picks)]
                e.sync()
                mark(tp)
                per_rank[str(r)].append((time.perf_counter() - t0) * 1e3)
                if phases and s > 2:
                    phase_log.setdefault(str(r), []).append(ph)
                if got != pk:
                    raise RuntimeError('step %d: rank %d of %d picked %s, the one-rank loop %s' % (s, r, nranks, got, pk))
                if s > 0 and (e.counter(4) != 0 or e.counter(1) != Npad - p0):
                    raise RuntimeError('step %d: rank %d fell back to the triangular solve (%d fall-backs, %d rows placed)'
                                       % (s, r, e.counter(4), e.counter(1)))
            static[pk] = True
            mob = cidx[rng.permutation(M)[:26]]
            mob = mob[~np.isin(mob, idx) & ~np.isin(mob, pk)]
            nnew = len(pk) + len(mob)
            nnew_log.append(nnew)
            idx = np.r_[idx, pk, mob]
            var = np.r_[var, np.full(len(pk), SS ** 2), np.full(len(mob), SM ** 2)]
            y = np.r_[y, rng.uniform(0, 1, nnew)]
        out = {'workload': 'one rank of %d of config 5\'s loop: N0 = %d train rows (replicated factor), %d of %d candidates (%s owner map), '
                           '%d picks + 26 mobile readings per step, %d incremental steps' % (nranks, N0, len(emus[0][2]), M, layout, picks, steps),
               'ranks': {}, 'rows_per_rank_in_the_exchange_median': float(np.median(caps)) if caps else None,
               'new_train_rows_per_step_median': float(np.median(nnew_log))}
        for (e, st, mine) in emus:
            t = np.array(per_rank[str(st['r'])][1:])
            # the first incremental steps allocate the exchange's buffers (pinned staging: tens of ms once)
            tt = t[2:] if len(t) > 4 else t
            out['ranks'][str(st['r'])] = {'candidates': int(len(mine)), 'from_scratch_step_ms': per_rank[str(st['r'])][0],
                                          'ms_per_step_median': float(np.median(tt)), 'ms_per_step_p95': _pct(tt, 95),
                                          'ms_per_step_max': float(np.max(tt)), 'step_ms': [round(v, 2) for v in t],
                                          'row_exchanges': e.counter(3), 'fallbacks': e.counter(4),
                                          'staging_fills_inside_the_timed_step': st['refills'],
                                          'callbacks_agree_pick_rows': st['calls']}
        if phases:
            out['phase_ms_median_set_train_factorize_set_candidates_solve_picks'] = {
                r: [round(float(v), 3) for v in np.median(np.array(p), axis=0)] for r, p in phase_log.items()}
        out['ms_per_step'] = max(v['ms_per_step_median'] for v in out['ranks'].values())
        return out
    finally:
        teacher.close()
        for e, _, _ in emus:
            e.close()


def extra_mi(_hip, device):
    """The mutual-information criterion (agent.py:330-339) at single-GPU sizes up to config 4's own pool: pool n, 1 000 sampled
    sites, every other site a candidate, 4 picks one by one.  The first pick factors the two pool-wide matrices and inverts
    each factor in place (4 n^3 / 3 flop); the later picks fold the previous winner into both inverse diagonals with one pass
    over each triangle (O(n^2))."""
    out = {'workload': 'MI criterion: pool n, |A| = 1 000 sampled, n - 1 000 candidates, 4 picks one at a time (first pick: two pool-wide '
                       'O(n^3) factorisations + triangular inverses; later picks: rank-1 updates of the two inverse diagonals, O(n^2))',
           'dtype': 'f64', 'by_pool': {}}
    # the last pool is config 4's own (10 000 train + 100 000 candidate sites): two pool-wide inverses of 96.8 GB each,
    # resident as triangles over their factors; its first pick is 4 n^3 / 3 = 1.8e15 flop and runs ONCE (no warm-up pass)
    for (R, C) in ((50, 100), (100, 200), (250, 200), (275, 400)):
        rng = np.random.RandomState(7)
        grid, field = mog_field(R, C, rng)
        n = len(grid)
        perm = rng.permutation(n)
        A = np.sort(perm[:1000])
        cand = np.sort(perm[1000:])
        c = _hip.Context(np.float64, device=device)
        try:
            c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
            c.set_pool(grid)
            c.set_train(A, field[A], np.full(len(A), 0.01))
            c.factorize()
            c.set_candidates(cand, prior_includes_noise=True)
            c.solve_candidates()
            if n < 100000:
                c.greedy(_hip.CRIT_MUTUAL_INFORMATION, 0.1, 1.0, 1)                   # warm-up (scratch allocation)
                c.solve_candidates()
            c.sync()
            ms, picks = [], []
            for _ in range(4):
                t0 = time.perf_counter()
                pk = c.greedy(_hip.CRIT_MUTUAL_INFORMATION, 0.1, 1.0, 1)
                c.sync()
                ms.append((time.perf_counter() - t0) * 1e3)
                picks.append(int(pk[0]))
            out['by_pool'][str(n)] = {'first_pick_ms': ms[0], 'later_picks_ms': ms[1:], 'first_over_later': ms[0] / max(np.mean(ms[1:]), 1e-9),
                                      'candidates_per_s_later_picks': len(cand) / (np.mean(ms[1:]) * 1e-3), 'picks': picks,
                                      'device_gb': c.device_bytes() / 1e9,
                                      'first_pick_tflops': 4.0 * n ** 3 / 3.0 / (ms[0] * 1e-3) / 1e12}
        finally:
            c.close()
    out['ms_per_pick'] = out['by_pool']['5000']['first_pick_ms']
    return out


def extra_fit(_hip, device):
    """One iteration of the reference's hyper-parameter fit (models.py:145-158: 200 Adam iterations of factorisation + MLL +
    gradient per update_model, agent.py:84-87) at N = 10 000: algp_set_hypers (an Adam step changed them) + algp_fit_step
    (S, its factor and L^-T in one task-list launch, S^-1 = X X^T as one triangular-aware launch beside the two
    substitutions, the pairwise gradient reduction).  N^3 flop in all (N^3/3 each for the factor, L^-T and X X^T)."""
    out = {'workload': 'one iteration of GPR.fit at N = 10 000 (100 x 100 MoG field, D = 2): set_hypers + fit_step = factorisation + MLL + '
                       'gradient w.r.t. the D + 2 log hyper-parameters; median of 5', 'by_dtype': {}}
    for name, dt, peak in (('f64', np.float64, FP64_MATRIX_PEAK_TFLOPS), ('f32', np.float32, FP32_MATRIX_PEAK_TFLOPS)):
        rng = np.random.RandomState(4)
        grid, field = mog_field(100, 100, rng)
        N = len(grid)
        c = _hip.Context(dt, device=device)
        try:
            c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
            c.set_pool(grid)
            c.set_train(np.arange(N), field + 0.1 * rng.standard_normal(N), np.full(N, 0.01))
            c.fit_step()
            ts = []
            for it in range(5):
                c.sync()
                t0 = time.perf_counter()
                c.set_hypers(np.log([3.0 + 0.01 * (it + 1), 3.0]), 0.0, np.log(1e-2))
                mll, g = c.fit_step()
                ts.append((time.perf_counter() - t0) * 1e3)
            c.prof_enable(True)
            c.prof_reset()
            c.fit_step()
            pr = {k: c.prof_get(k) for k in ('kmat', 'dag_panel', 'chol_dag', 'gemm_other', 'trsv')}
            c.prof_enable(False)
            ms = float(np.median(ts))
            tf = float(N) ** 3 / (ms * 1e-3) / 1e12
            out['by_dtype'][name] = {
                'ms_per_iteration': ms, 'iterations_ms': [round(t, 3) for t in ts], 'mll': mll, 'grad': [float(v) for v in g],
                'factor_and_inverse_in_one_launch_ms': pr['dag_panel']['ms'] if pr['dag_panel']['launches'] else None,
                'xxt_launch_ms': pr['gemm_other']['ms'], 'xxt_launches': pr['gemm_other']['launches'],
                'substitutions_ms_beside_xxt': pr['trsv']['ms'], 'kernel_matrix_ms': pr['kmat']['ms'],
                'roofline': {'bound': 'mfma', 'kernel': 'chol_dag_kernel + gemm_nt_kernel_dma4 (factor, L^-T, X X^T)', 'unit': 'TFLOP/s',
                             'achieved': tf, 'peak': peak, 'frac': tf / peak,
                             'note': 'N^3 flop (N^3/3 each for the factorisation, L^-T and X X^T) / the whole iteration\'s wall time'}}
        finally:
            c.close()
    out['ms_per_iteration'] = out['by_dtype']['f64']['ms_per_iteration']
    return out


def sources_sha16():
    """Fingerprint of the kernel sources the PMC traffic figure belongs to (the GEMM and the solve that launches it)."""
    import hashlib
    import re
    h = hashlib.sha256()
    for f in ('gemm.hip', 'potrf.hip', 'mfma.h'):
        # the CODE: comments and white space do not change a kernel (round 6: a comment edit had voided the PMC figures)
        src = open(os.path.join(REPO, 'algp_amd', 'csrc', f), 'r').read()
        src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
        src = re.sub(r'//[^\n]*', '', src)
        h.update(re.sub(r'\s+', ' ', src).encode())
    return h.hexdigest()[:16]


def strong_emulation(ctx, _hip, res, args):
    """What ONE rank of a 2 / 4 / 8-rank strong-scaling run of config 4 does, measured on this GPU through the product
    path itself: the same train set, the rank's contiguous share of the candidates, algp_fit_and_solve, then
    algp_greedy_sharded over algp_comm_init_host(n, r, fn) -- pack -> gather -> first maximum -> commit, remote commits
    included.  `fn` stands in for the n - 1 absent ranks: the global winners, their utilities, statistics and rows of V^T
    are known from the one-rank run (picks are identical by construction, and that run holds every candidate's row), so it
    answers (-inf, -1, 0) for every absent rank except the winner's owner, whose true contribution it supplies.  Timed for
    the first and the last rank of each n.  Left out: the wire time of RCCL's all-gather (n x 80 KB per pick) and skew
    between ranks; the host transport used here costs one stream synchronisation and ~0.7 MB of PCIe traffic per pick
    instead, plus the Python callback."""
    import ctypes
    import struct
    from algp_amd.sharded import partition
    w0, N0, total = res['w'], res['N'], res['total_c']
    k = args.picks
    allc = np.arange(N0, N0 + total)
    out = {'by_gpus': {}}
    try:
        # the one-rank run once more, keeping what the absent owners would send: utility, statistic and row per pick
        ctx.set_candidates(allc, prior_includes_noise=True)
        ctx.fit_and_solve()
        picks, ut = ctx.greedy(_hip.CRIT_ENTROPY, w0['static_std'], w0['mobile_std'], k, want_utilities=True)
        picks = [int(p) for p in picks]
        util = [float(np.nanmax(ut[q])) for q in range(k)]
        del ut
        rows = [ctx.debug_get_pick(q) for q in range(k)]
        if picks != [int(p) for p in res['picks']]:
            raise RuntimeError('the picks of the reference run changed: %s vs %s' % (picks, res['picks']))
        # the replicated fit on its own (what every rank repeats): median of 5
        ts = []
        for _ in range(6):
            ctx.sync()
            t0 = time.perf_counter()
            ctx.factorize()
            ctx.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        fit_alone = float(np.median(ts[1:]))
        out['fit_alone_ms'] = fit_alone
        es = ctx.dtype.itemsize
        for n in (2, 4, 8):
            parts = partition(total, n)
            per_rank = {}
            for r in sorted(set((0, n - 1))):
                lo, hi = parts[r]
                share = allc[lo:hi]
                owners = [next(s for s, (a, b) in enumerate(parts) if a <= (p - N0) < b) for p in picks]
                state = {'q': 0, 'calls': 0}
                absent = struct.pack('<4d', float('-inf'), -1.0, 0.0, 0.0)

                def fn(send, recv, pb, r=r, state=state):
                    # raw form of the callback: the library's pinned staging itself.  (Round 4 went through Python bytes objects:
                    # ~1.5 ms of copies per step that belong to neither the product nor a real transport.)
                    state['calls'] += 1
                    q = min(state['q'], k - 1)
                    for s_ in range(n):
                        if s_ != r:
                            ctypes.memmove(recv + pb * s_, absent, 32)
                    if owners[q] != r:
                        row, d = rows[q]
                        o = recv + pb * owners[q]
                        ctypes.memmove(o, struct.pack('<3d', util[q], float(picks[q]), 0.0), 24)
                        ctypes.memmove(o + 24, np.asarray([d], dtype=ctx.dtype).tobytes(), es)
                        ctypes.memmove(o + 32, row.ctypes.data, row.nbytes)
                    ctypes.memmove(recv + pb * r, send, pb)
                    if struct.unpack('<d', ctypes.string_at(send + 16, 8))[0] == 0.0:     # a settled round: the next call is the next pick
                        state['q'] += 1
                    return 0

                ctx.set_candidates(share, prior_includes_noise=True)
                ctx.comm_init_host(n, r, fn, raw=True)
                try:
                    def one():
                        state['q'] = 0
                        ctx.fit_and_solve()
                        return ctx.greedy_sharded(_hip.CRIT_ENTROPY, w0['static_std'], w0['mobile_std'], k)
                    got = [int(p) for p in one()]
                    if got != picks:
                        raise RuntimeError('rank %d of %d picked %s, the one-rank run %s' % (r, n, got, picks))
                    ctx.prof_enable(True)
                    ctx.prof_reset()
                    ctx.sync()
                    reps = 4
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        one()
                    ctx.sync()
                    ms = (time.perf_counter() - t0) / reps * 1e3
                    pr = {kk: ctx.prof_get(kk) for kk in ('cholesky', 'trsm', 'dag_panel')}
                    ctx.prof_enable(False)
                finally:
                    ctx.comm_destroy()
                folded = pr['dag_panel']['launches'] > 0 and pr['trsm']['launches'] == 0
                fs_ms = (pr['dag_panel']['ms'] if folded else pr['cholesky']['ms'] + pr['trsm']['ms']) / reps
                fl = float(N0) ** 3 / 3.0 + float(N0) ** 2 * len(share)
                per_rank[str(r)] = {
                    'candidates': int(len(share)), 'ms_per_step_with_remote_commits': ms,
                    'remote_commits_per_step': int(sum(1 for o in owners if o != r)),
                    'fit_and_solve_ms': fs_ms, 'fit_and_solve_in_one_launch': bool(folded),
                    'fit_and_solve_tflops': fl / (fs_ms * 1e-3) / 1e12 if fs_ms > 0 else None,
                    'exchanges_per_step': state['calls'] / float(reps + 1)}
            worst = max(v['ms_per_step_with_remote_commits'] for v in per_rank.values())
            out['by_gpus'][str(n)] = {'ranks': per_rank, 'ms_per_step': worst}
    except Exception as e:
        import traceback
        traceback.print_exc(file=sys.stderr)
        print('bench: strong-scaling emulation failed: %s' % e, file=sys.stderr)
        return None
    finally:
        try:
            ctx.set_candidates(allc, prior_includes_noise=True)
        except Exception:
            pass
    return out


def loop_mode(args, world, rank, local_rank, dist, torch, progress=None):
    """`bench.py --loop K [--gpus N]`: BASELINE config 5's active-learning loop itself (reference agent.py:125-229) on N real
    ranks -- every rank replicates the factor of the growing train set and owns 1/N of the candidates (sharded.ShardLink,
    strided owner map); per planning step: factor update with the new train sites' rows arriving in the row exchange
    (algp_comm_set_owners), the new columns of the rank's rows of V^T, 4 picks through algp_greedy_sharded; then the picks
    + 26 mobile readings join the train set.  One from-scratch step, then K incremental ones, timed per step between
    barriers on rank 0 and as a whole (max over ranks).  Prints ONE JSON line (metric: ms per incremental step)."""
    from algp_amd import _hip
    from algp_amd.sharded import ShardLink
    rng = np.random.RandomState(5)
    R, C = args.loop_field
    grid, field, pool = _c5_field(rng, R, C, args.cand)
    N0, M, n = len(grid), args.cand, len(grid) + args.cand
    ctx = _hip.Context(np.float64, device=local_rank)
    ctx.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
    ctx.set_pool(pool)
    link, transport = None, 'none (one rank)'
    if world > 1:
        if args.backend == 'nccl':
            uid = [_hip.Context.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            link = ShardLink(rank, world, unique_id=uid[0])
            transport = 'RCCL inside libalgp_hip.so (algp_comm_init): one all-gather per pick + agreement word and row all-gather per step'
        else:
            def gloo_gather(send):
                t = torch.frombuffer(bytearray(send), dtype=torch.uint8)
                out = torch.empty(world * len(send), dtype=torch.uint8)
                dist.all_gather_into_tensor(out, t)
                return out.numpy().tobytes()
            link = ShardLink(rank, world, all_gather=gloo_gather)
            transport = 'gloo all-gather supplied by the caller (algp_comm_init_host)'
        link.attach(ctx, n)
        mine = link.mine(n)
        mine = mine[mine >= N0]
    else:
        mine = np.arange(N0, n)
    idx = np.arange(N0)
    var = np.where(rng.uniform(size=N0) < 0.5, 0.01, 1.0)
    y = np.maximum(field + rng.standard_normal(N0) * np.sqrt(var), 0.0)
    static = np.zeros(n, bool)
    static[:N0] = var == 0.01
    cidx = np.arange(N0, n)

    def note(where):
        if progress is not None:
            progress['where'] = where

    def barrier():
        ctx.sync()
        if dist is not None:
            note('barrier between steps')
            dist.barrier()
    times, picks_log = [], []
    t_all = None
    for s in range(args.loop + 1):
        if s == 1:
            barrier()
            t_all = time.perf_counter()
        barrier()
        t0 = time.perf_counter()
        ctx.set_train(idx, y, var)
        note('step %d of %d: factorize(incremental) -- agreement word + row all-gather' % (s, args.loop))
        ctx.factorize(incremental=True)
        ctx.set_candidates(mine, prior_includes_noise=True)
        note('step %d of %d: solve_candidates (rank-local)' % (s, args.loop))
        ctx.solve_candidates(incremental=True, alive=~static[mine])
        note('step %d of %d: greedy -- one all-gather per pick' % (s, args.loop))
        if world > 1:
            pk = [int(p) for p in ctx.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, args.picks)]
        else:
            pk = [int(p) for p in ctx.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, args.picks)]
        barrier()
        times.append((time.perf_counter() - t0) * 1e3)
        picks_log.append(pk)
        static[pk] = True
        mob = cidx[rng.permutation(M)[:26]]
        mob = mob[~np.isin(mob, idx) & ~np.isin(mob, pk)]
        idx = np.r_[idx, pk, mob]
        var = np.r_[var, np.full(len(pk), 0.01), np.full(len(mob), 1.0)]
        y = np.r_[y, rng.uniform(0, 1, len(pk) + len(mob))]
    barrier()
    total = time.perf_counter() - t_all
    if dist is not None:
        t = torch.tensor([total], dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        total = float(t.item())
    out = None
    if rank == 0:
        inc = np.array(times[1:])
        out = {'metric': 'active-learning loop, ms per planning step (config 5: factor update + new columns of V^T + %d picks)' % args.picks,
               'value': float(np.median(inc)), 'unit': 'ms/step', 'n_gpus': world, 'steps': args.loop, 'warmup': 1,
               'ms_per_step': 1e3 * total / args.loop, 'higher_is_better': False, 'scaling': 'strong', 'vs_baseline': None,
               'dtype': 'f64', 'data': 'synthetic',
               'config': {'workload': '%d-point MoG field (%d x %d) growing by %d picks + 26 mobile readings per step, %d candidates '
                                      'sharded over %d rank(s), entropy criterion' % (N0, R, C, args.picks, M, world),
                          'n_train_first_last': [N0, int(len(idx))], 'candidates_per_gpu': int(len(mine)), 'collective': transport},
               'from_scratch_step_ms': times[0], 'step_ms_p95': _pct(inc, 95), 'step_ms_max': float(inc.max()),
               'loop_total_s': total, 'picks_last_step': picks_log[-1],
               'row_exchanges': ctx.counter(3), 'fallbacks_to_the_triangular_solve': ctx.counter(4), 'roofline': None, 'cpu_baseline': None}
    ctx.close()
    return out


def self_spawn(args):
    """`python bench.py --gpus N` launched plainly: start the N ranks as a CHILD `python -m torch.distributed.run`
    (never an exec, and before this process has imported torch or touched HIP), relay rank 0's JSON line, exit with the
    child's return code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('bench: no WORLD_SIZE in the environment; starting %d ranks: %s' % (args.gpus, ' '.join(cmd)), file=sys.stderr)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE)
    line = None
    for ln in proc.stdout.decode('utf-8', 'replace').splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
        sys.stdout.flush()
    return proc.returncode if proc.returncode != 0 or line is not None else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--train', type=int, default=10000)
    ap.add_argument('--cand', type=int, default=100000, help='candidates in total (strong) / per GPU (weak)')
    ap.add_argument('--scaling', default='both', choices=['weak', 'strong', 'both'],
                    help='strong (BASELINE config 4: the 100 000 candidates are split over the ranks), weak (100 000 per rank), both '
                         '(default: the headline is the strong run; for N > 1 the weak one follows and is reported under "weak_scaling")')
    ap.add_argument('--dtype', default='f64', choices=['f64', 'f32'])
    ap.add_argument('--picks', type=int, default=4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the C3 / C5 / MI legs (they run after the timed region)')
    ap.add_argument('--no-emulation', action='store_true', help='skip the one-stream leg and the strong-scaling emulation (profiling runs: '
                    'only the workload\'s own launches in the trace)')
    ap.add_argument('--force-dist', action='store_true', help='use the torch.distributed/RCCL path even with one rank')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='gloo only for rehearsing several ranks on ONE card (RCCL refuses duplicate devices)')
    ap.add_argument('--collective', default='abi', choices=['abi', 'torch'],
                    help='abi: the all-gather inside the library (algp_greedy_sharded; RCCL with --backend nccl, the caller\'s '
                         'gloo all-gather handed to algp_comm_init_host with --backend gloo); torch: ShardedGreedy over '
                         'torch.distributed (the cross-check)')
    ap.add_argument('--loop', type=int, default=0, help='K > 0: run config 5\'s active-learning loop for K incremental steps on the --gpus ranks '
                    'instead of the config-4 step (candidates sharded, factor replicated; see loop_mode)')
    ap.add_argument('--loop-field', type=lambda v: tuple(int(x) for x in v.split('x')), default=(250, 200), help='RxC of the loop\'s field (default 250x200 = 50 000 sites)')
    ap.add_argument('--extra-loop-timeout', type=int, default=300, help='seconds after which the N > 1 extra leg is abandoned (the line is printed without it)')
    ap.add_argument('--extra-loop-steps', type=int, default=200, help='with --gpus N > 1 the line also carries extra.c5_loop: config 5\'s loop '
                    'for this many incremental steps on the N ranks (0 or --no-extras: skipped)')
    ap.add_argument('--cpu-train', type=int, default=6000)
    ap.add_argument('--traffic-json', default=os.path.join(REPO, 'profiles', 'r06_traffic_pmc.json'))
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_spawn(args))

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
        raise SystemExit('bench: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
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

    if args.loop > 0:
        res = loop_mode(args, world, rank, local_rank, dist, torch)
        if rank == 0:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            print(json.dumps(res))
            sys.stdout.flush()
            os.dup2(2, 1)
        if dist is not None:
            dist.destroy_process_group()
        return

    from algp_amd import _hip
    from algp_amd.sharded import LocalComm, ShardedGreedy, TorchComm

    dt = np.float64 if args.dtype == 'f64' else np.float32
    hyp_vals = dict(ls=[3.0, 3.0], os=1.0, noise=1e-2)
    ctx = _hip.Context(dt, device=local_rank)
    ctx.set_hypers(np.log(hyp_vals['ls']), np.log(hyp_vals['os']), np.log(hyp_vals['noise']))
    comm = TorchComm(torch.device('cuda', local_rank)) if dist is not None else LocalComm()
    collective = 'none'
    if dist is not None:
        collective = 'torch.distributed all_gather_into_tensor (%s)' % args.backend
        if args.collective == 'abi':
            try:
                if args.backend == 'nccl':
                    # the library's own communicator: rank 0 creates the id, torch.distributed only carries the 128 bytes
                    uid = [_hip.Context.comm_unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(uid, src=0)
                    ctx.comm_init(world, rank, uid[0])
                    collective = 'algp_greedy_sharded: ncclAllGather of (utility, pool index, status, statistic + that candidate\'s row of V^T) inside libalgp_hip.so'
                else:
                    def gloo_gather(send):
                        t = torch.frombuffer(bytearray(send), dtype=torch.uint8)
                        out = torch.empty(world * len(send), dtype=torch.uint8)
                        dist.all_gather_into_tensor(out, t)
                        return out.numpy().tobytes()
                    ctx.comm_init_host(world, rank, gloo_gather)
                    collective = 'algp_greedy_sharded over algp_comm_init_host (gloo all-gather supplied by the caller)'
            except Exception as e:                                 # keep the run: fall back to the torch collective
                print('bench: algp_comm_init failed (%s); using the torch.distributed collective' % e, file=sys.stderr)
        flags = [1 if collective.startswith('algp') else 0]
        t = torch.tensor(flags, dtype=torch.int32, device='cuda' if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MIN)                   # every rank takes the same path
        if int(t.item()) == 0 and collective.startswith('algp'):
            ctx.comm_destroy()
            collective = 'torch.distributed all_gather_into_tensor (%s)' % args.backend
    use_abi = collective.startswith('algp')

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def run_case(scaling):
        """W warm-up steps, K timed steps with the stage timers, K more without, on the workload of `scaling`."""
        args.scaling = scaling
        w = build_workload(args, world)
        ctx.set_pool(w['pool'])
        N = w['N']
        offs = np.concatenate([[0], np.cumsum(w['counts'])])
        total_c = int(offs[-1])
        ctx.set_train(np.arange(N), w['y'], w['var'])
        all_cand = np.arange(N, N + total_c)
        mine = all_cand[offs[rank]:offs[rank + 1]]
        ctx.set_candidates(mine, prior_includes_noise=True)
        picks_log = []

        def step():
            ctx.fit_and_solve()                                       # = algp_factorize + algp_solve_candidates
            if dist is None:
                picks = list(ctx.greedy(_hip.CRIT_ENTROPY, w['static_std'], w['mobile_std'], args.picks))
            elif use_abi:
                picks = list(ctx.greedy_sharded(_hip.CRIT_ENTROPY, w['static_std'], w['mobile_std'], args.picks))
            else:
                sg = ShardedGreedy(ctx, comm, all_cand)
                picks, _ = sg.greedy(_hip.CRIT_ENTROPY, w['static_std'], w['mobile_std'], args.picks)
            picks_log.append([int(p) for p in picks])

        step_ms = []                                              # every timed step's own duration: a step ends with its picks on the host

        def timed(nsteps, log=None):
            barrier()
            t0 = time.perf_counter()
            t_prev = t0
            for _ in range(nsteps):
                step()
                if log is not None:                               # (no synchronisation of its own: the last pick's read-back is one)
                    t_now = time.perf_counter()
                    log.append(1e3 * (t_now - t_prev))
                    t_prev = t_now
            barrier()
            el = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([el], dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            return el

        for _ in range(args.warmup):
            step()
        # the timed region: K steps with the library's HIP-event profiling on (the roofline figures below come from these
        # very launches); the same K steps are then repeated without the ~600 event pairs per step
        ctx.prof_enable(True)
        ctx.prof_reset()
        s0 = ctx.sync_count()
        elapsed = timed(args.steps, step_ms)
        syncs = ctx.sync_count() - s0
        prof = {k: ctx.prof_get(k) for k in _hip.PROF}
        chol_stats = ctx.cholesky_task_stats()
        ctx.prof_enable(False)
        elapsed_unprof = timed(args.steps)
        # SURVEY 8(d)'s second clock: host wall time around the ABI with the inputs handed over as host NumPy buffers every
        # step -- pool coordinates, train set and candidate list up (H2D inside the calls), the picks back on the host -- where
        # the headline's steps start from resident inputs (reference utils.py:293-319 is host-in / host-out)
        host_ms = []
        if world == 1 and not args.no_emulation:                  # (profiling runs, --no-emulation, hold only the workload's own 7 solves)
            picks_resident = picks_log[-1]
            for _ in range(max(3, min(args.steps, 10))):
                barrier()
                t0 = time.perf_counter()
                ctx.set_pool(w['pool'])
                ctx.set_train(np.arange(N), w['y'], w['var'])
                ctx.set_candidates(mine, prior_includes_noise=True)
                step()
                host_ms.append(1e3 * (time.perf_counter() - t0))
            assert picks_log[-1] == picks_resident, 'the host-inclusive step picked other candidates'
        return dict(w=w, N=N, total_c=total_c, Mloc=int(w['counts'][rank]), M0=int(w['counts'][0]), elapsed=elapsed,
                    elapsed_unprof=elapsed_unprof, prof=prof, chol_stats=chol_stats, picks=picks_log[-1], step=step,
                    syncs_per_step=syncs / float(args.steps), step_ms=step_ms, host_ms=host_ms)

    want = args.scaling
    res = run_case('weak' if want == 'weak' else 'strong')
    weak = None
    if want == 'both' and world > 1:
        weak = run_case('weak')
        args.scaling = 'strong'
    elif want == 'both':
        args.scaling = 'strong'

    serial = None
    if rank == 0 and world == 1 and not args.no_emulation:
        # the same solve with ONE row-chunk stream: the launches run back to back, so the sum of their own durations is
        # the GEMM kernel's rate without help from overlapping another stream's tail (what a --pmc pass also measures)
        try:
            ctx.set_trsm_chunks(1)
            res['step']()
            ctx.prof_enable(True)
            ctx.prof_reset()
            for _ in range(2):
                res['step']()
            ctx.sync()
            g1, sp1, tl1 = ctx.prof_get('gemm_trsm'), ctx.prof_get('trsm'), ctx.prof_get('tail_cols')
            ctx.prof_enable(False)
            # (the narrow last tile's tail-kernel launch belongs to the solve: its duration counts with the GEMM launches')
            serial = dict(sum_launch_ms=(g1['ms'] + tl1['ms']) / 2, wall_ms=sp1['ms'] / 2, launches=(g1['launches'] + tl1['launches']) // 2,
                          flops_executed=(g1['flops'] + tl1['flops']) / 2)
        except Exception as e:
            print('bench: serial (1 chunk stream) leg failed: %s' % e, file=sys.stderr)
        finally:
            ctx.set_trsm_chunks(0)

    emu = None
    if rank == 0 and world == 1 and want != 'weak' and not args.no_emulation:
        emu = strong_emulation(ctx, _hip, res, args)

    if rank == 0:
        K = args.steps
        w, N, total_c, prof, chol_stats = res['w'], res['N'], res['total_c'], res['prof'], res['chol_stats']
        elapsed, elapsed_unprof = res['elapsed'], res['elapsed_unprof']
        ms_step = 1e3 * elapsed / K
        peak = FP64_MATRIX_PEAK_TFLOPS if args.dtype == 'f64' else FP32_MATRIX_PEAK_TFLOPS
        Mloc = res['Mloc']
        g = prof['gemm_trsm']
        span = prof['trsm']                                       # wall time of the solves (row chunks overlap on 3 streams)
        alg_flops_step = float(N) ** 2 * Mloc                     # N^2 flop per candidate
        ach = alg_flops_step * K / (span['ms'] * 1e-3) / 1e12 if span['ms'] > 0 else 0.0
        ach_padded = g['flops'] / (span['ms'] * 1e-3) / 1e12 if span['ms'] > 0 else 0.0
        # HBM bytes per launch from the PMC passes (profiles/): only when they were taken on THESE kernel sources
        traffic, traffic_info = None, {'source': os.path.relpath(args.traffic_json, REPO), 'sources_sha16_now': sources_sha16()}
        try:
            tj = json.load(open(args.traffic_json))
            traffic_info['sources_sha16_measured'] = tj.get('sources_sha16')
            traffic_info['commit_measured'] = tj.get('commit')
            if tj.get('sources_sha16') == traffic_info['sources_sha16_now'] and args.train == 10000 and args.cand == 100000 and world == 1:
                traffic = tj.get('gemm_nt_%s_bytes_per_launch' % args.dtype)
                traffic_info['bytes_per_solve'] = tj.get('bytes_per_solve_%s' % args.dtype)
                traffic_info['fetch_factor'] = tj.get('fetch_factor')          # FETCH_SIZE x this = bytes: calibrated, see the source
                traffic_info['fetch_factor_source'] = tj.get('fetch_factor_source')
                if traffic_info['bytes_per_solve']:
                    # algorithmic minimum of a solve: V^T written once + read as the A operand of 20 block columns' updates is
                    # NOT minimal; the minimum is s * (N^2 / 2 + 2 N M)
                    traffic_info['over_algorithmic'] = traffic_info['bytes_per_solve'] / ((8 if args.dtype == 'f64' else 4) * (N * N / 2.0 + 2.0 * N * Mloc))
            else:
                traffic_info['note'] = 'null: the PMC passes were taken on other kernel sources or another workload than this run'
        except Exception as e:
            traffic_info['note'] = 'null: %s' % e
        dag = prof['chol_dag']
        fit_ms = prof['cholesky']['ms'] / K                       # wall time of the fit (kernel build, factorisation, z)
        dag_ms = dag['ms'] / K if dag['launches'] else None       # the factorisation proper (one launch)
        chol_tf = (N ** 3 / 3.0) / (fit_ms * 1e-3) / 1e12 if fit_ms > 0 else 0.0
        upd_tf = None
        if chol_stats['update_us'] > 0:
            # rank-k update ("panel update") tasks inside the one-launch factorisation: flops / time the workgroups spent in
            # them; two workgroups share a CU, so the chip-level rate while they compute is over time / 2 / CUs
            upd_tf = 2 * 128.0 ** 3 * chol_stats['update_steps'] / (chol_stats['update_us'] * 1e-6 / (2 * CUS)) / 1e12
        stage = {k: v['ms'] / K for k, v in prof.items()}
        # per-stage split of one step (ms): what is replicated on every rank vs what shards with the candidates
        replicated = fit_ms
        sharded_ms = max(ms_step - replicated, 0.0)
        proj = {}
        for n in (1, 2, 4, 8):
            proj[str(n)] = {'ms_per_step': replicated + sharded_ms / n,
                            'speedup_vs_1': ms_step / (replicated + sharded_ms / n),
                            'scoring_only_speedup_vs_1': float(n)}
        roof = {'bound': 'mfma', 'kernel': 'gemm_nt_kernel_dma4<%s> (candidate TRSM; the N %% 128 = %d columns of a narrow last tile by tail_cols_kernel, inside the solve\'s span)'
                                           % ('double' if args.dtype == 'f64' else 'float', N % 128),
                'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
                'traffic': traffic, 'traffic_provenance': traffic_info,
                'algorithmic_flops_per_step': alg_flops_step,
                'algorithmic_flops_per_launch': alg_flops_step * K / max(1, g['launches']),
                'launches': g['launches'], 'avg_launch_ms': g['ms'] / max(1, g['launches']),
                'wall_ms_all_launches': span['ms'], 'sum_launch_ms': g['ms'],
                'achieved_padded': ach_padded, 'padded_flops_per_step': g['flops'] / K,
                'note': 'achieved = N^2 M flop (SURVEY 8d) / wall time of the solve; the launches of the 3 row chunks '
                        'overlap on 3 streams, so sum_launch_ms > wall_ms_all_launches and avg_launch_ms (what rocprofv3 '
                        '--stats reports per launch) is not a denominator by itself; achieved_padded counts the '
                        'flops actually executed (128-padding + full-square diagonal-block products); serial_kernel_frac is the '
                        'same algorithmic flop over the SUM of the launches\' own durations when they run back to back on one '
                        'stream: the kernel\'s rate on these shapes without another stream filling its partial last rounds'}
        if serial and serial['sum_launch_ms'] > 0:
            roof['serial_kernel_frac'] = alg_flops_step / (serial['sum_launch_ms'] * 1e-3) / 1e12 / peak
            roof['serial_kernel_achieved'] = alg_flops_step / (serial['sum_launch_ms'] * 1e-3) / 1e12
            roof['serial_sum_launch_ms'] = serial['sum_launch_ms']
            roof['serial_wall_ms'] = serial['wall_ms']
            roof['serial_avg_launch_ms'] = serial['sum_launch_ms'] / max(1, serial['launches'])
        out = {
            'metric': 'GP-fit+MI-score throughput (N train x M candidates)',
            'value': total_c / (elapsed / K),
            'unit': 'candidates/s',
            'n_gpus': world, 'steps': K, 'warmup': args.warmup,
            'ms_per_step': ms_step,
            'ms_per_step_median': float(np.median(res['step_ms'])) if res['step_ms'] else None,
            'ms_per_step_p95': _pct(np.array(res['step_ms']), 95) if res['step_ms'] else None,
            'ms_per_step_host_inclusive': float(np.median(res['host_ms'])) if res['host_ms'] else None,
            'host_inclusive_note': ('median of %d steps that hand the pool coordinates (%d x 2), the train set and the candidate list over as host '
                                    'NumPy buffers inside the clock (algp_set_pool / algp_set_train / algp_set_candidates: H2D in the calls) and '
                                    'return the picks to the host; ms_per_step / _median / _p95: inputs resident, rank 0\'s own clock per step'
                                    % (len(res['host_ms']), N + total_c)) if res['host_ms'] else None,
            'ms_per_step_unprofiled': 1e3 * elapsed_unprof / K,
            'value_unprofiled': total_c / (elapsed_unprof / K),
            'higher_is_better': True, 'scaling': 'weak' if want == 'weak' else 'strong', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': '%d-point MoG field (train) x %d candidates %s, %d greedy picks, entropy criterion, D=2'
                                   % (N, args.cand, 'per GPU' if want == 'weak' else 'in total (BASELINE config 4: sharded over the ranks)',
                                      args.picks),
                       'criterion': 'entropy (agent.py:125 default); the MI criterion (agent.py:330-339) needs pool-wide inverse '
                                    'diagonals and does not shard: it is timed in extra.mi_criterion',
                       'n_train': N, 'candidates_per_gpu': res['M0'], 'candidates_total': total_c,
                       'parallelism': 'candidate shards x%d, one all-gather per pick (each rank\'s best utility + pool index + status + statistic and that candidate\'s row of V^T)' % world,
                       'collective': collective},
            'roofline': roof,
            'cholesky': {'fit_ms': fit_ms, 'tflops_n3_over_3_per_fit_ms': chol_tf, 'kernel': 'chol_dag_kernel (one launch)',
                         'kernel_ms': dag_ms,
                         'kernel_tflops': (N ** 3 / 3.0) / (dag_ms * 1e-3) / 1e12 if dag_ms else None,
                         'frac_of_peak': ((N ** 3 / 3.0) / (dag_ms * 1e-3) / 1e12 / peak) if dag_ms else None,
                         'panel_update': {'achieved_while_computing': upd_tf, 'peak': peak, 'unit': 'TFLOP/s',
                                          'frac': upd_tf / peak if upd_tf else None,
                                          'update_task_us_summed_over_workgroups': chol_stats['update_us'] / K,
                                          'update_k128_steps': chol_stats['update_steps'] / K,
                                          'note': 'rank-k update tile tasks of the one-launch factorisation: 2*128^3 flop per K=128 step / '
                                                  '(time inside the tasks / 2 workgroups per CU / 256 CUs), in-kernel 100 MHz stamps'}},
            'cholesky_tflops': chol_tf, 'cholesky_ms': fit_ms,
            'stage_ms_per_step': stage,
            'host_syncs_per_step': res['syncs_per_step'],
            'throughput': {'with_factorisation_candidates_per_s': total_c / (elapsed / K),
                           'scoring_only_candidates_per_s': total_c / (sharded_ms * 1e-3) if sharded_ms > 0 else None,
                           'replicated_fit_ms': replicated, 'sharded_scoring_ms': sharded_ms},
            'picks_last_step': res['picks'],
        }
        if world == 1:
            out['strong_projection'] = {'note': 'arithmetic on this run\'s stage split, NOT measured: every rank repeats the fit '
                                                '(replicated_fit_ms), the candidate work is assumed to divide by the number of ranks '
                                                '-- optimistic: a solve of 1/n of the candidates runs at a lower rate, see strong_emulation',
                                        'by_gpus': proj}
            if emu:
                for n, e in emu['by_gpus'].items():
                    e['speedup_vs_1'] = ms_step / e['ms_per_step']
                    e['scoring_only_ms'] = max(e['ms_per_step'] - emu['fit_alone_ms'], 0.0)
                    e['scoring_only_speedup_vs_1'] = sharded_ms / e['scoring_only_ms'] if e['scoring_only_ms'] > 0 else None
                emu['note'] = ('MEASURED on this one GPU through algp_fit_and_solve + algp_greedy_sharded over algp_comm_init_host(n, r, fn): the step '
                               'of the first and of the last rank of an n-rank strong-scaling run (same train set, the rank\'s 1/n of the '
                               'candidates, pack -> gather -> first maximum -> commit with the REMOTE commits of winners other ranks own); fn '
                               'fabricates the absent ranks\' contributions from the one-rank run (the true utility, statistic and row of '
                               'each winner).  ms_per_step = the slower of the two ranks; speedup_vs_1 = this run\'s ms_per_step / that; '
                               'scoring_only_* subtract the replicated fit timed on its own (fit_alone_ms) from the rank\'s step and '
                               'stage_ms_per_step.cholesky from the one-GPU step.  Not in it: RCCL\'s wire time for n x 80 KB per pick (the '
                               'host transport\'s synchronisation, its PCIe staging copies and a raw Python callback are in it instead) and skew between '
                               'ranks.  Up to 51 200 rows the fit and the solve are ONE task-list launch (fit_and_solve_in_one_launch).')
                out['strong_emulation'] = emu
        if weak is not None:
            out['weak_scaling'] = {'value': weak['total_c'] / (weak['elapsed'] / K), 'unit': 'candidates/s',
                                   'ms_per_step': 1e3 * weak['elapsed'] / K,
                                   'ms_per_step_unprofiled': 1e3 * weak['elapsed_unprof'] / K,
                                   'candidates_per_gpu': weak['M0'], 'candidates_total': weak['total_c'],
                                   'picks_last_step': weak['picks']}

    # N > 1: BASELINE config 5 in its stated form as well -- the active-learning loop on the N ranks (every rank takes part).
    # The headline line is complete at this point and must not be lost to this leg: every rank runs it under a watchdog that,
    # when no result has come after --extra-loop-timeout seconds (a rank stuck in a collective), lets rank 0 print the line
    # with the failure noted and ends the process (each rank its own: all of them leave).
    c5_multi = None
    if world > 1 and not args.no_extras and args.extra_loop_steps > 0:
        import threading
        if ctx is not None:
            ctx.close()
            ctx = None
        done = threading.Event()
        line_lock = threading.Lock()                               # exactly one of {watchdog, main thread} prints the line
        progress = {'where': 'starting'}                          # what the loop was doing when the watchdog fired (loop_mode updates it)

        def watchdog():
            if done.wait(args.extra_loop_timeout):
                return
            if not line_lock.acquire(blocking=False):              # the main thread is printing already: let it finish
                return
            print('bench: rank %d: extra loop leg gave no result after %d s (%s); leaving with status %d'
                  % (rank, args.extra_loop_timeout, progress['where'], WATCHDOG_EXIT), file=sys.stderr)
            if rank == 0:
                out['extra'] = {'c5_loop': {'error': 'no result after %d s: abandoned so that the headline line is not lost; rank 0 was at: %s; '
                                                     'every rank exits with status %d' % (args.extra_loop_timeout, progress['where'], WATCHDOG_EXIT)}}
                out['cpu_baseline'] = None
                sys.stdout.flush()
                os.dup2(saved_stdout, 1)
                print(json.dumps(out))
                sys.stdout.flush()
            # a process that has touched the GPU and is abandoned mid-collective did NOT succeed: non-zero on every
            # rank; self_spawn relays the line and this code.  Rank 0 leaves first: the launcher ends every rank as soon as one
            # has failed, and the line must be out by then.
            if rank != 0:
                time.sleep(5.0)
            os._exit(WATCHDOG_EXIT)
        threading.Thread(target=watchdog, daemon=True).start()
        largs = argparse.Namespace(**vars(args))
        largs.loop, largs.cand = args.extra_loop_steps, 100000 if args.cand == 100000 or want == 'weak' else args.cand
        t0 = time.perf_counter()
        try:
            c5_multi = loop_mode(largs, world, rank, local_rank, dist, torch, progress=progress)
        except Exception as e:                                   # must not cost the headline line (every rank raises alike or none)
            c5_multi = {'error': '%s: %s' % (type(e).__name__, e)}
        if not line_lock.acquire(blocking=False):                  # the watchdog fired first and is printing: it ends the process
            time.sleep(3600)
        done.set()
        if c5_multi is not None:
            c5_multi['leg_wall_s'] = time.perf_counter() - t0

    if rank == 0:
        extras = {}
        if world == 1 and not args.no_extras:
            ctx.close()
            ctx = None
            for name, fn in (('fit_iteration', lambda: extra_fit(_hip, local_rank)),
                             ('c3_fp32_10k', lambda: extra_c3(_hip, local_rank)),
                             ('c5_fp64_50k', lambda: extra_c5(_hip, local_rank, args.picks)),
                             ('mi_criterion', lambda: extra_mi(_hip, local_rank))):
                t0 = time.perf_counter()
                try:
                    extras[name] = fn()
                except Exception as e:                         # a failed extra must not cost the headline line
                    extras[name] = {'error': '%s: %s' % (type(e).__name__, e)}
                extras[name]['leg_wall_s'] = time.perf_counter() - t0
        if c5_multi is not None:
            extras['c5_loop'] = c5_multi
        out['extra'] = extras
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(w, hyp_vals, args)
        else:
            out['cpu_baseline'] = None
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out))
        sys.stdout.flush()
        os.dup2(2, 1)
    if ctx is not None:
        ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
