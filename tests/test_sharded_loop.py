"""BASELINE config 5 in its stated form -- the active-learning LOOP on sharded candidates (reference agent.py:125-229:
greedy :141 -> _add_samples :66-82 -> refit / predict :196-210, with the candidate loop of agent.py:313-354 cut into shards)
-- as far as ONE card allows: two real ranks of the library share the GPU over a caller-supplied gloo all-gather
(algp_comm_init_host; RCCL refuses duplicate devices).  Every planning step appends the picks (static readings) and the
mobile readings of a path to the train set; with an owner map attached (algp_comm_set_owners) the new rows of the replicated
factor travel from their owners' V^T in ONE all-gather per step instead of being solved against the kept factor on every
rank.  Checked at every step against the one-rank loop (same picks, posterior, log-determinant), against the NumPy oracle
at the end, that the fall-back solve was NOT taken, that a failure injected into one rank's agreement word comes back
from BOTH ranks, and that a rank whose V^T cannot supply its rows makes both fall back (same results).  The Agent itself
(`Agent(env, args, comm=ShardLink(...))`) runs its mission loop sharded and must reproduce the one-rank agent."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(repo)r)
import torch
import torch.distributed as dist
from algp_amd import _hip
from algp_amd.sharded import ShardLink, partition

dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()

def gather(send):
    t = torch.frombuffer(bytearray(send), dtype=torch.uint8)
    out = torch.empty(world * len(send), dtype=torch.uint8)
    dist.all_gather_into_tensor(out, t)
    return out.numpy().tobytes()
'''

LOOP_WORKER = PRELUDE + r'''
from oracle import gp_oracle as O
LAYOUT = os.environ.get('LOOP_LAYOUT', 'strided')
HYP = O.Hypers(np.log([3.0, 2.5]), 0.0, np.log(1e-2))
rng = np.random.RandomState(21)
N0, NC, STEPS = 3000, 8000, 20
n = N0 + NC
X = rng.uniform(0, 60, (n, 2))
SS, SM = 0.1, 1.0
is_static0 = rng.uniform(size=N0) < 0.5

def make():
    c = _hip.Context(np.float64)
    c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
    c.set_pool(X)
    return c

link = ShardLink(rank, world, all_gather=gather, layout=LAYOUT)
mine = link.mine(n)
ref = make()                        # the one-rank loop: every pool site a candidate of this one context
sh = make()                         # this rank's share
link.attach(sh, n)

# one train row per (site, kind of reading), as algp_amd/agent.py keeps them
rows_site = list(range(N0))
rows_static = list(is_static0)
static = np.zeros(n, bool); static[:N0] = is_static0
mobile = np.zeros(n, bool); mobile[:N0] = ~is_static0
y_rows = list(rng.uniform(0, 1, N0))
r2 = np.random.RandomState(5)
allc = np.arange(n)
peers_total = 0
for step in range(STEPS + 1):
    A = np.array(rows_site, dtype=np.int64)
    var = np.where(np.array(rows_static), SS ** 2, SM ** 2)
    y = np.array(y_rows)
    out = []
    for c, cand in ((ref, allc), (sh, mine)):
        c.set_train(A, y, var)
        kept = c.factorize(incremental=True)
        if step > 0:
            assert kept == (len(A) - nnew) // 128 * 128, (kept, len(A), nnew)
            # the new rows of L were placed, not solved: all rows from the first changed one to the end of the padding
            assert c.counter(1) == (len(A) + 127) // 128 * 128 - (len(A) - nnew), (step, c.counter(1))
        c.set_candidates(cand, prior_includes_noise=True)
        c.solve_candidates(incremental=True, alive=~static[cand])
        mu, pv = c.posterior()
        out.append((mu, pv, c.logdet()))
    if step > 0:
        assert sh.counter(4) == 0, 'a sharded factor update fell back to the triangular solve'
        assert sh.counter(3) == step
        peers_total += sh.counter(2)
    (mu1, pv1, ld1), (mu2, pv2, ld2) = out
    assert abs(ld1 - ld2) < 1e-9 * abs(ld1), (step, ld1, ld2)
    assert np.max(np.abs(mu1[mine] - mu2)) < 1e-9 and np.max(np.abs(pv1[mine] - pv2)) < 1e-9, step
    want = [int(p) for p in ref.greedy(_hip.CRIT_ENTROPY, SS, SM, 4)]
    got = [int(p) for p in sh.greedy_sharded(_hip.CRIT_ENTROPY, SS, SM, 4)]
    assert got == want, (step, got, want)
    # the step's new readings: static at the picks, mobile along a "path" (some sites new, some static-sampled already,
    # some with a mobile reading already -> no new row)
    path = [int(q) for q in r2.permutation(n)[:14]]
    new_rows = [(q, True) for q in got] + [(q, False) for q in path if not mobile[q] and q not in got]
    nnew = len(new_rows)
    for q, st in new_rows:
        rows_site.append(q); rows_static.append(st); y_rows.append(float(r2.uniform(0, 1)))
        (static if st else mobile)[q] = True
assert peers_total > 0, 'no row ever came from the other rank'
# oracle anchor at the end state, in the REFERENCE's form: one row per site with the static and the mobile reading fused
# (agent.py:100-109) -- the same GP as the two rows the loop keeps, the constant mean being the mean of the fused targets
A = np.array(rows_site); var = np.where(np.array(rows_static), SS ** 2, SM ** 2); y = np.array(y_rows)
ys = np.full(n, np.nan); ym = np.full(n, np.nan)
for q, st, v in zip(rows_site, rows_static, y_rows):
    (ys if st else ym)[q] = v
sites = np.where(static | mobile)[0]
both = static[sites] & mobile[sites]
yf = np.where(both, (SM ** 2 * ys[sites] + SS ** 2 * ym[sites]) / (SM ** 2 + SS ** 2), np.where(static[sites], ys[sites], ym[sites]))
vf = np.where(both, 1.0 / (1.0 / SS ** 2 + 1.0 / SM ** 2), np.where(static[sites], SS ** 2, SM ** 2))
assert both.sum() > 10 and not np.isnan(yf).any()
test = np.where(~(static | mobile))[0][rank::world][:300]
sh.set_constant_mean(float(np.mean(yf)))
sh.set_train(A, y, var)
sh.factorize(incremental=True)
sh.set_constant_mean(None)
sh.set_candidates(test, prior_includes_noise=False)
sh.solve_candidates()
mu, pv = sh.posterior()
o = O.posterior_chol(HYP, X[sites], yf, X[test], vf)
assert np.max(np.abs(mu - o['mu'])) < 1e-7 and np.max(np.abs(pv - o['var'])) < 1e-8, (np.max(np.abs(mu - o['mu'])), np.max(np.abs(pv - o['var'])))

# a failure in ONE rank's agreement word: BOTH ranks return it from the same call, nobody entered the row gather
def relock(c, cand):
    c.set_train(A, y, var); c.factorize(); c.set_candidates(cand, prior_includes_noise=True); c.solve_candidates(incremental=True, alive=~static[cand])
relock(sh, mine)
A2 = np.r_[A, np.where(~(static | mobile))[0][:9]]; y2 = np.r_[y, np.zeros(9)]; var2 = np.r_[var, np.full(9, SM ** 2)]
sh.set_train(A2, y2, var2)
if rank == 1:
    sh.debug_fail_at(3, _hip.ERR_OOM)
try:
    sh.factorize(incremental=True)
    raise SystemExit('rank %%d: the injected failure was lost' %% rank)
except MemoryError as e:
    assert rank == 1 or 'rank 1' in str(e), str(e)
# ... and a rank whose V^T cannot supply its rows (here: it holds another candidate list than it solved): both fall back
relock(sh, mine)
fb = sh.counter(4)
if rank == 0:
    sh.set_candidates(mine[:-1], prior_includes_noise=True)
sh.set_train(A2, y2, var2)
kept = sh.factorize(incremental=True)
assert kept > 0 and sh.counter(4) == fb + 1 and sh.counter(1) == 0
relock(ref, allc)
ref.set_train(A2, y2, var2); ref.factorize(incremental=True)
assert abs(ref.logdet() - sh.logdet()) < 1e-9 * abs(ref.logdet())
# ... and a rank whose resident factor is void (its hyper-parameter stamp moved: keep = 0) while its peer keeps its blocks:
# the incremental call is a collective all the same (ADVICE r5: the rank used to skip the agreement its peer waits in) --
# the peer falls back to the solve, this rank factorises from scratch, both end with the one-rank factor
relock(sh, mine)
fb = sh.counter(4)
if rank == 1:
    sh.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)    # the same values: only the stamp moves
sh.set_train(A2, y2, var2)
kept = sh.factorize(incremental=True)
assert (kept == 0) == (rank == 1), (rank, kept)
assert sh.counter(4) == fb + 1, (rank, sh.counter(4), fb)
assert abs(ref.logdet() - sh.logdet()) < 1e-9 * abs(ref.logdet())
# a new pool drops the owner map (it belongs to the pool it was given for): the next incremental call is rank-local again
sh.set_pool(X)
sh.set_train(A2, y2, var2)
sh.factorize(incremental=True)
assert abs(ref.logdet() - sh.logdet()) < 1e-9 * abs(ref.logdet())
dist.barrier()
if rank == 0:
    print('SHARDED_LOOP_OK', LAYOUT, peers_total)
dist.destroy_process_group()
'''

AGENT_WORKER = PRELUDE + r'''
sys.path.insert(0, os.path.join(%(repo)r, 'tests'))
from test_agent_loops import ManhattanField
from algp_amd.agent import Agent
from algp_amd.arguments import get_args

def run(comm):
    np.random.seed(3)
    args = get_args([])
    args.kernel, args.max_iterations, args.num_samples_per_batch, args.fraction_pretrain = 'rbf', 20, 3, 0.5
    env = ManhattanField(30, 24, num_test=40)
    agent = Agent(env, args, static_std=args.static_std, mobile_std=10 * args.static_std, comm=comm)
    out = agent.run_ipp(num_runs=4, criterion='entropy', strategy='MaxEnt', disp=False)
    return agent, out

one, out1 = run(None)
two, out2 = run(ShardLink(rank, world, all_gather=gather, layout=os.environ.get('LOOP_LAYOUT', 'strided')))
assert np.array_equal(one.static_locations, two.static_locations), (one.static_locations, two.static_locations)
assert np.array_equal(one.path, two.path)
assert np.allclose(out1['error'], out2['error'], rtol=0, atol=1e-9), (out1['error'], out2['error'])
assert np.max(np.abs(out1['mean'] - out2['mean'])) < 1e-8
c = two.gp.ctx
assert c.counter(3) > 0 and c.counter(4) == 0, (c.counter(3), c.counter(4))       # rows travelled, no fall-back
dist.barrier()
if rank == 0:
    print('SHARDED_AGENT_OK', c.counter(3))
dist.destroy_process_group()
'''


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _run(tmp_path, text, env_extra, token):
    script = tmp_path / 'worker.py'
    script.write_text(text % {'repo': REPO})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', **env_extra)
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                          '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), str(script)],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert token in out.stdout


@pytest.mark.parametrize('layout', ['strided', 'contiguous'])
def test_twenty_incremental_steps_on_two_ranks_equal_the_one_rank_loop(tmp_path, layout):
    _run(tmp_path, LOOP_WORKER, {'LOOP_LAYOUT': layout}, 'SHARDED_LOOP_OK')


def test_agent_mission_loop_sharded_over_two_ranks(tmp_path):
    _run(tmp_path, AGENT_WORKER, {}, 'SHARDED_AGENT_OK')


def _bench(args, env_extra=None, want_rc=0):
    import json
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + args, capture_output=True, text=True, timeout=900, env=env)
    assert (r.returncode == 0) == (want_rc == 0) and (want_rc == 0 or r.returncode != 0), (r.returncode, r.stdout[-2000:] + r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_loop_mode_two_ranks_on_one_card_equals_one_rank():
    """`bench.py --loop K --gpus 2` (what the driver can run on a node: config 5's loop on real ranks) rehearsed on one card
    over gloo at a reduced size: same picks as the one-rank loop at the last step, every factor update of the sharded run
    went through the row exchange, none fell back."""
    common = ['--loop', '6', '--loop-field', '60x50', '--cand', '9000']
    one = _bench(['--gpus', '1'] + common)
    two = _bench(['--gpus', '2', '--backend', 'gloo'] + common, {'ALGP_BENCH_DEVICE': '0'})
    assert one['n_gpus'] == 1 and two['n_gpus'] == 2 and two['steps'] == 6
    assert two['picks_last_step'] == one['picks_last_step']
    assert two['row_exchanges'] == 6 and two['fallbacks_to_the_triangular_solve'] == 0
    assert two['config']['candidates_per_gpu'] == 4500 and one['config']['candidates_per_gpu'] == 9000
    assert two['value'] > 0 and two['higher_is_better'] is False


def test_bench_multi_rank_line_carries_the_loop_as_an_extra():
    """With --gpus N > 1 the default bench line also times config 5's loop on the N ranks (extra.c5_loop): two ranks on one card
    over gloo at a reduced size; the headline stays config 4's."""
    two = _bench(['--gpus', '2', '--backend', 'gloo', '--steps', '1', '--warmup', '1', '--train', '2500', '--cand', '9000',
                  '--loop-field', '60x50', '--extra-loop-steps', '4', '--no-cpu-baseline'], {'ALGP_BENCH_DEVICE': '0'})
    assert two['n_gpus'] == 2 and two['config']['candidates_total'] == 9000
    lp = two['extra']['c5_loop']
    assert 'error' not in lp, lp
    assert lp['n_gpus'] == 2 and lp['steps'] == 4 and lp['row_exchanges'] == 4 and lp['fallbacks_to_the_triangular_solve'] == 0
    assert lp['config']['candidates_per_gpu'] == 4500 and lp['value'] > 0


def test_bench_multi_rank_line_survives_a_stuck_extra_leg():
    """The N > 1 extra leg runs under a watchdog on every rank: with a limit the leg cannot meet (0 s) the headline line is
    still printed -- exactly once, by rank 0, with the failure and the place the loop had reached noted under extra.c5_loop --
    and every rank ends with a NON-ZERO status (a process abandoned in a collective did not succeed), which `bench.py --gpus N`
    relays together with the line."""
    two = _bench(['--gpus', '2', '--backend', 'gloo', '--steps', '1', '--warmup', '1', '--train', '2500', '--cand', '9000',
                  '--loop-field', '60x50', '--extra-loop-steps', '50', '--extra-loop-timeout', '0', '--no-cpu-baseline'],
                 {'ALGP_BENCH_DEVICE': '0'}, want_rc=3)
    assert two['n_gpus'] == 2 and two['config']['candidates_total'] == 9000 and two['value'] > 0
    assert 'abandoned' in two['extra']['c5_loop']['error'] and 'rank 0 was at' in two['extra']['c5_loop']['error']

