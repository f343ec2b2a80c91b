"""EIGHT real ranks of the library on the one card a session has: eight contexts, eight shards, eight
payloads through the pick's first maximum, seven remote commits per pick, the row exchange's buffers sized for eight owners --
the reference loop agent.py:313-354 cut eight ways, every rank a real `algp_ctx` doing real work (no fabricated peers).

Why threads: this pool allows at most 6 processes on a card (gpurun's process guard), so the eight ranks are FOUR worker
processes x TWO rank threads (+ the pytest process = 5 processes on the GPU).  A context belongs to one thread, as the ABI
asks; ctypes releases the GIL around every library call, so the two ranks of a process do run side by side.  The transports:
  * the caller-supplied all-gather (algp_comm_init_host): the two threads of a process meet at a barrier, one of them runs
    the gloo all-gather of the process's two payloads over the four processes, both take the result (rank = 2 p + t);
  * the RCCL code path (algp_comm_init) against tests/fake_rccl.cpp: eight communicators over one shared-memory segment.
Checked: config 4 at FULL size (10 000 train x 100 003 candidates -- a count that does not divide by 8) gives the one-rank
run's four picks, their utilities, and after the commits every shard's utilities; config 5's loop (160 x 125 field, 10
steps, both owner maps) gives the one-rank loop's picks, posterior and log-determinant at every step with one row exchange
per step and no fall-back; the same loop and the pick exchange through the RCCL path."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ---- a worker process hosting R rank threads ---------------------------------------------------
HOST = r'''
import os, sys, threading, time, traceback
import numpy as np
sys.path.insert(0, %(repo)r)
from algp_amd import _hip
from algp_amd.sharded import ShardLink, partition
R = int(os.environ['RANKS_PER_PROC'])

class Meet(object):
    """the R rank threads of this process: a barrier with a time limit (a broken protocol fails the test, it does not hang)"""
    def __init__(self):
        self.bar = threading.Barrier(R, timeout=300)
        self.slots = [None] * R
        self.out = None
        self.errs = []

def run_ranks(meet, body, first_rank):
    def tgt(t):
        try:
            body(first_rank + t, t)
        except BaseException:
            meet.errs.append('rank %%d:\n%%s' %% (first_rank + t, traceback.format_exc()))
            meet.bar.abort()
    ths = [threading.Thread(target=tgt, args=(t,)) for t in range(R)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    if meet.errs:
        print('\n'.join(meet.errs))
        sys.stdout.flush()
        os._exit(1)
'''

GLOO = HOST + r'''
import torch
import torch.distributed as dist
dist.init_process_group('gloo')
P, p = dist.get_world_size(), dist.get_rank()
WORLD = P * R
MEET = Meet()

def gather(t, send):
    """bytes of rank 2 p + t -> the bytes of all WORLD ranks in rank order"""
    MEET.slots[t] = bytes(send)
    if MEET.bar.wait() == 0:                     # exactly one thread of the process talks to gloo
        local = b''.join(MEET.slots)
        tt = torch.frombuffer(bytearray(local), dtype=torch.uint8)
        out = torch.empty(P * len(local), dtype=torch.uint8)
        dist.all_gather_into_tensor(out, tt)
        MEET.out = out.numpy().tobytes()
    MEET.bar.wait()
    return MEET.out
'''

C4_WORKER = GLOO + r'''
# BASELINE config 4 at its own size, the candidate count one that 8 does not divide
N, M = 10000, 100003
rng = np.random.RandomState(17)
X = np.vstack([np.stack(np.meshgrid(np.arange(100.0), np.arange(100.0)), -1).reshape(-1, 2), rng.uniform(0, 99, (M, 2))])
var = rng.choice([0.01, 1.0], N)
y = rng.uniform(0, 1, N)
cand = np.arange(N, N + M)
parts = partition(M, WORLD)
assert len(set(h - l for l, h in parts)) == 2                         # uneven shards

def make(idx):
    c = _hip.Context(np.float64)
    c.set_hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
    c.set_pool(X)
    c.set_train(np.arange(N), y, var)
    c.set_candidates(idx, prior_includes_noise=True)
    return c

# the one-rank run (once per process): picks, their utilities, every candidate's utility after the four commits
full = make(cand)
full.factorize(); full.solve_candidates()
want, ut = full.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4, want_utilities=True)
want = [int(q) for q in want]
best = [float(np.nanmax(u)) for u in ut]
uref = full.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
full.close()

def body(rank, t):
    lo, hi = parts[rank]
    c = make(cand[lo:hi])
    c.comm_init_host(WORLD, rank, lambda b: gather(t, b))
    for rep in range(2):
        c.fit_and_solve()                                               # the rank's share: factor + solve in ONE launch
        s0 = c.sync_count()
        got, gut = c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 4, want_utilities=True)
        assert c.sync_count() - s0 == 4, ('one stream synchronisation per pick', c.sync_count() - s0)
        assert [int(q) for q in got] == want, (rank, got, want)
        for k in range(4):
            assert abs(gut[k] - best[k]) <= 1e-10 * max(1.0, abs(best[k])), (rank, k, gut[k], best[k])
    ush = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
    fin = np.isfinite(uref[lo:hi])
    assert np.array_equal(fin, np.isfinite(ush)), rank
    assert np.max(np.abs(uref[lo:hi][fin] - ush[fin])) <= 1e-10 * max(1.0, np.max(np.abs(uref[lo:hi][fin]))), rank
    c.comm_destroy(); c.close()

run_ranks(MEET, body, p * R)
owners = sorted(set(next(r for r, (a, b) in enumerate(parts) if a <= q - N < b) for q in want))
dist.barrier()
if p == 0:
    print('EIGHT_RANKS_C4_OK world %%d picks %%s owners %%s' %% (WORLD, want, owners))
dist.destroy_process_group()
'''

LOOP_BODY = r'''
# config 5's loop on a 160 x 125 field: 20 000 pool sites, every one a candidate; 3 000 sampled at the start; per step 4
# picks (static readings) + the mobile readings of a path join the train set
ROWS, COLS, N0, STEPS = 160, 125, 3000, 10
n = ROWS * COLS
rng = np.random.RandomState(23)
X = np.stack(np.meshgrid(np.arange(float(ROWS)), np.arange(float(COLS)), indexing='ij'), -1).reshape(-1, 2)
start = np.sort(rng.permutation(n)[:N0])
is_static0 = rng.uniform(size=N0) < 0.5
SS, SM = 0.1, 1.0
LAYOUT = os.environ.get('LOOP_LAYOUT', 'strided')

def make():
    c = _hip.Context(np.float64)
    c.set_hypers(np.log([3.0, 2.5]), 0.0, np.log(1e-2))
    c.set_pool(X)
    return c

def loop(c, cand, rank, pick_fn, check):
    """the loop on context c holding `cand` as its candidates; check(step, mu, pv, logdet, picks, c)"""
    rows_site, rows_static = list(start), list(is_static0)
    static = np.zeros(n, bool); static[start] = is_static0
    mobile = np.zeros(n, bool); mobile[start] = ~is_static0
    y_rows = list(np.random.RandomState(1).uniform(0, 1, N0))
    r2 = np.random.RandomState(5)
    for step in range(STEPS + 1):
        A = np.array(rows_site, dtype=np.int64)
        var = np.where(np.array(rows_static), SS ** 2, SM ** 2)
        c.set_train(A, np.array(y_rows), var)
        c.factorize(incremental=True)
        c.set_candidates(cand, prior_includes_noise=True)
        c.solve_candidates(incremental=True, alive=~static[cand])
        mu, pv = c.posterior()
        picks = [int(q) for q in pick_fn(c)]
        check(step, mu, pv, c.logdet(), picks, c)
        path = [int(q) for q in r2.permutation(n)[:14]]
        for q, st in [(q, True) for q in picks] + [(q, False) for q in path if not mobile[q] and q not in picks]:
            rows_site.append(q); rows_static.append(st); y_rows.append(float(r2.uniform(0, 1)))
            (static if st else mobile)[q] = True

# the one-rank loop, once per process
REF = []
ref = make()
loop(ref, np.arange(n), 0, lambda c: c.greedy(_hip.CRIT_ENTROPY, SS, SM, 4),
     lambda step, mu, pv, ld, picks, c: REF.append((mu.copy(), pv.copy(), ld, picks)))
ref.close()

def make_body(attach):
    def body(rank, t):
        link = attach(rank, t)
        mine = link.mine(n)
        sh = make()
        link.attach(sh, n)
        seen = {'peers': 0}
        def check(step, mu, pv, ld, picks, c):
            mu1, pv1, ld1, want = REF[step]
            assert picks == want, (rank, step, picks, want)
            assert abs(ld - ld1) < 1e-9 * abs(ld1), (rank, step, ld, ld1)
            assert np.max(np.abs(mu1[mine] - mu)) < 1e-9 and np.max(np.abs(pv1[mine] - pv)) < 1e-9, (rank, step)
            if step > 0:
                assert c.counter(3) == step, ('row exchanges == steps', rank, step, c.counter(3))
                assert c.counter(4) == 0, ('a sharded factor update fell back to the triangular solve', rank, step)
                seen['peers'] += c.counter(2)
        loop(sh, mine, rank, lambda c: c.greedy_sharded(_hip.CRIT_ENTROPY, SS, SM, 4), check)
        assert seen['peers'] > 0, 'no row ever came from another rank'
        sh.comm_destroy(); sh.close()
    return body
'''

LOOP_WORKER = GLOO + LOOP_BODY + r'''
run_ranks(MEET, make_body(lambda rank, t: ShardLink(rank, WORLD, all_gather=lambda b: gather(t, b), layout=LAYOUT)), p * R)
dist.barrier()
if p == 0:
    print('EIGHT_RANKS_LOOP_OK world %%d %%s' %% (WORLD, LAYOUT))
dist.destroy_process_group()
'''

# ---- the RCCL code path against the shared-memory double: P processes started directly, no torch in them ----
RCCL_WORKER = HOST + r'''
assert 'torch' not in sys.modules
p, P, tmp = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
WORLD = P * R
MEET = Meet()

def exchange(tag, payload=None):
    path = os.path.join(tmp, tag)
    if p == 0:
        with open(path + '.tmp', 'wb') as f:
            f.write(payload)
        os.rename(path + '.tmp', path)
        return payload
    t0 = time.time()
    while not os.path.exists(path):
        assert time.time() - t0 < 120, 'process 0 never published ' + tag
        time.sleep(0.01)
    return open(path, 'rb').read()
''' + LOOP_BODY + r'''
STEPS_RCCL = 4
UID = exchange('uid', _hip.Context.comm_unique_id() if p == 0 else None)
assert len(UID) == 128

def picks_body(rank, t):
    # the pick exchange alone first: 900 train rows x 4 001 candidates in 8 contiguous shards, six picks
    rng = np.random.RandomState(11)
    Nn, Mm = 900, 4001
    Xp = rng.uniform(0, 40, (Nn + Mm, 2))
    vv = np.where(rng.uniform(size=Nn) < 0.5, 0.01, 1.0)
    cd = np.arange(Nn, Nn + Mm)
    def mk(idx):
        c = _hip.Context(np.float64)
        c.set_hypers(np.log([3.0, 2.5]), 0.0, np.log(1e-2))
        c.set_pool(Xp); c.set_train(np.arange(Nn), np.zeros(Nn), vv)
        c.set_candidates(idx, prior_includes_noise=True); c.fit_and_solve()
        return c
    full = mk(cd)
    want = [int(q) for q in full.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 6)]
    full.close()
    lo, hi = partition(Mm, WORLD)[rank]
    c = mk(cd[lo:hi])
    c.comm_init(WORLD, rank, UID)
    s0 = c.sync_count()
    got = [int(q) for q in c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 6)]
    assert c.sync_count() - s0 == 6 and got == want, (rank, got, want)
    c.comm_destroy(); c.close()

run_ranks(MEET, picks_body, p * R)
maps = open('/proc/self/maps').read()
assert os.path.basename(os.environ['ALGP_RCCL_PATH']) in maps and 'librccl' not in maps
# ... then config 5's loop through algp_comm_init + the owner map (a fresh id: an id is single-use)
STEPS = STEPS_RCCL
UID2 = exchange('uid2', _hip.Context.comm_unique_id() if p == 0 else None)
run_ranks(MEET, make_body(lambda rank, t: ShardLink(rank, WORLD, unique_id=UID2, layout=LAYOUT)), p * R)
print('EIGHT_RANKS_RCCL_OK process %%d of %%d' %% (p, P))
'''

# ---- the Agent over ShardLink(unique_id=...) with hyper-parameters that change between steps ----
AGENT_RCCL_WORKER = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %(repo)r)
sys.path.insert(0, os.path.join(%(repo)r, 'tests'))
from algp_amd import _hip
from algp_amd.sharded import ShardLink
from algp_amd.agent import Agent
from algp_amd.arguments import get_args
from test_agent_loops import ManhattanField
rank, world, tmp = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]

def exchange(tag, payload=None):
    path = os.path.join(tmp, tag)
    if rank == 0:
        with open(path + '.tmp', 'wb') as f:
            f.write(payload)
        os.rename(path + '.tmp', path)
        return payload
    t0 = time.time()
    while not os.path.exists(path):
        assert time.time() - t0 < 120
        time.sleep(0.01)
    return open(path, 'rb').read()

def run(comm):
    np.random.seed(3)
    args = get_args([])
    args.kernel, args.max_iterations, args.num_samples_per_batch, args.fraction_pretrain = 'rbf', 15, 3, 0.5
    args.update_every = 1
    env = ManhattanField(30, 24, num_test=40)
    agent = Agent(env, args, static_std=args.static_std, mobile_std=10 * args.static_std, comm=comm)
    out = agent.run_ipp(num_runs=3, criterion='entropy', strategy='MaxEnt', update=True, disp=False)    # refit after every step
    return agent, out

one, out1 = run(None)
uid = exchange('uid', _hip.Context.comm_unique_id() if rank == 0 else None)
link = ShardLink(rank, world, unique_id=uid)
two, out2 = run(link)
maps = open('/proc/self/maps').read()
assert os.path.basename(os.environ['ALGP_RCCL_PATH']) in maps              # the double served algp_comm_init, not PyTorch's librccl
assert np.array_equal(one.static_locations, two.static_locations), (one.static_locations, two.static_locations)
assert np.allclose(out1['error'], out2['error'], rtol=0, atol=1e-8), (out1['error'], out2['error'])
c = two.gp.ctx
assert c.pool_generation >= 3, c.pool_generation                           # the pool WAS reloaded (new hyper-parameters) ...
assert link._id_used and c._shard_link is link                             # ... and the transport joined once
# a second ShardLink on the same id must not be able to join: the id is spent
try:
    ShardLink(rank, world, unique_id=uid).attach(_hip.Context(np.float64), 1)
    raise SystemExit('a used unique id initialised a second communicator')
except _hip.AlgpError:
    pass
print('AGENT_RCCL_OK rank %%d generations %%d' %% (rank, c.pool_generation))
'''


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _torchrun(tmp_path, text, token, env_extra=None, procs=4, ranks_per_proc=2):
    script = tmp_path / 'worker.py'
    script.write_text(text % {'repo': REPO})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', RANKS_PER_PROC=str(ranks_per_proc), **(env_extra or {}))
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % procs,
                          '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), str(script)],
                         capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-6000:]
    assert token in out.stdout, out.stdout[-2000:]
    return out.stdout


def test_config4_at_full_size_on_eight_ranks(tmp_path):
    out = _torchrun(tmp_path, C4_WORKER, 'EIGHT_RANKS_C4_OK world 8')
    assert 'owners [' in out


@pytest.mark.parametrize('layout', ['strided', 'contiguous'])
def test_config5_loop_on_eight_ranks_equals_the_one_rank_loop(tmp_path, layout):
    _torchrun(tmp_path, LOOP_WORKER, 'EIGHT_RANKS_LOOP_OK world 8 ' + layout, {'LOOP_LAYOUT': layout})


@pytest.fixture(scope='module')
def fake_rccl(tmp_path_factory):
    out = tmp_path_factory.mktemp('fake_rccl8') / 'libalgp_test_gather.so'
    r = subprocess.run(['/opt/rocm/bin/hipcc', '-shared', '-fPIC', '-O2', os.path.join(REPO, 'tests', 'fake_rccl.cpp'), '-o', str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(out)


def _spawn(tmp_path, text, args_of, n, env, token):
    script = tmp_path / 'worker.py'
    script.write_text(text % {'repo': REPO})
    procs = [subprocess.Popen([sys.executable, str(script)] + args_of(r), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(n)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=1200))
    finally:
        for p in procs:                                        # exactly the processes started here
            if p.poll() is None:
                p.kill()
    failed = ['process %d (rc %s):\n%s\n%s' % (r, p.returncode, so[-2500:], se[-2500:]) for r, (p, (so, se)) in enumerate(zip(procs, outs))
              if p.returncode != 0]
    assert not failed, '\n'.join(failed)
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert token % r in so, so[-2000:]


def test_eight_ranks_over_the_rccl_code_path(tmp_path, fake_rccl):
    """algp_comm_init / ncclAllGather (comm.hip) with eight communicators: the pick exchange, then four steps of config 5's
    loop with the agreement word and the row all-gather on the device path; ids are single-use (the double refuses a second
    join, as RCCL's bootstrap would)."""
    env = dict(os.environ, ALGP_RCCL_PATH=fake_rccl, RANKS_PER_PROC='2', LOOP_LAYOUT='strided')
    _spawn(tmp_path, RCCL_WORKER, lambda r: [str(r), '4', str(tmp_path)], 4, env, 'EIGHT_RANKS_RCCL_OK process %d of 4')


def test_agent_over_the_rccl_transport_survives_pool_reloads(tmp_path, fake_rccl):
    """`Agent(env, args, comm=ShardLink(rank, world, unique_id=...))` with a refit after every planning step: every refit
    changes the hyper-parameters, the Agent reloads its pool and re-attaches the link -- which must re-send the owner map only,
    never run ncclCommInitRank on the spent id again (round 5 did, and a real RCCL hangs there).  Two ranks on one card
    against the double, which refuses a second join; the mission equals the one-rank agent's."""
    env = dict(os.environ, ALGP_RCCL_PATH=fake_rccl)
    _spawn(tmp_path, AGENT_RCCL_WORKER, lambda r: [str(r), '2', str(tmp_path)], 2, env, 'AGENT_RCCL_OK rank %d')
