"""The sharded greedy loop behind the C ABI (algp_greedy_sharded, comm.hip) beyond a world of one, as far as ONE card
allows: the reduction kernel of the gather on fabricated 8-rank buffers (np.argmax semantics, agent.py:349; status
agreement), one host round trip per pick, two real ranks of the library sharing the GPU over a caller-supplied gloo
all-gather (algp_comm_init_host: RCCL refuses duplicate devices) -- incl. an empty shard and a rank that fails, which
must make EVERY rank return the error instead of leaving its peers in the collective -- and bench.py started plainly
with --gpus 2.  N > 1 over RCCL/xGMI needs one GPU per rank: unmeasured inside a session."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from algp_amd import _hip

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def ctx():
    c = _hip.Context(np.float64)
    yield c
    c.close()


def _argmax_ref(triples):
    """np.argmax over the ranks that offer a candidate; NaN utilities never win (the local argmax skips them too)."""
    best = None
    for r, (v, i, s) in enumerate(triples):
        if i >= 0 and v == v and (best is None or v > triples[best][0]):
            best = r
    return best


def test_first_max_on_fabricated_eight_rank_buffers(ctx):
    inf = np.inf
    cases = {
        'ties across ranks go to the first rank': [(1.5, 10, 0), (2.5, 20, 0), (2.5, 30, 0), (0.1, 40, 0), (2.5, 50, 0), (-1.0, 60, 0), (2.0, 70, 0), (2.5, 80, 0)],
        'empty ranks (-1) and -inf ranks': [(-inf, -1, 0), (-inf, 11, 0), (-inf, -1, 0), (0.25, 33, 0), (-inf, 44, 0), (0.25, 55, 0), (-inf, -1, 0), (0.2, 77, 0)],
        'NaN at rank 0 never wins': [(np.nan, 5, 0), (-3.0, 6, 0), (-2.0, 7, 0), (np.nan, 8, 0), (-2.0, 9, 0), (-5.0, 10, 0), (-inf, 11, 0), (-inf, -1, 0)],
        'only -inf on offer: the first rank that has a candidate': [(-inf, -1, 0), (-inf, 21, 0), (-inf, 22, 0), (-inf, -1, 0)] * 2,
        'nobody has a candidate': [(-inf, -1, 0)] * 8,
        'winner on the last rank': [(float(r), 100 + r, 0) for r in range(8)],
        'large pool indices are exact': [(0.0, 2 ** 40 + 1, 0), (1.0, 2 ** 52 + 3, 0)] + [(-inf, -1, 0)] * 6,
    }
    for name, tr in cases.items():
        out = ctx.debug_first_max(np.array(tr, dtype=np.float64))
        want = _argmax_ref(tr)
        if want is None:
            assert out[1] == -1 and out[2] == -1 and out[0] == -inf, (name, out)
        else:
            assert (out[0], int(out[1]), int(out[2])) == (tr[want][0], tr[want][1], want), (name, out)
        assert out[3] == 0 and out[4] == -1, (name, out)
    # against np.argmax proper on random per-rank maxima (no NaN, every rank non-empty)
    rng = np.random.RandomState(0)
    for _ in range(50):
        v = rng.randint(0, 5, 8).astype(float)            # many ties
        tr = [(v[r], 1000 + r, 0) for r in range(8)]
        out = ctx.debug_first_max(np.array(tr))
        assert int(out[2]) == int(np.argmax(v)) and out[0] == v.max()
    # status agreement: an error code outranks "one more round", the first failing rank is reported, the winner is still there
    tr = [(1.0, 10, 0), (2.0, 20, 1), (-inf, -1, _hip.ERR_OOM), (3.0, 40, 0), (-inf, -1, _hip.ERR_STATE), (0.0, 60, 0), (0.0, 70, 1), (0.0, 80, 0)]
    out = ctx.debug_first_max(np.array(tr, dtype=np.float64))
    assert out[3] == max(_hip.ERR_OOM, _hip.ERR_STATE) and int(out[4]) == 1 and int(out[1]) == 40
    out = ctx.debug_first_max(np.array([(1.0, 10, 0), (2.0, 20, 1)] + [(-inf, -1, 0)] * 6, dtype=np.float64))
    assert out[3] == 1 and int(out[4]) == 1
    out = ctx.debug_first_max(np.array([(1.0, 10, np.nan)] + [(0.5, 3, 0)] * 7, dtype=np.float64))
    assert out[3] >= 2                                     # a garbled status word is a failure, not "fine"


def _field(seed=11, n_train=900, n_cand=4000):
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 40, (n_train + n_cand, 2))
    static = rng.uniform(size=n_train) < 0.5
    var = np.where(static, 0.01, 1.0)
    cand = np.r_[np.where(~static)[0][:150], np.arange(n_train, n_train + n_cand)]
    return X, n_train, var, cand


def _setup(c, X, N, var, cand):
    c.set_hypers(np.log([3.0, 2.5]), 0.0, np.log(1e-2))
    c.set_pool(X)
    c.set_train(np.arange(N), np.zeros(N), var)
    c.factorize()
    c.set_candidates(cand, prior_includes_noise=True)
    c.solve_candidates()


def test_one_host_round_trip_per_pick(ctx):
    """algp_greedy (picks only, entropy) and algp_greedy_sharded resolve a pick with ONE stream synchronisation: the 40-byte
    winner record.  (there were >= 3 -- best_candidate per lazy round, gather, commit.)"""
    X, N, var, cand = _field()
    _setup(ctx, X, N, var, cand)
    want, ut = ctx.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 6, want_utilities=True)
    ctx.factorize()
    ctx.solve_candidates()
    s0 = ctx.sync_count()
    got = ctx.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 6)
    assert ctx.sync_count() - s0 == 6
    assert [int(p) for p in got] == [int(p) for p in want]
    # the same chain through algp_greedy_sharded with a host transport of one rank: ONE, in front of the caller's gather (the
    # payload has to be in host memory; the winner record is then computed there, and the winner's row goes back up as an
    # asynchronous copy from pinned staging)
    ctx.comm_init_host(1, 0, lambda b: b)
    try:
        ctx.factorize()
        ctx.solve_candidates()
        s0 = ctx.sync_count()
        got2, gut = ctx.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 6, want_utilities=True)
        assert ctx.sync_count() - s0 == 6
        assert [int(p) for p in got2] == [int(p) for p in want]
        for p in range(6):
            assert gut[p] == np.nanmax(ut[p])
        # a failure injected into this rank's next pick comes back as that error, and the context stays usable
        ctx.factorize()
        ctx.solve_candidates()
        ctx.debug_fail_next_pick(_hip.ERR_OOM)
        with pytest.raises(MemoryError):
            ctx.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 2)
        got3 = ctx.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 6)
        assert [int(p) for p in got3] == [int(p) for p in want]
    finally:
        ctx.comm_destroy()


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(repo)r)
import torch
import torch.distributed as dist
from algp_amd import _hip
from algp_amd.sharded import partition

dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.RandomState(11)
N, M = 900, 4001
X = rng.uniform(0, 40, (N + M, 2))
static = rng.uniform(size=N) < 0.5
var = np.where(static, 0.01, 1.0)
cand = np.r_[np.where(~static)[0][:150], np.arange(N, N + M)]

def gather(send):
    t = torch.frombuffer(bytearray(send), dtype=torch.uint8)
    out = torch.empty(world * len(send), dtype=torch.uint8)
    dist.all_gather_into_tensor(out, t)
    return out.numpy().tobytes()

def make(idx_slice, solve=True):
    c = _hip.Context(np.float64)
    c.set_hypers(np.log([3.0, 2.5]), 0.0, np.log(1e-2))
    c.set_pool(X)
    c.set_train(np.arange(N), np.zeros(N), var)
    c.factorize()
    c.set_candidates(idx_slice, prior_includes_noise=True)
    if solve:
        c.solve_candidates()
    return c

full = make(cand)
want, ut = full.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 6, want_utilities=True)
want = [int(p) for p in want]
full.close()

# 1) balanced shards: same picks, same utilities, winners from both shards (remote commits on both ranks: the winner's
#    row arrives in the gather, or -- ALGP_GATHER_ROWS=0 -- is rebuilt from the replicated factor)
lo, hi = partition(len(cand), world)[rank]
c = make(cand[lo:hi])
c.comm_init_host(world, rank, gather)
for rep in range(2):
    c.factorize(); c.solve_candidates()
    got, gut = c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 6, want_utilities=True)
    assert [int(p) for p in got] == want, (rank, got, want)
    for p in range(6):          # not bit-equal: a shard of <= 4096 candidates takes another solve order than the full list
        assert abs(gut[p] - np.nanmax(ut[p])) < 1e-11, (p, gut[p], np.nanmax(ut[p]))
owners = [0 if int(np.where(cand == p)[0][0]) < partition(len(cand), world)[0][1] else 1 for p in want]
assert len(set(owners)) == 2, owners
# the state after the six commits: every row of this shard has received every pick -- its utilities are those of the
# one-rank run after the same six picks (remote rows copied from the gather are the owner's bits)
ref = make(cand)
ref.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 6)
uref = ref.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)[lo:hi]
ush = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
fin = np.isfinite(uref)
assert np.array_equal(fin, np.isfinite(ush)) and np.max(np.abs(uref[fin] - ush[fin])) < 1e-11
ref.close()

# 1b) a commit that fails on ONE rank after the exchange: with a pick still ahead in the call, BOTH ranks return the error
#     from that call (it travels in the failing rank's next status word) ...
c.factorize(); c.solve_candidates()
if rank == 1:
    c.debug_fail_at(1, _hip.ERR_OOM)
try:
    c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 3)
    raise SystemExit('rank %%d: the failed commit was lost' %% rank)
except MemoryError as e:
    assert ('rank 1' in str(e)) or rank == 1, str(e)
#     ... and after the LAST pick of a call the failing rank returns it at once and reports it again in the first exchange
#     of its next call, where every rank sees it
c.factorize(); c.solve_candidates()
if rank == 1:
    c.debug_fail_at(1, _hip.ERR_OOM)
try:
    c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 1)
    assert rank == 0
except MemoryError:
    assert rank == 1
try:
    c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 2)
    raise SystemExit('rank %%d: the pending commit failure was not reported in the next call' %% rank)
except MemoryError:
    pass
# 1c) the launch that packs a rank's contribution fails: its status word says so, the gather still runs, both ranks return
c.factorize(); c.solve_candidates()
if rank == 0:
    c.debug_fail_at(2, _hip.ERR_HIP)
try:
    c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 2)
    raise SystemExit('rank %%d: the failed pack was lost' %% rank)
except _hip.AlgpError as e:
    assert e.code == _hip.ERR_HIP
c.factorize(); c.solve_candidates()
got = c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 6)
assert [int(p) for p in got] == want, (rank, got, want)

# 2) one rank fails while resolving its pick: BOTH ranks return that error, nobody hangs, nobody committed
c.factorize(); c.solve_candidates()
if rank == 1:
    c.debug_fail_next_pick(_hip.ERR_OOM)
try:
    c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 3)
    raise SystemExit('rank %%d: the injected failure was lost' %% rank)
except MemoryError as e:
    assert ('rank 1' in str(e)) or rank == 1, str(e)
got = c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 6)          # the very next call works: no pick was half-committed
assert [int(p) for p in got] == want, (rank, got, want)

# 3) a rank that never solved its candidates: an ALGP_ERR_STATE on every rank
c.factorize()
if rank == 0:
    c.solve_candidates()
try:
    c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 2)
    raise SystemExit('rank %%d: a rank without a candidate solve went unnoticed' %% rank)
except ValueError:
    pass
c.comm_destroy(); c.close()

# 4) an EMPTY shard (rank 1 owns no candidate): not an error, rank 0's candidates win every pick
e = make(cand if rank == 0 else cand[:0])
e.comm_init_host(world, rank, gather)
got = e.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 6)
assert [int(p) for p in got] == want, (rank, got, want)
e.comm_destroy(); e.close()
dist.barrier()
if rank == 0:
    print('SHARDED_ABI_OK', want, owners)
dist.destroy_process_group()
'''


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize('rows_in_gather', ['1', '0'], ids=['rows-travel', 'rows-rebuilt'])
def test_two_ranks_through_the_abi_collective(tmp_path, rows_in_gather):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'repo': REPO})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', ALGP_GATHER_ROWS=rows_in_gather)
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                          '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), str(script)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert 'SHARDED_ABI_OK' in out.stdout


def _bench(args, env_extra=None):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + args, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_plain_launch_with_two_ranks_on_one_card():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the script starts its ranks itself (child torch.distributed.run),
    prints ONE JSON line with n_gpus 2, config 4's strong reading as the headline (the candidates are split, so the picks
    equal the one-rank run's) and the weak figure beside it."""
    common = ['--steps', '1', '--warmup', '1', '--train', '2500', '--cand', '20000', '--no-extras', '--no-cpu-baseline']
    one = _bench(['--gpus', '1'] + common)
    two = _bench(['--gpus', '2', '--backend', 'gloo'] + common, {'ALGP_BENCH_DEVICE': '0'})
    assert one['n_gpus'] == 1 and two['n_gpus'] == 2
    assert two['scaling'] == 'strong' and two['config']['candidates_total'] == 20000 and two['config']['candidates_per_gpu'] == 10000
    assert two['picks_last_step'] == one['picks_last_step']
    assert 'algp_greedy_sharded' in two['config']['collective']
    assert two['weak_scaling']['candidates_total'] == 40000 and two['weak_scaling']['value'] > 0
    assert one['host_syncs_per_step'] <= 4 + 6            # 4 picks + the fit/solve's own


def test_comm_set_owners_argument_checks(ctx):
    """algp_comm_set_owners: needs a communicator and a pool, one entry per pool site, ranks inside the communicator (-1 = nobody);
    NULL clears the map; a map of one rank changes nothing (the factor update stays local, counters untouched)."""
    X, N, var, cand = _field()
    _setup(ctx, X, N, var, cand)
    with pytest.raises(ValueError):
        ctx.comm_set_owners(np.zeros(len(X), np.int32))             # no communicator attached
    ctx.comm_init_host(3, 1, lambda b: b * 3)
    try:
        with pytest.raises(ValueError):
            ctx.comm_set_owners(np.zeros(len(X) - 1, np.int32))     # one entry per pool site
        bad = np.zeros(len(X), np.int32)
        bad[5] = 3
        with pytest.raises(ValueError):
            ctx.comm_set_owners(bad)                                # rank outside the communicator
        ok = (np.arange(len(X)) % 3).astype(np.int32)
        ok[7] = -1                                                  # nobody holds site 7 as a candidate: allowed
        ctx.comm_set_owners(ok)
        ctx.comm_set_owners(None)                                   # cleared
    finally:
        ctx.comm_destroy()
    ctx.comm_init_host(1, 0, lambda b: b)
    try:
        ctx.comm_set_owners(np.zeros(len(X), np.int32))
        idx = np.r_[np.arange(N), cand[-3:]]
        ctx.set_train(idx, np.zeros(len(idx)), np.r_[var, np.full(3, 1.0)])
        assert ctx.factorize(incremental=True) == N // 128 * 128
        assert ctx.counter(3) == 0 and ctx.counter(4) == 0          # one rank: no exchange, no agreed fall-back
    finally:
        ctx.comm_destroy()
