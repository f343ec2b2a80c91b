"""The RCCL transport of the sharded greedy loop (algp_comm_unique_id / algp_comm_init / algp_greedy_sharded: comm.hip) with
MORE THAN ONE RANK, on the one GPU a session has.  Real RCCL refuses two ranks on one device, so these tests hand the
library a test double for the five RCCL entry points it binds with dlsym (tests/fake_rccl.cpp: an all-gather over POSIX
shared memory, built here with hipcc, found through $ALGP_RCCL_PATH).  Everything on the library's side is the shipped code
path of a multi-GPU run -- the unique id travelling between processes, the device-resident payloads (utility, pool index,
status, statistic + the best row of V^T), the ncclChar all-gather on the library's stream, first maximum in rank order
(np.argmax over the concatenated scores, agent.py:349), remote commits from the gathered rows, ONE stream synchronisation
per pick -- only the wire is not xGMI (timing says nothing here).  The workers must not import torch:
a process that has PyTorch's librccl mapped uses that copy."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %(repo)r)
from algp_amd import _hip
from algp_amd.sharded import partition
assert 'torch' not in sys.modules
rank, world, tmp, dtname = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
dt = np.float64 if dtname == 'f64' else np.float32

def exchange(tag, payload=None):
    """rank 0 publishes bytes under `tag`, the others read them (a file: any out-of-band channel does, INTEGRATION.md)"""
    path = os.path.join(tmp, tag)
    if rank == 0:
        with open(path + '.tmp', 'wb') as f:
            f.write(payload)
        os.rename(path + '.tmp', path)
        return payload
    t0 = time.time()
    while not os.path.exists(path):
        assert time.time() - t0 < 120, 'rank 0 never published ' + tag
        time.sleep(0.01)
    return open(path, 'rb').read()

rng = np.random.RandomState(11)
N, M = 900, 4001
X = rng.uniform(0, 40, (N + M, 2))
static = rng.uniform(size=N) < 0.5
var = np.where(static, 0.01, 1.0)
cand = np.r_[np.where(~static)[0][:150], np.arange(N, N + M)]

def make(idx):
    c = _hip.Context(dt)
    c.set_hypers(np.log([3.0, 2.5]), 0.0, np.log(1e-2))
    c.set_pool(X)
    c.set_train(np.arange(N), np.zeros(N), var)
    c.set_candidates(idx, prior_includes_noise=True)
    c.fit_and_solve()
    return c

full = make(cand)
want, ut = full.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 6, want_utilities=True)
want = [int(p) for p in want]
tol = 1e-11 if dt == np.float64 else 2e-4

# shards: contiguous slices; with three ranks the LAST one owns nothing (an empty shard is not an error)
parts = partition(len(cand), world if world < 3 else world - 1)
lo, hi = parts[rank] if rank < len(parts) else (len(cand), len(cand))
c = make(cand[lo:hi])
uid = exchange('uid', _hip.Context.comm_unique_id() if rank == 0 else None)
assert len(uid) == 128
c.comm_init(world, rank, uid)
maps = open('/proc/self/maps').read()
assert os.path.basename(os.environ['ALGP_RCCL_PATH']) in maps and 'librccl' not in maps      # the double, and only it
for rep in range(2):
    c.fit_and_solve()
    s0 = c.sync_count()
    got, gut = c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 6, want_utilities=True)
    assert c.sync_count() - s0 == 6, ('one read-back per pick over the RCCL transport', c.sync_count() - s0)
    assert [int(p) for p in got] == want, (rank, got, want)
    for p in range(6):
        assert abs(gut[p] - np.nanmax(ut[p])) <= tol * max(1.0, abs(np.nanmax(ut[p]))), (p, gut[p], np.nanmax(ut[p]))
owners = sorted(set(next(r for r, (a, b) in enumerate(parts) if a <= int(np.where(cand == p)[0][0]) < b) for p in want))
# the state after the six commits (local and remote): this shard's utilities are the one-rank run's after the same picks
ref = make(cand)
ref.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 6)
uref = ref.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)[lo:hi]
ush = c.scores(_hip.CRIT_ENTROPY, 0.1, 1.0)
fin = np.isfinite(uref)
assert np.array_equal(fin, np.isfinite(ush))
if fin.any():
    assert np.max(np.abs(uref[fin] - ush[fin])) <= tol * max(1.0, np.max(np.abs(uref[fin])))
ref.close()

# a rank fails while resolving its pick: EVERY rank returns that error from the same call, nobody is left in the collective
c.fit_and_solve()
if rank == 1:
    c.debug_fail_next_pick(_hip.ERR_OOM)
try:
    c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 3)
    raise SystemExit('rank %%d: the injected failure was lost' %% rank)
except MemoryError:
    pass
got = c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 6)            # nothing was half-committed
assert [int(p) for p in got] == want, (rank, got, want)
# a failed pack launch travels in the status word as well
c.fit_and_solve()
if rank == 0:
    c.debug_fail_at(2, _hip.ERR_HIP)
try:
    c.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 2)
    raise SystemExit('rank %%d: the failed pack was lost' %% rank)
except _hip.AlgpError as e:
    assert e.code == _hip.ERR_HIP
c.comm_destroy()
c.close()
full.close()
print('RCCL_TRANSPORT_OK rank %%d owners %%s picks %%s' %% (rank, owners, want))
'''


@pytest.fixture(scope='module')
def fake_rccl(tmp_path_factory):
    out = tmp_path_factory.mktemp('fake_rccl') / 'libalgp_test_gather.so'
    r = subprocess.run(['/opt/rocm/bin/hipcc', '-shared', '-fPIC', '-O2', os.path.join(REPO, 'tests', 'fake_rccl.cpp'), '-o', str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(out)


@pytest.mark.parametrize('world,dtname,rows', [(2, 'f64', '1'), (2, 'f64', '0'), (3, 'f64', '1'), (2, 'f32', '1')],
                         ids=['two-ranks', 'two-ranks-rows-rebuilt', 'three-ranks-one-empty', 'two-ranks-fp32'])
def test_ranks_over_the_rccl_transport(tmp_path, fake_rccl, world, dtname, rows):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'repo': REPO})
    env = dict(os.environ, ALGP_RCCL_PATH=fake_rccl, ALGP_GATHER_ROWS=rows)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(tmp_path), dtname], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:                                        # exactly the processes started here
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d:\n%s\n%s' % (r, so[-2000:], se[-4000:])
        assert 'RCCL_TRANSPORT_OK rank %d' % r in so
    if world == 2:
        assert 'owners [0, 1]' in outs[0][0], outs[0][0]       # winners from both shards: both ranks commit remotely


LOOP_WORKER = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %(repo)r)
from algp_amd import _hip
from algp_amd.sharded import ShardLink
assert 'torch' not in sys.modules
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
        assert time.time() - t0 < 120, 'rank 0 never published ' + tag
        time.sleep(0.01)
    return open(path, 'rb').read()

rng = np.random.RandomState(31)
N0, NC = 2300, 5000
n = N0 + NC
X = rng.uniform(0, 55, (n, 2))
is_static = rng.uniform(size=N0) < 0.5

def make():
    c = _hip.Context(np.float64)
    c.set_hypers(np.log([3.0, 2.5]), 0.0, np.log(1e-2))
    c.set_pool(X)
    return c

uid = exchange('uid', _hip.Context.comm_unique_id() if rank == 0 else None)
link = ShardLink(rank, world, unique_id=uid, layout=sys.argv[4])
mine = link.mine(n)
ref, sh = make(), make()
link.attach(sh, n)                                   # algp_comm_init over the double + the owner map
rows_site, rows_static = list(range(N0)), list(is_static)
static = np.zeros(n, bool); static[:N0] = is_static
mobile = np.zeros(n, bool); mobile[:N0] = ~is_static
y_rows = list(rng.uniform(0, 1, N0))
r2 = np.random.RandomState(6)
peers = 0
for step in range(7):
    A = np.array(rows_site, dtype=np.int64)
    var = np.where(np.array(rows_static), 0.01, 1.0)
    y = np.array(y_rows)
    res = []
    for c, cand in ((ref, np.arange(n)), (sh, mine)):
        c.set_train(A, y, var)
        c.factorize(incremental=True)               # sharded: agreement word + ONE ncclAllGather of the new rows
        c.set_candidates(cand, prior_includes_noise=True)
        c.solve_candidates(incremental=True, alive=~static[cand])
        mu, pv = c.posterior()
        res.append((mu, pv, c.logdet()))
    if step > 0:
        assert sh.counter(3) == step and sh.counter(4) == 0, (sh.counter(3), sh.counter(4))
        peers += sh.counter(2)
    (mu1, pv1, ld1), (mu2, pv2, ld2) = res
    assert abs(ld1 - ld2) < 1e-9 * abs(ld1) and np.max(np.abs(mu1[mine] - mu2)) < 1e-9 and np.max(np.abs(pv1[mine] - pv2)) < 1e-9, step
    want = [int(p) for p in ref.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4)]
    got = [int(p) for p in sh.greedy_sharded(_hip.CRIT_ENTROPY, 0.1, 1.0, 4)]
    assert got == want, (step, got, want)
    path = [int(q) for q in r2.permutation(n)[:12]]
    for q, st in [(q, True) for q in got] + [(q, False) for q in path if not mobile[q] and q not in got]:
        rows_site.append(q); rows_static.append(st); y_rows.append(float(r2.uniform(0, 1)))
        (static if st else mobile)[q] = True
assert peers > 0
# an error in one rank's agreement word: every rank returns it, nobody enters the row gather
A = np.array(rows_site, dtype=np.int64); var = np.where(np.array(rows_static), 0.01, 1.0); y = np.array(y_rows)
sh.set_train(A, y, var)
if rank == world - 1:
    sh.debug_fail_at(3, _hip.ERR_OOM)
try:
    sh.factorize(incremental=True)
    raise SystemExit('rank %%d: the injected failure was lost' %% rank)
except MemoryError:
    pass
sh.comm_destroy(); sh.close(); ref.close()
print('RCCL_LOOP_OK rank %%d peers %%d' %% (rank, peers))
'''


@pytest.mark.parametrize('world,layout', [(2, 'strided'), (3, 'contiguous')])
def test_sharded_loop_over_the_rccl_transport(tmp_path, fake_rccl, world, layout):
    """Config 5's loop on sharded candidates through the RCCL code path of the row exchange (comm.hip: comm_agree's device
    all-gather + read-back, comm_rows_gather's ncclAllGather of the rows): 2 and 3 ranks on one card against the test double,
    seven steps each -- picks, posterior and log-determinant of the one-rank loop at every step, every factor update through
    the exchange, an injected failure returned by every rank (reference agent.py:125-229 with :313-354 sharded)."""
    script = tmp_path / 'worker.py'
    script.write_text(LOOP_WORKER % {'repo': REPO})
    env = dict(os.environ, ALGP_RCCL_PATH=fake_rccl)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(tmp_path), layout], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d:\n%s\n%s' % (r, so[-2000:], se[-4000:])
        assert 'RCCL_LOOP_OK rank %d' % r in so
