"""Two ranks with the real HIP library (both on the one visible GPU; gloo process group because RCCL
refuses duplicate devices) shard a fixed candidate list and must reproduce the single-context greedy:
same picks, same winning utilities.  Exercises algp_commit_pick for winners owned by the other rank."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(repo)r)
import torch.distributed as dist
from algp_amd import _hip
from algp_amd.sharded import ShardedGreedy, TorchComm, partition

dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.RandomState(4)
N, M = 1200, 5001                      # ragged split: 2501 + 2500
X = rng.uniform(0, 50, (N + M, 2))
static = rng.uniform(size=N) < 0.5
var = np.where(static, 0.01, 1.0)
cand = np.r_[np.where(~static)[0][:200], np.arange(N, N + M)]      # some mobile-sampled train sites are candidates

def make(idx_slice):
    c = _hip.Context(np.float64)
    c.set_hypers(np.log([3.0, 2.5]), 0.0, np.log(1e-2))
    c.set_pool(X)
    c.set_train(np.arange(N), np.zeros(N), var)
    c.factorize()
    c.set_candidates(idx_slice, prior_includes_noise=True)
    c.solve_candidates()
    return c

lo, hi = partition(len(cand), world)[rank]
comm = TorchComm()
c = make(cand[lo:hi])
picks, vals = ShardedGreedy(c, comm, cand).greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 6)    # (utility, position) pairs
c.factorize(); c.solve_candidates()
picks_v, vals_v = ShardedGreedy(c, comm, cand, lazy=False).greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 6)   # score vectors
assert picks_v == picks and vals_v == vals, (picks_v, picks)
if rank == 0:
    full = make(cand)
    p2, ut = full.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 6, want_utilities=True)
    assert [int(p) for p in p2] == picks, (list(p2), picks)
    assert np.allclose([np.max(u) for u in ut], vals, rtol=0, atol=1e-11)
    owners = [int(np.searchsorted([h for _, h in partition(len(cand), world)], int(np.where(cand == p)[0][0]), side='right')) for p in picks]
    assert len(set(owners)) == 2, owners          # winners came from both shards
    print('MULTIPROC_OK', picks, owners)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def test_two_ranks_shard_one_gpu(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'repo': REPO})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                          '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), str(script)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert 'MULTIPROC_OK' in out.stdout
