"""CPU tests of host logic: sensor fusion vs the reference golden, candidate partitioning, and the
sharded greedy (one all-gather per pick) over gloo with world_size 2, driven by a CPU stand-in
backend built on the oracle (tests may use the oracle; the product never does)."""
import os
import sys
import types

import numpy as np
import pytest

from oracle import gp_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_get_sampled_dataset_matches_reference(golden):
    from algp_amd.agent import Agent
    g = golden('g5_fusion')
    sd, md, os_, om = [], [], 0, 0
    for ls, lm in zip(g['g5_lens_s'], g['g5_lens_m']):
        sd.append(list(g['g5_flat_s'][os_:os_ + ls]))
        md.append(list(g['g5_flat_m'][om:om + lm]))
        os_ += ls
        om += lm
    a = Agent.__new__(Agent)
    a.env = types.SimpleNamespace(num_samples=len(sd))
    a.static_data, a.mobile_data, a.static_std, a.mobile_std = sd, md, 0.1, 1.0
    idx, y, var = a.get_sampled_dataset()
    assert idx == list(g['g5_idx'])
    assert np.allclose(y, g['g5_y'], rtol=1e-14, atol=0) and np.allclose(var, g['g5_var'], rtol=1e-14, atol=0)
    # the per-site means are cached between calls: appended readings, a replaced list of the same length and a
    # first reading at a new site must all show up
    empty = [i for i in range(len(sd)) if not sd[i] and not md[i]]
    full = [i for i in range(len(sd)) if sd[i]]
    sd[full[0]].append(0.75)
    md[full[1]].append(-0.5)
    sd[full[2]] = [v + 1.0 for v in sd[full[2]]]
    md[empty[0]].append(0.25)
    idx2, y2, var2 = a.get_sampled_dataset()
    idx_o, y_o, var_o = O.get_sampled_dataset_ref(sd, md, 0.1, 1.0)
    assert idx2 == [int(i) for i in idx_o] and np.array_equal(y2, y_o) and np.array_equal(var2, var_o)


def test_synthetic_field_generator_matches_reference(golden):
    from algp_amd.utils import generate_gaussian_data
    g = golden('g6_field')
    np.random.seed(1)
    grid, y = generate_gaussian_data(20, 20, k=5)
    assert np.array_equal(grid, g['g6_s1_grid']) and np.allclose(y, g['g6_s1_y'], rtol=1e-14, atol=0)


def test_partition():
    from algp_amd.sharded import partition
    assert partition(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert partition(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert partition(0, 2) == [(0, 0), (0, 0)]
    for n in (1, 7, 100001):
        for w in (1, 2, 8):
            p = partition(n, w)
            assert p[0][0] == 0 and p[-1][1] == n and all(a[1] == b[0] for a, b in zip(p, p[1:]))
            assert max(h - l for l, h in p) - min(h - l for l, h in p) <= 1


def test_shard_link_owner_maps():
    """sharded.ShardLink: the owner map handed to algp_comm_set_owners and the rank's share of the pool (host logic of the
    sharded active-learning loop).  Every site has exactly one owner, the shares are balanced, 'contiguous' is the
    partition of SURVEY 8(e), 'strided' spreads neighbouring sites (a path's readings) over all owners."""
    from algp_amd.sharded import ShardLink, partition
    for n in (1, 7, 360, 100001):
        for w in (2, 3, 8):
            for layout in ('strided', 'contiguous'):
                links = [ShardLink(r, w, all_gather=lambda b: b, layout=layout) for r in range(w)]
                own = links[0].owners(n)
                assert own.dtype == np.int32 and len(own) == n and own.min() >= 0 and own.max() < w
                assert all(np.array_equal(l.owners(n), own) for l in links)           # the same map on every rank
                shares = [l.mine(n) for l in links]
                assert np.array_equal(np.sort(np.concatenate(shares)), np.arange(n))  # a partition of the pool
                assert max(map(len, shares)) - min(map(len, shares)) <= 1
                for r, sh in enumerate(shares):
                    assert np.all(own[sh] == r) and np.all(np.diff(sh) > 0)
                if layout == 'contiguous':
                    assert [(int(sh[0]), int(sh[-1]) + 1) if len(sh) else None for sh in shares] == \
                        [(lo, hi) if hi > lo else None for lo, hi in partition(n, w)]
                elif n >= 4 * w:
                    run = np.arange(n // 2, n // 2 + 2 * w)                          # 2w neighbouring sites: every rank owns two
                    assert np.all(np.bincount(own[run], minlength=w) == 2)
    with pytest.raises(ValueError):
        ShardLink(0, 2)                                                               # no transport
    with pytest.raises(ValueError):
        ShardLink(0, 2, unique_id=b'x' * 128, all_gather=lambda b: b)                 # two transports
    with pytest.raises(ValueError):
        ShardLink(0, 2, all_gather=lambda b: b, layout='random')


def test_sharded_pairs_tie_break_is_the_smaller_position():
    """Equal utilities on several ranks: the winner is the smaller global position whichever rank offers it (np.argmax's
    first maximum in pool order, agent.py:349) -- the rule first_max_kernel applies on the device (comm.hip)."""
    from algp_amd.sharded import ShardedGreedy

    class Comm(object):
        rank, world_size = 0, 3

        def __init__(self, pairs):
            self.pairs = np.array(pairs, dtype=np.float64)

        def all_gather_pairs(self, value, position):
            return self.pairs

    class Backend(object):
        M = 2
        committed = None

        def best_candidate(self, criterion, a, b):
            return 1, 11, 0.5

        def commit_pick(self, idx, a, b):
            self.committed = idx
    cand = np.arange(100, 106)
    for pairs, want in (([[0.5, 5], [0.5, 2], [0.25, 0]], 102), ([[0.5, 1], [0.5, 2], [0.5, 4]], 101),
                        ([[-np.inf, -1], [0.1, 3], [0.1, 2]], 102)):
        b = Backend()
        sg = ShardedGreedy(b, Comm(pairs), cand)
        w, v = sg.step(0, 0.1, 1.0)
        assert (w, b.committed) == (want, want), (pairs, w)


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(repo)r)
import torch.distributed as dist
from oracle import gp_oracle as O
from algp_amd.sharded import ShardedGreedy, TorchComm, partition

class OracleBackend(object):
    """CPU stand-in with the _hip.Context scoring surface (scores / commit_pick / M)."""
    def __init__(self, C, static, mobile, cand, ss, sm):
        from scipy.linalg import solve_triangular
        self.C, self.cand, self.ss, self.sm = C, np.asarray(cand), ss, sm
        vf = 1 / (1 / ss + 1 / sm)
        self.delta = vf - sm
        A = np.where(static | mobile)[0]
        self.A = A
        D = np.where(static[A] & mobile[A], vf, np.where(static[A], ss, sm))
        self.pos = -np.ones(len(C), dtype=int); self.pos[A] = np.arange(len(A))
        self.L = np.linalg.cholesky(C[np.ix_(A, A)] + np.diag(D))
        self.in_A = mobile[self.cand]
        self.V = self._col(self.cand)
        self.d = np.where(self.in_A, 0.0, C[self.cand, self.cand]) + np.where(self.in_A, 1, -1) * np.sum(self.V ** 2, 0)
        self.alive = np.ones(len(self.cand), bool)
        self.M = len(self.cand)
        self.rows = []          # (l, in_train, scale, pool_idx)
        self.solve = solve_triangular
    def _col(self, idx):
        from scipy.linalg import solve_triangular
        idx = np.atleast_1d(idx)
        B = np.zeros((len(self.A), len(idx)))
        for k, i in enumerate(idx):
            if self.pos[i] >= 0: B[self.pos[i], k] = 1.0
            else: B[:, k] = self.C[self.A, i]
        return solve_triangular(self.L, B, lower=True)
    def scores(self, criterion, static_std, mobile_std, out_device_ptr=None):
        with np.errstate(all='ignore'):
            u = np.where(self.in_A, .5 * np.log1p(self.delta * self.d), O.CONST + .5 * np.log(self.d + self.ss))
        return np.where(self.alive, u, -np.inf)
    def best_candidate(self, criterion, static_std, mobile_std):
        s = self.scores(criterion, static_std, mobile_std)
        j = int(np.argmax(s))
        return j, int(self.cand[j]), float(s[j])
    def commit_pick(self, pool_idx, static_std, mobile_std):
        in_tr = self.pos[pool_idx] >= 0
        l = self._col(pool_idx)[:, 0]
        for (lp, itp, sc, pp) in self.rows:       # entries appended by earlier picks
            bp = 0.0 if (itp or in_tr) else self.C[pp, pool_idx]
            l = np.r_[l, (bp - lp @ l) * sc]
        dc = (l @ l) if in_tr else self.C[pool_idx, pool_idx] - l @ l
        scale = np.sqrt(-(self.delta / (1 + self.delta * dc))) if in_tr else 1 / np.sqrt(dc + self.ss)
        t = l @ self.V
        bp = np.where(self.in_A | in_tr, 0.0, self.C[pool_idx, self.cand])
        r = (bp - t) * scale
        self.d = self.d + np.where(self.in_A, 1, -1) * r * r
        self.V = np.vstack([self.V, r[None, :]])
        self.rows.append((l, in_tr, scale, pool_idx))
        hit = np.where(self.cand == pool_idx)[0]
        if len(hit): self.alive[hit[0]] = False

dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.RandomState(5)
X = rng.uniform(0, 15, (240, 2))
hyp = O.Hypers(np.log([2.0, 2.0]), 0.0, np.log(1e-2))
C = O.kernel_matrix(hyp, X) + hyp.noise * np.eye(len(X))
static = np.zeros(len(X), bool); mobile = np.zeros(len(X), bool)
perm = rng.permutation(len(X)); static[perm[:40]] = True; mobile[perm[30:90]] = True
cand = np.where(~static)[0]
lo, hi = partition(len(cand), world)[rank]
want, ut = O.greedy_fast(C, static, mobile, 0.1, 1.0, 5, 'entropy')
comm = TorchComm()
for lazy in (True, False):          # (utility, position) pairs | whole score vectors
    backend = OracleBackend(C, static, mobile, cand[lo:hi], 0.01, 1.0)
    sg = ShardedGreedy(backend, comm, cand, lazy=lazy)
    picks, vals = sg.greedy(0, 0.1, 1.0, 5)
    assert picks == want, (lazy, picks, want)
    assert np.allclose(vals, [ut[p][want[p]] for p in range(5)], rtol=1e-9)
if rank == 0:
    print('SHARDED_OK', picks)
dist.destroy_process_group()
'''


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def test_sharded_greedy_two_ranks_gloo(tmp_path):
    import subprocess
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'repo': REPO})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                          '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert 'SHARDED_OK' in out.stdout
