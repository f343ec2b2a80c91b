"""Candidate scoring sharded over the GPUs of one node (one process per GPU).

The reference's greedy loop (agent.py:313-354) evaluates every candidate independently given
the factor of the sampled set, so candidates shard embarrassingly:

  * every rank holds the full factor L of the train set (factorised redundantly, no comms),
  * rank r owns a contiguous slice of the candidate list and its V^T rows,
  * per pick: ONE all-gather (RCCL over xGMI when the backend is "nccl") -> the same first-max
    argmax on every rank (np.argmax semantics, agent.py:349) -> every rank commits the global
    winner to its own shard.  A rank that does not own the winner rebuilds the winner's row from
    the replicated factor (algp_commit_pick), so no second collective is needed.
    What is gathered: with the entropy criterion each rank resolves its own best candidate lazily
    (algp_best_candidate: only the rows whose upper bound can still win are brought up to date) and
    contributes the pair (utility, global position) -- 16 bytes per rank; otherwise (MI, or a
    backend without best_candidate) the per-shard score vectors as before.

``backend`` is any object with the `_hip.Context` scoring surface (scores / commit_pick / M);
tests drive the same logic over gloo with a CPU stand-in backend.

Two routes exist for the exchange.  On a GPU node the one to use is behind the C ABI:
`_hip.Context.comm_init` + `greedy_sharded` (include/algp_hip.h: algp_comm_init / algp_greedy_sharded) -- the library
packs each rank's (utility, global position), calls RCCL's all-gather on its own stream, takes the first maximum and
commits the winner on the device, one 16-byte read-back per pick, no host arithmetic between kernels (what
`bench.py --collective abi`, the default, times).  This module is the torch.distributed route (`bench.py --collective
torch`, and the gloo route the CPU tests cover): same partition, same tie-break, the collective issued by the caller.
"""
import numpy as np


def partition(n_items, world_size):
    """Contiguous, balanced split: the first (n % w) ranks get one extra item."""
    base, rem = divmod(int(n_items), int(world_size))
    counts = [base + (1 if r < rem else 0) for r in range(world_size)]
    offs = np.concatenate([[0], np.cumsum(counts)])
    return [(int(offs[r]), int(offs[r + 1])) for r in range(world_size)]


class ShardLink(object):
    """How an Agent's pool context joins the other ranks of a sharded run (one process and one context per GPU): the
    rank's place in the job, the transport of the library's collectives, and which rank holds which pool site as a
    candidate.  `Agent(env, args, comm=ShardLink(...))` then scores only its share of the pool in `greedy`
    (algp_greedy_sharded: one all-gather per pick) and its factor updates take the rows of the new train sites from
    their owners (algp_comm_set_owners: one all-gather per planning step) -- agent.py:125-229 with the loop of
    agent.py:313-354 cut into shards; every rank ends each step with the same picks and the same factor.

    unique_id: the 128 bytes of `_hip.Context.comm_unique_id()` (RCCL over xGMI; one rank creates them, the caller
    hands them to the others), or all_gather: a callable bytes -> bytes concatenating every rank's bytes in rank order
    (MPI, gloo, shared memory; also what lets two ranks share one card).
    layout: 'strided' (site q on rank q mod n: a path's neighbouring sites spread over all owners, so a step's row
    exchange carries ~1/n of the new rows per rank, and retired static sites thin every shard alike) or 'contiguous'
    (rank r owns `partition(n_pool, n)[r]`).  Picks are the same either way: equal utilities go to
    the smaller pool index, which is np.argmax's first maximum (agent.py:349)."""

    def __init__(self, rank, world_size, unique_id=None, all_gather=None, layout='strided'):
        if (unique_id is None) == (all_gather is None):
            raise ValueError('give exactly one transport: unique_id (RCCL) or all_gather (host)')
        if layout not in ('strided', 'contiguous'):
            raise ValueError("layout must be 'strided' or 'contiguous'")
        self.rank, self.world_size = int(rank), int(world_size)
        self.unique_id, self.all_gather, self.layout = unique_id, all_gather, layout
        self._id_used = False

    def owners(self, n_pool):
        if self.layout == 'strided':
            return (np.arange(n_pool) % self.world_size).astype(np.int32)
        own = np.empty(n_pool, dtype=np.int32)
        for r, (lo, hi) in enumerate(partition(n_pool, self.world_size)):
            own[lo:hi] = r
        return own

    def mine(self, n_pool):
        """Pool sites this rank holds as candidates, ascending."""
        return np.nonzero(self.owners(n_pool) == self.rank)[0].astype(np.int64)

    def attach(self, ctx, n_pool):
        """Join `ctx` (its pool already set) to the job: the transport ONCE per context, then the owner map.

        A context keeps its communicator across pool reloads (algp_set_pool drops only the owner map), and an RCCL
        unique id is single-use -- the root's bootstrap listener is gone after the first ncclCommInitRank -- so a second
        attach of the same context (the Agent reloads its pool whenever the hyper-parameters change or another caller
        has loaded one in between) must only re-send the map.  A context that needs a NEW communicator needs a new
        ShardLink with a fresh id from rank 0."""
        if getattr(ctx, '_shard_link', None) is not self:
            if self.unique_id is not None:
                if self._id_used:
                    raise RuntimeError('ShardLink: this RCCL unique id has already initialised a communicator; '
                                       'create a new ShardLink with a fresh id (rank 0: Context.comm_unique_id())')
                ctx.comm_init(self.world_size, self.rank, self.unique_id)
                self._id_used = True
            else:
                ctx.comm_init_host(self.world_size, self.rank, self.all_gather)
            ctx._shard_link = self
        ctx.comm_set_owners(self.owners(n_pool))


class LocalComm(object):
    """world_size == 1: no collective."""
    rank, world_size = 0, 1
    device_buffers = False

    def all_gather(self, local, max_len):
        return local[None, :]

    def all_gather_pairs(self, value, position):
        return np.array([[value, position]], dtype=np.float64)


class TorchComm(object):
    """torch.distributed all-gather.  With the "nccl" backend (= RCCL on ROCm) the score
    vectors stay on the GPU: the library writes them straight into the send buffer."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
        self.device_buffers = dist.get_backend() == 'nccl'
        self.device = device
        self._send = self._recv = None

    def buffers(self, max_len):
        t = self.torch
        if self._send is None or self._send.numel() != max_len:
            dev = self.device if self.device_buffers else 'cpu'
            self._send = t.full((max_len,), float('-inf'), dtype=t.float64, device=dev)
            self._recv = t.empty((self.world_size, max_len), dtype=t.float64, device=dev)
        return self._send, self._recv

    def all_gather(self, local, max_len):
        """Returns the gathered (world, max_len) score matrix: a torch tensor on the GPU with the
        nccl backend (it never visits the host), a NumPy array otherwise."""
        send, recv = self.buffers(max_len)
        if local is not None:                       # host scores (gloo)
            send.fill_(float('-inf'))
            send[:len(local)] = self.torch.from_numpy(np.ascontiguousarray(local))
        self.dist.all_gather_into_tensor(recv.view(-1), send)
        return recv if self.device_buffers else recv.numpy()

    def all_gather_pairs(self, value, position):
        """(world, 2) array of every rank's (value, position); exact for positions < 2**53."""
        t = self.torch
        dev = self.device if self.device_buffers else 'cpu'
        if getattr(self, '_pair_send', None) is None:
            self._pair_send = t.empty(2, dtype=t.float64, device=dev)
            self._pair_recv = t.empty((self.world_size, 2), dtype=t.float64, device=dev)
        self._pair_send.copy_(t.tensor([float(value), float(position)], dtype=t.float64))
        self.dist.all_gather_into_tensor(self._pair_recv.view(-1), self._pair_send)
        return self._pair_recv.cpu().numpy()

    def first_max(self, g, parts):
        """(global position, value) of the first maximum in rank order (np.argmax semantics,
        agent.py:349) of the gathered scores; on the GPU for device buffers (two scalars come back)."""
        if not self.device_buffers:
            return _first_max_numpy(g, parts)
        t = self.torch
        flat = g.view(-1)
        m = flat.max()
        pos = int(t.nonzero(flat == m)[0].item()) if not bool(t.isnan(m)) else 0
        r, j = divmod(pos, g.shape[1])              # padding is -inf and only wins when everything is -inf
        lo, hi = parts[r]
        if j >= hi - lo:                            # every score is -inf: fall back to the first candidate
            r = next(k for k, (l, h) in enumerate(parts) if h > l)
            lo, j = parts[r][0], 0
        return lo + j, float(m.item())


def _first_max_numpy(g, parts):
    best_v, best_pos = -np.inf, -1
    for r, (lo, hi) in enumerate(parts):                # first maximum in global order
        if hi > lo:
            seg = np.asarray(g[r, :hi - lo])
            j = int(np.argmax(seg))
            if best_pos < 0 or seg[j] > best_v:
                best_v, best_pos = float(seg[j]), lo + j
    return best_pos, best_v


class ShardedGreedy(object):
    """k greedy picks over a sharded candidate list.

    all_cand_idx: the GLOBAL candidate list (pool indices) in rank order; rank r owns
    all_cand_idx[lo_r:hi_r] per ``partition``.  The backend must already hold exactly that
    slice as its candidates (set_candidates + solve_candidates done).
    """

    def __init__(self, backend, comm, all_cand_idx, lazy=True):
        self.b = backend
        self.lazy = lazy                              # entropy criterion (0): gather (utility, position) pairs
        self.comm = comm
        self.all_idx = np.ascontiguousarray(all_cand_idx, dtype=np.int64)
        self.parts = partition(len(self.all_idx), comm.world_size)
        self.lo, self.hi = self.parts[comm.rank]
        self.max_len = max(h - l for l, h in self.parts)
        if getattr(backend, 'M', self.hi - self.lo) != self.hi - self.lo:
            raise ValueError('backend holds %d candidates, this rank owns %d' % (backend.M, self.hi - self.lo))

    def _gather_scores(self, criterion, static_std, mobile_std):
        if isinstance(self.comm, LocalComm):
            return self.b.scores(criterion, static_std, mobile_std)[None, :]
        if self.comm.device_buffers:
            send, _ = self.comm.buffers(self.max_len)
            self.b.scores(criterion, static_std, mobile_std, out_device_ptr=send.data_ptr())
            return self.comm.all_gather(None, self.max_len)
        return self.comm.all_gather(self.b.scores(criterion, static_std, mobile_std), self.max_len)

    def _step_pairs(self, criterion, static_std, mobile_std):
        """Each rank resolves its own best candidate; the winners are compared, first maximum in rank order."""
        if self.hi > self.lo:
            pos, _, val = self.b.best_candidate(criterion, static_std, mobile_std)
            if pos < 0 or val != val:
                pos, val = 0, -np.inf
        else:
            pos, val = -1, -np.inf                    # a rank without candidates never wins
        pairs = self.comm.all_gather_pairs(val, self.lo + pos if pos >= 0 else -1)
        best_pos, best_v = -1, -np.inf
        for r in range(len(pairs)):
            v, gp = float(pairs[r, 0]), int(pairs[r, 1])
            if gp >= 0 and (best_pos < 0 or v > best_v or (v == best_v and gp < best_pos)):
                best_v, best_pos = v, gp
        return best_pos, best_v

    def step(self, criterion, static_std, mobile_std):
        """One pick: returns (pool index of the winner, its utility)."""
        if self.lazy and criterion == 0 and hasattr(self.b, 'best_candidate') and hasattr(self.comm, 'all_gather_pairs'):
            best_pos, best_v = self._step_pairs(criterion, static_std, mobile_std)
            winner = int(self.all_idx[best_pos])
            self.b.commit_pick(winner, static_std, mobile_std)
            return winner, best_v
        g = self._gather_scores(criterion, static_std, mobile_std)
        if hasattr(self.comm, 'first_max') and not isinstance(g, np.ndarray):
            best_pos, best_v = self.comm.first_max(g, self.parts)
        else:
            best_pos, best_v = _first_max_numpy(g, self.parts)
        winner = int(self.all_idx[best_pos])
        self.b.commit_pick(winner, static_std, mobile_std)
        return winner, best_v

    def greedy(self, criterion, static_std, mobile_std, k):
        picks, vals = [], []
        for _ in range(k):
            w, v = self.step(criterion, static_std, mobile_std)
            picks.append(w)
            vals.append(v)
        return picks, vals
