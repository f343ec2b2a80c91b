"""BASELINE config 5's loop: 200 planning steps that each append ~30 rows to the train set through
algp_factorize_update + algp_solve_candidates_update + 4 lazily resolved picks -- once at N0 = 20 000 x 12 000 candidates
(checked every 50th step) and once at config 5's own size, N0 = 50 000 x 100 000 candidates on one GPU (L 20 GB + V^T 45 GB
resident; checked at steps 100 and 200).  At a check the carried state (row sums acc3, u / w vectors, appended rows of L,
lazily refreshed V^T) is compared with a from-scratch context on the same train set: posterior mean / variance,
log-determinant and picks (the reference refactorises from scratch at every step, agent.py:210, 295).
That comparison is HIP against HIP, so the smaller case is also anchored to the ORACLE at its last step: the incremental
state's posterior mean / variance of 256 sampled candidates and its log-determinant against O.posterior_chol -- a NumPy /
LAPACK from-scratch factorisation of the same ~26 000-row train set (agent.py:210 -> utils.py:293-319), <= 1e-8 relative.
At config 5's size (a 56 000-row CPU factorisation would take minutes) the anchor is the Schur-complement identity
that tests/test_full_size.py checks against SciPy at that size."""
import numpy as np
import pytest

from algp_amd import _hip
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

HYP = O.Hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))


def _ctx(pool):
    c = _hip.Context(np.float64)
    c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
    c.set_pool(pool)
    return c


@pytest.mark.parametrize('R,C,M,check_every', [(160, 125, 12000, 50), (250, 200, 100000, 100)],
                         ids=['n20000_m12000', 'c5_n50000_m100000'])
def test_200_incremental_steps_do_not_drift(R, C, M, check_every):
    rng = np.random.RandomState(11)
    grid, field = O.generate_gaussian_data(R, C, k=5, rng=rng)    # R x C grid sites
    grid = grid.astype(np.float64)
    N0 = len(grid)
    cw = int(np.ceil(np.sqrt(M * C / R)))
    ch = int(np.ceil(M / cw))
    ii, jj = np.meshgrid(np.arange(ch), np.arange(cw), indexing='ij')
    cand_xy = np.vstack([(ii.ravel() + 0.37) * (R / ch), (jj.ravel() + 0.41) * (C / cw)]).T[:M]
    cand_xy = cand_xy + 0.03 * rng.standard_normal(cand_xy.shape)          # generic positions: no exact lattice ties
    pool = np.vstack([grid, cand_xy])
    cidx = np.arange(N0, N0 + M)

    def truth(xy):                                                # the field at arbitrary sites (for appended targets)
        return np.exp(-((xy[:, 0] - R / 2) ** 2 + (xy[:, 1] - C / 2) ** 2) / 800.0)

    idx = np.arange(N0)
    var = np.where(rng.uniform(size=N0) < 0.5, 0.01, 1.0)
    y = np.maximum(field + rng.standard_normal(N0) * np.sqrt(var), 0.0)
    static = np.zeros(len(pool), bool)
    static[:N0] = var == 0.01
    c = _ctx(pool)
    checked = 0
    for step in range(1, 201):
        inc = step > 1 or M > 50000            # (the large case asks for row-stride headroom from its first step, as Agent does)
        c.set_train(idx, y, var)
        kept = c.factorize(incremental=inc)
        c.set_candidates(cidx, prior_includes_noise=True)
        c.solve_candidates(incremental=inc, alive=~static[cidx])
        if step == 200 and M <= 50000:
            # oracle anchor: the state 199 incremental steps have carried, against a from-scratch CPU fit
            mu0, pv0 = c.posterior()
            ld0 = c.logdet()
            free = np.where(~np.isin(cidx, idx))[0]
            samp = free[rng.permutation(len(free))[:256]]
            ref = O.posterior_chol(HYP, pool[idx], y, pool[cidx[samp]], var, test_var=np.full(len(samp), HYP.noise))
            assert np.max(np.abs(mu0[samp] - ref['mu'])) <= 1e-8 * max(1.0, np.max(np.abs(ref['mu']))), 'mean vs oracle'
            assert np.max(np.abs(pv0[samp] - ref['var'])) <= 1e-8 * max(1.0, np.max(np.abs(ref['var']))), 'variance vs oracle'
            assert abs(ld0 - ref['logdet']) <= 1e-8 * abs(ref['logdet']), (ld0, ref['logdet'])
            del ref
        picks = c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4)
        if inc and step > 1:
            assert kept >= (len(idx) - 64) // 128 * 128 - 128, (step, kept, len(idx))    # the prefix really is reused
        if step % check_every == 0:
            mu, pv = c.posterior()
            ld = c.logdet()
            f = _ctx(pool)
            f.set_train(idx, y, var)
            f.factorize()
            f.set_candidates(cidx, prior_includes_noise=True)
            f.solve_candidates(alive=~static[cidx])
            fpicks = f.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 4)
            fmu, fpv = f.posterior()
            fld = f.logdet()
            f.close()
            assert [int(p) for p in picks] == [int(p) for p in fpicks], (step, picks, fpicks)
            assert np.max(np.abs(mu - fmu)) <= 1e-8 * max(1.0, np.max(np.abs(fmu))), step
            assert np.max(np.abs(pv - fpv)) <= 1e-8 * max(1.0, np.max(np.abs(fpv))), step
            assert abs(ld - fld) <= 1e-8 * abs(fld), (step, ld, fld)
            checked += 1
        # the planning step's samples: the 4 picks (static) + ~26 mobile readings at fresh candidate sites
        static[picks] = True
        mob = cidx[rng.permutation(M)[:26]]
        mob = mob[~np.isin(mob, idx) & ~np.isin(mob, picks)]
        new = np.r_[np.asarray(picks, dtype=np.int64), mob]
        idx = np.r_[idx, new]
        nv = np.r_[np.full(len(picks), 0.01), np.full(len(mob), 1.0)]
        var = np.r_[var, nv]
        y = np.r_[y, np.maximum(truth(pool[new]) + rng.standard_normal(len(new)) * np.sqrt(nv), 0.0)]
    assert checked == 200 // check_every
    assert len(idx) > N0 + 200 * 20
    c.close()
