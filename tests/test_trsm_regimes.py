"""The candidate solve V^T = B^T L^-T picks its order by the number of candidate rows: up to 4 096 rows right-looking with
K = 128 steps (potrf.hip), up to 51 200 rows ONE launch of the task list without the factorisation's own tasks
(chol_dag.hip; $ALGP_SOLVE_DAG=0: right-looking over 512-column blocks on two streams, the "push"), beyond that
left-looking in row chunks on three streams.  Every regime, at its edges, against the oracle's
posterior (utils.py:293-319 as O.posterior_chol) on sampled candidates, in fp64 and fp32 -- and the regimes against each
other: a candidate's posterior must not depend on how many other candidates were solved with it."""
import numpy as np
import pytest

from algp_amd import _hip
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

HYP = O.Hypers(np.log([3.0, 3.0]), 0.0, np.log(1e-2))
N = 1400                                   # train rows: 11 tiles of 128, the last one ragged


def _setup(dtype, M, rng, side=40):
    xx, yy = np.meshgrid(np.arange(side), np.arange(side))
    grid = np.vstack([yy.ravel(), xx.ravel()]).T.astype(np.float64)
    A = np.sort(rng.permutation(len(grid))[:N])
    cand = rng.uniform(0, side, (M, 2))
    pool = np.vstack([grid, cand])
    var = rng.choice([0.01, 1.0], N)
    y = rng.uniform(0, 1, N)
    c = _hip.Context(dtype)
    c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
    c.set_pool(pool)
    c.set_train(A, y, var)
    c.factorize()
    return c, pool, A, y, var, np.arange(len(grid), len(grid) + M)


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-9), (np.float32, 2e-3)], ids=['f64', 'f32'])
def test_every_solve_order_matches_the_oracle_and_the_others(dtype, tol):
    rng = np.random.RandomState(5)
    sizes = [4096, 4097, 12500, 51200, 51201]          # last of the short order, first / middle / last of the task list, first of the chunks
    c, pool, A, y, var, cidx = _setup(dtype, max(sizes), rng)
    samp = np.sort(rng.permutation(4096)[:192])       # candidates that every size contains
    ref = O.posterior_chol(HYP, pool[A], y, pool[cidx[samp]], var)
    seen = {}
    for M in sizes:
        c.set_candidates(cidx[:M], prior_includes_noise=False)
        c.solve_candidates()
        mu, pv = c.posterior()
        assert mu.shape == (M,)
        assert np.max(np.abs(mu[samp] - ref['mu'])) <= tol * max(1.0, np.max(np.abs(ref['mu']))), (M, 'mean vs oracle')
        assert np.max(np.abs(pv[samp] - ref['var'])) <= tol * max(1.0, np.max(np.abs(ref['var']))), (M, 'variance vs oracle')
        assert np.all(np.isfinite(mu)) and np.all(np.isfinite(pv))
        assert np.min(pv) > -tol
        seen[M] = (mu[samp].copy(), pv[samp].copy())
    loose = 2e-11 if dtype == np.float64 else 2e-4       # the orders against each other: the same products summed in another order
    for M in sizes[1:]:
        assert np.max(np.abs(seen[M][0] - seen[sizes[0]][0])) <= loose
        assert np.max(np.abs(seen[M][1] - seen[sizes[0]][1])) <= loose
    c.close()


def test_row_chunk_counts_give_the_same_solution():
    """algp_debug_set_trsm_chunks (ALGP_TRSM_CHUNKS): one to four row-chunk streams of the left-looking order solve the
    same rows with the same arithmetic -- bit-identical posteriors."""
    rng = np.random.RandomState(6)
    M = 51500                                  # beyond the task list's 400 tile rows
    c, pool, A, y, var, cidx = _setup(np.float64, M, rng)
    c.set_candidates(cidx, prior_includes_noise=False)
    out = []
    for chunks in (1, 2, 3, 4):
        c.set_trsm_chunks(chunks)
        c.solve_candidates()
        out.append(c.posterior())
    c.set_trsm_chunks(0)
    for mu, pv in out[1:]:
        assert np.array_equal(mu, out[0][0]) and np.array_equal(pv, out[0][1])
    c.close()


def test_row_statistics_come_out_of_the_solves_own_launches(monkeypatch):
    """Beyond the task list's range the launch that writes a column tile of V^T for the last time also leaves the tile's row
    sums of v^2 and v z (gemm.hip, STATS), so variance and mean (utils.py:301-304) need no second pass over V^T: against the
    pass ($ALGP_ROW_STATS=0) to rounding, in both precisions, and by the bytes the library books for the statistics."""
    for dtype, tol in ((np.float64, 1e-12), (np.float32, 2e-5)):
        rng = np.random.RandomState(9)
        M = 51500
        c, pool, A, y, var, cidx = _setup(dtype, M, rng)
        c.set_candidates(cidx, prior_includes_noise=False)
        c.prof_enable(True)
        out, booked = [], []
        for flag in ('1', '0'):
            monkeypatch.setenv('ALGP_ROW_STATS', flag)
            c.prof_reset()
            c.solve_candidates()
            booked.append(c.prof_get('rows')['bytes'])
            out.append(c.posterior())
        monkeypatch.delenv('ALGP_ROW_STATS')
        c.prof_enable(False)
        assert booked[0] < 0.05 * booked[1]                  # 2 x 11 tiles of sums per row against the 1 408 columns of V^T
        assert np.max(np.abs(out[0][0] - out[1][0])) <= tol * max(1.0, np.max(np.abs(out[1][0])))
        assert np.max(np.abs(out[0][1] - out[1][1])) <= tol
        c.close()


def test_explicit_inverses_of_the_512_column_blocks(monkeypatch):
    """The left-looking order solves inside a full 512-column block J >= 1 with ONE product against the block's explicit
    inverse (potrf.hip: build_inv512 from the factorisation's 128-block inverses, gemm.hip: kcut) instead of seven
    128-column launches ($ALGP_TRSM_INV512=0).  3 000 train rows = five full blocks + a ragged one: both ways against the
    oracle (utils.py:293-319) and against each other, in both precisions; fewer launches with the inverses."""
    global N
    old_n = N
    N = 3000
    try:
        for dtype, tol, loose in ((np.float64, 1e-9, 1e-10), (np.float32, 3e-3, 5e-4)):
            rng = np.random.RandomState(12)
            M = 51300
            side = 60
            xx, yy = np.meshgrid(np.arange(side), np.arange(side))
            grid = np.vstack([yy.ravel(), xx.ravel()]).T.astype(np.float64)
            A = np.sort(rng.permutation(len(grid))[:N])
            pool = np.vstack([grid, rng.uniform(0, side, (M, 2))])
            var = rng.choice([0.01, 1.0], N)
            y = rng.uniform(0, 1, N)
            cidx = np.arange(len(grid), len(grid) + M)
            c = _hip.Context(dtype)
            c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
            c.set_pool(pool)
            c.set_train(A, y, var)
            c.factorize()
            c.set_candidates(cidx, prior_includes_noise=False)
            samp = np.sort(rng.permutation(M)[:160])
            ref = O.posterior_chol(HYP, pool[A], y, pool[cidx[samp]], var)
            out, launches = [], []
            c.prof_enable(True)
            for flag in ('1', '0'):
                monkeypatch.setenv('ALGP_TRSM_INV512', flag)
                c.prof_reset()
                c.solve_candidates()
                launches.append(c.prof_get('gemm_trsm')['launches'])
                mu, pv = c.posterior()
                assert np.max(np.abs(mu[samp] - ref['mu'])) <= tol * max(1.0, np.max(np.abs(ref['mu']))), (dtype, flag)
                assert np.max(np.abs(pv[samp] - ref['var'])) <= tol, (dtype, flag)
                out.append((mu, pv))
            monkeypatch.delenv('ALGP_TRSM_INV512')
            c.prof_enable(False)
            assert launches[0] < launches[1] - 30, launches
            assert np.max(np.abs(out[0][0] - out[1][0])) <= loose * max(1.0, np.max(np.abs(out[1][0])))
            assert np.max(np.abs(out[0][1] - out[1][1])) <= loose
            c.close()
    finally:
        N = old_n


def test_push_order_six_blocks_deep_against_the_oracle_and_itself(monkeypatch):
    """the right-looking order over 512-column blocks (potrf.hip; the fallback of the task-list solve for train
    sets beyond its range, selected here with ALGP_SOLVE_DAG=0) rotates four pairs of events between its two streams; with
    N = 1 400 it is only three blocks deep.  Here 3 000 train rows = 6 blocks of 512 (event reuse, a helper stream several
    blocks behind): against the oracle, against a second run of itself (bit-identical: the streams only overlap
    launches that touch different columns) and against the task list (rounding)."""
    global N
    rng = np.random.RandomState(8)
    old_n = N
    N = 3000
    try:
        side = 60
        xx, yy = np.meshgrid(np.arange(side), np.arange(side))
        grid = np.vstack([yy.ravel(), xx.ravel()]).T.astype(np.float64)
        A = np.sort(rng.permutation(len(grid))[:N])
        M = 4097
        cand = rng.uniform(0, side, (M, 2))
        pool = np.vstack([grid, cand])
        var = rng.choice([0.01, 1.0], N)
        y = rng.uniform(0, 1, N)
        cidx = np.arange(len(grid), len(grid) + M)
        c = _hip.Context(np.float64)
        c.set_hypers(HYP.log_lengthscale, HYP.log_outputscale, HYP.log_noise)
        c.set_pool(pool)
        c.set_train(A, y, var)
        c.factorize()
        c.set_candidates(cidx, prior_includes_noise=False)
        c.solve_candidates()
        mu_dag, pv_dag = c.posterior()
        monkeypatch.setenv('ALGP_SOLVE_DAG', '0')
        c.prof_enable(True)
        c.prof_reset()
        c.solve_candidates()
        assert c.prof_get('dag_panel')['launches'] == 0 and c.prof_get('gemm_trsm')['launches'] > 30
        c.prof_enable(False)
        mu2, pv2 = c.posterior()
        c.solve_candidates()                                      # a second run of the two-stream order: the same bits (every column block
        mu1, pv1 = c.posterior()                                  # receives its pushes in ascending order whatever the streams' timing)
        assert np.array_equal(mu1, mu2) and np.array_equal(pv1, pv2)
        samp = np.sort(rng.permutation(M)[:200])
        ref = O.posterior_chol(HYP, pool[A], y, pool[cidx[samp]], var)
        assert np.max(np.abs(mu2[samp] - ref['mu'])) <= 1e-9 * max(1.0, np.max(np.abs(ref['mu'])))
        assert np.max(np.abs(pv2[samp] - ref['var'])) <= 1e-9
        assert np.max(np.abs(mu2 - mu_dag)) <= 2e-11 and np.max(np.abs(pv2 - pv_dag)) <= 2e-11
        c.close()
    finally:
        N = old_n


@pytest.mark.parametrize('n_train', [2064, 2100, 2112, 2113], ids=['r16', 'r52', 'r64', 'r65_not_narrow'])
def test_a_narrow_last_tile_is_solved_by_the_tail_kernel(monkeypatch, n_train):
    """Round 6: a train set that ends r <= 64 columns into its last 128-column tile (config 4: N = 10 000, r = 16) has those
    columns solved by the tail kernel -- as an append to a factor of N - r rows would -- and the full tiles by the chunked
    sweep, whose launches no longer pay a whole tile for them; the last tile's row statistics come from a 128-column
    reduction.  Against the oracle's posterior (utils.py:293-319) on sampled candidates and against the sweep alone
    ($ALGP_TAIL_COLS=0), for r = 16, 52, 64 and -- unchanged path -- 65."""
    global N
    rng = np.random.RandomState(n_train)
    old_n = N
    N = n_train
    try:
        M = 51300                                                  # > 400 tile rows: beyond the task list, the chunked left-looking sweep
        c, pool, A, y, var, cidx = _setup(np.float64, M, rng, side=50)
        c.set_candidates(cidx, prior_includes_noise=False)
        c.prof_enable(True)
        c.prof_reset()
        c.solve_candidates()
        tail_launches = c.prof_get('tail_cols')['launches']
        c.prof_enable(False)
        assert tail_launches == (1 if n_train % 128 <= 64 else 0), tail_launches
        mu, pv = c.posterior()
        samp = np.sort(rng.permutation(M)[:160])
        ref = O.posterior_chol(HYP, pool[A], y, pool[cidx[samp]], var)
        assert np.max(np.abs(mu[samp] - ref['mu'])) <= 1e-9 * max(1.0, np.max(np.abs(ref['mu'])))
        assert np.max(np.abs(pv[samp] - ref['var'])) <= 1e-9
        monkeypatch.setenv('ALGP_TAIL_COLS', '0')
        c.solve_candidates()
        mu0, pv0 = c.posterior()
        assert np.max(np.abs(mu - mu0)) <= 2e-11 and np.max(np.abs(pv - pv0)) <= 2e-11
        # the rows themselves: greedy picks (which read V^T, not only its sums) agree between the two routes
        c.set_candidates(cidx, prior_includes_noise=True)
        c.solve_candidates()
        p0 = [int(q) for q in c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 3)]
        monkeypatch.delenv('ALGP_TAIL_COLS')
        c.solve_candidates()
        p1 = [int(q) for q in c.greedy(_hip.CRIT_ENTROPY, 0.1, 1.0, 3)]
        assert p0 == p1
        c.close()
    finally:
        N = old_n
