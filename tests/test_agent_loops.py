"""The agent's mission loops (reference agent.py:125-287 run_ipp / run_greedy_ipp, :405-473 run_naive) executed end to end.

The loops only orchestrate: greedy picks -> waypoints -> the environment's path planner -> best_path -> samples ->
predict.  The reference's planner (env.py / map.py / graph_utils.py: integer graph search) is out of scope,
so the test drives the loops with a stand-in that has the planner's INTERFACE and trivial contents: every map cell is a
field cell, a path is a chain of L-shaped Manhattan legs through the waypoints in some order, its cost is its length.
What is checked is the bookkeeping the loops are there for: paths are connected, every pick becomes a static sample, every
cell passed becomes a mobile sample, the models predict after every run and the error series has the right length.
parity unpinned: the reference holds no recorded output of these loops (they need its planner and a data file)."""
import itertools
import types

import numpy as np
import pytest

from algp_amd.agent import Agent
from algp_amd.arguments import get_args
from algp_amd.field import SyntheticField

pytestmark = pytest.mark.gpu


def _leg(a, b):
    """cells after `a` up to and including `b`: rows first, then columns"""
    out, (r, c) = [], a
    while r != b[0]:
        r += 1 if b[0] > r else -1
        out.append((r, c))
    while c != b[1]:
        c += 1 if b[1] > c else -1
        out.append((r, c))
    return out


class ManhattanField(SyntheticField):
    """SyntheticField + the planner half of the reference's FieldEnv interface (env.py:115-421), Manhattan style."""

    def __init__(self, rows, cols, num_test):
        super().__init__(rows, cols, num_test=num_test)
        self.map = types.SimpleNamespace(shape=(rows, cols), nearest_waypoint_path_cost=self._nearest)

    @staticmethod
    def _dist(a, b):
        return abs(a[0] - b[0]) + abs(a[1] - b[1])

    def get_heuristic_cost(self, pose, heading, waypoints):
        cost, cur = 0, pose
        for w in waypoints:
            cost, cur = cost + self._dist(cur, w), w
        return cost

    def get_path_from_checkpoints(self, checkpoints):
        path = [tuple(checkpoints[0])]
        for nxt in checkpoints[1:]:
            path += _leg(path[-1], tuple(nxt))
        return [np.array(p) for p in path]

    def get_all_paths(self, pose, heading, waypoints, upper_bound, slack=0):
        """(checkpoints, gp indices of the cells each path passes, costs) of every visiting order within the bound."""
        checkpoints, indices, costs = [], [], []
        for order in itertools.islice(itertools.permutations(waypoints), 6):
            cps = [tuple(pose)] + list(order)
            cells = [tuple(p) for p in self.get_path_from_checkpoints(cps)][1:]
            cost = len(cells)
            if cost > upper_bound + slack and checkpoints:
                continue
            gi = [self.map_pose_to_gp_index_matrix[c] for c in cells]
            checkpoints.append(cps)
            indices.append([g for g in gi if g is not None and tuple(self.gp_index_to_map_pose(g)) not in waypoints])
            costs.append(cost)
        return checkpoints, indices, costs

    def _nearest(self, pose, heading, waypoints, return_seq=False):
        left, cur, costs, seq = list(range(len(waypoints))), tuple(pose), [], []
        while left:
            j = min(left, key=lambda q: self._dist(cur, waypoints[q]))
            costs.append(self._dist(cur, waypoints[j]))
            seq.append(j)
            cur = waypoints[j]
            left.remove(j)
        return (costs, seq) if return_seq else costs


def _make(seed=0, rows=14, cols=12):
    np.random.seed(seed)
    args = get_args([])
    args.kernel, args.max_iterations, args.num_samples_per_batch = 'rbf', 30, 3
    env = ManhattanField(rows, cols, num_test=24)
    agent = Agent(env, args, static_std=args.static_std, mobile_std=10 * args.static_std)
    agent.reset()
    return env, agent


def _connected(path):
    d = np.abs(np.diff(np.asarray(path), axis=0)).sum(axis=1)
    return bool(np.all(d == 1))


@pytest.mark.parametrize('strategy', ['MaxEnt', 'Shortest', 'Equi-Sample'])
def test_run_ipp_bookkeeping(strategy):
    env, agent = _make(seed=1)
    out = agent.run_ipp(num_runs=3, criterion='entropy', strategy=strategy, disp=False)
    assert len(out['error']) == 3 and np.all(np.isfinite(out['error'])) and len(out['mean']) == len(env.test_X)
    assert _connected(agent.path) and len(agent.static_locations) == 9
    static, mobile = agent._masks()
    # every pick was sampled with the static sensor -- except a pick that is the cell the vehicle stands on when the
    # batch is planned: the path to follow starts AFTER that cell (agent.py:181, same in the reference), at most one per run
    missed = [tuple(p) for p in agent.static_locations if not static[env.map_pose_to_gp_index_matrix[tuple(p)]]]
    assert len(missed) <= 3 and all(any(np.array_equal(m, q) for q in agent.path) for m in missed), missed
    passed = {env.map_pose_to_gp_index_matrix[tuple(p)] for p in agent.path[1:]} - {None}
    sampled = set(np.where(static | mobile)[0].tolist())
    assert passed <= sampled                                            # every field cell passed has a reading
    assert len(agent.collected['ind']) == len(agent.path) - 1             # one entry per cell moved (-1 = held-out cell)
    assert sum(1 for g in agent.collected['ind'] if g != -1) == \
        sum(1 for p in agent.path[1:] if env.map_pose_to_gp_index_matrix[tuple(p)] is not None)


def test_run_greedy_ipp_and_mutual_information_criterion():
    env, agent = _make(seed=2)
    out = agent.run_greedy_ipp(num_runs=2, criterion='entropy', disp=False)
    assert len(out['error']) == 1 and np.isfinite(out['error'][0]) and _connected(agent.path)
    assert len(agent.static_locations) == 6
    env2, agent2 = _make(seed=2)
    out2 = agent2.run_ipp(num_runs=2, criterion='mutual_information', disp=False)
    assert len(out2['error']) == 2 and np.all(np.isfinite(out2['error']))


@pytest.mark.parametrize('metric', ['distance', 'samples'])
def test_run_naive_lawn_mower(metric):
    env, agent = _make(seed=3)
    out = agent.run_naive(agent.mobile_std, [10, 10, 10], metric=metric)
    assert len(out['error']) == 3 and len(out['mi']) == 3 and np.all(np.isfinite(out['mi']))
    assert np.all(np.diff(out['mean_var']) <= 1e-9)                      # more samples never raise the mean predictive variance
    assert len(agent.path) >= 31
