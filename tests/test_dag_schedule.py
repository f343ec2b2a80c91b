"""The ticket list of the one-launch Cholesky (chol_dag.hip: dag_build_schedule) replayed on the host, no GPU needed:
for every matrix size the kernel serves (8 <= N/128 <= 192), the window the library ships (W = 2) and three residencies
(512, 64 and 11 workgroups) the
list, executed strictly in ticket order by ONE bulk worker beside the chain team, finds every task's inputs already
produced, applies every update exactly once in ascending k and completes the factorisation (tools/dag_sched_probe.hip
restates the kernel's waits and publishes independently of the builder's graph).  A list that passes cannot deadlock at
any residency or dispatch order: the lowest unfinished ticket can always run.  (the invariant was only
argued, nothing pinned it against a retune of the schedule's constants.)  Replaces nothing in the reference -- the
factorisation itself replaces the LU inside np.linalg.inv / slogdet, utils.py:193, 300."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ticket_list_is_a_valid_sequential_order_for_every_size(tmp_path):
    exe = str(tmp_path / 'dag_sched_probe')
    src = os.path.join(REPO, 'tools', 'dag_sched_probe.hip')
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-std=c++17', '-w', src, '-o', exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, '--check', '8', '192'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'CHECK OK: 555 schedules (N/128 = 8..192, W = 2)' in r.stdout, (r.stdout[-2000:], r.stderr[-1000:])
    # lists that carry a row panel below the factor (the candidates' rows of algp_fit_and_solve, the identity that becomes
    # L^-T), with and without the factorisation's own tasks: same replay, panel rows starting at their first column
    r = subprocess.run([exe, '--check-panel', '8', '40'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'CHECK OK: 2574 panel schedules' in r.stdout, (r.stdout[-2000:], r.stderr[-1000:])
    # the panel at its shipped maximum (400 tile rows = 51 200 candidates) under config 4's factor, folded and alone, and the
    # identity panel of a fit iteration at N = 10 000
    # (round 6: and with the panel's tile rows -- all but the one that carries z, or all -- leaving their last column tile to the tail kernel)
    for shape in (('79', '400', '1', '0'), ('79', '400', '1', '1'), ('79', '79', '2', '0'), ('79', '80', '2', '0'), ('192', '193', '2', '0'),
                  ('79', '98', '1', '0', '97'), ('79', '400', '1', '0', '399'), ('79', '79', '1', '0', '79')):
        r = subprocess.run([exe, '--check-shape'] + list(shape), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and 'CHECK OK: shape' in r.stdout, (shape, r.stdout[-2000:], r.stderr[-1000:])
