"""profiles/r04_traffic_pmc.json from the per-kernel counter summaries (tools/collect_profiles.sh -> pmc_by_kernel.json):
HBM-side bytes per launch of the dominant kernel (the MFMA GEMM of the candidate solve) and of the one-launch Cholesky,
with the fingerprint of the kernel sources they were measured on -- bench.py reports `roofline.traffic` only when that
fingerprint matches the sources it runs.
  python tools/make_traffic_json.py <f64 pmc_by_kernel.json> <f32 pmc_by_kernel.json> <out.json>"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import importlib.util
spec = importlib.util.spec_from_file_location('bench', os.path.join(REPO, 'bench.py'))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

out = {'sources_sha16': bench.sources_sha16(),
       'commit': subprocess.run(['git', 'rev-parse', 'HEAD'], cwd=REPO, capture_output=True, text=True).stdout.strip() or None,
       'command': 'bench.py --dtype <dt> --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-emulation (7 candidate solves per run: 1 warm-up + 3 timed with the stage timers + 3 without)',
       'fetch_factor': 2.0,
       'fetch_factor_source': 'profiles/r06_fetch_calibration.txt (tools/fetch_calibrate.sh): FETCH_SIZE x 2 = the true byte count to 0.1 % for '
                              '64 / 128 / 256 / 1024-byte row pieces, plain loads and LDS-DMA, all three lane orders of the GEMM\'s 64-byte pieces',
       'note': 'bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): gfx950 tallies the 128-byte requests of wide streaming reads at 64 '
               'bytes (MI355X_MICROARCH.md, HBM; calibrated for this library\'s request patterns in round 6, see fetch_factor_source); '
               'separate --pmc passes; FETCH/WRITE count the L2\'s fabric-side requests, so '
               'Infinity-Cache hits are included.  WRITE_SIZE is calibrated for 16-byte-per-lane streaming stores; the Cholesky\'s '
               'tile stores are 8-byte (fp64) / 4-byte (fp32) write-through stores, its write figure is uncalibrated.'}
for dt, path in (('f64', sys.argv[1]), ('f32', sys.argv[2])):
    if not os.path.exists(path):
        continue
    bk = json.load(open(path))['by_kernel']
    sym = 'double' if dt == 'f64' else 'float'
    # the GEMM is two instantiations since round 4 (<T, false>; <T, true> also leaves the row statistics of the tile it
    # writes): every launch of the candidate solve counts, the per-launch figure is the mean over both
    gs = [v for k, v in bk.items() if k.startswith('gemm_nt_kernel_dma4<%s' % sym) and 'hbm_bytes_per_launch' in v]
    if gs:
        ns = [g.get('n_FETCH_SIZE', g.get('calls', 0)) for g in gs]
        total = sum(g['hbm_bytes_per_launch'] * n for g, n in zip(gs, ns))
        big = max(zip(gs, ns), key=lambda gn: gn[0]['hbm_bytes_per_launch'] * gn[1])[0]
        out['gemm_nt_%s_bytes_per_launch' % dt] = total / max(1, sum(ns))
        out['gemm_nt_%s_launches_counted' % dt] = sum(ns)
        out['bytes_per_solve_%s' % dt] = total / 7.0
        out['gemm_nt_%s_gb_per_s_in_counter_pass' % dt] = big.get('hbm_gb_per_s')
        out['gemm_nt_%s_mfma_busy_pct' % dt] = big.get('mfma_busy_pct')
    d = bk.get('chol_dag_kernel<%s>' % sym)
    if d and 'hbm_bytes_per_launch' in d:
        out['chol_dag_%s_bytes_per_launch' % dt] = d['hbm_bytes_per_launch']
        out['chol_dag_%s_fetch_bytes_x2' % dt] = d.get('fetch_bytes_per_launch_x2')
        out['chol_dag_%s_write_bytes' % dt] = d.get('write_bytes_per_launch')
        out['chol_dag_%s_mfma_busy_pct' % dt] = d.get('mfma_busy_pct')
        out['chol_dag_%s_avg_launch_ms_stats_pass' % dt] = d.get('avg_launch_ms')
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1))
