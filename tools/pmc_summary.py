"""Summarise rocprofv3 --pmc passes of bench.py per kernel symbol (profiles/r02_pmc_by_kernel.json).

usage: python tools/pmc_summary.py <out.json> <stats_dir> <pmc_dir> [<pmc_dir> ...]
  stats_dir : a `rocprofv3 --kernel-trace --stats` run of the same command (per-symbol calls / average duration)
  pmc_dir   : runs with --pmc (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE | FETCH_SIZE | WRITE_SIZE), in own passes
Derived per symbol (MI355X_MICROARCH.md): MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs);
HBM bytes = 2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B; calibrated in round 6 against known byte counts in this library's own
request patterns -- 64 / 128 / 256 / 1024-byte row pieces, plain and LDS-DMA: 2.000 each, profiles/r06_fetch_calibration.txt) + WRITE_SIZE, both in KiB;
GB/s = bytes / the launches' own durations in the counter pass (counter passes serialise dispatches that overlap on several
streams in the stats run)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r'algp::(\w+)(<[^>]*>)?', name)
    if not m:
        return name[:60]
    s = m.group(1)
    if m.group(2) and ('double' in m.group(2) or 'float' in m.group(2)):
        s += '<%s>' % ('double' if 'double' in m.group(2) else 'float')
    return s


def main():
    out, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    res = defaultdict(dict)
    f = glob.glob(stats_dir + '/**/*kernel_stats.csv', recursive=True)
    if f:
        for r in csv.DictReader(open(f[0])):
            k = short(r['Name'])
            d = res[k]
            d['calls'] = d.get('calls', 0) + int(r['Calls'])
            d['total_ms'] = d.get('total_ms', 0.0) + float(r['TotalDurationNs']) * 1e-6
    for k, d in res.items():
        d['avg_launch_ms'] = d['total_ms'] / max(1, d['calls'])
    for pd in pmc_dirs:
        for f in glob.glob(pd + '/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                c = r['Counter_Name']
                d = res[k]
                d[c] = d.get(c, 0.0) + float(r['Counter_Value'])
                d['n_' + c] = d.get('n_' + c, 0) + 1
                # the launch's own duration in THIS pass (counter passes serialise the dispatches: launches that overlap on
                # several streams in the stats run do not here)
                d['ms_' + c] = d.get('ms_' + c, 0.0) + (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
    for k, d in res.items():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and d.get('GRBM_GUI_ACTIVE', 0) > 0:
            d['mfma_busy_pct'] = 100.0 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8.0 * 256 * 4)
        if 'FETCH_SIZE' in d or 'WRITE_SIZE' in d:
            fb = 2.0 * 1024.0 * d.get('FETCH_SIZE', 0.0) / max(1, d.get('n_FETCH_SIZE', 1))
            wb = 1024.0 * d.get('WRITE_SIZE', 0.0) / max(1, d.get('n_WRITE_SIZE', 1))
            d['hbm_bytes_per_launch'] = fb + wb
            d['fetch_bytes_per_launch_x2'] = fb
            d['write_bytes_per_launch'] = wb
            gbs = 0.0
            if d.get('ms_FETCH_SIZE'):
                gbs += 2.0 * 1024.0 * d['FETCH_SIZE'] / (d['ms_FETCH_SIZE'] * 1e-3) / 1e9
            if d.get('ms_WRITE_SIZE'):
                gbs += 1024.0 * d['WRITE_SIZE'] / (d['ms_WRITE_SIZE'] * 1e-3) / 1e9
            d['hbm_gb_per_s'] = gbs
            d['avg_launch_ms_in_counter_pass'] = d.get('ms_FETCH_SIZE', d.get('ms_WRITE_SIZE', 0.0)) / max(1, d.get('n_FETCH_SIZE', d.get('n_WRITE_SIZE', 1)))
    json.dump({'by_kernel': res,
               'note': 'counters summed over the dispatches of a symbol in the profiled bench command; mfma_busy_pct and the GB/s use '
                       'the formulas in the header of tools/pmc_summary.py'}, open(out, 'w'), indent=1, sort_keys=True)
    for k, d in sorted(res.items(), key=lambda kv: -kv[1].get('total_ms', 0)):
        print('%-42s calls %6d avg %9.4f ms  mfma %5s %%  hbm %8s GB/s' % (k, d.get('calls', 0), d.get('avg_launch_ms', 0),
              ('%.1f' % d['mfma_busy_pct']) if 'mfma_busy_pct' in d else '-', ('%.0f' % d['hbm_gb_per_s']) if 'hbm_gb_per_s' in d else '-'))


if __name__ == '__main__':
    main()
