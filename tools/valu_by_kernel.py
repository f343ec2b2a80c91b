"""VALU issue share per kernel symbol from a rocprofv3 --pmc pass of bench.py (SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_INST_ANY): summed over a symbol's dispatches; valu_issue_pct = SQ_INSTS_VALU / (256 CUs x 4 SIMDs x 2.4 GHz / 4
cycles per wave64 instruction x the launches' own durations in that pass).
usage: python tools/valu_by_kernel.py <pmc_dir> <out.json>"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

res = defaultdict(lambda: defaultdict(float))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'algp::(\w+)(<[^>]*>)?', r['Kernel_Name'])
        k = m.group(1) if m else r['Kernel_Name'][:50]
        res[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_INSTS_VALU':
            res[k]['launches'] += 1
            res[k]['ms'] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
out = {'note': __doc__.split('usage')[0].strip(), 'by_kernel': {}}
for k, d in sorted(res.items(), key=lambda kv: -kv[1]['ms']):
    d = dict(d)
    cap = 256 * 4 * 2.4e9 / 4 * d.get('ms', 0) * 1e-3
    d['valu_issue_capacity_wave_insts'] = cap
    d['valu_issue_pct'] = 100.0 * d.get('SQ_INSTS_VALU', 0) / cap if cap else None
    out['by_kernel'][k] = d
json.dump(out, open(sys.argv[2], 'w'), indent=1)
for k, d in list(out['by_kernel'].items())[:8]:
    print('%-28s launches %5d  %9.3f ms  valu issue %5.1f %%' % (k, d.get('launches', 0), d.get('ms', 0), d['valu_issue_pct'] or 0))
