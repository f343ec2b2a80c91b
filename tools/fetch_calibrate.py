"""FETCH_SIZE (KiB, as rocprofv3 reports it) per dispatch of tools/row_piece_probe.hip's calibration set, divided by the bytes
the launch really read (rows x cols x 8): the factor by which the counter must be multiplied for that request pattern.
usage: python tools/fetch_calibrate.py <pmc_dir> <rows> <cols>"""
import csv
import glob
import re
import sys
from collections import defaultdict


def main():
    d, rows, cols = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    true = rows * cols * 8.0
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'FETCH_SIZE':
            continue
        m = re.search(r'(probe(?:_dma)?)<(\d+), *(\d+), *(\d+)>', r['Kernel_Name'])
        if not m:
            continue
        acc[(m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)))].append(float(r['Counter_Value']) * 1024.0)
    print('matrix %d x %d fp64: %.3f GB read exactly once per launch' % (rows, cols, true / 1e9))
    print('%-44s %14s %10s %12s' % ('request pattern', 'FETCH_SIZE B', 'ctr/true', 'multiply by'))
    for (kind, a, b, c), v in sorted(acc.items()):
        mean = sum(v) / len(v)
        if kind == 'probe_dma':
            what = 'LDS-DMA  %4d-byte row pieces, lane order %d' % (a, b)
        else:
            what = 'plain    %4d-byte row pieces, %3d rows / WG' % (a, b)
        print('%-44s %14.0f %10.4f %12.3f' % (what, mean, mean / true, true / mean))


if __name__ == '__main__':
    main()
