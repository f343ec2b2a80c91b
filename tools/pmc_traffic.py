"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles/traffic.json.

HBM bytes per launch of the dominant kernel (the fp64 GEMM launches of the candidate TRSM, told
apart from the Cholesky's GEMM launches by their grid: a multiple of ceil(M/128) workgroups),
corrected as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE under-reports wide (16 B/lane)
coalesced reads by exactly 2x on gfx950; WRITE_SIZE is exact; both are in KiB.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <row-tile counts of the TRSM chunks, comma separated> <out.json>"""
import csv
import glob
import json
import sys


def load(d, counter):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    out = []
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter and 'gemm_nt_kernel' in r['Kernel_Name'] and 'double' in r['Kernel_Name']:
            out.append((int(r['Dispatch_Id']), int(r['Grid_Size']) // int(r['Workgroup_Size']), float(r['Counter_Value'])))
    return sorted(out)


def main():
    fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[4]
    tiles = [int(t) for t in sys.argv[3].split(',')]
    ok = lambda wg: any(wg % t == 0 and wg // t in (1, 2, 3, 4) for t in tiles)
    fe = [v for (_, wg, v) in load(fetch_dir, 'FETCH_SIZE') if ok(wg)]
    wr = [v for (_, wg, v) in load(write_dir, 'WRITE_SIZE') if ok(wg)]
    n = min(len(fe), len(wr))
    fetch_b = 2.0 * 1024.0 * sum(fe) / len(fe)
    write_b = 1024.0 * sum(wr) / len(wr)
    res = {'gemm_nt_f64_bytes_per_launch': fetch_b + write_b,
           'fetch_bytes_per_launch_corrected_x2': fetch_b, 'write_bytes_per_launch': write_b,
           'launches_counted': n, 'bytes_per_solve': (fetch_b + write_b) * n,
           'note': 'TRSM GEMM launches only; FETCH_SIZE doubled per MI355X_MICROARCH.md (16 B/lane streaming reads)'}
    json.dump(res, open(out, 'w'), indent=1)
    print(res)


if __name__ == '__main__':
    main()
