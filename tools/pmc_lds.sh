# LDS bank-conflict cycles by kernel (SQ_LDS_BANK_CONFLICT = extra cycles, SQ_LDS_IDX_ACTIVE = all LDS-array cycles) for the
# bench's kernels and for config 5's loop:  bash tools/pmc_lds.sh OUTDIR
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=${1:-gpurun_out/pmc_lds}
mkdir -p $OUT
R=$PWD
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $OUT/bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-emulation > /dev/null 2> $OUT/bench.err
export C5_ONLY=loop C5_STEPS=8
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $OUT/c5 -- python3 $R/tools/c5_leg.py > /dev/null 2> $OUT/c5.err
unset C5_ONLY C5_STEPS
python3 - $OUT <<'PY' > $OUT/lds_conflicts_by_kernel.txt
import csv, glob, sys, collections
out = sys.argv[1]
print('LDS-array cycles by kernel: SQ_LDS_BANK_CONFLICT (extra cycles) / SQ_LDS_IDX_ACTIVE (all cycles), rocprofv3 --pmc, own pass')
for leg in ('bench', 'c5'):
    f = glob.glob(out + '/' + leg + '/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0].replace('void algp::', '')][r['Counter_Name']] += float(r['Counter_Value'])
    print('--', 'bench.py --steps 2 --warmup 1 (config 4)' if leg == 'bench' else 'config 5 loop, 8 steps (tools/c5_leg.py)')
    for k in sorted(agg, key=lambda k: -agg[k]['SQ_LDS_IDX_ACTIVE'])[:12]:
        a = agg[k]
        if a['SQ_LDS_IDX_ACTIVE'] <= 0:
            continue
        print('%-52s conflict %10.3e  active %10.3e  = %5.1f %%   (%.3e LDS instructions)' % (k[:52], a['SQ_LDS_BANK_CONFLICT'], a['SQ_LDS_IDX_ACTIVE'],
              100 * a['SQ_LDS_BANK_CONFLICT'] / a['SQ_LDS_IDX_ACTIVE'], a['SQ_INSTS_LDS']))
PY
rm -rf $OUT/bench $OUT/c5
cat $OUT/lds_conflicts_by_kernel.txt
