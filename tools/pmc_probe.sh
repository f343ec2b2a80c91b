# a look at the SQ's wait / LDS counters for the bench's kernels (which unit do the GEMM's waves wait for?)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pmc_probe
mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | grep -i "LDS\|WAIT\|MFMA\|BUSY\|INSTS_VALU\|ACTIVE_INST\|INST_CYCLES\|IFETCH\|WAVE_CYCLES\|WAVES" | tr '\n' ' ' > $OUT/avail.txt
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-emulation"
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $ARGS > /dev/null 2> $OUT/p$i.err || echo "pass $i failed"
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][-44:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    n[k] += 1
for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:4]:
    print(k, {c: '%.4g' % v for c, v in agg[k].items()})
PY
done
cat $OUT/avail.txt | head -c 3000
