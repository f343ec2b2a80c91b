# Fabric-side read traffic (FETCH_SIZE) and L2 hit / miss requests of the fp64 GEMM launches of the bench's candidate solve, for
# the product library and for the build with the LDS image of rounds 1-4 (ab_tmp/lib_vxor3.so: -DALGP_GEMM_XOR_MASK=3): does
# the lane order of the 64-byte row pieces change what the XCDs' L2s miss?  
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/xor_ab
rm -rf $OUT && mkdir -p $OUT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-emulation"
for v in base xor3; do
  if [ "$v" = base ]; then unset ALGP_LIB; else export ALGP_LIB=$PWD/ab_tmp/lib_v$v.so; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$v -- python3 $ARGS > /dev/null 2> $OUT/fetch_$v.err || echo "fetch pass $v failed"
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2_$v -- python3 $ARGS > /dev/null 2> $OUT/l2_$v.err || echo "l2 pass $v failed"
done
unset ALGP_LIB
python3 - $OUT <<'PY' | tee $OUT/xor_traffic_ab.txt
import csv, glob, sys, collections
out = sys.argv[1]
for v in ('base', 'xor3'):
    row = {}
    for kind in ('fetch', 'l2'):
        fs = glob.glob('%s/%s_%s/**/*counter_collection.csv' % (out, kind, v), recursive=True)
        if not fs:
            continue
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(fs[0])):
            if 'gemm_nt_kernel_dma4<double' in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
        for k in acc:
            row[k] = (acc[k] / n[k], n[k])
    print(v, {k: ('%.4g per launch over %d launches' % row[k]) for k in row},
          'FETCH x2 KiB -> GB per launch: %.3f' % (2 * 1024 * row['FETCH_SIZE'][0] / 1e9) if 'FETCH_SIZE' in row else '')
PY
