# A/B timing of library builds under ab_tmp/ (ALGP_LIB selects the build; "base" = the product library): the headline run, short
for v in "$@"; do
  if [ "$v" = base ]; then unset ALGP_LIB; else export ALGP_LIB=$PWD/ab_tmp/lib_v$v.so; fi
  python3 bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python3 -c "
import json,sys
o=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1])
r=o['roofline']
print('v$v', round(o['ms_per_step'],2), 'frac', round(r['frac'],4), 'avg launch ms', round(r.get('avg_launch_ms',0),4), 'serial', r.get('serial_kernel_frac'))"
done
