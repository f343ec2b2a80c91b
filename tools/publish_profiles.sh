#!/bin/bash
# copies what tools/collect_profiles.sh (both dtypes) and tools/collect_c5_profiles.sh left under gpurun_out/ into profiles/
# under this round's names (gpurun_out/ is scratch; profiles/ is what is committed):  bash tools/publish_profiles.sh r05
set -e
R=${1:-r06}
G=gpurun_out
for DT in f64 f32; do
  D=$G/prof_${R}_$DT
  [ -d $D ] || continue
  cp $D/bench_plain.json profiles/${R}_bench_${DT}_no_extras.json
  cp $D/bench_under_rocprof.json profiles/${R}_bench_under_rocprof_$DT.json
  cp $D/bench_under_rocprof_1stream.json profiles/${R}_bench_under_rocprof_${DT}_1stream.json
  cp $D/kernel_stats.csv profiles/${R}_bench_kernel_stats_$DT.csv
  cp $D/kernel_stats_1stream.csv profiles/${R}_bench_kernel_stats_${DT}_1stream.csv
  cp $D/pmc_by_kernel.json profiles/${R}_pmc_by_kernel_$DT.json
  cp $D/pmc_summary.txt profiles/${R}_pmc_summary_$DT.txt
  cp $D/trsm_span_3stream.txt profiles/${R}_trsm_span_3stream_$DT.txt
  cp $D/trsm_launch_shapes_1stream.txt profiles/${R}_trsm_launch_shapes_1stream_$DT.txt
done
[ -f $G/prof_${R}_traffic/traffic_pmc.json ] && cp $G/prof_${R}_traffic/traffic_pmc.json profiles/${R}_traffic_pmc.json
C=$G/prof_${R}_c5
if [ -d $C ]; then
  cp $C/c5_loop.json profiles/${R}_c5_loop.json
  cp $C/c5_loop_kernel_stats.csv profiles/${R}_c5_loop_kernel_stats.csv
  cp $C/c5_pmc_by_kernel.json profiles/${R}_c5_pmc_by_kernel.json
  cp $C/c5_pmc_summary.txt profiles/${R}_c5_pmc_summary.txt
  cp $C/c5_rank_of_8_step_trace.txt profiles/${R}_c5_rank_of_8_step_trace.txt
  cp $C/c4_rank_step.json profiles/${R}_c4_rank_step.json
  cp $C/c4_rank_step_pmc_by_kernel.json profiles/${R}_c4_rank_step_pmc_by_kernel.json
  cp $C/c4_rank_step_pmc_summary.txt profiles/${R}_c4_rank_step_pmc_summary.txt
  cp $C/tail_cols_by_width.json profiles/${R}_tail_cols_by_width.json
  cp $C/tail_cols_by_width_m12500.json profiles/${R}_tail_cols_by_width_m12500.json
  for DT in f64 f32; do [ -s $C/fold_task_breakdown_$DT.txt ] && cp $C/fold_task_breakdown_$DT.txt profiles/${R}_fold_task_breakdown_$DT.txt; done
fi
# the tail kernel's HBM bytes (config 5 loop, own --pmc passes) into the traffic record, with tail.hip's fingerprint
if [ -d $C ]; then
python3 - $R $C <<'PY'
import hashlib, json, sys
R, C = sys.argv[1], sys.argv[2]
tj_path = 'profiles/%s_traffic_pmc.json' % R
tj = json.load(open(tj_path))
k = json.load(open(C + '/c5_pmc_by_kernel.json'))['by_kernel']['tail_part_kernel<double>']
run = json.load(open(C + '/c5_loop_under_rocprof.json'))
first, last = run['train_rows_first_last']
steps = int(run['incremental_steps'])
n_old = first + (last - first) * (steps - 1) / (2.0 * steps)          # mean train size in front of an append
mpad = 100096
alg = 8.0 * (mpad * n_old + 64 * n_old)
tj['tail_part_f64_bytes_per_launch'] = k['hbm_bytes_per_launch']
tj['tail_part_f64_algorithmic_bytes_per_launch_same_run'] = alg
tj['tail_part_f64_note'] = ('config 5 loop, steps 1..%d of tools/c5_leg.py (N = %d .. %d, Mpad = %d): 2 x FETCH_SIZE + WRITE_SIZE per launch '
                            'of tail_part_kernel<double> (%d launches); algorithmic s * (Mpad * N_old + 64 * N_old) = %.1f GB at these sizes: '
                            '%.2f x (V^T is streamed once; the new rows of L come from L2 / Infinity Cache)'
                            % (steps, first, last, mpad, k['calls'], alg / 1e9, k['hbm_bytes_per_launch'] / alg))
import subprocess
tj['commit'] = subprocess.run(['git', 'log', '-1', '--format=%h', '--', 'algp_amd/csrc'], capture_output=True, text=True).stdout.strip() or None
tj['tail_sources_sha16'] = hashlib.sha256(open('algp_amd/csrc/tail.hip', 'rb').read()).hexdigest()[:16]
json.dump(tj, open(tj_path, 'w'), indent=1)
print('tail traffic: %.2f x algorithmic' % (k['hbm_bytes_per_launch'] / alg))
PY
fi
git status --short profiles | head -40
