#!/bin/bash
# Config 5's profiles for profiles/ (on the GPU box; OUT under gpurun_out/): the loop's kernel statistics and HBM counters
# (tail_part_kernel / tail_cols_kernel: the kernel that dominates an incremental step), one rank of eight's step as a kernel
# timeline, the folded rank step of config 4 with its MFMA-busy counters, the tail kernel by width, the fold's task breakdown.
#   bash tools/collect_c5_profiles.sh OUTDIR
set -e -o pipefail
OUT=${1:-gpurun_out/prof_r06_c5}
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
R=$PWD
# the 200-step loop + one rank of eight, plain (the numbers of DESIGN.md)
python3 tools/c5_leg.py > $OUT/c5_loop.json 2> $OUT/c5_loop.err
echo "loop done"
# kernel statistics of a shorter loop (40 steps), loop only
export C5_ONLY=loop C5_STEPS=40
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/c5_leg.py > $OUT/c5_loop_under_rocprof.json 2> $OUT/stats.err
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/c5_loop_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/c5_leg.py > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/c5_leg.py > /dev/null 2> $OUT/pmc_write.err
python3 tools/pmc_summary.py $OUT/c5_pmc_by_kernel.json $OUT/stats $OUT/pmc_fetch $OUT/pmc_write > $OUT/c5_pmc_summary.txt
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
unset C5_ONLY C5_STEPS
head -12 $OUT/c5_pmc_summary.txt
echo "loop counters done"
# one rank of eight: kernel trace of 10 steps -> the last step's kernels in time order
export C5_ONLY=rank8 C5_EMU_STEPS=10
rocprofv3 --kernel-trace --output-format csv -d $OUT/r8 -- python3 $R/tools/c5_leg.py > /dev/null 2> $OUT/r8.err
unset C5_ONLY C5_EMU_STEPS
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + '/r8/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the emulated rank's launches are the ones with 98-workgroup candidate grids; its last step = from the last-but-one teacher kernel on
idx = [i for i, r in enumerate(rows) if 'tail_' in r['Kernel_Name'] and int(r['Grid_Size_X']) <= 131072 and 'finish' not in r['Kernel_Name']]
i0 = idx[-1]
j = i0
while j > 0 and 'lazy_refresh' not in rows[j]['Kernel_Name'] or int(rows[j]['Grid_Size_X']) < 3000000:
    j -= 1
    if j == 0:
        break
t0 = int(rows[j + 1]['Start_Timestamp'])
with open(out + '/c5_rank_of_8_step_trace.txt', 'w') as fo:
    fo.write('one planning step of rank 7 of 8 (config 5: N = 50 000 replicated, 12 500 candidates), rocprofv3 --kernel-trace:\n'
             'start us / duration us / kernel / grid (threads)\n')
    for r in rows[j + 1:j + 110]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        fo.write('%9.1f %8.1f  %-56s %9s\n' % ((s - t0) / 1e3, (e - s) / 1e3, r['Kernel_Name'].split('(')[0][-56:], r['Grid_Size_X']))
PY
rm -rf $OUT/r8
head -50 $OUT/c5_rank_of_8_step_trace.txt
echo "rank-of-8 trace done"
# config 4's rank step (fit + solve folded): timing, then MFMA-busy counters of the folded launch
python3 tools/rank_step.py > $OUT/c4_rank_step.json 2> $OUT/c4_rank_step.err
REPS=2 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rs_stats -- python3 $R/tools/rank_step.py > /dev/null 2> $OUT/rs_stats.err
REPS=2 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/rs_pmc -- python3 $R/tools/rank_step.py > /dev/null 2> $OUT/rs_pmc.err
python3 tools/pmc_summary.py $OUT/c4_rank_step_pmc_by_kernel.json $OUT/rs_stats $OUT/rs_pmc > $OUT/c4_rank_step_pmc_summary.txt
rm -rf $OUT/rs_stats $OUT/rs_pmc
head -6 $OUT/c4_rank_step_pmc_summary.txt
cat $OUT/c4_rank_step.json
echo "rank step done"
python3 tools/tail_sweep.py > $OUT/tail_cols_by_width.json 2> $OUT/tail_sweep.err
TAIL_M=12500 python3 tools/tail_sweep.py > $OUT/tail_cols_by_width_m12500.json 2>> $OUT/tail_sweep.err
cat $OUT/tail_cols_by_width.json
if [ -x build/dag_test ]; then timeout -k 10 120 build/dag_test 79 3 d 98 > $OUT/fold_task_breakdown_f64.txt 2>&1 || true; timeout -k 10 120 build/dag_test 79 3 f 98 > $OUT/fold_task_breakdown_f32.txt 2>&1 || true; fi
echo "all done"
