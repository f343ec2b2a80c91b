#!/bin/bash
# rocprofv3 passes of bench.py for profiles/: kernel trace + stats, then the counter passes -- each in a run of its own
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; counters are never combined with tracing).
#   bash tools/collect_profiles.sh f64|f32 OUTDIR        (on the GPU box; OUTDIR under gpurun_out/)
set -e -o pipefail
DT=${1:-f64}
OUT=${2:-gpurun_out/prof_r04_$DT}
ARGS="bench.py --dtype $DT --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-emulation"
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
python3 $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err
echo "plain done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats done"
# the same command with the candidate solve on ONE stream (the launches back to back): the trace from which
# roofline.serial_kernel_frac can be recomputed, joined with the library's launch log for a table by launch shape
export ALGP_TRSM_CHUNKS=1
export ALGP_LAUNCH_LOG=$PWD/$OUT/launch_log_1stream.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -- python3 $ARGS > $OUT/bench_under_rocprof_1stream.json 2> $OUT/stats1.err
unset ALGP_TRSM_CHUNKS ALGP_LAUNCH_LOG
cp $(find $OUT/stats1 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_1stream.csv
python3 tools/trace_shapes.py $(find $OUT/stats1 -name '*kernel_trace.csv' | head -1) $OUT/launch_log_1stream.txt 7 $OUT/bench_under_rocprof_1stream.json > $OUT/trsm_launch_shapes_1stream.txt
cat $OUT/trsm_launch_shapes_1stream.txt
rm -rf $OUT/stats1
echo "1-stream stats done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $ARGS > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err
echo "mfma done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "write done"
python3 tools/pmc_summary.py $OUT/pmc_by_kernel.json $OUT/stats $OUT/pmc_mfma $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.txt
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
# the raw per-dispatch CSVs are large: only the summaries travel back
rm -rf $OUT/stats $OUT/pmc_mfma $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/pmc_summary.txt
