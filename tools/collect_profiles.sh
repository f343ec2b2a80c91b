#!/bin/bash
# rocprofv3 passes of bench.py for profiles/: the counter passes first -- each in a run of its own (MI355X_MICROARCH.md:
# FETCH_SIZE and WRITE_SIZE do not fit one pass; counters are never combined with tracing) -- so that the traffic file exists
# when the traced runs print their bench lines (round 4 took them in the other order: `traffic: null` in the traced lines);
# then kernel trace + stats of the default run (three row-chunk streams: per-solve SPAN of the launches beside the HIP-event
# wall time) and of the one-stream run (launches back to back: table by launch shape).
#   bash tools/collect_profiles.sh f64|f32 OUTDIR [TRAFFIC_JSON]      (on the GPU box; OUTDIR under gpurun_out/)
set -e -o pipefail
DT=${1:-f64}
OUT=${2:-gpurun_out/prof_r06_$DT}
TRAFFIC=${3:-$OUT/traffic_pmc.json}
ARGS="bench.py --dtype $DT --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-emulation --traffic-json $TRAFFIC"
mkdir -p $OUT $(dirname $TRAFFIC)
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
python3 $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err
echo "plain done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats0 -- python3 $ARGS > /dev/null 2> $OUT/stats0.err
echo "stats (for the counter join) done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $ARGS > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err
echo "mfma done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "write done"
python3 tools/pmc_summary.py $OUT/pmc_by_kernel.json $OUT/stats0 $OUT/pmc_mfma $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.txt
rm -rf $OUT/stats0 $OUT/pmc_mfma $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/pmc_summary.txt
# the traffic file of THIS dtype (the other dtype's entries are added when its pass runs into the same file)
if [ "$DT" = f64 ]; then python3 tools/make_traffic_json.py $OUT/pmc_by_kernel.json /nonexistent $TRAFFIC; else
  F64=$(dirname $OUT)/$(basename $OUT | sed s/f32/f64/)/pmc_by_kernel.json; python3 tools/make_traffic_json.py $F64 $OUT/pmc_by_kernel.json $TRAFFIC; fi
# default run, three streams: trace + stats + the launch log -> per-solve span
export ALGP_LAUNCH_LOG=$PWD/$OUT/launch_log_3stream.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
unset ALGP_LAUNCH_LOG
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
python3 tools/trace_shapes.py $(find $OUT/stats -name '*kernel_trace.csv' | head -1) $OUT/launch_log_3stream.txt 7 $OUT/bench_under_rocprof.json > $OUT/trsm_span_3stream.txt
tail -8 $OUT/trsm_span_3stream.txt
rm -rf $OUT/stats
echo "3-stream stats done"
# the same command with the candidate solve on ONE stream (the launches back to back): the trace from which
# roofline.serial_kernel_frac can be recomputed, joined with the library's launch log for a table by launch shape
export ALGP_TRSM_CHUNKS=1
export ALGP_LAUNCH_LOG=$PWD/$OUT/launch_log_1stream.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -- python3 $ARGS > $OUT/bench_under_rocprof_1stream.json 2> $OUT/stats1.err
unset ALGP_TRSM_CHUNKS ALGP_LAUNCH_LOG
cp $(find $OUT/stats1 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_1stream.csv
python3 tools/trace_shapes.py $(find $OUT/stats1 -name '*kernel_trace.csv' | head -1) $OUT/launch_log_1stream.txt 7 $OUT/bench_under_rocprof_1stream.json > $OUT/trsm_launch_shapes_1stream.txt
tail -12 $OUT/trsm_launch_shapes_1stream.txt
rm -rf $OUT/stats1
echo "1-stream stats done"
