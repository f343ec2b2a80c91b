# FETCH_SIZE against a KNOWN byte count, per request pattern: tools/row_piece_probe.hip in its
# calibration mode (one launch per shape, each reading rows x cols x 8 bytes exactly once) under a FETCH_SIZE counter pass.
#   bash tools/fetch_calibrate.sh [rows] [cols]     (binary: build/tools/row_piece_probe, built on the CPU side)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ROWS=${1:-50048}; COLS=${2:-20096}
OUT=gpurun_out/fetch_cal
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -- build/tools/row_piece_probe $ROWS $COLS 1 > $OUT/run.txt 2> $OUT/run.err || { echo "counter pass failed"; tail -5 $OUT/run.err; exit 1; }
python3 tools/fetch_calibrate.py $OUT/pmc $ROWS $COLS | tee $OUT/fetch_calibration.txt
