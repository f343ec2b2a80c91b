# Build an A/B variant of the library: tools/ab_build.sh <name> <file.hip | replacement.hip:file.hip | change.patch:file.hip> "<extra hipcc flags>"  ->  ab_tmp/lib_v<name>.so
# e.g. tools/ab_build.sh t256 tools/experiments/gemm_r06_variants.patch:gemm.hip "-DALGP_GEMM_T256=1"
# (a .patch is a unified diff against the shipped algp_amd/csrc/<file.hip>: the rejected experiments of EXPERIMENTS.md are kept that way)
# (every other object is the product build's; tools/ab_libs.sh times the variants on one box through $ALGP_LIB)
set -e
NAME=$1; SRC=$2; FLAGS=$3
cd "$(dirname "$0")/.."
SRCPATH=algp_amd/csrc/$SRC
case "$SRC" in *:*) SRCPATH=${SRC%%:*}; SRC=${SRC##*:};; esac
mkdir -p ab_tmp
case "$SRCPATH" in *.patch) cp algp_amd/csrc/$SRC ab_tmp/patched_$SRC; patch -s ab_tmp/patched_$SRC < $SRCPATH; SRCPATH=ab_tmp/patched_$SRC;; esac
make -C algp_amd/csrc -j8 > /dev/null
mkdir -p ab_tmp
OBJ=ab_tmp/$(basename $SRC .hip)_v$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result $FLAGS -Ialgp_amd/csrc -c $SRCPATH -o $OBJ
OTHERS=$(ls algp_amd/csrc/*.o | grep -v "/$(basename $SRC .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ $OTHERS -ldl -o ab_tmp/lib_v$NAME.so
echo built ab_tmp/lib_v$NAME.so
