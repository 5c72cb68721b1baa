#!/bin/bash
# A/B builds of the library with one translation unit compiled differently (replaces tools/r3/exp*.sh's ad-hoc compile lines):
#   tools/build_variant.sh NAME SRC_OBJ "EXTRA FLAGS"
# SRC_OBJ = a Makefile object of savitzky-golay-filter_amd/build (e.g. sg_2d_roll_g1.o, sg_stream_roll.o); its compile line is taken from
# `make -n`, EXTRA FLAGS are appended, and tools/ab/lib_NAME.so links that object with the rest of the current build.
set -e
cd "$(dirname "$0")/../savitzky-golay-filter_amd"
NAME=$1; OBJ=$2; FLAGS=$3
mkdir -p ../tools/ab
LINE=$(make -n -W csrc/$(echo $OBJ | sed -E 's/_g[0-9]+\.o$|_t[0-9]+\.o$|_f(32|64)_g[0-9]\.o$|\.o$//').hip build/$OBJ 2>/dev/null | grep -E "hipcc|^g\+\+|^cc " | grep -- "-c " | tail -1)
[ -n "$LINE" ] || { echo "no compile line for $OBJ"; exit 1; }
LINE=$(echo "$LINE" | sed -E "s# -o build/$OBJ# $FLAGS -o ../tools/ab/obj_${NAME}.o#")
echo "$LINE"
eval "$LINE"
OTHERS=$(ls build/*.o | grep -v "build/$OBJ")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/ab/lib_${NAME}.so ../tools/ab/obj_${NAME}.o $OTHERS -Wl,-soname,libsavgol_hip.so -Wl,--version-script=exports.map -lm -lpthread
echo "built tools/ab/lib_${NAME}.so"
