#!/bin/bash
# Build a side copy of libmi355seg.so for A/B timing: tools/build_variant.sh NAME [make variables, e.g. TUNE=1 "EXTRA=-DX3S_WD=2"]
# -> ab/NAME.so (sources copied to ab/NAME_src so that the in-tree objects are untouched).  Select it with MI355SEG_LIB_PATH=$PWD/ab/NAME.so.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/ab/${name}_src
mkdir -p "$src/pkg/csrc" "$src/include"
cp -pu "$root"/general-medical-image-segmentation-cnn-framework_amd/csrc/{*.hip,*.h,*.inc,Makefile} "$src/pkg/csrc/"
cp -pu "$root"/include/*.h "$src/include/"
make -C "$src/pkg/csrc" -j8 OUT="$root/ab/$name.so" "$@" 2>&1 | grep -v "^make\|hipcc" | tail -5
ls -la "$root/ab/$name.so"
