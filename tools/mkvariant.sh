#!/bin/bash
# tools/mkvariant.sh <name> [--rev <git rev>] [-D<macro>...] [<sed-expr>...]: build ab/libskx_<name>.so from skx_device.hip (the working file, or
# the one of a revision) patched with the sed expressions / compiled with the macros, linked with the other objects of the current build (A/B experiments)
set -e
cd "$(dirname "$0")/../ska.rust_amd"
name=$1; shift
unit=${SRC:-skx_device}                 # SRC=skx_reads tools/mkvariant.sh ...: the variant is of another translation unit
src=csrc/$unit.hip; defs=(); args=()
while [ $# -gt 0 ]; do
  case "$1" in
    --rev) git show "$2:ska.rust_amd/csrc/$unit.hip" > csrc/_base.hip; src=csrc/_base.hip; shift 2;;
    -D*|-W*) defs+=("$1"); shift;;
    *) args+=(-e "$1"); shift;;
  esac
done
if [ ${#args[@]} -gt 0 ]; then sed "${args[@]}" $src > csrc/_v.hip; else cp $src csrc/_v.hip; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result "${defs[@]}" -c csrc/_v.hip -o build/_v.o
rm -f csrc/_v.hip csrc/_base.hip
mkdir -p ../ab
objs=$(ls build/*.o | grep -v -e '/_v.o' -e "/$unit.o" -e '/ska_main.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../ab/libskx_$name.so build/_v.o $objs -lz -lpthread
echo built ab/libskx_$name.so
