#!/bin/bash
# tools/mkvariant.sh <name> <sed-expr>...: build ab/libskx_<name>.so from skx_device.hip patched with the sed expressions (A/B experiments)
set -e
cd "$(dirname "$0")/../ska.rust_amd"
name=$1; shift
args=(); for e in "$@"; do args+=(-e "$e"); done
if [ ${#args[@]} -gt 0 ]; then sed "${args[@]}" csrc/skx_device.hip > csrc/_v.hip; else cp csrc/skx_device.hip csrc/_v.hip; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -c csrc/_v.hip -o build/_v.o
rm csrc/_v.hip
mkdir -p ../ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../ab/libskx_$name.so build/_v.o build/skx_reads.o build/skx_setops.o build/skx_snappy.o build/skx_api.o build/skx_api_io.o build/fastx.o build/skf_codec.o build/ska_host.o -lz -lpthread
echo built ab/libskx_$name.so
