#!/bin/bash
# usage: scripts/build_variant.sh NAME "sed-expr" [more sed-exprs...]   -> variants/NAME.so (hem.hip patched by sed, icp.o reused)
set -euo pipefail
mkdir -p variants
NAME=$1; shift
C=gaussiansplattingregistration_amd/csrc
cp $C/hem.hip /tmp/hem_$NAME.hip
for e in "$@"; do sed -i "$e" /tmp/hem_$NAME.hip; done
cp /tmp/hem_$NAME.hip $C/hem_variant_$NAME.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -c $C/hem_variant_$NAME.hip -o /tmp/hem_$NAME.o
rm -f $C/hem_variant_$NAME.hip
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/hem_$NAME.o $C/hem_select.o $C/icp.o $C/voxel.o $C/model.o $C/comm.o -o variants/$NAME.so
echo built variants/$NAME.so
