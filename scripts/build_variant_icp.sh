#!/bin/bash
# usage: scripts/build_variant_icp.sh NAME "-DFOO=1"   -> variants/NAME.so (icp.hip compiled with the extra flags, the other objects reused)
set -euo pipefail
mkdir -p variants
NAME=$1; FLAGS=${2:-}
C=gaussiansplattingregistration_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $FLAGS -c $C/icp.hip -o /tmp/icp_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/icp_$NAME.o $C/hem.o $C/hem_select.o $C/voxel.o $C/model.o $C/comm.o -o variants/$NAME.so
echo built variants/$NAME.so
