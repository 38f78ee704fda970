#!/bin/bash
# A/B of variant libraries on the three level sizes of a 5 M cloud, interleaved on one device: k_select / k_mstep / level per level.
# usage: bash scripts/ab_libs.sh SHAPE lib ...      lib = "-" (the in-tree library) or a name under variants/
SHAPE=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset GSR_HIP_LIB; else export GSR_HIP_LIB=$PWD/variants/$lib.so; fi
    echo "round $round $SHAPE $lib: $(python scripts/prof_hem.py 5000000 3 3 $SHAPE 2>&1 | grep 'rep2 L. kernels' | grep -oE " select [0-9.]+| mstep [0-9.]+|level [0-9.]+" | tr '\n' ' ')"
  done
done
