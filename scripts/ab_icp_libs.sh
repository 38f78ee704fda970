#!/bin/bash
# A/B of variant libraries on the bench's coarse-to-fine ICP (scripts/prof_icp.py: per level the kernels' time of one registration), interleaved.
# usage: bash scripts/ab_icp_libs.sh lib ...      lib = "-" (the in-tree library) or a name under variants/
for round in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset GSR_HIP_LIB; else export GSR_HIP_LIB=$PWD/variants/$lib.so; fi
    echo "round $round $lib: $(python scripts/prof_icp.py 5000000 2 2>&1 | grep -E '^rep1' | grep -oE 'kernels [0-9.]+|total [0-9.]+ ms' | tr '\n' ' ')"
  done
done
