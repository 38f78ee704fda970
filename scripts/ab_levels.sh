#!/bin/bash
# A/B of variant libraries on the three level sizes of the bench (5 M cloud, 3 levels), interleaved.  usage: bash scripts/ab_levels.sh v1 v2 ...
for round in 1 2 3; do
  for v in "$@"; do
    echo "round $round $v: $(GSR_HIP_LIB=$PWD/variants/$v.so python scripts/prof_hem.py 5000000 3 3 2>&1 | grep 'rep2 L. kernels' | sed -e 's/.*level \([0-9.]*\)/\1/' | tr '\n' ' ')"
  done
done
