#!/bin/bash
# A/B several variant libraries in ONE gpurun session (same device), interleaved rounds.
# usage: bash scripts/ab_hem.sh N LEVELS v1 v2 ...
N=$1; L=$2; shift; shift
for round in 1 2 3; do
  for v in "$@"; do
    echo "round $round $v: $(GSR_HIP_LIB=$PWD/variants/$v.so python scripts/prof_hem.py $N $L 2 2>&1 | grep 'rep1 L1' | sed -e 's/.*ms_select/ms_select/' | cut -c1-175)"
  done
done
