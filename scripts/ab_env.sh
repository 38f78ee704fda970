#!/bin/bash
# A/B an environment knob in ONE gpurun session (same device), interleaved rounds.
# usage: bash scripts/ab_env.sh N LEVELS VAR v1 v2 ...        (value "-" = unset)
N=$1; L=$2; VAR=$3; shift; shift; shift
for round in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    echo "round $round $VAR=$v: $(python scripts/prof_hem.py $N $L 2 2>&1 | grep "rep1 L$L" | sed -e 's/.*ms_select/ms_select/' | cut -c1-175)"
  done
done
