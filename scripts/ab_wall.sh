#!/bin/bash
# wall time of the three levels of a 5 M cloud under an environment knob, interleaved.  usage: bash scripts/ab_wall.sh SHAPE VAR v1 v2 ... ("-" = unset)
SHAPE=$1; VAR=$2; shift; shift
for round in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    echo "round $round $SHAPE $VAR=$v: $(python scripts/prof_hem.py 5000000 3 4 $SHAPE 2>&1 | grep -E 'rep3 L.: wall' | grep -oE 'wall [0-9.]+ ms' | tr '\n' ' ')"
  done
done
