#!/bin/bash
# A/B of an ICP environment knob on the bench's schedule (scripts/prof_icp.py 5000000 2: ms of the iteration kernels per entry, total of the phase), interleaved.
# usage: bash scripts/ab_icp_env.sh VAR v1 v2 ...   ("-" = unset)
VAR=$1; shift
for r in 1 2 3; do for v in "$@"; do if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi; echo "round $r $VAR=$v: $(python scripts/prof_icp.py 5000000 2 2>&1 | grep rep1 | grep -oE 'ns= *[0-9]+|iters [0-9]+, kernels [0-9.]+|total [0-9.]+' | tr '\n' ' ')"; done; done
