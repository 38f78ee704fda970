#!/bin/bash
# k_mstep A/B in one session: default library vs variants, prefetch knob; usage: bash scripts/mstep_ab.sh
for round in 1 2; do
  echo "round $round default:";            python scripts/mstep_check.py 2>&1 | grep seed | cut -c1-150
  echo "round $round prefetch=1:";         GSR_HEM_MSTEP_PREFETCH=1 python scripts/mstep_check.py 2>&1 | grep seed | cut -c1-150
  echo "round $round small kernel w/o waves_per_eu(5):"; GSR_HIP_LIB=$PWD/variants/smallnoattr.so python scripts/mstep_check.py 2>&1 | grep seed | cut -c1-150
done
