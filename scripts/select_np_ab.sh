#!/bin/bash
# A/B of the selection's parents-per-wave (GSR_HEM_SELECT_NP) and of variant libraries, interleaved on one device.
# usage: bash scripts/select_np_ab.sh SHAPE "lib:np" ...      lib = "-" (the in-tree library) or a name under variants/
SHAPE=$1; shift
for round in 1 2 3; do
  for cfg in "$@"; do
    lib=${cfg%%:*}; np=${cfg##*:}
    if [ "$lib" = "-" ]; then unset GSR_HIP_LIB; else export GSR_HIP_LIB=$PWD/variants/$lib.so; fi
    export GSR_HEM_SELECT_NP=$np
    echo "round $round $SHAPE $cfg: $(python scripts/prof_hem.py 5000000 3 3 $SHAPE 2>&1 | grep 'rep2 L. kernels' | grep -oE " select [0-9.]+| mstep [0-9.]+|level [0-9.]+" | tr '\n' ' ')"
  done
done
