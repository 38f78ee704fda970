#!/bin/bash
# one HEM level of a 20 M- and a 40 M-splat cloud on one GPU (the sizes of BASELINE configs[4]'s source): times, counters
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
for n in 20000000 40000000; do timeout 600 python scripts/prof_hem.py $n 1 2 2>&1 | grep -E "rep1 L1|Error|error" | cut -c1-700; done
