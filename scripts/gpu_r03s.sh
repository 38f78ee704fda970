#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
timeout 120 python scripts/prof_icp_small.py 20000 1 0.5 2>&1 | tail -5
echo "rc=$?"
