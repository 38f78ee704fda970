#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
export GSR_ICP_WIDE_BELOW=0
PMC="TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" bash scripts/pmc_one.sh r03r/ta accumulate_dev scripts/prof_icp_small.py 185000 2 0.5
PMC="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" bash scripts/pmc_one.sh r03r/tcp accumulate_dev scripts/prof_icp_small.py 185000 2 0.5
