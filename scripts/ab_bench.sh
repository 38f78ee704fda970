for round in 1 2 3; do for v in "$@"; do echo "round $round $v: $(GSR_HIP_LIB=$PWD/variants/$v.so python bench.py --no-cpu-baseline --steps 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['hem_s_per_step']*1e3,3), round(d['icp_s_per_step']*1e3,3))")"; done; done
