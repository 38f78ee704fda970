#!/bin/bash
# usage: bash scripts/pmc_one.sh <out-subdir> <kernel-substring> <python script> [args...]
# One SQ counter pass over a short script; prints the per-launch averages of the kernels whose name contains the substring.
set -uo pipefail
OUT=${GRAFT_REPO_ROOT:?}/gpurun_out/$1; KSUB=$2; shift; shift; mkdir -p "$(dirname "$OUT")"
SCRIPT=$(realpath "$1"); shift; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc ${PMC:-SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS} --output-format csv -d $OUT -- python3 "$SCRIPT" "$@" > $OUT.log 2>&1; tail -3 $OUT.log
cd "${GRAFT_REPO_ROOT:?}"
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
d, sub = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if sub not in k: continue
    key = (r["Dispatch_Id"], r["Counter_Name"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen: seen.add(r["Dispatch_Id"]); cnt[k] += 1
for k in agg:
    print(k[:80], "launches", cnt[k])
    for c, v in sorted(agg[k].items()): print(f"   {c:24s} {v / cnt[k]:16.0f}")
PY
