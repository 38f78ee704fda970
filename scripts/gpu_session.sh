#!/bin/bash
# ONE parameterised runner for GPU sessions (replaces round 5's 45 one-shot gpu_r05*.sh scripts).  Run through gpurun:
#
#   gpurun --timeout 1500 -- 'bash scripts/gpu_session.sh r06a tests bench stats'
#
#   usage: bash scripts/gpu_session.sh TAG STEP [STEP ...]        output under gpurun_out/TAG/ (copy what is to be judged into profiles/)
#
# Steps (each may carry arguments after a colon, comma separated):
#   tests[:PYTEST_ARGS]      pytest -q -m gpu (default: tests -x, the whole suite); e.g. tests:tests/test_hem_gpu.py,--maxfail=5
#   smoke                    __graft_entry__.smoke()
#   bench[:ARGS]             python bench.py ARGS            -> bench.json (e.g. bench:--steps,10)
#   c2                       bench.py --splats 1000000 (BASELINE configs[1])
#   stats                    rocprofv3 --kernel-trace --stats of the bench command -> kernel_stats.csv, bench_prof.json
#   pmc[:aniso]              the separate --pmc passes (FETCH_SIZE | WRITE_SIZE | L2 | SQ | SQ2) over scripts/prof_hem.py 5000000 1 2 [aniso]
#   pmc_icp                  the same over scripts/prof_icp.py 5000000 2
#   summary                  scripts/summarize_profiles.py: TAG_kernel_stats.csv, TAG_pmc.json, TAG_pmc_aniso.json, TAG_pmc_icp.json, ... in gpurun_out/TAG/
#   kstats:N,SHAPE           per-kernel averages of ONE level (rocprofv3 --stats over scripts/prof_hem.py N 1 3 SHAPE)
#   timeline:N               kernel timeline (durations, idle gaps) of one level of N splats (scripts/trace_timeline.py)
#   icp_timeline             kernel timeline of the bench's ICP phase, last repetition (scripts/prof_bench_icp.py)
#   ladder                   scripts/small_levels.py + scripts/level_ladder.py (small levels, async / sync)
#   levels:SHAPE             per-kernel table of the three levels of a 5 M cloud (scripts/prof_hem.py 5000000 3 3 SHAPE)
#   icp                      scripts/prof_bench_icp.py (the ICP phase of the step, per entry)
#   ab_libs:SHAPE,lib,...    scripts/ab_libs.sh  (variant libraries under variants/, "-" = the in-tree one)
#   ab_env:N,LEVELS,VAR,v..  scripts/ab_env.sh   (an environment knob, "-" = unset)
#   ab_bench:lib,...         scripts/ab_bench.sh (whole step: ms_per_step, HEM, ICP)
#   stress                   tests/stress_*.py sweeps (parity, tiny, knobs, icp, partition)
#   fullsize                 scripts/fullsize_modes.py c5 40 M / 8 ranks and c4 2 x 5 M
#   run:CMD                  any command (spaces as commas)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
TAG=${1:?tag}; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
ABS=$PWD
SCRATCH=$(mktemp -d)            # no fixed /tmp names: two sessions never share scratch files
trap 'rm -rf "$SCRATCH"' EXIT
export TMPDIR=/tmp
args() { echo "${1#*:}" | tr ',' ' '; }
pmc_passes() {                  # $1 = directory prefix (pmc_ | pmc_aniso_ | pmc_icp_: what scripts/summarize_profiles.py reads), rest = program after python3
    local pre=$1; shift
    ( cd /tmp                   # counters in their own passes, with --kernel-trace only (never combined with a trace domain)
      rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${pre}fetch" -- python3 "$@" > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${pre}write" -- python3 "$@" > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d "$OUT/${pre}l2" -- python3 "$@" > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS \
          --output-format csv -d "$OUT/${pre}sq" -- python3 "$@" > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR \
          --output-format csv -d "$OUT/${pre}sq2" -- python3 "$@" > /dev/null 2>&1 )
}
for step in "$@"; do
    name=${step%%:*}
    echo "=== $TAG $step"
    case $name in
    tests)    a=$( [ "$step" = "$name" ] && echo "tests -x" || args "$step" )
              python -m pytest $a -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; echo "exit $?"; tail -5 "$OUT/pytest_gpu.log" ;;
    smoke)    python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log" ;;
    bench)    a=$( [ "$step" = "$name" ] && echo "" || args "$step" )
              python bench.py $a > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 700 "$OUT/bench.json"; echo ;;
    c2)       python bench.py --splats 1000000 --no-cpu-baseline --no-aniso > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"; tail -c 300 "$OUT/bench_c2.json"; echo ;;
    stats)    ( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- python3 "$ABS/bench.py" --no-cpu-baseline --no-aniso \
                  > "$OUT/bench_prof.json" 2> "$OUT/bench_prof.err" )
              f=$(ls "$OUT"/prof_stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -12 "$OUT/kernel_stats.csv" | cut -c1-150 ;;
    pmc)      if [ "$step" = "pmc:aniso" ]; then pmc_passes pmc_aniso_ "$ABS/scripts/prof_hem.py" 5000000 1 2 aniso; else pmc_passes pmc_ "$ABS/scripts/prof_hem.py" 5000000 1 2; fi ;;
    pmc_icp)  ( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/icp_stats" -- python3 "$ABS/scripts/prof_icp.py" 5000000 2 > "$OUT/prof_icp.log" 2>&1 )
              pmc_passes pmc_icp_ "$ABS/scripts/prof_icp.py" 5000000 2 ;;
    summary)  python scripts/summarize_profiles.py "$OUT" "$OUT/$TAG" 2>&1 | tail -2
              rm -rf "$OUT"/pmc_*/ "$OUT"/prof_stats "$OUT"/icp_stats ;;           # the raw rocprofv3 output (gpurun merges <= 64 MiB back)
    kstats)   set -- $(args "$step"); ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$SCRATCH/ks" -- python3 "$ABS/scripts/prof_hem.py" ${1:-5000000} 1 3 ${2:-iso} > "$OUT/kstats.log" 2>&1 )
              python3 - "$SCRATCH/ks" > "$OUT/kstats_${1:-5000000}_${2:-iso}.txt" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))[:70]:
    if r["Name"].startswith("void at::") or "rocclr" in r["Name"] or "distribution" in r["Name"]: continue
    print(f'{r["Name"][:72]:72s} calls/level {int(r["Calls"])/3:5.1f}  avg us {float(r["AverageNs"])/1e3:8.1f}  per level us {float(r["TotalDurationNs"])/3e3:8.1f}')
PY
              head -45 "$OUT"/kstats_*.txt ;;
    timeline) n=$(args "$step"); ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$SCRATCH/tl" -- python3 "$ABS/scripts/prof_hem.py" $n 1 3 > "$OUT/timeline.log" 2>&1 )
              # the level's own first kernel is k_keys (its prologue travelled with the level's input): the last level from there on
              python scripts/trace_timeline.py "$SCRATCH/tl" "gsr::k_keys" > "$OUT/timeline_$n.txt"; tail -40 "$OUT/timeline_$n.txt" | grep -E "window|before" | head -40 ;;
    icp_timeline) ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$SCRATCH/it" -- python3 "$ABS/scripts/prof_icp.py" 5000000 2 > "$OUT/icp_timeline.log" 2>&1 )
              python scripts/trace_timeline.py "$SCRATCH/it" "gsr::k_debug_logf" -4 > "$OUT/icp_timeline.txt"; grep -E "window|before" "$OUT/icp_timeline.txt" | head -60 ;;
    ladder)   python scripts/small_levels.py > "$OUT/small_levels.txt" 2>&1; tail -12 "$OUT/small_levels.txt" ;;
    levels)   python scripts/prof_hem.py 5000000 3 3 $(args "$step") 2>&1 | grep -E "rep2 L. " > "$OUT/levels_$(args "$step" | tr ' ' '_').txt"; cut -c1-260 "$OUT"/levels_*.txt ;;
    icp)      python scripts/prof_bench_icp.py > "$OUT/prof_bench_icp.txt" 2>&1; tail -20 "$OUT/prof_bench_icp.txt" ;;
    ab_libs)  bash scripts/ab_libs.sh $(args "$step") 2>&1 | tee -a "$OUT/ab_libs.txt" ;;
    ab_env)   bash scripts/ab_env.sh $(args "$step") 2>&1 | tee -a "$OUT/ab_env.txt" ;;
    ab_bench) bash scripts/ab_bench.sh $(args "$step") 2>&1 | tee -a "$OUT/ab_bench.txt" ;;
    stress)   for s in parity tiny knobs icp partition; do timeout 900 python tests/stress_$s.py > "$OUT/stress_$s.txt" 2>&1; tail -2 "$OUT/stress_$s.txt"; done ;;
    fullsize) python scripts/fullsize_modes.py c5 --splats 40000000 --world 8 --out "$OUT/c5_40m_8ranks.json" > "$OUT/c5.log" 2>&1; tail -2 "$OUT/c5.log"
              python scripts/fullsize_modes.py c4 --splats 5000000 --out "$OUT/c4_2x5m.json" > "$OUT/c4.log" 2>&1; tail -2 "$OUT/c4.log" ;;
    run)      $(args "$step") 2>&1 | tee -a "$OUT/run.txt" | tail -40 ;;
    *)        echo "unknown step $name"; exit 2 ;;
    esac
done
ls "$OUT" | head -40
