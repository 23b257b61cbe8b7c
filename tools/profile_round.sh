#!/bin/bash
# Round profile of the headline bench (run on the GPU box through gpurun):
#   tools/profile_round.sh r02      -> gpurun_out/prof_r02/{bench.json, stats/, fetch/, write/, summary files}
# Separate rocprofv3 passes: --kernel-trace --stats ; --pmc FETCH_SIZE ; --pmc WRITE_SIZE
# (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2 -- they do not fit one pass).
TAG=${1:-r02}
[ $# -gt 0 ] && shift
EXTRA="$*"   # e.g. --format ds --samples 200000 --variants 300000 (tag r02_ds)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 $EXTRA"
# the line as the driver reads it (compact, last stdout line) and the full object with every secondary beside it;
# NPS_PROFILE_SWEEPS=1: all eighteen size x distribution cases and six layout cohorts (minutes longer)
SWEEP=""; [ -n "$NPS_PROFILE_SWEEPS" ] && SWEEP="--full-sweeps --extras-budget 900 --extras-deadline 1000"
python3 "$R/bench.py" $ARGS $SWEEP --full-out "$O/bench_full.json" > "$O/bench.json" 2> "$O/bench.err" || exit 1
rocprofv3 --kernel-trace --stats -d "$O/stats" -o stats --output-format csv -- python3 "$R/bench.py" $ARGS --no-cpu-baseline --no-extras > "$O/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$O/fetch" -o fetch --output-format csv -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras $EXTRA > "$O/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$O/write" -o write --output-format csv -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras $EXTRA > "$O/write.log" 2>&1
python3 "$R/tools/profile_summary.py" "$O" "$TAG"
