#!/bin/bash
# round 6, final tree: the headline's profile on ONE box (bench, rocprofv3 --stats, FETCH / WRITE passes, SQ counters)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
( while true; do sleep 60; echo "[progress] $(date +%T) $(ls gpurun_out/prof_r06 2>/dev/null | tr '\n' ' ')"; done ) &
PP=$!
bash tools/profile_round.sh r06; rc1=$?
bash tools/pmc1.sh r06; rc2=$?
kill $PP
echo "profile_round rc=$rc1 pmc1 rc=$rc2"; tail -c 600 gpurun_out/prof_r06/bench.json; ls gpurun_out/prof_r06/summary 2>/dev/null
exit $rc1
