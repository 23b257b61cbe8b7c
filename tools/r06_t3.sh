#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
( while true; do sleep 60; echo "[progress] $(tail -c 100 gpurun_out/r06_gputests2.log | tr '\n' ' ')"; done ) &
PP=$!
timeout -k 10 1150 python3 -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests2.log 2>&1; rc=$?
kill $PP
echo rc=$rc; tail -12 gpurun_out/r06_gputests2.log
exit $rc
