#!/bin/bash
# round 6, final tree: soak under NPS_MODE_AUTO at sizes that keep their tallies, then the eighteen-case sweep (another box)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
( while true; do sleep 60; echo "[progress] $(date +%T)"; done ) &
PP=$!
( timeout -k 10 200 python3 tools/soak.py --format strip --samples 300000 --mode 0 --passes 3000 --variants 100000 && timeout -k 10 200 python3 tools/soak.py --format strip --samples 400000 --mode 0 --passes 3000 --variants 100000 ) > gpurun_out/r06_soak_auto_raw.txt 2>&1; echo "soak rc=$?"; grep -h "soak\|auto:" gpurun_out/r06_soak_auto_raw.txt | tail -4
timeout -k 10 1000 python3 bench.py --steps 5 --warmup 1 --full-sweeps --extras-budget 900 --extras-deadline 1000 --full-out gpurun_out/r06_bench_sweeps.json > gpurun_out/r06_bench_sweeps_line.json 2> gpurun_out/r06_bench_sweeps.err; echo "sweeps rc=$?"
kill $PP
tail -c 1500 gpurun_out/r06_bench_sweeps_line.json
