#!/bin/bash
# round 6, final tree: the counted test run, soak runs, the FORMAT/DS profile (one box)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
( while true; do sleep 60; echo "[progress] $(date +%T)"; done ) &
PP=$!
timeout -k 10 400 python3 -m pytest tests/test_gpu_mx.py -q -s > gpurun_out/r06_test_counts_raw.txt 2>&1; echo "counted tests rc=$?"; tail -3 gpurun_out/r06_test_counts_raw.txt
( timeout -k 10 200 python3 tools/soak.py --format strip --passes 4000 --variants 100000 && timeout -k 10 200 python3 tools/soak.py --format strip --samples 300000 --mode 0 --passes 3000 --variants 100000 && timeout -k 10 200 python3 tools/soak.py --format strip --samples 400000 --mode 0 --passes 3000 --variants 100000 ) > gpurun_out/r06_soak_raw.txt 2>&1; echo "soak rc=$?"; grep -h "soak" gpurun_out/r06_soak_raw.txt | tail -4
bash tools/profile_round.sh r06_ds --format ds --samples 200000 --variants 300000; echo "ds profile rc=$?"
kill $PP
tail -c 400 gpurun_out/prof_r06_ds/bench.json
timeout -k 10 200 python3 tools/qb_harvest.py > gpurun_out/r06_harvest.txt 2>&1; tail -1 gpurun_out/r06_harvest.txt
timeout -k 10 200 python3 tools/qb_harvest.py --samples 300000 >> gpurun_out/r06_harvest.txt 2>&1; tail -1 gpurun_out/r06_harvest.txt
