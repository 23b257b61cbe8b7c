#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_mx.py -x -q -k "auto_counts or expect_passes or two_threads or given" > gpurun_out/r06_t2.log 2>&1; echo rc=$?; tail -8 gpurun_out/r06_t2.log
for n in 200000 250000 300000; do timeout -k 5 120 python3 tools/qb_harvest.py --samples $n 2>&1 | grep -v amdgpu | tail -1; done
