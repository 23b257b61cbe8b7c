#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_mx.py -x -q -k "auto_counts or expect_passes or two_threads or given" > gpurun_out/r06_t1.log 2>&1; echo rc=$?; tail -15 gpurun_out/r06_t1.log
