#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_mx.py -x -q -k "expect_passes or banded" > gpurun_out/r06_t4.log 2>&1; echo rc=$?; tail -6 gpurun_out/r06_t4.log
