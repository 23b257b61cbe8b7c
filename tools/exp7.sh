#!/bin/bash
mkdir -p gpurun_out/e7
{
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for i in 1 2 3; do
  NPS_TELEMETRY=1 python tools/qb.py --samples 500000 --variants 400000 --steps 8 --warmup 2 --mode fused | tail -2
done
} > gpurun_out/e7/log.txt 2>&1
cat gpurun_out/e7/log.txt
