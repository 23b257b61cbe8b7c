#!/bin/bash
mkdir -p gpurun_out/e6
{
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for V in 2 27 35; do
  echo "== parity VAR=$V"; NPS_FUSED_VAR=$V NPS_FUSED_THREADS=1024 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "fused or resident or cohort" 2>&1 | tail -2
done
for V in 0 2 19 27 35 0 2; do
  echo "== VAR=$V"; NPS_FUSED_VAR=$V python tools/qb.py --samples 500000 --variants 400000 --steps 8 --warmup 2 --mode fused
done
} > gpurun_out/e6/log.txt 2>&1
cat gpurun_out/e6/log.txt
