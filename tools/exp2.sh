#!/bin/bash
# dev tool: experiment batch 2 (lookup FIFO depth, DPP bank-mask reduce) -- parity first, then timing
mkdir -p gpurun_out/e2
{
./build/ubench_valu | grep -i "cndmask\|bank_mask\|s_nop\|ds_add"
for V in 2 17 21 19 23; do
  echo "== parity VAR=$V"; NPS_FUSED_VAR=$V NPS_FUSED_THREADS=1024 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "fused or resident or cohort" 2>&1 | tail -2
done
for V in 0 2 17 21 19 23 0; do
  echo "== VAR=$V"; NPS_FUSED_VAR=$V python tools/qb.py --samples 500000 --variants 400000 --steps 8 --warmup 2 --mode fused
done
} > gpurun_out/e2/log.txt 2>&1
tail -40 gpurun_out/e2/log.txt
