#!/bin/bash
mkdir -p gpurun_out/e4
{
for V in 0 2; do
  echo "== VAR=$V"; NPS_TELEMETRY=1 NPS_FUSED_VAR=$V python tools/qb.py --samples 500000 --variants 400000 --steps 3 --warmup 1 --mode fused
done
for D in 1 2 3; do
  echo "== DBG=$D"; NPS_TELEMETRY=1 NPS_DEBUG_FLAGS=$D python tools/qb.py --samples 500000 --variants 400000 --steps 3 --warmup 1 --mode fused
done
for Q in 1 3 5; do
  echo "== MAXQ=$Q"; NPS_TELEMETRY=1 NPS_FUSED_MAXQ=$Q python tools/qb.py --samples 500000 --variants 100000 --steps 3 --warmup 1 --mode fused
done
} > gpurun_out/e4/log.txt 2>&1
cat gpurun_out/e4/log.txt
