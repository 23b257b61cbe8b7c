#!/bin/bash
# dev tool: experiment batch 1 (workgroup sizes, telemetry, diagnostics builds, instruction costs)
mkdir -p gpurun_out/e1
{
./build/ubench_valu
for T in 1024 960 896 832; do
  echo "== T=$T"; NPS_FUSED_THREADS=$T NPS_TELEMETRY=1 python tools/qb.py --samples 500000 --variants 400000 --steps 8 --warmup 2 --mode fused
done
for D in 1 2 3; do
  echo "== DBG=$D (T=1024)"; NPS_DEBUG_FLAGS=$D python tools/qb.py --samples 500000 --variants 400000 --steps 8 --warmup 2 --mode fused
done
} > gpurun_out/e1/log.txt 2>&1
tail -40 gpurun_out/e1/log.txt
