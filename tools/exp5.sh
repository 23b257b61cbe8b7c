#!/bin/bash
mkdir -p gpurun_out/e5
{
echo "== P=1 (no cross-workgroup coupling): 15360 samples x 4M rows"
NPS_TELEMETRY=1 python tools/qb.py --samples 15360 --variants 4000000 --steps 3 --warmup 1 --mode fused
echo "== P=2: 30720 samples x 2M rows"
NPS_TELEMETRY=1 python tools/qb.py --samples 30720 --variants 2000000 --steps 3 --warmup 1 --mode fused
echo "== P=8: 122880 samples x 1M rows"
NPS_TELEMETRY=1 python tools/qb.py --samples 122880 --variants 1000000 --steps 3 --warmup 1 --mode fused
echo "== P=33 reference"
NPS_TELEMETRY=1 python tools/qb.py --samples 500000 --variants 400000 --steps 3 --warmup 1 --mode fused
} > gpurun_out/e5/log.txt 2>&1
cat gpurun_out/e5/log.txt
