#!/bin/bash
# dev tool: sample GPU clocks/power while a bench runs.  usage: clkwatch.sh <Q> 
Q=$1
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clk_q$Q.log &
W=$!
NPS_FUSED_MAXQ=$Q python tools/qb.py --samples 500000 --variants 400000 --steps 400 --warmup 2 --mode fused
wait $W
sort gpurun_out/clk_q$Q.log | uniq -c | sort -rn | head -8
