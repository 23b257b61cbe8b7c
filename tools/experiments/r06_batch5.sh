#!/bin/bash
# round 6, batch 5: at 300 000 / 400 000 samples the step is the data waves of SIMD 0/1 (19 units) and the control waves wait
# 1 600 cycles at the barrier: splits that move units to the control waves' SIMDs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_b5; mkdir -p $O
cd $R
run() { d=$1; tag=$2; shift 2
  ( cd $R/$d && env "$@" timeout -k 10 240 python3 tools/qb_mx.py $ARGS ) > $O/$tag.txt 2>&1
  echo "== $tag: $(grep -h 'best' $O/$tag.txt | tail -1)"; }
for n in 300000 400000; do
ARGS="--mode 2 --samples $n"
run exp/u1 base_$n X=1 && run exp/u1 s9_10_9_3_$n NPS_MX_SPLIT=9,10,9,3 && run exp/u1 s9_10_8_4_$n NPS_MX_SPLIT=9,10,8,4 && run exp/u1 s8_10_9_4_$n NPS_MX_SPLIT=8,10,9,4 && run exp/u1 s8_10_8_5_$n NPS_MX_SPLIT=8,10,8,5 && run exp/u1 s7_10_8_6_$n NPS_MX_SPLIT=7,10,8,6 && run exp/u1 s8_12_8_3_$n NPS_MX_SPLIT=8,12,8,3 && run exp/u1 s8_11_8_4_$n NPS_MX_SPLIT=8,11,8,4 && run exp/u1 base2_$n X=1 || exit 1
done
ARGS="--mode 2"
run exp/u1 base_500k X=1 && run exp/u1 s8_10_9_4_500k NPS_MX_SPLIT=8,10,9,4 && run exp/u1 s8_10_8_5_500k NPS_MX_SPLIT=8,10,8,5 && run exp/u1 s9_10_8_4_500k NPS_MX_SPLIT=9,10,8,4 && run exp/u1 base2_500k X=1
