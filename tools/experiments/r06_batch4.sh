#!/bin/bash
# round 6, batch 4: control waves WITHOUT units of their own (interleaved accumulate: 11 / 12 units on a data wave fit)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_b4; mkdir -p $O
cd $R
run() { d=$1; tag=$2; shift 2
  ( cd $R/$d && env "$@" timeout -k 10 240 python3 tools/qb_mx.py $ARGS ) > $O/$tag.txt 2>&1
  echo "== $tag: $(grep -h 'best' $O/$tag.txt | tail -1)"; }
for n in 300000 400000; do
ARGS="--mode 2 --samples $n"
run exp/u0 base_$n X=1 && run exp/u0 s10_11_10_0_$n NPS_MX_SPLIT=10,11,10,0 && run exp/u0 s9_12_10_0_$n NPS_MX_SPLIT=9,12,10,0 && run exp/u0 s10_12_9_0_$n NPS_MX_SPLIT=10,12,9,0 && run exp/u0 s11_10_10_0_$n NPS_MX_SPLIT=11,10,10,0 && run exp/u0 base2_$n X=1 || exit 1
done
ARGS="--mode 2"
run exp/u0 base_500k X=1 && run exp/u0 s10_12_9_0_500k NPS_MX_SPLIT=10,12,9,0 && run exp/u0 s9_12_10_0_500k NPS_MX_SPLIT=9,12,10,0 && run exp/u0 s10_11_10_0_500k NPS_MX_SPLIT=10,11,10,0 || exit 1
ARGS="--mode 2 --samples 300000"
run exp/u0t t_base_300k X=1 && run exp/u0t t_10_11_10_0_300k NPS_MX_SPLIT=10,11,10,0
