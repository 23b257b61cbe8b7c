#!/bin/bash
# round 6, batch 1: unit splits of the strip kernel, parts switched off, timers, given-tallies kernel with one bank
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_b1; mkdir -p $O
cd $R
run() { # dir, tag, env..., -- args
  d=$1; tag=$2; shift 2
  ( cd $R/$d && env "$@" timeout -k 10 240 python3 tools/qb_mx.py $ARGS ) > $O/$tag.txt 2>&1
  echo "== $tag: $(grep -h 'best' $O/$tag.txt | tail -1)"
}
ARGS="--mode 2"
run . base X=1 && run exp/d d_default X=1 && run exp/d s10_10_8_3 NPS_MX_SPLIT=10,10,8,3 && run exp/d s9_10_9_3 NPS_MX_SPLIT=9,10,9,3 && run exp/d s9_10_8_4 NPS_MX_SPLIT=9,10,8,4 && run . base2 X=1 && \
run exp/d dbg1 NPS_MX_DEBUG=1 && run exp/d dbg4 NPS_MX_DEBUG=4 && run exp/d dbg5 NPS_MX_DEBUG=5 && \
run exp/t t_base X=1 && run exp/t t_10_10_8_3 NPS_MX_SPLIT=10,10,8,3 && run exp/t t_9_10_9_3 NPS_MX_SPLIT=9,10,9,3 || exit 1
ARGS="--mode 1"
run . given_b2 X=1 && run exp/g1 given_b1 X=1 && run exp/t t_given X=1 || exit 1
ARGS="--mode 2 --samples 400000"
run . base_400k X=1 && run exp/d s10_10_8_3_400k NPS_MX_SPLIT=10,10,8,3 && run exp/d s9_10_9_3_400k NPS_MX_SPLIT=9,10,9,3
