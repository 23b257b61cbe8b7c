#!/bin/bash
# round 6, batch 2: MFMA / VALU interleaved accumulate (-DNPS_MX_INTERLEAVE), with 11- and 12-unit splits
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_b2; mkdir -p $O
cd $R
run() { d=$1; tag=$2; shift 2
  ( cd $R/$d && env "$@" timeout -k 10 240 python3 tools/qb_mx.py $ARGS ) > $O/$tag.txt 2>&1
  echo "== $tag: $(grep -h 'best' $O/$tag.txt | tail -1)"; }
ARGS="--mode 2"
run exp/d base X=1 && run exp/i i_default X=1 && run exp/i i_9_10_9_3 NPS_MX_SPLIT=9,10,9,3 && run exp/i i_9_11_9_2 NPS_MX_SPLIT=9,11,9,2 && \
run exp/i i_9_12_8_2 NPS_MX_SPLIT=9,12,8,2 && run exp/i i_8_12_9_2 NPS_MX_SPLIT=8,12,9,2 && run exp/i i_9_11_8_3 NPS_MX_SPLIT=9,11,8,3 && run exp/d base2 X=1 && \
run exp/i i_dbg4 NPS_MX_DEBUG=4 && run exp/i i_dbg5 NPS_MX_DEBUG=5 && \
run exp/it it_default X=1 && run exp/it it_9_12_8_2 NPS_MX_SPLIT=9,12,8,2 && run exp/it it_9_11_9_2 NPS_MX_SPLIT=9,11,9,2 || exit 1
ARGS="--mode 2 --samples 400000"
run exp/d base_400k X=1 && run exp/i i_400k X=1 && run exp/i i_9_10_9_3_400k NPS_MX_SPLIT=9,10,9,3 && run exp/i i_9_12_8_2_400k NPS_MX_SPLIT=9,12,8,2 && run exp/i i_8_12_9_2_400k NPS_MX_SPLIT=8,12,9,2 || exit 1
ARGS="--mode 2 --samples 300000"
run exp/i i_300k_inpass X=1
ARGS="--mode 0 --samples 300000"
run exp/i i_300k_auto X=1
ARGS="--mode 2 --samples 200000"
run exp/d base_200k X=1 && run exp/i i_200k X=1
