#!/bin/bash
# round 6, batch 7: FOUR control waves (one per SIMD, 32 rows each): 10|10|6|5 etc. -- parity first, then timing
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_b7; mkdir -p $O
cd $R/exp/c4 && NPS_MX_SPLIT=10,10,6,5,4 timeout -k 10 400 python3 -m pytest tests/test_gpu_mx.py -x -q -k "not two_threads" > $O/tests_c4.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests_c4.log
cd $R
run() { d=$1; tag=$2; shift 2
  ( cd $R/$d && env "$@" timeout -k 10 240 python3 tools/qb_mx.py $ARGS ) > $O/$tag.txt 2>&1
  echo "== $tag: $(grep -h 'best' $O/$tag.txt | tail -1)"; }
ARGS="--mode 2"
run exp/c4 base X=1 && run exp/c4 c4_10_10_6_5 NPS_MX_SPLIT=10,10,6,5,4 && run exp/c4 c4_10_10_6_5_prio NPS_MX_SPLIT=10,10,6,5,4 NPS_MX_PRIO=1 && run exp/c4 c4_10_9_6_6 NPS_MX_SPLIT=10,9,6,6,4 && run exp/c4 c4_9_9_7_6 NPS_MX_SPLIT=9,9,7,6,4 && run exp/c4 c4_10_10_5_6 NPS_MX_SPLIT=10,10,5,6,4 && run exp/c4 base2 X=1 && \
run exp/c4t t_c4_10_10_6_5 NPS_MX_SPLIT=10,10,6,5,4 && run exp/c4t t_c4_10_10_6_5_prio NPS_MX_SPLIT=10,10,6,5,4 NPS_MX_PRIO=1 || exit 1
ARGS="--mode 2 --samples 400000"
run exp/c4 base_400k X=1 && run exp/c4 c4_400k NPS_MX_SPLIT=10,10,6,5,4 && run exp/c4 c4_400k_noprio NPS_MX_SPLIT=10,10,6,5,4 NPS_MX_PRIO=0
ARGS="--mode 2 --samples 300000"
run exp/c4 base_300k X=1 && run exp/c4 c4_300k NPS_MX_SPLIT=10,10,6,5,4
