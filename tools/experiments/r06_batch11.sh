#!/bin/bash
# round 6, batch 11: what would one more step of slack for the hand-over buy?  NPS_MX_DEBUG=32 looks at the PREVIOUS superblock's
# tally words (published a step earlier): wrong results, the timing of a kernel whose tallies are counted one step further ahead
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_b11; mkdir -p $O
cd $R
run() { d=$1; tag=$2; shift 2
  ( cd $R/$d && env "$@" timeout -k 10 240 python3 tools/qb_mx.py $ARGS ) > $O/$tag.txt 2>&1
  echo "== $tag: $(grep -h 'best' $O/$tag.txt | tail -1 | cut -c60-190)"; }
ARGS="--mode 2"
for i in 1 2 3; do run exp/sl base_$i X=1 && run exp/sl slack_$i NPS_MX_DEBUG=32 || exit 1; done
run exp/sl nohand NPS_MX_DEBUG=4
run exp/slt t_slack NPS_MX_DEBUG=32
ARGS="--mode 2 --samples 400000"
run exp/sl base_400 X=1 && run exp/sl slack_400 NPS_MX_DEBUG=32
