#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_b10; mkdir -p $O
cd $R
run() { d=$1; tag=$2; shift 2
  ( cd $R/$d && env "$@" timeout -k 10 240 python3 tools/qb_mx.py $ARGS ) > $O/$tag.txt 2>&1
  echo "== $tag: $(grep -h 'best' $O/$tag.txt | tail -1 | cut -c60-190)"; }
ARGS="--mode 1"
for i in 1 2 3; do run exp/rel rel_given_$i X=1 && run exp/ilp ilp_given_$i X=1 || exit 1; done
ARGS="--mode 0 --samples 400000"
for i in 1 2; do run exp/rel rel_400_$i X=1 && run exp/ilp ilp_400_$i X=1 || exit 1; done
