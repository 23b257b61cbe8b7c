#!/bin/bash
# round 6, batch 9: LLVM scheduling strategies for the whole library (max-ilp, max-memory-clause) against the default
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_b9; mkdir -p $O
cd $R
run() { d=$1; tag=$2; shift 2
  ( cd $R/$d && env "$@" timeout -k 10 240 python3 tools/qb_mx.py $ARGS ) > $O/$tag.txt 2>&1
  echo "== $tag: $(grep -h 'best' $O/$tag.txt | tail -1 | cut -c60-210)"; }
ARGS="--mode 2"
run exp/rel rel_a X=1 && run exp/ilp ilp_a X=1 && run exp/mmc mmc_a X=1 && run exp/rel rel_b X=1 && run exp/ilp ilp_b X=1 && run exp/mmc mmc_b X=1 || exit 1
ARGS="--mode 1"
run exp/rel rel_given X=1 && run exp/ilp ilp_given X=1 && run exp/mmc mmc_given X=1
