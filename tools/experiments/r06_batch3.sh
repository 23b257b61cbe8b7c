#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_b3; mkdir -p $O
cd $R
run() { d=$1; tag=$2; shift 2
  ( cd $R/$d && env "$@" timeout -k 10 240 python3 tools/qb_mx.py $ARGS ) > $O/$tag.txt 2>&1
  echo "== $tag: $(grep -h 'best' $O/$tag.txt | tail -1)"; }
ARGS="--mode 1"
run . given_t2 X=1 && run exp/g8 given_t8 X=1 && run . given_t2b X=1 && run exp/g8 given_t8b X=1 || exit 1
ARGS="--mode 0 --samples 300000"
run . auto300_t2 X=1 && run exp/g8 auto300_t8 X=1 || exit 1
ARGS="--mode 0 --samples 400000"
run . auto400_t2 X=1 && run exp/g8 auto400_t8 X=1 || exit 1
ARGS="--mode 0 --samples 100000"
run . auto100 X=1
timeout -k 10 600 python3 bench.py --full-out $O/bench_full.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 3000 $O/bench.json
