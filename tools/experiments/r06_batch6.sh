#!/bin/bash
# round 6, batch 6: 9|10|8|4 (release tree, up to 176 layout strips) against 10|10|9|2 everywhere (exp/f0), alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_b6; mkdir -p $O
cd $R
run() { d=$1; tag=$2; shift 2
  ( cd $R/$d && env "$@" timeout -k 10 240 python3 tools/qb_mx.py $ARGS ) > $O/$tag.txt 2>&1
  echo "== $tag: $(grep -h 'best' $O/$tag.txt | tail -1)"; }
for n in 300000 350000 270000; do
ARGS="--mode 2 --samples $n"
run exp/f0 old_a_$n X=1 && run . new_a_$n X=1 && run exp/f0 old_b_$n X=1 && run . new_b_$n X=1 && run exp/f0 old_c_$n X=1 && run . new_c_$n X=1 || exit 1
done
timeout -k 10 300 python3 -m pytest tests/test_gpu_mx.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
