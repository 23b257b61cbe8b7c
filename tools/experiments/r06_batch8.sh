#!/bin/bash
# round 6, batch 8: is the per-process "fast / slow mode" a matter of where the cohort lies in VRAM?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_b8; mkdir -p $O
cd $R
run() { d=$1; tag=$2; shift 2
  ( cd $R/$d && env "$@" timeout -k 10 240 python3 tools/qb_mx.py $ARGS ) > $O/$tag.txt 2>&1
  echo "== $tag: $(grep -h 'nps\]' $O/$tag.txt | tail -1 | cut -c1-120) | $(grep -h 'best' $O/$tag.txt | tail -1 | cut -c60-200)"; }
ARGS="--mode 2 --samples 300000"
for i in 1 2 3; do run exp/ct def300_$i NPS_COHORT_CONTIG=0 && run exp/ct con300_$i NPS_COHORT_CONTIG=1 || exit 1; done
ARGS="--mode 2"
for i in 1 2 3; do run exp/ct def500_$i NPS_COHORT_CONTIG=0 && run exp/ct con500_$i NPS_COHORT_CONTIG=1 || exit 1; done
