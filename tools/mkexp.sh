#!/bin/bash
# dev tool: freeze a copy of the tree's runnable parts (bench.py, the package with a libnps.so built with
# extra compiler flags, the oracle) under exp/NAME/, so that a gpurun call measures exactly that build
# while the main tree moves on.   tools/mkexp.sh NAME [extra hipcc flags...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
D=$R/exp/$N
rm -rf "$D"; mkdir -p "$D/nimpress_amd" "$D/oracle" "$D/tools"
cp "$R"/bench.py "$D"/
cp "$R"/nimpress_amd/*.py "$D"/nimpress_amd/
cp "$R"/oracle/*.py "$R"/oracle/*.so "$D"/oracle/ 2>/dev/null || true
cp "$R"/tools/qb.py "$R"/tools/qb_multi.py "$R"/tools/qb_mx.py "$D"/tools/
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off "$@" \
    -o "$D"/nimpress_amd/libnps.so "$R"/nimpress_amd/csrc/*.hip
echo "built $D"
