#!/bin/bash
# dev tool: like tools/mkexp.sh, but of a git revision:   tools/mkexp_rev.sh NAME REV [extra hipcc flags...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; REV=$2; shift 2
D=$R/exp/$N
rm -rf "$D"; mkdir -p "$D"
git -C "$R" archive "$REV" bench.py nimpress_amd oracle tools include | tar -x -C "$D"
cp "$R"/oracle/*.so "$D"/oracle/ 2>/dev/null || true
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off "$@" \
    -o "$D"/nimpress_amd/libnps.so "$D"/nimpress_amd/csrc/*.hip
echo "built $D from $REV"
