#!/bin/bash
# dev tool: instruction histogram of ONE batch step (between two s_barrier) of the data waves' loop of
# fused_cw_kernel<960,0,1>, from the compiler's own assembly.   tools/isa_histogram.sh > profiles/rNN_isa_histogram.txt
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only \
    -o "$T/f.s" "$R/nimpress_amd/csrc/nps_fused.hip" 2>/dev/null
python3 - "$T/f.s" <<'PY'
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3nps15fused_cw_kernelILi960ELi0ELi1EEEvNS_9FusedArgsE:"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
# the data loop: the innermost loop with exactly four s_barrier (ring of 4 steps)
loops = [i for i, l in enumerate(body) if "Inner Loop Header" in l]
best = None
for h in loops:
    bars = []
    for j in range(h, len(body)):
        t = body[j].split()
        if t[:1] == ["s_barrier"]:
            bars.append(j)
        if t[:1] and t[0].startswith("s_cbranch") and len(bars) >= 4:
            break
        if j > h and "Loop Header" in body[j]:
            break
    if len(bars) == 4:
        best = bars
        break
a, b = best[0], best[1]
hist = collections.Counter()
for l in body[a + 1:b + 1]:
    t = l.strip()
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    hist[t.split()[0]] += 1
valu = sum(c for k, c in hist.items() if k.startswith("v_"))
print("# fused_cw_kernel<960,0,1>, one batch step of a data wave (16 rows x 16 samples per lane), hipcc -O3 gfx950")
print("# VALU %d  LDS %d  VMEM %d  SALU %d  s_waitcnt %d" % (
    valu, sum(c for k, c in hist.items() if k.startswith("ds_")),
    sum(c for k, c in hist.items() if k.startswith(("buffer_", "global_"))),
    sum(c for k, c in hist.items() if k.startswith("s_") and k not in ("s_waitcnt", "s_barrier", "s_nop")),
    hist["s_waitcnt"]))
for k, c in hist.most_common():
    print("%6d %s" % (c, k))
PY
rm -rf "$T"
