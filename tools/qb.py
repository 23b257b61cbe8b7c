#!/usr/bin/env python3
"""quick bench helper: runs bench.py once and prints kernel ms / roofline fraction (dev tool)"""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + sys.argv[1:],
                     capture_output=True, text=True)
for line in out.stderr.splitlines():
    if line.startswith("[nps]"):
        print(line, flush=True)
try:
    d = json.loads(out.stdout.strip().splitlines()[-1])
    k = d["roofline"]["kernel_ms_per_step"]
    print("ms", {a: round(b, 4) for a, b in k.items() if b}, "frac %.4f" % d["roofline"]["frac"],
          "value %.4g" % d["value"], flush=True)
except Exception as e:
    print("FAILED", e, out.stdout[-500:], out.stderr[-1500:], flush=True)
