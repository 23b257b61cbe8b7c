#!/usr/bin/env python3
"""Register / LDS / scratch use of the hot kernels as the compiler reports it (no GPU needed):
    python tools/kernel_resources.py r05   ->  profiles/r05_kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
WANT = ("fused_mx", "mx_given", "ds_fused", "multi_mfma", "fused_cw", "mx_ops", "mx_tally", "mx_fold", "ds_accumulate", "ds_tally")
out = ["# round %s: registers, LDS and scratch of the hot kernels as `hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage`" % tag[1:].lstrip("0"),
       "# reports them (tools/kernel_resources.py; no GPU involved).  Wave64 code on gfx950 allocates vector registers in granules of 8:",
       "# 'allocated' is the count rounded up to a multiple of 8.  rocprofv3's VGPR_Count column shows HALF of that (it decodes the kernel",
       "# descriptor's granule count with a granule of 4): 124 there = 31 granules x 8 = 248 registers for a kernel that uses 241 .. 248.",
       "%-18s %-62s %5s %9s %5s %7s %8s %8s %6s" % ("source", "kernel", "VGPRs", "allocated", "AGPRs", "spilled", "scratch", "LDS", "waves")]
for src in ("nps_mx.hip", "nps_mxg.hip", "nps_ds_fused.hip", "nps_ds.hip", "nps_multi.hip", "nps_fused.hip"):
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-parameter",
                        "--cuda-device-only", "-c", os.path.join(ROOT, "nimpress_amd", "csrc", src), "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    recs, cur = [], None
    for ln in r.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", ln)
        if m:
            cur = {"name": m.group(1)}
            recs.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z][^:]*): (\S+) \[-Rpass", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    for rec in recs:
        if not any(w in rec["name"] for w in WANT):
            continue
        dem = subprocess.run(["c++filt", rec["name"]], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("void ", "")
        v = int(rec.get("VGPRs", 0))
        out.append("%-18s %-62s %5d %9d %5s %7s %8s %8s %6s" % (src, dem[:62], v, (v + 7) // 8 * 8, rec.get("AGPRs", "?"),
                                                               rec.get("VGPRs Spill", "?"), rec.get("ScratchSize [bytes/lane]", "?"),
                                                               rec.get("LDS Size [bytes/block]", "?"), rec.get("Occupancy [waves/SIMD]", "?")))
path = os.path.join(ROOT, "profiles", "%s_kernel_resources.txt" % tag)
open(path, "w").write("\n".join(out) + "\n")
print("wrote", path)
