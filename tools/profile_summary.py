#!/usr/bin/env python3
"""Summarise a tools/profile_round.sh directory into the files kept under profiles/:
<tag>_bench_full.json, <tag>_kernel_stats.csv, <tag>_pmc_{FETCH,WRITE}_SIZE.csv (hot kernel rows only),
traffic.json (HBM bytes per launch of the dominant kernel, corrected as MI355X_MICROARCH.md says)."""
import csv
import glob
import json
import os
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out_dir, "summary")
os.makedirs(dst, exist_ok=True)


def find(pattern):
    hits = glob.glob(os.path.join(out_dir, pattern), recursive=True)
    return hits[0] if hits else None


line = json.loads(open(os.path.join(out_dir, "bench.json")).read().strip().splitlines()[-1])   # the compact line
full_path = os.path.join(out_dir, "bench_full.json")
bench = json.load(open(full_path)) if os.path.exists(full_path) else line                       # every secondary
json.dump(bench, open(os.path.join(dst, "%s_bench_full.json" % tag), "w"))
json.dump(line, open(os.path.join(dst, "%s_bench_line.json" % tag), "w"))
st = find("stats/**/*kernel_stats.csv")
if st:
    open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w").write(open(st).read())


kernel_names = set()


def pmc(name, counter):
    f = find("%s/**/*counter_collection.csv" % name)
    vals, rows = [], []
    if not f:
        return None
    rd = csv.DictReader(open(f))
    for r in rd:
        if "fused" in r["Kernel_Name"] and "fold" not in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
            rows.append(r)
            kernel_names.add(r["Kernel_Name"].split("(")[0].replace("void ", ""))
    with open(os.path.join(dst, "%s_pmc_%s.csv" % (tag, counter)), "w", newline="") as fo:
        if rows:
            w = csv.DictWriter(fo, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    vals = vals[1:] if len(vals) > 1 else vals  # drop the warm-up launch
    return (sum(vals) / len(vals), len(vals)) if vals else None


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
cfg = bench["config"]
is_ds = "FORMAT/DS" in cfg["workload"]
traffic = {"round": tag, "samples": cfg["samples"], "variants": cfg["variants"],
           "kernel": "nps::ds_fused_kernel" if is_ds else "nps::fused_cw_kernel<1024, 0>",
           "algorithmic_bytes_per_step": bench["roofline"]["algorithmic_bytes_per_step"]}
if fetch and write:
    traffic.update({
        "FETCH_SIZE_KB_per_launch": fetch[0], "WRITE_SIZE_KB_per_launch": write[0],
        "launches_averaged": [fetch[1], write[1]],
        "correction": "MI355X_MICROARCH.md HBM section: gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide "
                      "(16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE taken as is; separate --pmc passes",
        "hbm_bytes_per_step": 2.0 * fetch[0] * 1024.0 + write[0] * 1024.0})
if kernel_names:
    traffic["kernel"] = sorted(kernel_names)[0]
is_strip = "NPS_FMT_GT2X" in cfg.get("cohort_layout", "")
json.dump(traffic, open(os.path.join(dst, "traffic_ds.json" if is_ds else ("traffic_strip.json" if is_strip else "traffic.json")), "w"), indent=1)
print(json.dumps(traffic))
print(open(os.path.join(out_dir, "bench.json")).read().strip()[:600])
