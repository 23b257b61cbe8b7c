#!/usr/bin/env python3
"""quick bench of the multi-score path alone (dev tool):  python tools/qb_multi.py [--samples N] [--variants M] [--scores S]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=500_000)
ap.add_argument("--variants", type=int, default=1_000_000)
ap.add_argument("--scores", type=int, default=8)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--zero", action="store_true", help="all genotypes 0 (clock / power experiment)")
ap.add_argument("--no-missing", action="store_true", help="a cohort without missing genotypes")
ap.add_argument("--missing-bits", type=int, default=0, help="nps_multi_set_missing_weight_bits (32 or 56)")
ap.add_argument("--weight-bits", type=int, default=0, help="nps_multidef_create_bits (49 or 41)")
a = ap.parse_args()
import torch
from nimpress_amd import capi
import bench
n, m, S, seed = a.samples, a.variants, a.scores, 20250103
_, eaf, miss = bench.synth_score(m, seed)
th, tm, tmi = bench.hwe_thresholds(eaf, miss)
if a.zero:
    th[:] = 0; tm[:] = 0; tmi[:] = 0
if a.no_missing:
    tmi[:] = 0
t0 = time.perf_counter()
co = capi.Cohort(n, m, fmt=capi.FMT_GT2M)
for x in range(0, m, 1 << 15):
    y = min(m, x + (1 << 15))
    co.synth_at(x, x, seed, th[x:y], tm[x:y], tmi[x:y])
print("cohort generated in %.2f s" % (time.perf_counter() - t0), flush=True)
descs = np.zeros((S, m), dtype=capi.ROW_DESC_DTYPE)
for s in range(S):
    descs[s]["beta"] = np.round(np.random.default_rng(seed + 1000 + s).normal(0.0, 0.02, m), 4)
    descs[s]["eaf"] = eaf
mdef = capi.MultiDef(descs, weight_bits=a.weight_bits)
msc = capi.MultiScorer(n, capi.make_params(), S)
if a.missing_bits:
    msc.set_missing_weight_bits(a.missing_bits)
d_scores = torch.empty((S, n), dtype=torch.float64, device="cuda")
off = np.zeros(S)
for i in range(a.steps + 1):
    msc.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    msc.score_cohort(co, mdef)
    nl = msc.finish_device(off, d_scores.data_ptr())
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    p = msc.timing()
    if i:
        alg = m * ((n + 15) // 16) * 4
        print("pass %d: wall %.2f ms  params %.3f  product %.3f  fold %.3f ms   %.2f TB/s  %.2f ms/score  checksum %.6g"
              % (i, wall * 1e3, p[0], p[1], p[2], alg / (p[1] * 1e-3) / 1e12, wall * 1e3 / S,
                 float(d_scores[:, ::977].sum().item())), flush=True)
