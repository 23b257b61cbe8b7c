#!/usr/bin/env python3
"""dev tool: what does keeping the tallies cost the pass that counts them?  A resident NPS_FMT_GT2X cohort scored under
NPS_MODE_AUTO: passes 1-3 in the pass (no hint), then nps_cohort_expect_passes(2): pass 4 keeps its tallies, passes 5-7 run
with them given.   python tools/qb_harvest.py [--samples N] [--variants M]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=500_000)
ap.add_argument("--variants", type=int, default=1_000_000)
a = ap.parse_args()
import torch
from nimpress_amd import capi
n, m, seed = a.samples, a.variants, 20250103
rng = np.random.default_rng(seed)
beta = np.round(rng.normal(0.0, 0.02, m), 4)
SC = 4294967296.0
f = lambda x: np.minimum(np.floor(np.asarray(x, dtype=np.float64) * SC), 4294967295.0).astype(np.uint32)
eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
miss = rng.uniform(0, 0.02, m)
th, tm, tmi = f(eaf * eaf + 2 * eaf * (1 - eaf)), f(eaf * eaf), f(miss)
co = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
for x in range(0, m, 1 << 15):
    y = min(m, x + (1 << 15))
    co.synth_at(x, x, seed, th[x:y], tm[x:y], tmi[x:y])
sdef = capi.ScoreDef(capi.row_descs(beta, 0.3 * np.ones(m)))
sc = capi.Scorer(n, capi.make_params())
d = torch.empty(n, dtype=torch.float64, device="cuda")
out = []
for i in range(8):
    if i == 4:
        co.expect_passes(2)
    sc.reset()
    sc.profile_enable(True)
    sc.profile_get(reset=True)
    sc.score_cohort_def(co, sdef, 0, capi.MODE_AUTO)
    sc.finish_device(0.0, d.data_ptr())
    p = sc.profile_get(reset=True)
    out.append("%s %.2f+%.3f" % ("in-pass" if p.n_fused else "given", p.ms_fused + p.ms_accumulate + p.ms_tally, p.ms_reduce))
print("%d x %d: kernel + fold ms per pass: %s   (the hint is given before pass 5; has_tallies now: %s)" % (n, m, " | ".join(out), co.has_tallies()), flush=True)
