#!/usr/bin/env python3
"""dev tool: the fused single-score pass over cohorts with different genotype distributions, plain layout vs
nps_cohort_optimize (the parity layout is data independent; what it is worth is not).
    python tools/qb_layouts.py [--samples N] [--variants M]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=500_000)
ap.add_argument("--variants", type=int, default=1_000_000)
ap.add_argument("--steps", type=int, default=4)
a = ap.parse_args()
import torch
from nimpress_amd import capi
n, m, seed = a.samples, a.variants, 20250103
rng = np.random.default_rng(seed)
beta = np.round(rng.normal(0.0, 0.02, m), 4)
SC = 4294967296.0


def thresholds(p_miss, p_hom, p_het):
    """per-row probabilities (of the non-missing part for hom / het) -> generator thresholds"""
    f = lambda x: np.minimum(np.floor(np.asarray(x, dtype=np.float64) * SC), 4294967295.0).astype(np.uint32)
    return f(p_hom + p_het), f(p_hom), f(p_miss)


def hwe(eaf, miss):
    return thresholds(miss, eaf * eaf, 2 * eaf * (1 - eaf))


one = np.ones(m)
cases = {
    "bench (eaf U(0.01,0.5), 1 % missing)": (hwe(np.round(rng.uniform(0.01, 0.5, m), 4), rng.uniform(0, 0.02, m)), {}),
    "low MAF (eaf U(0.001,0.05))": (hwe(rng.uniform(0.001, 0.05, m), rng.uniform(0, 0.02, m)), {}),
    "eaf 0.5 everywhere": (hwe(0.5 * one, rng.uniform(0, 0.02, m)), {}),
    "all heterozygous (non-HWE)": (thresholds(0.0 * one, 0.0 * one, 1.0 * one), {}),
    "20 % missing, --maxmis=1": (hwe(np.round(rng.uniform(0.01, 0.5, m), 4), 0.2 * one), dict(maxmis=1.0)),
    "uniform codes (random .bed bytes), --maxmis=1": (thresholds(0.25 * one, one / 3, one / 3), dict(maxmis=1.0)),
}
for name, ((th, tm, tmi), kw) in cases.items():
    co = capi.Cohort(n, m)
    for x in range(0, m, 1 << 15):
        y = min(m, x + (1 << 15))
        co.synth_at(x, x, seed, th[x:y], tm[x:y], tmi[x:y])
    sdef = capi.ScoreDef(capi.row_descs(beta, 0.3 * one))
    sc = capi.Scorer(n, capi.make_params(**kw))
    d = torch.empty(n, dtype=torch.float64, device="cuda")
    res = []
    for layout in ("plain", "optimized"):
        if layout == "optimized":
            co.optimize()
        best = 1e9
        for i in range(a.steps + 1):
            sc.reset()
            sc.profile_enable(True)
            sc.profile_get(reset=True)
            sc.score_cohort_def(co, sdef, 0, capi.MODE_FUSED)
            sc.finish_device(0.0, d.data_ptr())
            p = sc.profile_get(reset=True)
            if i:
                best = min(best, p.ms_fused)
        res.append(best)
    alg = m * ((n + 15) // 16) * 4 + 40 * m + 8 * n
    print("%-48s plain %.2f ms (%.1f %%)   optimized %.2f ms (%.1f %%)" % (
        name, res[0], alg / res[0] / 8e9 * 100, res[1], alg / res[1] / 8e9 * 100), flush=True)
    sc.close(); sdef.close(); co.close()
    torch.cuda.empty_cache()
