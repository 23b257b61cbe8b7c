#!/usr/bin/env python3
"""dev tool: the single-score pass over a NPS_FMT_GT2X cohort (matrix-core kernel), by genotype distribution.
    python tools/qb_mx.py [--samples N] [--variants M] [--cases bench,eaf05,...] [--fmt 3|0]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=500_000)
ap.add_argument("--variants", type=int, default=1_000_000)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--cases", default="bench")
ap.add_argument("--fmt", type=int, default=3)
ap.add_argument("--mode", type=int, default=0, help="0 auto, 1 two reads (tally + accumulation), 2 single read")
ap.add_argument("--imp-sample", default=None, help="ps | homref | fail | int_ps | int_fail (default: the CLI default, int_ps)")
a = ap.parse_args()
import torch
from nimpress_amd import capi
n, m, seed = a.samples, a.variants, 20250103
rng = np.random.default_rng(seed)
beta = np.round(rng.normal(0.0, 0.02, m), 4)
SC = 4294967296.0


def thresholds(p_miss, p_hom, p_het):
    f = lambda x: np.minimum(np.floor(np.asarray(x, dtype=np.float64) * SC), 4294967295.0).astype(np.uint32)
    return f(p_hom + p_het), f(p_hom), f(p_miss)


def hwe(eaf, miss):
    return thresholds(miss, eaf * eaf, 2 * eaf * (1 - eaf))


one = np.ones(m)
bench_miss = rng.uniform(0, 0.02, m)
bench_miss[::1000] = 0.10
cases = {
    "bench": ("bench (eaf U(0.01,0.5), 1 % missing)", hwe(np.round(rng.uniform(0.01, 0.5, m), 4), bench_miss), {}),
    "lowmaf": ("low MAF (eaf U(0.001,0.05))", hwe(rng.uniform(0.001, 0.05, m), rng.uniform(0, 0.02, m)), {}),
    "eaf05": ("eaf 0.5 everywhere", hwe(0.5 * one, rng.uniform(0, 0.02, m)), {}),
    "het": ("all heterozygous (non-HWE)", thresholds(0.0 * one, 0.0 * one, 1.0 * one), {}),
    "miss20": ("20 % missing, --maxmis=1", hwe(np.round(rng.uniform(0.01, 0.5, m), 4), 0.2 * one), dict(maxmis=1.0)),
    "uniform": ("uniform codes, --maxmis=1", thresholds(0.25 * one, one / 3, one / 3), dict(maxmis=1.0)),
}
for key in a.cases.split(","):
    name, (th, tm, tmi), kw = cases[key]
    co = capi.Cohort(n, m, fmt=a.fmt)
    for x in range(0, m, 1 << 15):
        y = min(m, x + (1 << 15))
        co.synth_at(x, x, seed, th[x:y], tm[x:y], tmi[x:y])
    sdef = capi.ScoreDef(capi.row_descs(beta, 0.3 * one))
    if a.imp_sample:
        kw = dict(kw, imp_sample=a.imp_sample)
    sc = capi.Scorer(n, capi.make_params(**kw))
    d = torch.empty(n, dtype=torch.float64, device="cuda")
    times = []
    for i in range(a.steps + 1):
        sc.reset()
        sc.profile_enable(True)
        sc.profile_get(reset=True)
        sc.score_cohort_def(co, sdef, 0, a.mode)
        nloci = sc.finish_device(0.0, d.data_ptr())
        p = sc.profile_get(reset=True)
        times.append((p.ms_fused + p.ms_tally + p.ms_accumulate, p.ms_reduce, p.ms_tally))
    alg = m * ((n + 15) // 16) * 4 + 40 * m + 8 * n
    best = min(t[0] for t in times[1:])
    print("%-44s %d x %d fmt %d: kernels %s ms (tally pass %.2f), reduce %.3f ms; best %.2f ms = %.1f %% of 8 TB/s; nloci %d, score[0] %.12g" % (
        name, n, m, a.fmt, " ".join("%.2f" % t[0] for t in times), times[-1][2], times[-1][1], best, alg / best / 8e9 * 100, nloci,
        float(d[0])), flush=True)
    sc.close(); sdef.close(); co.close()
    torch.cuda.empty_cache()
