#!/usr/bin/env python3
"""dev tool: soak test of the fused kernels -- many passes over a resident cohort, every pass must give
bit-identical scores and nloci (no float atomics, fixed combine order) and no bounded wait may expire.
    python tools/soak.py [--format ds] [--samples N] [--variants M] [--passes K]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from nimpress_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--format", default="gt")
ap.add_argument("--samples", type=int, default=500_000)
ap.add_argument("--variants", type=int, default=200_000)
ap.add_argument("--passes", type=int, default=1000)
a = ap.parse_args()
n, m = a.samples, a.variants
_, eaf, miss = bench.synth_score(m, 7)
th, tm, tmi = bench.hwe_thresholds(eaf, miss)
is_ds = a.format == "ds"
co = capi.Cohort(n, m, fmt=capi.FMT_DS32 if is_ds else capi.FMT_GT2)
for r0 in range(0, m, 1 << 15):
    r1 = min(m, r0 + (1 << 15))
    co.synth(r0, 7, th[r0:r1], tm[r0:r1], tmi[r0:r1])
co.optimize()
beta = np.round(np.random.default_rng(8).normal(0, 0.02, m), 4)
sdef = capi.ScoreDef(capi.row_descs(beta, eaf))
sc = capi.Scorer(n, capi.make_params())
d = torch.empty(n, dtype=torch.float64, device="cuda")
ref = None
t0 = time.time()
for k in range(a.passes):
    sc.reset()
    sc.score_cohort_def(co, sdef, 0, capi.MODE_FUSED)
    nloci = sc.finish_device(0.0, d.data_ptr())     # raises on NPS_E_TIMEOUT
    cur = d.clone()
    if ref is None:
        ref, ref_nloci = cur, nloci
    else:
        assert nloci == ref_nloci, (k, nloci, ref_nloci)
        assert bool(torch.equal(cur.view(torch.int64), ref.view(torch.int64))), "pass %d differs" % k
    if k % 200 == 0:
        print("pass %d ok (%.1f s)" % (k, time.time() - t0), flush=True)
print("soak ok: %d passes of %s %d x %d bit-identical, nloci %d, %.1f s" % (a.passes, a.format, n, m, ref_nloci,
                                                                          time.time() - t0))
