#!/usr/bin/env python3
"""dev tool: soak test of the fused kernels -- many passes over a resident cohort, alternating between TWO
score definitions (so that a value handed over too early -- a stale partial sum, a stale tally word of the
previous pass -- is a different number, not the same bits again); every pass must give bit-identical
scores and nloci for its definition (no float atomics, fixed combine order), no bounded wait may expire,
and the fused result is compared with the two-pass kernels at the start.  Build with -DNPS_DIAGNOSTICS
(tools/mkexp.sh) to have the DS kernel's partial sums poisoned with NaNs before every launch as well.
    python tools/soak.py [--format ds|multi|strip] [--samples N] [--variants M] [--passes K]
--format strip: the matrix-core single-score kernel on a NPS_FMT_GT2X cohort (two-stage tally hand-over).
--format multi: the multi-score product kernel (LDS-DMA staging with hand-counted waits): two sets of 8
definitions alternate, every pass bit-identical per set, set 0 checked against the single-score fused kernel."""
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
ap.add_argument("--mode", type=int, default=2, help="mode of the soaked passes: 2 single read (default), 0 auto, 1 two reads")
a = ap.parse_args()
n, m = a.samples, a.variants
_, eaf, miss = bench.synth_score(m, 7)
th, tm, tmi = bench.hwe_thresholds(eaf, miss)
if a.format == "multi":
    S = 8
    co = capi.Cohort(n, m, fmt=capi.FMT_GT2M)
    gt = capi.Cohort(n, m)
    for r0 in range(0, m, 1 << 15):
        r1 = min(m, r0 + (1 << 15))
        co.synth_at(r0, r0, 7, th[r0:r1], tm[r0:r1], tmi[r0:r1])
        gt.synth_at(r0, r0, 7, th[r0:r1], tm[r0:r1], tmi[r0:r1])
    sets = []
    for j in range(2):
        d = np.zeros((S, m), dtype=capi.ROW_DESC_DTYPE)
        for s_ in range(S):
            d[s_]["beta"] = np.round(np.random.default_rng(100 * j + s_).normal(0, 0.02 * (1 + j), m), 4)
            d[s_]["eaf"] = eaf
        sets.append((d, capi.MultiDef(d)))
    msc = capi.MultiScorer(n, capi.make_params(), S)
    out = torch.empty((S, n), dtype=torch.float64, device="cuda")
    first = [None, None]
    t0 = time.time()
    for k in range(a.passes):
        j = k & 1
        msc.reset()
        msc.score_cohort(co, sets[j][1])
        nl = msc.finish_device(np.zeros(S), out.data_ptr())
        cur = out.clone()
        if first[j] is None:
            first[j] = (cur, nl.copy())
            if j == 0:
                sc = capi.Scorer(n, capi.make_params())
                d1 = torch.empty(n, dtype=torch.float64, device="cuda")
                for s_ in (0, S - 1):
                    sc.reset()
                    sc.score_cohort(gt, sets[0][0][s_])
                    sc.finish_device(0.0, d1.data_ptr())
                    scale = float(d1.abs().max().item()) + 1e-300
                    assert float((cur[s_] - d1).abs().max().item()) <= 1e-9 * scale, "multi != single-score path"
        else:
            assert np.array_equal(nl, first[j][1]), (k, nl)
            assert bool(torch.equal(cur.view(torch.int64), first[j][0].view(torch.int64))), "pass %d differs" % k
        if k % 100 == 0:
            print("pass %d ok (%.1f s)" % (k, time.time() - t0), flush=True)
    print("soak ok: %d multi-score passes of 8 scores x %d x %d, two definition sets alternating, bit-identical per "
          "set and equal to the single-score kernel, %.1f s" % (a.passes, n, m, time.time() - t0))
    sys.exit(0)
is_ds = a.format == "ds"
is_strip = a.format == "strip"   # NPS_FMT_GT2X: the matrix-core kernel, checked against the table kernels on a NPS_FMT_GT2 copy
co = capi.Cohort(n, m, fmt=capi.FMT_DS32 if is_ds else (capi.FMT_GT2X if is_strip else capi.FMT_GT2))
for r0 in range(0, m, 1 << 15):
    r1 = min(m, r0 + (1 << 15))
    co.synth(r0, 7, th[r0:r1], tm[r0:r1], tmi[r0:r1])
co.optimize()
ref_co = co
if is_strip:
    ref_co = capi.Cohort(n, m)
    for r0 in range(0, m, 1 << 15):
        r1 = min(m, r0 + (1 << 15))
        ref_co.synth(r0, 7, th[r0:r1], tm[r0:r1], tmi[r0:r1])
params = capi.make_params(imp_locus="ps") if is_ds else capi.make_params()
defs, refs = [], []
for j in range(2):
    beta = np.round(np.random.default_rng(8 + j).normal(0, 0.02 * (1 + j), m), 4)
    eaf_j = eaf if j == 0 else np.round(np.clip(eaf * 1.5, 0.0, 1.0), 4)
    defs.append(capi.ScoreDef(capi.row_descs(beta, eaf_j)))
sc = capi.Scorer(n, params)
d = torch.empty(n, dtype=torch.float64, device="cuda")
t0 = time.time()
for j in range(2):   # reference per definition: the two-pass kernels
    sc.reset()
    sc.score_cohort_def(ref_co, defs[j], 0, capi.MODE_TWOPASS)
    nl = sc.finish_device(0.0, d.data_ptr())
    refs.append([d.clone(), nl, None])
if is_strip:
    ref_co.close()
if a.mode == 0 and is_strip:
    # NPS_MODE_AUTO may keep the tallies the FIRST pass counts and run every later pass with them given (round 6): two kernels
    # whose locus constants are summed in different orders (last-bit differences).  One pass up front, so that every compared
    # pass runs the kernel the steady state runs.
    sc.reset()
    sc.score_cohort_def(co, defs[0], 0, a.mode)
    sc.finish_device(0.0, d.data_ptr())
    print("auto: tallies kept after the first pass: %s" % co.has_tallies(), flush=True)
for k in range(a.passes):
    j = k & 1
    sc.reset()
    sc.score_cohort_def(co, defs[j], 0, a.mode)
    nloci = sc.finish_device(0.0, d.data_ptr())     # raises on NPS_E_TIMEOUT
    cur = d.clone()
    twopass, ref_nloci, first = refs[j]
    assert nloci == ref_nloci, (k, nloci, ref_nloci)
    if first is None:
        refs[j][2] = cur
        scale = float(twopass.abs().max().item()) + 1e-300
        assert float((cur - twopass).abs().max().item()) <= 1e-9 * scale, "fused != two-pass for definition %d" % j
    else:
        assert bool(torch.equal(cur.view(torch.int64), first.view(torch.int64))), "pass %d differs" % k
    if k % 200 == 0:
        print("pass %d ok (%.1f s)" % (k, time.time() - t0), flush=True)
print("soak ok: %d passes of %s %d x %d, two definitions alternating, bit-identical per definition and equal to "
      "the two-pass kernels, nloci %d / %d, %.1f s" % (a.passes, a.format, n, m, refs[0][1], refs[1][1], time.time() - t0))
