import sys
sys.path.insert(0, '.')
import numpy as np
variant = sys.argv[1]
if variant != "notorch":
    import torch
    torch.cuda.set_device(0)
from nimpress_amd import capi
import bench
n, m = 50000, 2000
_, eaf, miss = bench.synth_score(m, 1)
th, tm, tmi = bench.hwe_thresholds(eaf, miss)
cohort = capi.Cohort(n, m, device=0)
cohort.synth(0, 1, th, tm, tmi)
beta = np.round(np.random.default_rng(5).normal(0.0, 0.02, m), 4)
sdef = capi.ScoreDef(capi.row_descs(beta, eaf), device=0)
sc = capi.Scorer(n, capi.make_params(), device=0)
sc.reset()
print("reset ok", flush=True)
sc.score_cohort_def(cohort, sdef, 0, capi.MODE_TWOPASS)
print("scored", flush=True)
if variant == "syncreset":
    sc.sync(); print("synced", flush=True)
    sc.reset(); print("reset after sync ok", flush=True)
s, nloci = sc.finish(0.0)
print("finish", nloci, flush=True)
sc.reset()
print("reset2 ok", flush=True)
