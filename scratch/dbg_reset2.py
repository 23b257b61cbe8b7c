import sys, faulthandler
#faulthandler.enable()
sys.path.insert(0, '.')
import numpy as np
import torch
torch.cuda.set_device(0)
from nimpress_amd import capi
import bench
n, m = 50000, 2000
_, eaf, miss = bench.synth_score(m, 1)
th, tm, tmi = bench.hwe_thresholds(eaf, miss)
cohort = capi.Cohort(n, m, device=0)
print("cohort", flush=True)
cohort.synth(0, 1, th, tm, tmi)
print("synth", flush=True)
beta = np.round(np.random.default_rng(5).normal(0.0, 0.02, m), 4)
sdef = capi.ScoreDef(capi.row_descs(beta, eaf), device=0)
print("sdef", sdef.n_present, flush=True)
sc = capi.Scorer(n, capi.make_params(), device=0)
print("scorer", flush=True)
d_scores = torch.empty(n, dtype=torch.float64, device="cuda")
print("torch alloc", flush=True)
sc.reset()
print("reset ok", flush=True)
for mode in (capi.MODE_TWOPASS, capi.MODE_FUSED):
    sc.reset()
    sc.score_cohort_def(cohort, sdef, 0, mode)
    print("scored", mode, flush=True)
    nloci = sc.finish_device(0.0, d_scores.data_ptr())
    print("finish", nloci, float(d_scores[:5].sum()), flush=True)
