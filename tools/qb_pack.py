import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import bench
from nimpress_amd import capi
n, mp, seed = 500_000, 1<<16, 20250103
_, eaf, miss = bench.synth_score(mp, seed)
th, tm, tmi = bench.hwe_thresholds(eaf, miss)
src = capi.Cohort(n, mp); src.synth_at(0,0,seed,th,tm,tmi)
for fmt,name in ((capi.FMT_GT2M,'GT2M'),(capi.FMT_GT2X,'GT2X')):
    dst = capi.Cohort(n, mp, fmt=fmt); dst.convert_from(src)
    t0=time.perf_counter(); dst.convert_from(src); dt=time.perf_counter()-t0
    print(name, "%.1f ms per 1M rows, %.2f TB/s r+w" % (dt*1e3*1e6/mp, 2.0*mp*((n+15)//16)*4/dt/1e12))
    dst.close()
