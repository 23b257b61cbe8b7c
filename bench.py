#!/usr/bin/env python3
"""bench.py -- headline benchmark: genotype-dosage accumulations/s on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

Workload (BASELINE.json configs[2], the configuration the north-star metric is quoted on):
synthetic 1 000 000-variant PRS on a 500 000-sample bit-packed GT matrix (125 GB of 2-bit codes)
resident in HBM; flags = CLI defaults (imp-locus ps, imp-missing homref, imp-sample int_ps,
maxmis 0.05, mincs 100); every 1000th row has 10 % missingness so the locus-imputation branch
runs.  One STEP = one full pass of the hot path over the cohort: tally -> per-row decision/LUT ->
accumulate -> /(2 nloci) + offset, inputs already in HBM, result left in a device buffer.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  Started under
torch.distributed.run the ranks come from the environment; started as plain `python bench.py --gpus N`
this process spawns the N rank processes itself (before anything here touches a GPU) and relays
rank 0's line.
  --scaling weak (default): each rank scores ITS OWN score definition (its own betas) against its own
      resident copy of the cohort -- multi-score evaluation sharded by score file, BASELINE.json
      north_star -- and each step ends with the one real exchange of that layout, an RCCL all-gather of
      the samples x scores matrix.
  --scaling strong: ONE 1M-row score, its rows sharded in contiguous blocks (every GPU holds all
      samples of its rows, so tallies stay local and exact); each step ends with the exchange of that
      layout, a sum all-reduce of the un-normalised scores and of nloci, then /(2 nloci) + offset.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

T_START = time.perf_counter()

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--samples", type=int, default=500_000, help="cohort size N (default: config 3)")
    ap.add_argument("--variants", type=int, default=1_000_000, help="score rows M (default: config 3)")
    ap.add_argument("--mode", choices=["auto", "twopass", "fused"], default="auto")
    ap.add_argument("--format", choices=["gt", "ds"], default="gt",
                    help="gt: 2-bit packed GT matrix (the headline, config 3); ds: float32 FORMAT/DS "
                         "matrix (config 5: --samples 200000 --variants 2000000 --chunk-rows 300000)")
    ap.add_argument("--chunk-rows", type=int, default=0,
                    help="score the rows in resident chunks of this many rows, each regenerated on the "
                         "device outside the timed segments (for matrices larger than HBM: config 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements (config 5 DS pass, streaming entry points, "
                         "config 2 end to end)")
    ap.add_argument("--layout", choices=["strip", "rows"], default="strip",
                    help="2-bit cohort layout: strip = NPS_FMT_GT2X (strips of 2048 samples x superblocks of 128 rows, "
                         "scored by the matrix-core single-read kernel; the headline), rows = NPS_FMT_GT2 (row groups, "
                         "scored by the table-lookup single-read kernel)")
    ap.add_argument("--no-optimize", action="store_true",
                    help="skip nps_cohort_optimize (the parity layout of the high-bit planes: the table lookups "
                         "spread over twice as many LDS banks)")
    ap.add_argument("--no-multi-legs", action="store_true",
                    help="N > 1: skip the two strong-scaling legs that follow the headline (one GT score with its rows "
                         "sharded over the GPUs = configs[2] at N GPUs; the FORMAT/DS score of configs[4] likewise)")
    ap.add_argument("--multi-legs-timeout", type=float, default=420.0,
                    help="N > 1: seconds after which the headline line is printed WITHOUT the strong-scaling legs and the "
                         "run ends with status 3 (a hung collective must not cost the measurement)")
    ap.add_argument("--ds-samples", type=int, default=200_000, help="cohort size of the N > 1 FORMAT/DS leg (configs[4])")
    ap.add_argument("--ds-variants", type=int, default=2_000_000, help="score rows of the N > 1 FORMAT/DS leg (configs[4])")
    ap.add_argument("--ds-chunk-rows", type=int, default=300_000,
                    help="resident chunk of the N > 1 FORMAT/DS leg (rows; 300 000 x 200 000 float32 = 240 GB)")
    ap.add_argument("--cpu-rows", type=int, default=2000, help="rows of the CPU-baseline sample")
    ap.add_argument("--full-sweeps", action="store_true",
                    help="every secondary leg at full width (all six genotype distributions, the size x distribution "
                         "sweep, the multi-score options): what tools/profile_round.sh runs; the default keeps the run "
                         "inside the driver's window")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"),
                    help="where the FULL object (headline + every secondary leg) is written; the last stdout line is the "
                         "compact one (< 4 KiB) the driver parses")
    ap.add_argument("--extras-budget", type=float, default=150.0,
                    help="seconds since process start after which no further secondary leg is STARTED (the ones skipped "
                         "are listed); --full-sweeps: no budget")
    ap.add_argument("--extras-deadline", type=float, default=420.0,
                    help="seconds since process start after which the compact line is printed without the unfinished "
                         "secondary legs and the run ends with status 3")
    ap.add_argument("--seed", type=int, default=20250103)
    ap.add_argument("--rank-timeout", type=float, default=1500.0,
                    help="--gpus N without a launcher: seconds after which all ranks are killed (exit status 124)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torch.distributed.run
def spawn_ranks(n, timeout_s):
    """Start the N rank processes (this process never touches a GPU), wait for them with a deadline, pass on the first
    failing rank's exit code (nimpress_amd/launch.py; importing it loads no GPU library)."""
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("nps_launch", os.path.join(ROOT, "nimpress_amd", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    launch.spawn_ranks(__file__, sys.argv[1:], n, timeout_s, name="bench.py")


# ------------------------------------------------------------------------------------------------
def synth_score(m, seed, fmt="gt"):
    """beta ~ N(0,0.02^2) and eaf ~ U(0.01,0.5), 4 decimals.  Config 3 (gt):
    missing rate U(0,0.02), every 1000th row 0.10.  Config 5 (ds): missing rate U(0,0.10), mean 5 %."""
    rng = np.random.default_rng(seed)
    beta = np.round(rng.normal(0.0, 0.02, m), 4)
    eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
    if fmt == "ds":
        miss = rng.uniform(0.0, 0.10, m)
    else:
        miss = rng.uniform(0.0, 0.02, m)
        miss[::1000] = 0.10
    return beta, eaf, miss


def hwe_thresholds(eaf, miss):
    """Integer thresholds of the counter-based generator (same formula as the oracle's)."""
    p_hom = eaf * eaf
    p_het = 2.0 * eaf * (1.0 - eaf)
    scale = 4294967296.0
    t_hom = np.minimum(np.floor(p_hom * scale), 4294967295.0).astype(np.uint32)
    t_het = np.minimum(np.floor((p_hom + p_het) * scale), 4294967295.0).astype(np.uint32)
    t_miss = np.minimum(np.floor(miss * scale), 4294967295.0).astype(np.uint32)
    return t_het, t_hom, t_miss


def host_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or 1, usable


def cpu_baseline(n, eaf, miss, seed, rows):
    """The oracle (CPU restatement of the reference's per-row path) timed on the host on bounded samples
    of the same workload, at the full cohort size:
      value          -- 1 core, dosage math only (decode, tally, impute, accumulate), `rows` rows
      with_binomtest -- 1 core, plus the O(N) binomTest enumeration the reference ALWAYS executes per
                        genotyped row (nimpress.nim:573 -> :155-188); fewer rows (it is ~10x slower)
      all_cores      -- the dosage math split over the samples with OpenMP (not a restatement: the
                        reference is single-threaded)"""
    model, nproc, usable = host_info()
    from oracle import refcpu
    threads = refcpu.host_threads(16)  # the GPU box's CPU share for one GPU
    n_distinct = 16
    th, tm, tmi = refcpu.hwe_thresholds(eaf[:n_distinct], miss[:n_distinct])
    codes = refcpu.synth_rows(n, 0, n_distinct, seed, th, tm, tmi)
    gts = np.stack([refcpu.codes_to_gt(codes[j], n) for j in range(n_distinct)])
    rng = np.random.default_rng(1)
    beta = np.round(rng.normal(0, 0.02, rows), 4)
    e = np.resize(eaf[:n_distinct], rows)
    prm = refcpu.make_params()
    secs, _, _ = refcpu.bench_gt(gts, n, rows, beta, e, prm)
    rows_b = max(8, min(rows, int(300 * 500_000 / max(n, 1))))
    secs_b, _, _, warned = refcpu.bench_gt_full(gts, n, rows_b, beta[:rows_b], e[:rows_b], prm, True, 0.001)
    refcpu.bench_gt_allcores(gts, n, min(rows, 64), beta[:64], e[:64], prm, threads)  # start the thread team
    secs_a, _, _, used = refcpu.bench_gt_allcores(gts, n, rows, beta, e, prm, threads)
    unit = "genotype-dosage accumulations/s"
    return {"value": n * rows / secs, "unit": unit, "cores": 1, "kind": "port",
            "sample": "%d score rows x %d samples (bcf_get_genotypes int32 buffers, %d distinct rows "
                      "cycled), literal decode+tally+impute+accumulate of nimpress.nim:561-583,"
                      "639-641, binomTest warnings off, %.1f s" % (rows, n, n_distinct, secs),
            "with_binomtest": {"value": n * rows_b / secs_b, "unit": unit, "cores": 1,
                               "sample": "%d rows x %d samples incl. binomTest(neffect, 2*ngenotyped, eaf) "
                                         "per row as the reference always runs it (nimpress.nim:573, "
                                         "155-188), %.1f s, %d rows below --afmisp 0.001"
                                         % (rows_b, n, secs_b, warned)},
            "all_cores": {"value": n * rows / secs_a, "unit": unit, "cores": used,
                          "sample": "%d rows x %d samples, the same passes split over the samples with "
                                    "OpenMP, %.2f s" % (rows, n, secs_a)},
            "nproc": nproc, "nproc_usable": usable, "cpu_model": model}


def cover_columns(n_per_thread_samples, n, samples_per_slice):
    """Samples that look at EVERY slice of the persistent grid: per slice the first, a middle and the
    last thread's samples (first / middle / last lane of a wave among them), plus the last, ragged
    group of the cohort (sample n-1)."""
    k = n_per_thread_samples
    units = (n + k - 1) // k            # thread-sized groups of samples (word columns for GT)
    per_slice = max(samples_per_slice // k, 1)
    cols = set()
    n_slices = (units + per_slice - 1) // per_slice
    for p in range(n_slices):
        first = p * per_slice
        last = min(units, first + per_slice) - 1
        for c in (first, first + 31, first + 63, (first + last) // 2, last):
            if first <= c <= last:
                cols.add(c)
    cols.add(units - 1)
    samples = np.concatenate([np.arange(c * k, min(n, (c + 1) * k)) for c in sorted(cols)])
    return samples.astype(np.uint64), n_slices, len(cols)


def score_delta(stats, got, nloci, fmt, beta, eaf, seed, n, m, th, tm, tmi, geometry, row0=0,
                recount=20000, params=None):
    """The second half of BASELINE.json's metric ("+ max-abs score delta vs reference"), outside the timed
    region, rank 0 only.  `stats` / `got` / `nloci` come from one more pass with per-row statistics.
    The oracle (oracle/refcpu.c, the checker -- nothing of it is timed or shipped) then
      * recounts the whole-row tallies of `recount` random rows over all n samples (decode + tallyAlleles)
        and compares them with the device's, bit for bit (GT) / 1e-9 (DS dosage sums);
      * scores samples from EVERY slice of the persistent grid over ALL rows with the restated procs
        (ref_score_subset: decode, the maxmis decision, imputeLocus/SampleDosages, accumulate in row
        order), fed with each row's whole-row tally."""
    from oracle import refcpu
    is_ds = 2 if fmt == "ds16" else (1 if fmt == "ds" else 0)   # (2: the NPS_FMT_DS16 generator of the oracle)
    slices, teams, sps = geometry
    samples, n_slices, n_cols = cover_columns(8 if is_ds else 16, n, sps if sps else (960 * 16))
    rng = np.random.default_rng(seed + 7)
    rows = np.unique(np.concatenate([rng.choice(m, size=min(recount, m), replace=False),
                                     [0, m - 1, min(1000, m - 1)]])).astype(np.uint64)
    ri = rows.astype(np.int64)
    g, ms, ne = refcpu.tally_synth_rows(rows + np.uint64(row0), n, seed, th[ri], tm[ri], tmi[ri], is_ds=is_ds)
    tally_ok = bool(np.array_equal(ms, stats["nmissing"][ri].astype(np.float64)) and
                    np.array_equal(g, stats["ngenotyped"][ri].astype(np.float64)))
    if is_ds:
        tally_ok &= bool(np.allclose(ne, stats["neffect"][ri], rtol=1e-9, atol=0.0))
    else:
        tally_ok &= bool(np.array_equal(ne, stats["neffect"][ri]))
    sums, ref_nloci = refcpu.score_subset(samples, n, row0, seed, th, tm, tmi, beta, eaf, 0,
                                          stats["ngenotyped"].astype(np.float64),
                                          stats["nmissing"].astype(np.float64), stats["neffect"],
                                          params or refcpu.make_params(), is_ds=is_ds)
    ref = sums / (2.0 * ref_nloci)
    idx = samples.astype(np.int64)
    delta = np.abs(got[idx] - ref)
    mean_w = float(np.sum(np.abs(beta))) / (2.0 * max(int(nloci), 1))
    floor = 1e-12 * mean_w
    # the north star's bar, plain: 1e-6 relative, |ref| floored at 1e-12 of the mean absolute weight as SURVEY.md 8(d)
    # prescribes (scores cancel towards 0).  bench.py exits non-zero when this, nloci or a recount fails (parity_failures).
    max_rel = float((delta / np.maximum(np.abs(ref), floor)).max())
    return {"max_abs": float(delta.max()), "max_rel": max_rel,
            "max_abs_over_mean_abs_weight": float(delta.max() / mean_w) if mean_w > 0 else 0.0,
            "within_1e-6_relative": bool(max_rel <= 1e-6),
            "nloci_equal": bool(int(nloci) == int(ref_nloci) == int(stats["used"].sum())),
            "tally_recount_equal": tally_ok, "samples_checked": int(samples.size),
            "slices_covered": "%d of %d" % (n_slices, max(slices, n_slices)), "rows_recounted": int(rows.size),
            "checked": "%d samples (%d thread columns: first / lane 31 / lane 63 / middle / last of each of "
                       "the %d slices of the %dx%d persistent grid, and the last ragged column) x all %d "
                       "rows scored by oracle/refcpu.c (ref_score_subset); whole-row tallies of %d random "
                       "rows recounted over all %d samples" % (samples.size, n_cols, n_slices, slices, teams,
                                                               m, rows.size, n)}


# ------------------------------------------------------------------------------------------------
# N > 1: the other two north-star curves, measured by ALL ranks right after the headline (none of them is `value`)
def strong_scaling_leg(torch, dist, multi, capi, args, rank, world, device, fmt, n, m, seed, chunk_rows, steps,
                       cohort=None):
    """ONE score whose rows are sharded over the GPUs (second layout): rank r holds all samples of its
    block of rows, so tallies stay local and exact; the one exchange is the RCCL sum all-reduce of the partial sums and
    of nloci, then nimpress.nim:643-649 in place.  fmt = NPS_FMT_GT2X (configs[2] at N GPUs; `cohort` = the
    headline's resident matrix, of which the rank scores its 128-aligned block where it lies) or NPS_FMT_DS32
    (configs[4]: the block is generated chunk by chunk on the device, outside the timed segments).  Time = the slowest
    rank's timed segments (scoring + exchange + normalisation) per pass; value = n m / that."""
    is_ds = fmt == capi.FMT_DS32
    beta, eaf, miss = synth_score(m, seed, "ds" if is_ds else "gt")
    th, tm, tmi = hwe_thresholds(eaf, miss)
    r0, r1 = multi.shard_rows(m, world, rank, align=128)
    own = cohort is None
    chunk = max(1, min(chunk_rows if chunk_rows > 0 else (r1 - r0), max(r1 - r0, 1)))
    chunks = [(a, min(r1, a + chunk)) for a in range(r0, r1, chunk)]
    if own:
        cohort = capi.Cohort(n, chunk, fmt=fmt, device=device)
    sdefs = [capi.ScoreDef(capi.row_descs(beta[a:b], eaf[a:b]), device=device) for a, b in chunks]
    sc = capi.Scorer(n, capi.make_params(imp_locus="ps") if is_ds else capi.make_params(), device=device)
    d = torch.empty(n, dtype=torch.float64, device="cuda")
    regenerate = own and len(chunks) > 1   # (a block that fits is generated once, before the timed passes)

    def fill(a, b):
        for x in range(a, b, 1 << 15):
            y = min(b, x + (1 << 15))
            cohort.synth_at(x - a, x, seed, th[x:y], tm[x:y], tmi[x:y])

    if own and chunks and not regenerate:
        fill(*chunks[0])

    def one_pass():
        sc.reset()
        seg = 0.0
        for (a, b), sdef in zip(chunks, sdefs):
            if regenerate:
                fill(a, b)
            sc.sync()
            t0 = time.perf_counter()
            sc.score_cohort_def(cohort, sdef, 0 if own else a, capi.MODE_AUTO)
            sc.sync()
            seg += time.perf_counter() - t0
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        nl = sc.partial_device(d.data_ptr())
        _, nl = multi.all_reduce_partial(d, nl)
        sc.normalize_device(d.data_ptr(), nl, 0.0)
        torch.cuda.synchronize()
        return seg, time.perf_counter() - t0, nl

    if not regenerate:
        one_pass()   # warm-up (a block that has to be regenerated chunk by chunk is scored once: 1.6 TB of generation per pass)
    seg = exch = 0.0
    for _ in range(steps):
        a_, b_, nloci = one_pass()
        seg += a_
        exch += b_
    t = torch.tensor([seg, exch], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    seg, exch = (float(x) for x in t.tolist())
    sc.close()
    for sd in sdefs:
        sd.close()
    if own:
        cohort.close()
    torch.cuda.empty_cache()
    per = (seg + exch) / max(steps, 1)
    return {"workload": "one %s score of %d rows on %d samples, rows sharded x%d (blocks of %d rows, %d resident chunk(s) "
                        "per GPU) + RCCL all-reduce of sums and nloci" % ("FORMAT/DS" if is_ds else "2-bit GT", m, n, world,
                                                                          r1 - r0, len(chunks)),
            "scaling": "strong", "n_gpus": world, "value": float(n) * float(m) / per if per > 0 else 0.0,
            "unit": "genotype-dosage accumulations/s", "ms_per_pass": per * 1e3, "scoring_ms": seg / max(steps, 1) * 1e3,
            "exchange_ms": exch / max(steps, 1) * 1e3, "nloci": int(nloci), "passes": steps}


# ------------------------------------------------------------------------------------------------
# secondary measurements (rank 0, N = 1, after the headline; none of them is `value`)
def size_sweep(capi, device, args, seed, steps=3, full=True):
    """The single-score pass by COHORT SIZE x GENOTYPE DISTRIBUTION (the strip kernel fitted one shape;
    round 4: NPS_FMT_GT_AUTO's worst case over size AND distribution was unreported).  The reference scores any N
    (nimpress.nim:626-628).  Sizes: 100 000 samples (BASELINE configs[1]'s cohort), 250 000, 300 000 (147 strips: the
    awkward middle), 400 000, 500 000, 1 000 000 (more strips than compute units); distributions: the bench cohort,
    eaf 0.5 in every row (the table-lookup kernel's worst case), uniformly random codes (--maxmis=1).
    What is measured is what a caller of nps_cohort_create(NPS_FMT_GT_AUTO) + NPS_MODE_AUTO gets: the strip layout at
    every size (round 5).  `first_run_ms`: the first scoring run on the fresh cohort -- where the resident grid does not
    cover the chip it also counts the cohort's tallies, once (reads 2); `ms_per_pass`: best of the later runs (reads 1
    everywhere).  auto_layout_worst_frac is over the LATER runs (a resident cohort exists to be scored many times);
    auto_first_run_worst_frac is reported beside it.  --full-sweeps adds the row-layout kernel on the bench distribution.
    HIP events on the library's stream; every case with the oracle's subset check.
    Default run (not --full-sweeps): 100 000 and 1 000 000 samples on the bench distribution and 300 000 on eaf 0.5."""
    import torch
    SC = 4294967296.0
    f = lambda x: np.minimum(np.floor(np.asarray(x, dtype=np.float64) * SC), 4294967295.0).astype(np.uint32)
    sizes = ((100_000, 1_000_000), (250_000, 1_000_000), (300_000, 1_000_000), (400_000, 1_000_000),
             (500_000, 1_000_000), (1_000_000, 500_000))
    dists = ("bench", "eaf 0.5", "uniform codes")
    if full:
        plan = [(n, m, d) for n, m in sizes for d in dists]
    else:
        plan = [(100_000, 1_000_000, "bench"), (300_000, 1_000_000, "eaf 0.5"), (1_000_000, 500_000, "bench")]
        steps = 2
    labels = {capi.FMT_GT2X: "strip_layout_matrix_cores", capi.FMT_GT2: "row_layout_table_lookups"}
    out = []
    for n, m, dist in plan:
        beta, eaf, miss = synth_score(m, seed, "gt")
        kw = {}
        if dist == "bench":
            th, tm, tmi = hwe_thresholds(eaf, miss)
        elif dist == "eaf 0.5":
            th, tm, tmi = hwe_thresholds(0.5 * np.ones(m), miss)
        else:
            one = np.ones(m)
            th, tm, tmi = f(2 * one / 3), f(one / 3), f(0.25 * one)
            kw = dict(maxmis=1.0)
        alg = m * ((n + 15) // 16) * 4 + 40 * m + 8 * n
        d = torch.empty(n, dtype=torch.float64, device="cuda")
        sdef = capi.ScoreDef(capi.row_descs(beta, eaf), device=device)
        row = {"samples": n, "rows": m, "cohort": dist}
        fmts = (capi.FMT_GT_AUTO, capi.FMT_GT2) if (full and dist == "bench") else (capi.FMT_GT_AUTO,)
        for fmt in fmts:
            co = capi.Cohort(n, m, fmt=fmt, device=device)
            for x in range(0, m, 1 << 15):
                y = min(m, x + (1 << 15))
                co.synth_at(x, x, seed, th[x:y], tm[x:y], tmi[x:y])
            if co.fmt == capi.FMT_GT2:
                co.optimize()
            prm = capi.make_params(**kw)
            sc = capi.Scorer(n, prm, device=device)
            geo = sc.fused_geometry(m, co.fmt)
            best, first, reads, first_reads = None, None, 1, 1
            for i in range(steps + 1):
                sc.reset()
                sc.profile_enable(True)
                sc.profile_get(reset=True)
                t0 = time.perf_counter()
                sc.score_cohort_def(co, sdef, 0, capi.MODE_AUTO)
                sc.finish_device(0.0, d.data_ptr())
                wall = (time.perf_counter() - t0) * 1e3
                p = sc.profile_get(reset=True)
                ms = p.ms_fused + p.ms_tally + p.ms_params + p.ms_accumulate
                if i == 0:   # (wall time: the one-time count of the cohort's tallies is not one of the context's launches)
                    # (two reads: a tally launch of the context's own, or -- no in-pass launch at all and the cohort now
                    #  carries tallies -- the cohort's one-time count ran before a given-tallies pass; ONE read where the
                    #  in-pass kernel ran and kept what it counted, round 6)
                    first, first_reads = wall, (2 if (p.n_tally or (co.has_tallies() and co.fmt == capi.FMT_GT2X and p.n_fused == 0)) else 1)
                else:
                    best = ms if best is None else min(best, ms)
                    reads = 2 if p.n_tally else 1
            r = {"ms_per_pass": best, "frac_of_8TBps": alg / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, "reads_of_the_matrix": reads,
                 "first_run_ms": first, "first_run_reads_of_the_matrix": first_reads,
                 "first_run_frac_of_8TBps": alg / (first * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "tallies": (("kept with the cohort by the first pass, which counts them in its one read" if first_reads == 1
                              else "counted once in a pass of their own, kept with the cohort")
                             if (co.fmt == capi.FMT_GT2X and co.has_tallies()) else "counted in the pass"),
                 "grid": {"slices": geo[0], "teams": geo[1], "samples_per_slice": geo[2]}}
            if not args.no_cpu_baseline:
                from oracle import refcpu
                sc.reset()
                sc.score_cohort_def(co, sdef, 0, capi.MODE_AUTO)
                stats = sc.flush()
                got, nl = sc.finish(0.0)
                sd = score_delta(stats, got, nl, "gt", beta, eaf, seed, n, m, th, tm, tmi, geo, recount=1000,
                                 params=refcpu.make_params(**kw))
                sd.pop("checked", None)
                r["score_delta_vs_reference"] = sd
            sc.close()
            row[labels[co.fmt] if fmt != capi.FMT_GT_AUTO else "NPS_FMT_GT_AUTO"] = r
            co.close()
            torch.cuda.empty_cache()
        sdef.close()
        row["auto_layout_frac_of_8TBps"] = row["NPS_FMT_GT_AUTO"]["frac_of_8TBps"]
        out.append(row)
    worst = min(out, key=lambda r: r["auto_layout_frac_of_8TBps"])
    return {"cases": out, "auto_layout_worst_frac": worst["auto_layout_frac_of_8TBps"],
            "auto_layout_worst_case": "%s x %d samples" % (worst["cohort"], worst["samples"]),
            "auto_first_run_worst_frac": min(r["NPS_FMT_GT_AUTO"]["first_run_frac_of_8TBps"] for r in out)}


def given_tallies(capi, sc, cohort, sdef, d_scores, n, m, headline_ms, steps=5):
    """Experiment B of the round-4 verdict, a SECONDARY (the headline counts its tallies in the pass): the same resident
    cohort after nps_cohort_keep_tallies -- tallyAlleles of every row counted once, kept with the cohort -- scored under
    NPS_MODE_AUTO with the tallies given: one read, no popcounts, no hand-over between the strips.  Legitimate for many
    score files over one cohort (configs[3]: tally once, score 8 times); the one-time cost is reported beside it.  The
    scores must equal the headline pass's (same kernel arithmetic, same tallies; 1e-9 relative because the two grids
    group the exact digit sums differently), the row statistics bit for bit."""
    import torch
    sc.reset()
    sc.score_cohort_def(cohort, sdef, 0, capi.MODE_AUTO)
    sc.finish_device(0.0, d_scores.data_ptr())
    want = d_scores.clone()
    st_want = sc.flush()
    # round 6: with nps_cohort_expect_passes(>= 2) the FIRST pass keeps the tallies it counts anyway (no extra read)
    cohort.expect_passes(2)
    sc.reset()
    sc.profile_enable(True)
    sc.profile_get(reset=True)
    sc.score_cohort_def(cohort, sdef, 0, capi.MODE_AUTO)
    sc.finish_device(0.0, d_scores.data_ptr())
    p = sc.profile_get(reset=True)
    harvest_ms = p.ms_fused + p.ms_tally + p.ms_params + p.ms_accumulate
    harvested = bool(cohort.has_tallies() and p.n_fused >= 1 and p.n_tally == 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cohort.keep_tallies()
    keep_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    cohort.keep_tallies()           # (the second call has its buffer: the cost of the count alone)
    keep_s = min(keep_s, time.perf_counter() - t0)
    best, nl = None, 0
    for i in range(steps + 1):
        sc.reset()
        sc.profile_enable(True)
        sc.profile_get(reset=True)
        sc.score_cohort_def(cohort, sdef, 0, capi.MODE_AUTO)
        nl = sc.finish_device(0.0, d_scores.data_ptr())
        p = sc.profile_get(reset=True)
        ms = p.ms_fused + p.ms_tally + p.ms_params + p.ms_accumulate
        if i:
            best = ms if best is None else min(best, ms)
    sc.profile_enable(False)
    st = sc.flush()
    # (teams differ between the two grids, so the float64 sums of the exact digit sums group differently: 1e-9
    # relative, |ref| floored at 1e-12 of the largest score, NaN positions equal)
    a, b = d_scores.cpu().numpy(), want.cpu().numpy()
    ok_nan = bool(np.array_equal(np.isnan(a), np.isnan(b)))
    fin = ~np.isnan(b)
    floor = 1e-12 * float(np.abs(b[fin]).max()) if fin.any() else 0.0
    worst = float((np.abs(a[fin] - b[fin]) / np.maximum(np.abs(b[fin]), max(floor, 1e-300))).max()) if fin.any() else 0.0
    same = ok_nan and worst <= 1e-9
    stats_same = bool(all(np.array_equal(st[k], st_want[k]) for k in ("ngenotyped", "nmissing", "neffect", "used", "reason")))
    alg = m * ((n + 15) // 16) * 4 + 40 * m + 8 * n
    return {"what": "the headline cohort carrying its whole-row tallies (nps_cohort_keep_tallies: counted once, one read of "
                    "the matrix), scored with the tallies given; NOT the headline, which counts them in the pass",
            "ms_per_pass": best, "frac_of_8TBps": alg / (best * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "one_time_tally_ms": keep_s * 1e3, "headline_ms_per_step": headline_ms, "nloci": int(nl),
            "first_pass_keeps_its_tallies": {"with": "nps_cohort_expect_passes(2)", "kept_by_the_in_pass_kernel": harvested,
                                             "that_pass_ms": harvest_ms, "extra_reads_of_the_matrix": 0 if harvested else 1},
            "outputs_equal_headline_within_1e-9_relative": same, "outputs_max_relative_difference": worst,
            "outputs_equal_row_statistics": stats_same,
            "passes_after_which_it_pays": (keep_s * 1e3) / max(headline_ms - best, 1e-9) if best < headline_ms else None}


def ds_config5(capi, device, args, n=200_000, m=2_000_000, chunk=300_000, seed=20250105, half=False):
    """BASELINE.json configs[4] at its stated size on ONE GPU: 2 000 000 x 200 000 float32 dosages are
    1.6 TB, so the rows are scored in resident chunks; each chunk is regenerated on the device (outside the
    timed segments), the partial scores and nloci carry across chunks inside the context, the timed
    segments (HIP events around the hot kernels + wall around each scoring call) are summed."""
    beta, eaf, miss = synth_score(m, seed, "ds")
    th, tm, tmi = hwe_thresholds(eaf, miss)
    chunk = min(chunk, m)
    fmt = capi.FMT_DS16 if half else capi.FMT_DS32
    co = capi.Cohort(n, chunk, fmt=fmt, device=device)
    sc = capi.Scorer(n, capi.make_params(imp_locus="ps"), device=device)
    geo = sc.fused_geometry(chunk, fmt)
    sc.profile_enable(True)
    sc.profile_get(reset=True)
    wall = 0.0
    stats = []
    sc.reset()
    for r0 in range(0, m, chunk):
        r1 = min(m, r0 + chunk)
        for a in range(r0, r1, 1 << 15):      # (the DS generator takes at most 65 535 rows per call)
            b = min(r1, a + (1 << 15))
            co.synth_at(a - r0, a, seed, th[a:b], tm[a:b], tmi[a:b])
        rows = capi.row_descs(beta[r0:r1], eaf[r0:r1])
        sdef = capi.ScoreDef(rows, device=device)
        sc.sync()
        t0 = time.perf_counter()
        sc.score_cohort_def(co, sdef, 0, capi.MODE_AUTO)
        sc.sync()
        wall += time.perf_counter() - t0
        stats.append(sc.flush())
        sdef.close()
    got, nloci = sc.finish(0.0)
    prof = sc.profile_get(reset=True)
    sc.close()
    co.close()
    hot_ms = prof.ms_tally + prof.ms_params + prof.ms_accumulate + prof.ms_fused
    alg = (2 if half else 4) * m * n + 40 * m + 8 * n
    out = {"workload": ("synthetic %d-variant PRS on %d-sample FORMAT/DS matrix held as NPS_FMT_DS16 (2 bytes per dosage: "
                        "three-decimal values, lossless; a NEW configuration beside configs[4], which stays float32) in %d "
                        "resident chunk(s) of %d rows, 5 %% mean missingness, --imp-locus=ps" if half else
                        "synthetic %d-variant PRS on %d-sample float32 FORMAT/DS matrix (BASELINE.json "
                        "configs[4], 1.6 TB) in %d resident chunks of %d rows, 5 %% mean missingness, "
                        "--imp-locus=ps") % (m, n, (m + chunk - 1) // chunk, chunk),
           "value": n * m / wall, "unit": "genotype-dosage accumulations/s", "wall_s_scoring": wall,
           "kernel_ms": hot_ms, "nloci": int(nloci),
           "roofline": {"bound": "hbm", "achieved": alg / (hot_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": alg / (hot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes": alg, "kernel": "ds_fused" if prof.n_fused else "ds two-pass"}}
    if not args.no_cpu_baseline:
        st = np.concatenate(stats)
        # the last chunk's rows stand for the matrix in the full-width recount (its generator rows are
        # the global row numbers, so any row can be recounted); all rows are scored for the subset
        from oracle import refcpu
        out["score_delta_vs_reference"] = score_delta(st, got, nloci, "ds16" if half else "ds", beta, eaf, seed, n, m, th, tm, tmi,
                                                      geo, recount=300, params=refcpu.make_params("ps"))
    return out


def multi_score(capi, device, args, n, m, seed, S=8, steps=3):
    """SURVEY.md section 8 f2 measured: S = 8 score definitions over the same 1M rows in ONE pass over the
    cohort (nps_score_cohort_multi: int8 MFMA over the NPS_FMT_GT2M layout, row tallies from the packer),
    against S passes of the single-score kernel.  One step = reset -> weights as base-256 digits ->
    the product -> fold -> /(2 nloci) + offset for all S scores, result in a device buffer.  Measured with
    full-width weights (the headline of this object), with nps_multi_set_missing_weight_bits(32) and with
    41-bit weights (nps_multidef_create_bits: six digits per weight instead of seven)."""
    import torch
    _, eaf, miss = synth_score(m, seed)
    th, tm, tmi = hwe_thresholds(eaf, miss)
    # the one-time repack of a row-major 2-bit cohort into this layout, whole-row tallies included (the tallies
    # of the multi-score pass come from here): timed on the first 65 536 rows
    mp = min(m, 1 << 16)
    src = capi.Cohort(n, mp, device=device)
    src.synth_at(0, 0, seed, th[:mp], tm[:mp], tmi[:mp])
    dst = capi.Cohort(n, mp, fmt=capi.FMT_GT2M, device=device)
    dst.convert_from(src)
    t0 = time.perf_counter()
    dst.convert_from(src)           # (synchronises at its end)
    pack_s = time.perf_counter() - t0
    dst.close()
    dstx = capi.Cohort(n, mp, fmt=capi.FMT_GT2X, device=device)   # ... and into the strip layout of the single-score kernel
    dstx.convert_from(src)
    t0 = time.perf_counter()
    dstx.convert_from(src)
    packx_s = time.perf_counter() - t0
    dstx.close()
    src.close()
    co = capi.Cohort(n, m, fmt=capi.FMT_GT2M, device=device)
    for a in range(0, m, 1 << 15):
        b = min(m, a + (1 << 15))
        co.synth_at(a, a, seed, th[a:b], tm[a:b], tmi[a:b])
    descs = np.zeros((S, m), dtype=capi.ROW_DESC_DTYPE)
    for s in range(S):
        descs[s]["beta"] = np.round(np.random.default_rng(seed + 1000 + s).normal(0.0, 0.02, m), 4)
        descs[s]["eaf"] = eaf
    mdefs = {49: capi.MultiDef(descs, device=device), 41: capi.MultiDef(descs, device=device, weight_bits=41)}
    msc = capi.MultiScorer(n, capi.make_params(), S, device=device)
    d_scores = torch.empty((S, n), dtype=torch.float64, device="cuda")
    off = np.zeros(S)
    checker = None
    if not args.no_cpu_baseline:
        from oracle import refcpu
        nm, ne = co.row_tallies()
        rng = np.random.default_rng(seed + 11)
        rows = np.unique(np.concatenate([rng.choice(m, 300, replace=False), [0, m - 1]])).astype(np.uint64)
        ri = rows.astype(np.int64)
        g, ms, neff = refcpu.tally_synth_rows(rows, n, seed, th[ri], tm[ri], tmi[ri])
        tally_ok = bool(np.array_equal(ms, nm[ri].astype(np.float64)) and np.array_equal(neff, ne[ri].astype(np.float64)))
        samples = np.unique(np.concatenate([rng.choice(n, 600, replace=False), np.arange(0, 64), np.arange(n - 64, n)])).astype(np.uint64)
        refs = {}
        for s in (0, S - 1):
            sums, ref_nloci = refcpu.score_subset(samples, n, 0, seed, th, tm, tmi, descs[s]["beta"], eaf, 0,
                                                  float(n) - nm.astype(np.float64), nm.astype(np.float64),
                                                  ne.astype(np.float64), refcpu.make_params())
            refs[s] = (sums / (2.0 * ref_nloci), int(ref_nloci))

        def checker(nloci):
            got = d_scores.cpu().numpy()
            worst_abs, worst_rel, ok = 0.0, 0.0, tally_ok
            for s, (ref, ref_nloci) in refs.items():
                d = np.abs(got[s][samples.astype(np.int64)] - ref)
                floor = 1e-12 * float(np.sum(np.abs(descs[s]["beta"]))) / (2.0 * ref_nloci)
                worst_abs = max(worst_abs, float(d.max()))
                worst_rel = max(worst_rel, float((d / np.maximum(np.abs(ref), floor)).max()))
                ok &= ref_nloci == int(nloci[s])
            return {"max_abs": worst_abs, "max_rel": worst_rel, "within_1e-6_relative": bool(worst_rel <= 1e-6),
                    "tallies_and_nloci_equal": bool(ok),
                    "checked": "scores 1 and %d: %d samples x all %d rows by oracle/refcpu.c (ref_score_subset); "
                               "%d whole-row tallies recounted" % (S, samples.size, m, rows.size)}

    def measure(bits, wbits=49):
        def step():
            msc.reset()
            msc.score_cohort(co, mdefs[wbits])
            return msc.finish_device(off, d_scores.data_ptr())

        msc.set_missing_weight_bits(bits)
        nloci = step()
        torch.cuda.synchronize()
        msc.reset()
        t0 = time.perf_counter()
        prod_ms = []
        for _ in range(steps):
            nloci = step()
            prod_ms.append(msc.timing())
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps
        ms_params, ms_prod, ms_fold = (float(np.mean([p[k] for p in prod_ms])) for k in range(3))
        nd = 7 if wbits == 49 else 6
        # tiles of 16 columns (nps_multi.hip multi_plan): the dosage matrix's digits; the is-missing matrix's, or its
        # four leading digits only
        td = (nd * S + 15) // 16
        int8_ops = 2.0 * n * m * 16 * (td + (min(td, ((4 if bits == 32 else 5) * S + 15) // 16) if bits in (32, 40) else td))
        r = {"ms_per_pass": wall * 1e3, "ms_per_score": wall * 1e3 / S, "value": S * float(n) * m / wall,
             "kernel_ms": {"weights_to_digits": ms_params, "product": ms_prod, "fold": ms_fold},
             "int8_TOPs": int8_ops / (ms_prod * 1e-3) / 1e12, "hbm_GBps": alg / (ms_prod * 1e-3) / 1e9}
        if checker:
            r["score_delta_vs_reference"] = checker(nloci)
        return r, nloci

    alg = m * ((n + 15) // 16) * 4 + 40 * m * S + 8 * n * S
    full, nloci = measure(56)
    fast, _ = measure(32)
    mid, _ = measure(40)
    w41, _ = measure(56, 41)
    # ONE definition through the same pass (one tile of 16 columns): what a single score costs on a cohort that
    # carries its whole-row tallies (the headline kernel counts them while it reads -- not the same work)
    msc1 = capi.MultiScorer(n, capi.make_params(), 1, device=device)
    mdef1 = capi.MultiDef(descs[:1], device=device)
    d1 = torch.empty((1, n), dtype=torch.float64, device="cuda")
    one_ms = []
    for i in range(steps + 1):
        msc1.reset()
        msc1.score_cohort(co, mdef1)
        msc1.finish_device(np.zeros(1), d1.data_ptr())
        torch.cuda.synchronize()
        if i:
            one_ms.append(sum(msc1.timing()))
    one = {"what": "ONE score definition through the multi-score pass on the same cohort (tallies from the packer, "
                   "one tile of 16 digit columns): kernel time per pass",
           "ms_per_pass": float(np.mean(one_ms)),
           "hbm_GBps": (m * ((n + 15) // 16) * 4) / (float(np.mean(one_ms)) * 1e-3) / 1e9}
    one["hbm_frac"] = one["hbm_GBps"] / HBM_PEAK_GBS
    msc1.close()
    mdef1.close()
    del d1
    # the same cohort without its missing genotypes (imputed hard calls have none): superblocks whose rows have no
    # missing sample skip the is-missing matrix (timing only; the parity of that path is tests/test_gpu_multi.py's)
    checker, zeros = None, np.zeros_like(tmi)
    for a in range(0, m, 1 << 15):
        b = min(m, a + (1 << 15))
        co.synth_at(a, a, seed, th[a:b], tm[a:b], zeros[a:b])
    nomiss, _ = measure(56)
    nomiss41, _ = measure(56, 41)
    out = {"workload": "%d score definitions x %d rows x %d samples in one pass (NPS_FMT_GT2M cohort, int8 MFMA, "
                       "7 base-256 digits per weight), CLI-default imputation flags" % (S, m, n),
           "scores": S, "value": full["value"], "unit": "genotype-dosage accumulations/s (x scores)",
           "ms_per_pass": full["ms_per_pass"], "ms_per_score": full["ms_per_score"], "kernel_ms": full["kernel_ms"],
           "hbm_bytes_per_score": alg / S, "nloci": [int(x) for x in nloci],
           "one_time_pack": {"what": "nps_cohort_convert: row-major 2-bit cohort -> NPS_FMT_GT2M units + whole-row "
                                     "tallies (tallyAlleles of every row; the multi-score pass reads them instead of "
                                     "recounting), measured on %d rows" % mp,
                             "ms_per_million_rows": pack_s * 1e3 * 1e6 / mp,
                             "GBps_read_plus_written": 2.0 * mp * ((n + 15) // 16) * 4 / pack_s / 1e9,
                             "into_strip_layout": {"what": "nps_cohort_convert: the same cohort -> NPS_FMT_GT2X strips (no "
                                                           "tallies: the single-score kernel counts while it reads)",
                                                   "ms_per_million_rows": packx_s * 1e3 * 1e6 / mp,
                                                   "GBps_read_plus_written": 2.0 * mp * ((n + 15) // 16) * 4 / packx_s / 1e9}},
           "roofline": {"bound": "mfma", "achieved": full["int8_TOPs"], "peak": 5000.0, "unit": "TOP/s (int8 dense)",
                        "frac": full["int8_TOPs"] / 5000.0, "hbm_GBps": full["hbm_GBps"],
                        "hbm_frac": full["hbm_GBps"] / HBM_PEAK_GBS,
                        "note": "power-limited: the pass takes the same cycles at whatever clock the board's "
                                "power cap allows under int8 MFMA load (DESIGN.md 4.3)"},
           "missing_weight_bits_32": {k: fast[k] for k in ("ms_per_pass", "ms_per_score", "value", "kernel_ms",
                                                            "int8_TOPs")},
           "missing_weight_bits_40": dict({k: mid[k] for k in ("ms_per_pass", "ms_per_score", "value", "kernel_ms", "int8_TOPs")},
                                          what="nps_multi_set_missing_weight_bits(40): five leading digits of the is-missing "
                                               "weights (VERDICT round 4 item 8); an option: 2^-32 x B per missing genotype"),
           "weight_bits_41": dict({k: w41[k] for k in ("ms_per_pass", "ms_per_score", "value", "kernel_ms", "int8_TOPs")},
                                  what="nps_multidef_create_bits(.., 41): six base-256 digits per weight, a quarter "
                                       "fewer matrix instructions for 8 scores; an option, not the default: a sample "
                                       "whose terms cancel is not within 1e-6 of its own score (include/nps.h)",
                                  cohort_without_missing_genotypes_ms_per_pass=nomiss41["ms_per_pass"]),
           "cohort_without_missing_genotypes": {k: nomiss[k] for k in ("ms_per_pass", "ms_per_score", "value",
                                                                       "kernel_ms")},
           "one_score_given_tallies": one}
    if "score_delta_vs_reference" in full:
        out["score_delta_vs_reference"] = full["score_delta_vs_reference"]
        out["missing_weight_bits_32"]["score_delta_vs_reference"] = fast["score_delta_vs_reference"]
        out["missing_weight_bits_40"]["score_delta_vs_reference"] = mid["score_delta_vs_reference"]
        out["weight_bits_41"]["score_delta_vs_reference"] = w41["score_delta_vs_reference"]
    msc.close()
    for d in mdefs.values():
        d.close()
    co.close()
    return out


def layout_sweep(capi, device, n, m, seed, steps=3, full=True):
    """The single-score pass by genotype distribution (the table-lookup kernel's time depends on
    how a wave's 64 table indices spread over the LDS banks; the matrix-core kernel's must not).  Six synthetic
    cohorts of the bench shape, each regenerated on the device; best of `steps` passes, HIP events on the
    library's stream.  Both kernels for the bench distribution and for the table kernel's worst case."""
    import torch
    rng = np.random.default_rng(seed)
    beta = np.round(rng.normal(0.0, 0.02, m), 4)
    SC = 4294967296.0
    f = lambda x: np.minimum(np.floor(np.asarray(x, dtype=np.float64) * SC), 4294967295.0).astype(np.uint32)
    one = np.ones(m)

    def hwe(eaf, miss):
        return f(eaf * eaf + 2 * eaf * (1 - eaf)), f(eaf * eaf), f(miss)

    bench_miss = rng.uniform(0, 0.02, m)
    bench_miss[::1000] = 0.10
    cases = [
        ("bench (eaf U(0.01,0.5), 1 % missing)", hwe(np.round(rng.uniform(0.01, 0.5, m), 4), bench_miss), {}, True),
        ("low MAF (eaf U(0.001,0.05))", hwe(rng.uniform(0.001, 0.05, m), rng.uniform(0, 0.02, m)), {}, False),
        ("eaf 0.5 everywhere", hwe(0.5 * one, rng.uniform(0, 0.02, m)), {}, True),
        ("all heterozygous (non-HWE)", (f(one), f(0 * one), f(0 * one)), {}, False),
        ("20 % missing, --maxmis=1", hwe(np.round(rng.uniform(0.01, 0.5, m), 4), 0.2 * one), dict(maxmis=1.0), False),
        ("uniform codes, --maxmis=1", (f(2 * one / 3), f(one / 3), f(0.25 * one)), dict(maxmis=1.0), False),
    ]
    if not full:    # the default run: the bench distribution and the table kernel's worst case (the driver's window)
        cases = [c for c in cases if c[3]]
        steps = 2
    alg = m * ((n + 15) // 16) * 4 + 40 * m + 8 * n
    d = torch.empty(n, dtype=torch.float64, device="cuda")
    sdef = capi.ScoreDef(capi.row_descs(beta, 0.3 * one), device=device)
    out = []
    for name, (th, tm, tmi), kw, both in cases:
        row = {"cohort": name}
        for label, fmt in (("strip_layout_matrix_cores", capi.FMT_GT2X), ("row_layout_table_lookups", capi.FMT_GT2)):
            if fmt == capi.FMT_GT2 and (not both or (not full and name.startswith("bench"))):
                continue
            co = capi.Cohort(n, m, fmt=fmt, device=device)
            for x in range(0, m, 1 << 15):
                y = min(m, x + (1 << 15))
                co.synth_at(x, x, seed, th[x:y], tm[x:y], tmi[x:y])
            if fmt == capi.FMT_GT2:
                co.optimize()
            sc = capi.Scorer(n, capi.make_params(**kw), device=device)
            best = None
            for i in range(steps + 1):
                sc.reset()
                sc.profile_enable(True)
                sc.profile_get(reset=True)
                sc.score_cohort_def(co, sdef, 0, capi.MODE_FUSED)
                sc.finish_device(0.0, d.data_ptr())
                ms = sc.profile_get(reset=True).ms_fused
                if i:
                    best = ms if best is None else min(best, ms)
            sc.close()
            co.close()
            torch.cuda.empty_cache()
            row[label] = {"ms_per_pass": best, "frac_of_8TBps": alg / (best * 1e-3) / 1e9 / HBM_PEAK_GBS}
        out.append(row)
    sdef.close()
    fr = [r["strip_layout_matrix_cores"]["frac_of_8TBps"] for r in out]
    return {"shape": "%d samples x %d rows" % (n, m), "cases": out, "strip_layout_worst_frac": min(fr),
            "strip_layout_best_frac": max(fr)}


def streaming_rates(capi, device, n, seed):
    """The drop-in entry points (what the command line uses): one call per score row with a HOST buffer, so
    these rates include the pinned copy + PCIe transfer + decode launch per row.  Never `value`."""
    rng = np.random.default_rng(seed)
    nd = 8
    eaf = np.round(rng.uniform(0.05, 0.5, nd), 4)
    # input rows made here with numpy (the oracle is the checker, never an input generator of a measurement):
    # NPS_CODE_* per sample -> packed words (sample i in bits 2i, 2i+1) and bcf_get_genotypes pairs
    u = rng.uniform(size=(nd, n))
    code = np.where(u < 0.01, 2, np.where(u < 0.01 + eaf[:, None] ** 2, 3,
                    np.where(u < 0.01 + eaf[:, None] ** 2 + 2 * eaf[:, None] * (1 - eaf[:, None]), 1, 0))).astype(np.uint32)
    pad = (-n) % 16
    cp = np.pad(code, ((0, 0), (0, pad))).reshape(nd, -1, 16)
    codes = np.zeros(cp.shape[:2], dtype=np.uint32)
    for k in range(16):
        codes |= cp[:, :, k] << np.uint32(2 * k)
    a0 = np.array([2, 2, 0, 4], dtype=np.int32)[code]      # 0/0, 0/1, ./., 1/1 in the (allele+1)<<1 encoding
    a1 = np.array([2, 4, 0, 4], dtype=np.int32)[code]
    gt32 = [np.ascontiguousarray(np.stack([a0[j], a1[j]], axis=1).reshape(-1)) for j in range(nd)]
    gt8 = [g.astype(np.int8) for g in gt32]
    out = {}
    for name, rows, nbytes, push in (
            ("nps_push_gt int32", 200, 8 * n, lambda sc, j: sc.push_gt(gt32[j % nd], 2, 1, False, 0.01, eaf[j % nd])),
            ("nps_push_gt_raw int8", 400, 2 * n, lambda sc, j: sc.push_gt_raw(gt8[j % nd], 2, 1, False, 0.01, eaf[j % nd])),
            ("nps_push_packed", 2000, 4 * ((n + 15) // 16), lambda sc, j: sc.push_packed(codes[j % nd], False, 0.01, eaf[j % nd]))):
        sc = capi.Scorer(n, capi.make_params(), device=device)
        for j in range(8):
            push(sc, j)
        sc.sync()
        sc.reset()
        t0 = time.perf_counter()
        for j in range(rows):
            push(sc, j)
        sc.sync()
        dt = time.perf_counter() - t0
        sc.close()
        out[name] = {"genotypes_per_s": n * rows / dt, "host_to_device_GBps": nbytes * rows / dt / 1e9,
                     "rows": rows, "samples": n, "seconds": dt}
    return out


def config2_e2e(tmpdir):
    """BASELINE.json configs[1] end to end through the `nimpress` command line: wood-height (697 loci) on a
    synthetic 100 000-sample BCF2 + CSI (written by tests/config2.py), process start to last output line."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import config2
    cli = os.path.join(ROOT, "nimpress_amd", "nimpress")
    wood = os.path.join(ROOT, "tests", "golden", "scores", "wood-25282103-height.scores")
    path, n_rec, n = config2.write_cohort(tmpdir, wood, level=1)[:3]
    best, breakdown = None, None
    for _ in range(2):
        t0 = time.perf_counter()
        r = subprocess.run([cli, "--afmisp=0", wood, path], capture_output=True, text=True,
                           env=dict(os.environ, NIMPRESS_TIMINGS="1"))
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": r.stderr[-300:]}
        if best is None or dt < best:
            best, breakdown = dt, timings_line(r.stderr, "nimpress_timings")
    gt_bytes = 2 * n_rec * n                                   # int8 FORMAT/GT bytes of the records scored
    out = {"config2_e2e_s": best, "genotypes_per_s": n_rec * n / best, "records": n_rec, "samples": n,
           "workload": "nimpress --afmisp=0 wood-25282103-height.scores cohort.bcf (100 000 samples, "
                       "int8 GT, CSI random access), process start to exit, best of 2"}
    if breakdown:
        # the HIP context comes up on its own thread beside open + inflate + parse: hip_init_s is its duration,
        # hip_init_wait_s what of it the run had to wait for; process_start_s = what the process spent outside the listed stages
        listed = sum(breakdown[k] for k in ("hip_init_wait_s", "open_s", "inflate_parse_s", "push_s", "kernel_s",
                                            "warnings_s", "write_s"))
        breakdown["process_start_and_exit_s"] = best - listed
        out["breakdown"] = breakdown
        stage = breakdown["inflate_parse_s"] + breakdown["push_s"]
        out["ingest_stage_GBps"] = gt_bytes / stage / 1e9 if stage > 0 else None
    return out


def timings_line(stderr_text, key):
    """the one-line JSON object {key: {...}} a run printed on stderr (NIMPRESS_TIMINGS=1 / score_many.py --timings)"""
    for line in stderr_text.splitlines():
        if line.startswith("{") and key in line:
            try:
                return json.loads(line)[key]
            except ValueError:
                pass
    return None


def matrices_equal(path_a, path_b, rel=1e-6):
    """two samples x scores TSVs: same sample names, same NaN positions, values within `rel` relative (floored at 1e-12
    of the column's mean magnitude): (ok, max relative difference)"""
    a = [l.split("\t") for l in open(path_a).read().split("\n") if l]
    b = [l.split("\t") for l in open(path_b).read().split("\n") if l]
    if len(a) != len(b) or any(x[0] != y[0] or len(x) != len(y) for x, y in zip(a, b)):
        return False, float("inf")
    va = np.array([[float(v) for v in x[1:]] for x in a])
    vb = np.array([[float(v) for v in x[1:]] for x in b])
    if not np.array_equal(np.isnan(va), np.isnan(vb)):
        return False, float("inf")
    ok = ~np.isnan(va)
    if not ok.any():
        return True, 0.0
    floor = 1e-12 * np.nanmean(np.abs(va), axis=0, keepdims=True)
    d = np.abs(va - vb) / np.maximum(np.abs(va), np.maximum(floor, 1e-300))
    worst = float(np.nanmax(np.where(ok, d, 0.0)))
    return bool(worst <= rel), worst


def config4_e2e(tmpdir, n=500_000):
    """BASELINE.json configs[3] end to end on one GPU: the 8 score-format files of the reference tree on ONE
    500 000-sample BCF2 (+CSI) holding the union of their loci (tests/config2.py), through tools/score_many.py --
    file by file (the reference's loop, 8 times) and with --one-pass (the union decoded once into a resident cohort,
    the 8 definitions applied together on the matrix cores).  Process start to the matrix written."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import config2
    g = os.path.join(ROOT, "tests", "golden")
    files = sorted(os.path.join(g, "scores", f) for f in os.listdir(os.path.join(g, "scores"))) + [os.path.join(g, "set1.score")]
    path, samples, recs = config2.write_union_cohort(tmpdir, files, n)
    gt_bytes = sum(int(r["gts"].size) for r in recs)            # int8 FORMAT/GT bytes in the file's records
    out = {"workload": "tools/score_many.py --gpus 1 --afmisp=0 <8 score files> union.bcf (%d samples, %d records, "
                       "int8 GT, CSI random access), process start to the samples x scores matrix written" % (n, len(recs)),
           "score_files": len(files), "records": len(recs), "samples": n}
    res, brk = {}, {}
    for label, extra in (("per_file", []), ("one_pass", ["--one-pass"])):
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "score_many.py"), "--gpus", "1", "--afmisp=0",
                                "--timings", "--out", os.path.join(tmpdir, label + ".tsv")] + extra + files + [path],
                               capture_output=True, text=True)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": r.stderr[-300:]}
            if best is None or dt < best:
                best, brk[label] = dt, timings_line(r.stderr, "score_many_timings")
        res[label] = best
    # the two matrices, value by value (1e-6 relative: the one-pass weights are 49-bit fixed point)
    same, worst = matrices_equal(os.path.join(tmpdir, "per_file.tsv"), os.path.join(tmpdir, "one_pass.tsv"))
    out.update({"e2e_s": res["one_pass"], "per_file_e2e_s": res["per_file"],
                "one_pass_vs_per_file": res["per_file"] / res["one_pass"],
                "ingest_GBps_end_to_end": gt_bytes / res["one_pass"] / 1e9,
                "outputs_equal_within_1e-6_relative": bool(same), "outputs_max_relative_difference": worst})
    b = brk.get("one_pass")
    if b:
        # python_start_s: interpreter start until score_many's first line; hip_init_s runs on a thread of its own beside
        # open + inflate + parse, hip_init_wait_s is what of it the run waited for
        b["python_start_and_exit_s"] = res["one_pass"] - b["total_in_process_s"]
        out["breakdown"] = b
        stage = b["inflate_parse_s"] + b["push_s"]
        out["ingest_stage_GBps"] = gt_bytes / stage / 1e9 if stage > 0 else None
    if not same:
        out["error"] = "the one-pass matrix differs from the file-by-file matrix (max relative difference %g)" % worst
    return out



# ------------------------------------------------------------------------------------------------
# output: the driver parses the LAST stdout line; round 4's 20 KB line (every secondary object inline) was not parsed.
# The full object goes to a side file and to stderr, the last stdout line is the compact one.
LINE_LIMIT = 4096


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _r(x, digits=6):
    """floats of the compact line at `digits` significant digits (the full object keeps every digit)"""
    if isinstance(x, float):
        return float("%.*g" % (digits, x)) if x == x and abs(x) != float("inf") else x
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def secondary_summary(sec):
    """one number per leg (at most ten keys): enough to see that a leg ran and roughly what it measured"""
    s = {}

    def get(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d

    for name, path in (("strip_worst_frac_over_distributions", ("layout_sweep", "strip_layout_worst_frac")),
                       ("auto_layout_worst_frac_over_sizes", ("size_sweep", "auto_layout_worst_frac")),
                       ("multi_score_ms_per_8_score_pass", ("multi_score", "ms_per_pass")),
                       ("strip_given_tallies_frac", ("given_tallies", "frac_of_8TBps")),
                       ("config5_ds_frac", ("config5_ds", "roofline", "frac")),
                       ("ds16_frac_of_its_2B_per_dosage", ("config5_ds16", "roofline", "frac")),
                       ("streaming_int8_genotypes_per_s", ("streaming", "nps_push_gt_raw int8", "genotypes_per_s")),
                       ("config2_e2e_s", ("config2", "config2_e2e_s")),
                       ("config4_e2e_s", ("config4", "e2e_s"))):
        v = get(sec, *path)
        if v is not None:
            s[name] = v
    bad = sorted(k for k, v in sec.items() if isinstance(v, dict) and ("error" in v or "skipped" in v))
    if bad:
        s["legs_failed_or_skipped"] = bad
    return s


def compact_line(out):
    """the object the driver parses: the contract's keys, `roofline`, `cpu_baseline`, the parity flags, and a summary of
    the secondary legs; kept under LINE_LIMIT bytes whatever the legs returned"""
    c = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "rccl_ranks"))
    cfg = out.get("config", {})
    c["config"] = _pick(cfg, ("workload", "samples", "variants", "nloci", "mode", "parallelism"))
    c["roofline"] = _pick(out.get("roofline", {}), ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel",
                                                   "algorithmic_bytes_per_step", "kernel_ms_per_launch"))
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        c["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample", "nproc", "cpu_model"))
        for k in ("with_binomtest", "all_cores"):
            if k in cb:
                c["cpu_baseline"][k] = _pick(cb[k], ("value", "cores"))
    if "score_delta_vs_reference" in out:
        c["score_delta_vs_reference"] = _pick(out["score_delta_vs_reference"],
                                              ("max_abs", "max_rel", "within_1e-6_relative", "nloci_equal",
                                               "tally_recount_equal", "samples_checked", "rows_recounted"))
    if "secondary" in out:
        c["secondary_summary"] = secondary_summary(out["secondary"])
    if "multi_gpu" in out:
        mg = out["multi_gpu"]
        c["multi_gpu"] = ({k: _pick(v, ("value", "scaling", "n_gpus", "ms_per_pass", "scoring_ms", "exchange_ms", "nloci"))
                           for k, v in mg.items() if isinstance(v, dict)} if "error" not in mg else _pick(mg, ("error",)))
    for k in ("parity_failures", "parity_unchecked", "full_object", "rehearsal", "rehearsal_normalised", "seconds"):
        if k in out:
            c[k] = out[k]
    c = _r(c)
    # never over the limit: shed the optional parts, longest first
    for drop in (("secondary_summary",), ("multi_gpu",), ("cpu_baseline", "sample"), ("config", "workload")):
        if len(json.dumps(c)) < LINE_LIMIT:
            break
        d = c
        for k in drop[:-1]:
            d = d.get(k, {})
        d.pop(drop[-1], None)
    return c


# secondary legs whose result carries a parity check against the oracle (parity_failures walks them)
PARITY_LEGS = ("given_tallies", "layout_sweep", "size_sweep", "config5_ds", "config5_ds16", "multi_score", "config2", "config4")


def emit(out, full_path):
    """full object -> side file (+ stderr); compact object -> the last stdout line"""
    where = None
    try:
        os.makedirs(os.path.dirname(os.path.abspath(full_path)), exist_ok=True)
        with open(full_path, "w") as f:
            json.dump(out, f)
            f.write("\n")
        where = os.path.relpath(full_path, ROOT)
    except OSError:
        pass
    print("bench_full " + json.dumps(out), file=sys.stderr, flush=True)
    out = dict(out, full_object=where or "stderr (line starting with bench_full)")
    print(json.dumps(compact_line(out)), flush=True)


# ------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus, args.rank_timeout)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" in os.environ and args.gpus != world and rank == 0 and args.gpus != 1:
        print("bench.py: --gpus %d ignored, the launcher's WORLD_SIZE=%d decides" % (args.gpus, world), file=sys.stderr)
    args.gpus = world   # under torch.distributed.run the ranks come from the environment, --gpus or not

    import torch
    import torch.distributed as dist
    from nimpress_amd import capi, multi   # (either import order works: capi.load() sees to one HIP runtime)
    capi.load()
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU path to time)")
    torch.cuda.set_device(local_rank)
    rccl_ranks = 1
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        one = torch.ones(1, dtype=torch.int64, device="cuda")
        dist.all_reduce(one)               # sanity: every rank takes part in an RCCL collective
        rccl_ranks = int(one.item())
        assert rccl_ranks == dist.get_world_size() == world, (rccl_ranks, dist.get_world_size(), world)

    n, m = args.samples, args.variants
    mode = {"auto": capi.MODE_AUTO, "twopass": capi.MODE_TWOPASS, "fused": capi.MODE_FUSED}[args.mode]
    is_ds = args.format == "ds"
    strong = args.scaling == "strong" and world > 1
    strip = args.layout == "strip" and not is_ds
    fmt = capi.FMT_DS32 if is_ds else (capi.FMT_GT2X if strip else capi.FMT_GT2)

    # synthetic cohort, generated on the device; identical on every rank (same seed)
    _, eaf, miss = synth_score(m, args.seed, args.format)
    t_het, t_hom, t_miss = hwe_thresholds(eaf, miss)
    if strong:   # this rank's block of the ONE score's rows (blocks start on a group of 4)
        r0, r1 = multi.shard_rows(m, world, rank)
    else:
        r0, r1 = 0, m
    chunk = args.chunk_rows if args.chunk_rows > 0 else (r1 - r0)
    chunks = [(a, min(r1, a + chunk)) for a in range(r0, r1, max(chunk, 1))]
    resident = len(chunks) == 1
    cohort = capi.Cohort(n, chunks[0][1] - chunks[0][0] if chunks else 0, fmt=fmt, device=local_rank)

    def fill(a, b):     # rows [a, b) of the matrix -> cohort rows [0, b - a)
        for x in range(a, b, 1 << 15):      # (the DS generator takes at most 65 535 rows per call)
            y = min(b, x + (1 << 15))
            cohort.synth_at(x - a, x, args.seed, t_het[x:y], t_hom[x:y], t_miss[x:y])
        if not args.no_optimize and not strip:
            cohort.optimize()  # one-time layout step of a resident NPS_FMT_GT2 cohort (nps_cohort_optimize), untimed

    # this rank's score definition.  weak: its own betas (score files sharded across GPUs); strong: all
    # ranks share ONE score, each holds its rows
    beta_seed = args.seed + 1000 + (0 if strong else rank)
    beta = np.round(np.random.default_rng(beta_seed).normal(0.0, 0.02, m), 4)
    sdefs = [capi.ScoreDef(capi.row_descs(beta[a:b], eaf[a:b]), device=local_rank) for a, b in chunks]
    params = capi.make_params(imp_locus="ps") if is_ds else capi.make_params()
    sc = capi.Scorer(n, params, device=local_rank)
    geometry = sc.fused_geometry(chunks[0][1] - chunks[0][0], fmt) if chunks else (0, 0, 0)
    d_scores = torch.empty(n, dtype=torch.float64, device="cuda")
    offset = 0.0
    if resident and chunks:
        fill(*chunks[0])

    def step(timed=False):
        sc.reset()
        seg = 0.0
        for (a, b), sdef in zip(chunks, sdefs):
            if not resident:        # regenerate the chunk (untimed), then time the scoring call alone
                fill(a, b)
                sc.sync()
                t0 = time.perf_counter()
            sc.score_cohort_def(cohort, sdef, 0, mode)
            if not resident:
                sc.sync()
                seg += time.perf_counter() - t0
        if strong:
            # the one exchange of the row-sharded layout: sum all-reduce of sums and nloci (RCCL)
            nl = sc.partial_device(d_scores.data_ptr())
            _, nl = multi.all_reduce_partial(d_scores, nl)
            sc.normalize_device(d_scores.data_ptr(), nl, offset)
        else:
            nl = sc.finish_device(offset, d_scores.data_ptr())
            if world > 1:
                # the one exchange of the score-sharded layout: samples x scores matrix over RCCL
                step.matrix = multi.gather_scores(d_scores.view(1, n), world)
                torch.cuda.current_stream().synchronize()  # d_scores is rewritten by the next step
        step.seg += seg
        return nl

    step.seg = 0.0

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    nloci = 0
    for _ in range(args.warmup):
        nloci = step()
    sc.profile_enable(True)
    sc.profile_get(reset=True)
    step.seg = 0.0
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nloci = step()
    fence()
    elapsed = time.perf_counter() - t0
    if not resident:
        elapsed = step.seg     # chunked: the sum of the timed scoring segments (generation excluded)
    prof = sc.profile_get(reset=True)
    sc.profile_enable(False)
    t_headline = time.perf_counter()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    failed = []   # secondary measurements that raised (rank 0): the line is printed, the exit status is 1
    out = None
    if rank == 0:
        steps = max(args.steps, 1)
        genotypes_per_step = float(n) * float(m)
        value = (1 if strong else world) * genotypes_per_step * args.steps / elapsed
        # algorithmic bytes per step of THIS rank: one read of the matrix + per-row
        # params + one write of the scores
        my_m = r1 - r0
        alg_bytes = (4 * my_m * n if is_ds else my_m * ((n + 15) // 16) * 4) + 40 * my_m + 8 * n
        kern_ms = {"tally": prof.ms_tally, "params": prof.ms_params,
                   "accumulate": prof.ms_accumulate, "fused": prof.ms_fused,
                   "finish": prof.ms_reduce}
        hot_ms_per_step = (prof.ms_tally + prof.ms_params + prof.ms_accumulate + prof.ms_fused) / steps
        dominant = max(("tally", "accumulate", "fused"), key=lambda k: kern_ms[k])
        achieved = alg_bytes / (hot_ms_per_step * 1e-3) / 1e9 if hot_ms_per_step > 0 else 0.0
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_ds.json" if is_ds else
                             ("traffic_strip.json" if strip else "traffic.json"))
        if os.path.exists(tpath) and not strong:
            try:
                tj = json.load(open(tpath))
                if tj.get("samples") == n and tj.get("variants") == m:
                    traffic = tj.get("hbm_bytes_per_step")
                    traffic_source = ("profiles/%s (replayed: PMC passes of this command, %s; not measured "
                                      "in this run)" % (os.path.basename(tpath), tj.get("round", "round 1")))
            except Exception:
                traffic = None
        if strong:
            parallelism = "one score, rows sharded x%d + RCCL all-reduce of sums and nloci" % world
        elif world > 1:
            parallelism = "score-sharded x%d + RCCL all-gather" % world
        else:
            parallelism = "single GPU"
        out = {
            "metric": "genotype-dosage accumulations/s (samples x variants / s)",
            "value": value,
            "unit": "genotype-dosage accumulations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "rccl_ranks": rccl_ranks,
            "config": {"workload": ("synthetic %d-variant PRS on %d-sample float32 FORMAT/DS matrix "
                                    "(BASELINE.json configs[4]), 5 %% mean missingness, --imp-locus=ps" if is_ds else
                                    "synthetic %d-variant PRS on %d-sample 2-bit GT matrix resident in "
                                    "HBM (BASELINE.json configs[2]), CLI-default imputation flags")
                                   % (m, n),
                       "samples": n, "variants": m, "nloci": int(nloci),
                       "resident_chunks": len(chunks), "chunk_rows": chunk,
                       "persistent_grid": {"slices": geometry[0], "teams": geometry[1],
                                           "samples_per_slice": geometry[2]},
                       "cohort_layout": ("NPS_FMT_GT2X: strips of 2048 samples x superblocks of 128 rows x 1 KiB units; "
                                         "one workgroup per strip, FP4 codes x FP6 weight digits on the matrix cores, "
                                         "popcount tallies, time independent of the genotypes") if strip else
                                        "plain" if (args.no_optimize or is_ds) else
                                        "nps_cohort_optimize (one-time, untimed, data independent: parity layout of "
                                        "the high-bit planes, table lookups over twice as many LDS banks)",
                       "mode": args.mode, "parallelism": parallelism},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": dominant,
                         "algorithmic_bytes_per_step": alg_bytes,
                         "kernel_ms_per_launch": (kern_ms[dominant] / max(1, {"tally": prof.n_tally, "accumulate": prof.n_accumulate,
                                                                              "fused": prof.n_fused}[dominant])),
                         "kernel_ms_per_step": {k: v / steps for k, v in kern_ms.items()},
                         "launches_per_step": {"tally": prof.n_tally / steps,
                                               "params": prof.n_params / steps,
                                               "accumulate": prof.n_accumulate / steps,
                                               "fused": prof.n_fused / steps}},
        }
    # N > 1: the two strong-scaling curves of the north star, by all ranks, in this order on every rank.  The headline
    # line exists already: if a leg hangs (a collective some rank never reaches), a watchdog prints the line without the
    # legs and ends the process instead of losing the measurement to the caller's timeout.
    multi_legs = None
    if world > 1 and not args.no_multi_legs and not strong and not is_ds and resident:
        import threading

        def give_up():
            if rank == 0:
                out["multi_gpu"] = {"error": "the strong-scaling legs did not finish within %.0f s" % args.multi_legs_timeout}
                emit(out, args.full_out)
            else:
                time.sleep(5.0)
            os._exit(3)

        watchdog = threading.Timer(args.multi_legs_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        multi_legs = {"configs2_gt_rows_sharded": strong_scaling_leg(
            torch, dist, multi, capi, args, rank, world, local_rank, fmt, n, m, args.seed, 0, max(1, min(args.steps, 5)),
            cohort=cohort)}
        sc.close()
        for sd in sdefs:
            sd.close()
        cohort.close()      # the DS chunk needs the room
        torch.cuda.empty_cache()
        multi_legs["configs4_ds_rows_sharded"] = strong_scaling_leg(
            torch, dist, multi, capi, args, rank, world, local_rank, capi.FMT_DS32, args.ds_samples, args.ds_variants,
            20250105, args.ds_chunk_rows, 1)
        watchdog.cancel()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            t_leg = time.perf_counter()
            if not is_ds:
                out["cpu_baseline"] = cpu_baseline(n, eaf, miss, args.seed, args.cpu_rows)
            if resident:
                # one more pass with per-row statistics for the checker
                sc.reset()
                sc.score_cohort_def(cohort, sdefs[0], 0, mode)
                stats = sc.flush()
                got, nl = sc.finish(0.0)
                out["score_delta_vs_reference"] = score_delta(
                    stats, got, nl, args.format, beta, eaf, args.seed, n, m, t_het, t_hom, t_miss, geometry)
            out["seconds"] = {"to_headline": t_headline - T_START, "cpu_baseline_and_check": time.perf_counter() - t_leg}
        if world == 1 and not args.no_extras and not is_ds and resident:
            import tempfile
            import threading
            secondary = {}
            out["secondary"] = secondary
            leg_s = out.setdefault("seconds", {})

            def give_up_extras():
                # a secondary leg that hangs must not cost the headline: print what there is, end with status 3.  This runs
                # on the timer's thread while the main thread may be adding keys: the line is made from a SNAPSHOT (retried
                # if a dictionary changed size under the copy), and the process ends whatever happens here
                try:
                    import copy
                    snap = None
                    for _ in range(20):
                        try:
                            snap = copy.deepcopy(out)
                            break
                        except RuntimeError:
                            time.sleep(0.05)
                    if snap is None:
                        snap = {k: v for k, v in list(out.items()) if k != "secondary"}
                    snap.setdefault("secondary", {})["deadline"] = {
                        "error": "secondary legs unfinished %.0f s after process start" % args.extras_deadline}
                    snap["parity_unchecked"] = sorted(set(PARITY_LEGS) - set(k for k, v in snap["secondary"].items()
                                                                            if isinstance(v, dict) and "skipped" not in v and "error" not in v))
                    emit(snap, args.full_out)
                finally:
                    os._exit(3)

            watchdog = threading.Timer(max(1.0, args.extras_deadline - (time.perf_counter() - T_START)), give_up_extras)
            watchdog.daemon = True
            if not args.full_sweeps:
                watchdog.start()

            def leg(name, fn):
                # a failing secondary measurement never takes the headline line down, but it does fail the run;
                # one that no longer fits the run's window is skipped and listed (not a failure)
                if not args.full_sweeps and time.perf_counter() - T_START > args.extras_budget:
                    secondary[name] = {"skipped": "time budget (%.0f s since process start; --full-sweeps runs it)"
                                                  % args.extras_budget}
                    return
                t0 = time.perf_counter()
                try:
                    secondary[name] = fn()
                except Exception as e:
                    secondary[name] = {"error": repr(e)[:300]}
                if isinstance(secondary[name], dict) and "error" in secondary[name]:
                    failed.append(name)
                leg_s[name] = time.perf_counter() - t0
                torch.cuda.empty_cache()

            # legs that use the headline's resident cohort come first
            leg("given_tallies", lambda: given_tallies(capi, sc, cohort, sdefs[0], d_scores, n, m, elapsed / steps * 1e3))
            sc.close()
            for d in sdefs:
                d.close()
            cohort.close()      # frees the 125 GB matrix: the other legs need the room
            torch.cuda.empty_cache()

            def config2_leg():
                with tempfile.TemporaryDirectory() as td:
                    return config2_e2e(td)

            def config4_leg():
                with tempfile.TemporaryDirectory() as td:
                    return config4_e2e(td)

            def multi_leg():
                r = multi_score(capi, local_rank, args, n, m, args.seed)
                r["vs_single_score_passes"] = args.steps and (elapsed / steps * 1e3) / r["ms_per_score"]
                return r

            leg("layout_sweep", lambda: layout_sweep(capi, local_rank, n, m, args.seed, full=args.full_sweeps))
            leg("size_sweep", lambda: size_sweep(capi, local_rank, args, args.seed, full=args.full_sweeps))
            leg("config5_ds", lambda: ds_config5(capi, local_rank, args))
            # the same cohort shape at 2 bytes per dosage, one resident chunk of 600 000 rows (240 GB of float32 in 120 GB)
            leg("config5_ds16", lambda: ds_config5(capi, local_rank, args, m=600_000, chunk=600_000, half=True))
            leg("multi_score", multi_leg)
            leg("streaming", lambda: streaming_rates(capi, local_rank, n, args.seed))
            leg("config2", config2_leg)
            leg("config4", config4_leg)
            watchdog.cancel()
            # legs that carry a parity check and did not run (time budget): named in the line, so that a slow box cannot
            # exit 0 having checked nothing but the headline without saying so (--require-all-legs makes it a failure)
            unchecked = [k for k in PARITY_LEGS if isinstance(secondary.get(k), dict) and "skipped" in secondary[k]]
            if unchecked:
                out["parity_unchecked"] = unchecked
                if args.require_all_legs:
                    failed += ["unchecked:" + k for k in unchecked]
        if multi_legs is not None:
            out["multi_gpu"] = multi_legs
        bad = parity_failures(out)
        if bad:
            out["parity_failures"] = bad
        emit(out, args.full_out)
        failed += ["parity:" + p for p in bad]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit("bench.py: measurement(s) failed (a leg raised, or a parity check of the line is false): %s" % ", ".join(failed))


def parity_failures(obj, path=""):
    """Every place of the bench line where a parity check did not hold: a `score_delta_vs_reference` (headline or
    secondary) with a false `within...`, `nloci_equal`, `tally_recount_equal` or `tallies_and_nloci_equal`, and any
    `..._equal` / `outputs_equal...` of the end-to-end legs.  The documented exceptions are the two off-by-default
    precision options of the multi-score pass, `missing_weight_bits_32` and `weight_bits_41` (not
    inside the bar for every sample, reported as such)."""
    bad = []
    if isinstance(obj, dict):
        for k, v in obj.items():
            here = path + "/" + str(k)
            if k in ("missing_weight_bits_32", "missing_weight_bits_40", "weight_bits_41"):
                continue
            if isinstance(v, bool) and not v and (
                    k.startswith("within") or k in ("nloci_equal", "tally_recount_equal", "tallies_and_nloci_equal")
                    or k.startswith("outputs_equal")):
                bad.append(here)
            else:
                bad += parity_failures(v, here)
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            bad += parity_failures(v, "%s[%d]" % (path, i))
    return bad


if __name__ == "__main__":
    main()
