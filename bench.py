#!/usr/bin/env python3
"""bench.py -- headline benchmark: genotype-dosage accumulations/s on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the north-star metric is quoted on):
synthetic 1 000 000-variant PRS on a 500 000-sample bit-packed GT matrix (125 GB of 2-bit codes)
resident in HBM; flags = CLI defaults (imp-locus ps, imp-missing homref, imp-sample int_ps,
maxmis 0.05, mincs 100); every 1000th row has 10 % missingness so the locus-imputation branch
runs.  One STEP = one full pass of the hot path over the cohort: tally -> per-row decision/LUT ->
accumulate -> /(2 nloci) + offset, inputs already in HBM, result left in a device buffer.

N > 1 (weak scaling): one process per GPU (torch.distributed, backend nccl = RCCL).  Each rank
scores ITS OWN score definition (its own betas) against its own resident copy of the cohort --
multi-score evaluation sharded by score file, BASELINE.json north_star -- and each step ends
with the one real exchange of the path: an RCCL all-gather of the samples x scores matrix.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=500_000, help="cohort size N (default: config 3)")
    ap.add_argument("--variants", type=int, default=1_000_000, help="score rows M (default: config 3)")
    ap.add_argument("--mode", choices=["auto", "twopass", "fused"], default="auto")
    ap.add_argument("--format", choices=["gt", "ds"], default="gt",
                    help="gt: 2-bit packed GT matrix (the headline, config 3); ds: float32 FORMAT/DS "
                         "matrix (config 5 shape: pass --samples 200000 and as many --variants as fit HBM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimize", action="store_true",
                    help="skip nps_cohort_optimize (the row of each group of 4 with the most dosage-2 / "
                         "missing codes in the bank-selecting slot)")
    ap.add_argument("--cpu-rows", type=int, default=2000, help="rows of the CPU-baseline sample")
    ap.add_argument("--seed", type=int, default=20250103)
    return ap.parse_args()


def synth_score(m, seed):
    """beta ~ N(0,0.02^2) and eaf ~ U(0.01,0.5), 4 decimals;
    missing rate U(0,0.02), every 1000th row 0.10."""
    rng = np.random.default_rng(seed)
    beta = np.round(rng.normal(0.0, 0.02, m), 4)
    eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
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


def cpu_baseline(n, eaf, miss, seed, rows):
    """The oracle (CPU restatement of the reference's per-row path) timed on one host core on a
    bounded sample of the same workload: `rows` score rows at the full cohort size."""
    from oracle import refcpu
    n_distinct = 16
    th, tm, tmi = refcpu.hwe_thresholds(eaf[:n_distinct], miss[:n_distinct])
    codes = refcpu.synth_rows(n, 0, n_distinct, seed, th, tm, tmi)
    gts = np.stack([refcpu.codes_to_gt(codes[j], n) for j in range(n_distinct)])
    rng = np.random.default_rng(1)
    beta = np.round(rng.normal(0, 0.02, rows), 4)
    secs, _, _ = refcpu.bench_gt(gts, n, rows, beta, np.resize(eaf[:n_distinct], rows),
                                 refcpu.make_params())
    return {"value": n * rows / secs, "unit": "genotype-dosage accumulations/s", "cores": 1,
            "kind": "port",
            "sample": "%d score rows x %d samples (bcf_get_genotypes int32 buffers, %d distinct rows "
                      "cycled), literal decode+tally+impute+accumulate of nimpress.nim:561-583,"
                      "639-641, binomTest warnings off, %.1f s" % (rows, n, n_distinct, secs)}


def score_delta(sc_factory, cohort, beta, eaf, miss, seed, n, m, th, tm, tmi):
    """The second half of BASELINE.json's metric ("+ max-abs score delta vs reference"), outside the
    timed region and on rank 0 only: one more pass with per-row statistics, then the oracle's
    arithmetic (nimpress.nim:565-583, 639-649) on the CPU for the first 16 samples over ALL rows, fed
    with the row tallies the GPU reports, four of which are recounted over the full width by the
    oracle's generator.  The oracle is the checker here, nothing of it is timed or shipped."""
    from oracle import refcpu
    from nimpress_amd import capi
    sc = sc_factory()
    sc.score_cohort(cohort, capi.row_descs(beta, eaf))
    stats = sc.flush()
    got, nloci = sc.finish(0.0)
    sc.close()
    recount_ok = True
    for j in sorted({0, min(1000, m - 1), m // 2, m - 1}):
        codes = refcpu.synth_rows(n, j, 1, seed, th[j:j + 1], tm[j:j + 1], tmi[j:j + 1])
        c = np.unpackbits(codes.view(np.uint8), bitorder="little").reshape(-1, 2)
        code = (c[:, 0] + 2 * c[:, 1])[:n]
        recount_ok &= int((code == 2).sum()) == int(stats["nmissing"][j])
        recount_ok &= int((code == 1).sum() + 2 * (code == 3).sum()) == int(stats["neffect"][j])
    k = min(16, n)
    codes16 = refcpu.synth_rows(k, 0, m, seed, th, tm, tmi)[:, 0]
    nmiss = stats["nmissing"].astype(np.float64)
    ngen = float(n) - nmiss
    imp = np.where(ngen >= 100.0, stats["neffect"] / np.maximum(ngen, 1.0), eaf * 2.0)
    locus = (nmiss / float(n)) > 0.05
    ref = np.empty(k)
    for i in range(k):
        code = (codes16 >> np.uint32(2 * i)) & np.uint32(3)
        d = np.choose(code, [np.zeros(m), np.ones(m), imp, np.full(m, 2.0)])
        d = np.where(locus, eaf * 2.0, d)
        ref[i] = np.cumsum(d * beta)[-1] / (2.0 * m)
    delta = np.abs(got[:k] - ref)
    floor = 1e-12 * float(np.sum(np.abs(beta))) / (2.0 * max(int(nloci), 1))
    return {"max_abs": float(delta.max()), "max_rel": float((delta / np.maximum(np.abs(ref), floor)).max()),
            "nloci_equal": bool(int(nloci) == int(stats["used"].sum()) == m), "tally_recount_equal": bool(recount_ok),
            "checked": "first %d samples x all %d rows vs the oracle's arithmetic on the CPU; row tallies of 4 "
                       "rows recounted over all %d samples" % (k, m, n)}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "--nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU path to time)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from nimpress_amd import capi, multi
    n, m = args.samples, args.variants
    mode = {"auto": capi.MODE_AUTO, "twopass": capi.MODE_TWOPASS, "fused": capi.MODE_FUSED}[args.mode]

    # synthetic cohort, generated on the device; identical on every rank (same seed)
    _, eaf, miss = synth_score(m, args.seed)
    t_het, t_hom, t_miss = hwe_thresholds(eaf, miss)
    is_ds = args.format == "ds"
    cohort = capi.Cohort(n, m, fmt=capi.FMT_DS32 if is_ds else capi.FMT_GT2, device=local_rank)
    for r0 in range(0, m, 1 << 15):      # (the DS generator takes at most 65 535 rows per call)
        r1 = min(m, r0 + (1 << 15))
        cohort.synth(r0, args.seed, t_het[r0:r1], t_hom[r0:r1], t_miss[r0:r1])
    if not args.no_optimize:
        cohort.optimize()  # one-time layout step of a resident cohort (nps_cohort_optimize), untimed
    # this rank's score definition: its own betas (score files sharded across GPUs)
    beta = np.round(np.random.default_rng(args.seed + 1000 + rank).normal(0.0, 0.02, m), 4)
    sdef = capi.ScoreDef(capi.row_descs(beta, eaf), device=local_rank)
    sc = capi.Scorer(n, capi.make_params(), device=local_rank)
    d_scores = torch.empty(n, dtype=torch.float64, device="cuda")
    offset = 0.0

    def step():
        sc.reset()
        sc.score_cohort_def(cohort, sdef, 0, mode)
        nloci = sc.finish_device(offset, d_scores.data_ptr())
        if world > 1:
            # the one real exchange of the path: samples x scores matrix over RCCL (one score per rank)
            step.matrix = multi.gather_scores(d_scores.view(1, n), world)
        return nloci

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
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nloci = step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = sc.profile_get(reset=True)
    sc.profile_enable(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        steps = max(args.steps, 1)
        genotypes_per_step = float(n) * float(m)
        value = world * genotypes_per_step * args.steps / elapsed
        # algorithmic bytes per step: one read of the matrix + per-row
        # params + one write of the scores
        alg_bytes = (4 * m * n if is_ds else m * ((n + 15) // 16) * 4) + 40 * m + 8 * n
        kern_ms = {"tally": prof.ms_tally, "params": prof.ms_params,
                   "accumulate": prof.ms_accumulate, "fused": prof.ms_fused,
                   "finish": prof.ms_reduce}
        hot_ms_per_step = (prof.ms_tally + prof.ms_params + prof.ms_accumulate + prof.ms_fused) / steps
        dominant = max(("tally", "accumulate", "fused"), key=lambda k: kern_ms[k])
        achieved = alg_bytes / (hot_ms_per_step * 1e-3) / 1e9 if hot_ms_per_step > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_ds.json" if is_ds else "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("samples") == n and tj.get("variants") == m:
                    traffic = tj.get("hbm_bytes_per_step")
            except Exception:
                traffic = None
        out = {
            "metric": "genotype-dosage accumulations/s (samples x variants / s)",
            "value": value,
            "unit": "genotype-dosage accumulations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("synthetic %d-variant PRS on %d-sample float32 FORMAT/DS matrix "
                                    "resident in HBM (BASELINE.json configs[4] shape, rows limited to "
                                    "what fits one GPU), CLI-default imputation flags" if is_ds else
                                    "synthetic %d-variant PRS on %d-sample 2-bit GT matrix resident in "
                                    "HBM (BASELINE.json configs[2]), CLI-default imputation flags")
                                   % (m, n),
                       "samples": n, "variants": m, "nloci": int(nloci),
                       "cohort_layout": "plain" if (args.no_optimize or is_ds) else
                                        "nps_cohort_optimize (one-time, untimed: per group of 4 rows the row "
                                        "with the most dosage-2/missing codes in the bank-selecting slot)",
                       "mode": args.mode, "parallelism": "score-sharded x%d + RCCL all-gather" % world
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": dominant,
                         "algorithmic_bytes_per_step": alg_bytes,
                         "kernel_ms_per_step": {k: v / steps for k, v in kern_ms.items()},
                         "launches_per_step": {"tally": prof.n_tally / steps,
                                               "params": prof.n_params / steps,
                                               "accumulate": prof.n_accumulate / steps,
                                               "fused": prof.n_fused / steps}},
        }
        if world == 1 and not args.no_cpu_baseline and not is_ds:
            out["cpu_baseline"] = cpu_baseline(n, eaf, miss, args.seed, args.cpu_rows)
            out["score_delta_vs_reference"] = score_delta(
                lambda: capi.Scorer(n, capi.make_params(), device=local_rank), cohort, beta, eaf, miss,
                args.seed, n, m, t_het, t_hom, t_miss)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
