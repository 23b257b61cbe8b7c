"""GPU parity tests: the HIP path behind the C-ABI (libnps.so) vs the CPU oracle.

Bars (BASELINE.json north_star): nmissing / ngenotyped / neffect / used / reason / nloci bit-exact;
scores within 1e-6 relative (floored) -- in practice ~1e-15.
Every test calls through the C-ABI; the oracle is only the checker.
"""
import json
import math
import os

import numpy as np
import pytest

from nimpress_amd import capi
from oracle import refcpu

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6


def rel_err(got, ref, beta, nloci):
    """max |d| / max(|ref|, 1e-12 * sum|beta| / (2 nloci))  -- SURVEY.md section 8(d)."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), "NaN positions differ"
    ok = ~np.isnan(ref)
    if not ok.any():
        return 0.0
    floor = 1e-12 * float(np.sum(np.abs(beta))) / max(2.0 * nloci, 1.0)
    return float(np.max(np.abs(got[ok] - ref[ok]) / np.maximum(np.abs(ref[ok]), max(floor, 1e-300))))


def assert_stats_equal(gpu_stats, ref_stats):
    assert len(gpu_stats) == len(ref_stats)
    for k, (g, r) in enumerate(zip(gpu_stats, ref_stats)):
        assert int(g["ngenotyped"]) == int(r[0]), (k, g, r)
        assert int(g["nmissing"]) == int(r[1]), (k, g, r)
        assert float(g["neffect"]) == float(r[2]), (k, g, r)
        assert int(g["used"]) == int(r[3]), (k, g, r)
        assert int(g["reason"]) == int(r[4]), (k, g, r)


# ------------------------------------------------------------------------------------------
# config 1: tests/set1 through the C-ABI, all 13 golden cases of the reference
@pytest.fixture(scope="module")
def set1(golden_dir):
    score = refcpu.read_score_file(os.path.join(golden_dir, "set1.score"))
    vcf = refcpu.read_vcf(os.path.join(golden_dir, "set1.vcf.gz"))
    bed = refcpu.read_bed(os.path.join(golden_dir, "set1.bed"))
    cases = json.load(open(os.path.join(golden_dir, "set1_cases.json")))["cases"]
    return score, vcf, bed, cases


def gpu_set1(score, vcf, bed, case):
    """The row loop of computePolygenicScores with the per-row work on the GPU."""
    sc = capi.Scorer(len(vcf.samples), capi.make_params(case["imp_locus"], case["imp_missing"],
                                                        case["imp_sample"], case["maxmis"],
                                                        case["mincs"]))
    for e in score.entries:
        rie = e.refseq == e.easeq
        if case["restrict_to_covered"] and not refcpu.is_variant_covered(e, bed):
            sc.push_locus(capi.ROW_UNCOVERED, rie, e.beta, e.eaf)
            continue
        rec = refcpu.find_variant(vcf, e)
        if rec is None:
            sc.push_locus(capi.ROW_ABSENT, rie, e.beta, e.eaf)
            continue
        if not case["ignore_filter"] and rec.filt not in (".", "PASS"):
            sc.push_locus(capi.ROW_FILTERED, rie, e.beta, e.eaf)
            continue
        eaidx = 0 if rie else rec.alts.index(e.easeq) + 1
        sc.push_gt(rec.gts, rec.ploidy, eaidx, rie, e.beta, e.eaf)
    stats = sc.flush()
    scores, nloci = sc.finish(score.offset)
    sc.close()
    return scores, nloci, stats


@pytest.mark.parametrize("idx", range(13))
def test_set1_golden_case(set1, idx):
    score, vcf, bed, cases = set1
    case = cases[idx]
    scores, nloci, stats = gpu_set1(score, vcf, bed, case)
    # (a) the reference's own expected values, tests/test_set1.nim tolerance 1e-4, NaN exact
    for got, exp in zip(scores, case["expected"]):
        assert (exp is None) == bool(np.isnan(got)), (scores, case["expected"])
        if exp is not None:
            assert abs(got - exp) <= 1e-4
    # (b) the oracle, to the north-star bar
    ref_scores, ref_nloci, ref_stats = refcpu.compute_polygenic_scores(
        score, vcf, case["restrict_to_covered"], bed, case["imp_locus"], case["imp_missing"],
        case["imp_sample"], case["maxmis"], case["mincs"], case["ignore_filter"])
    assert nloci == ref_nloci
    assert_stats_equal(stats, ref_stats)
    assert rel_err(scores, ref_scores, [e.beta for e in score.entries], max(nloci, 1)) <= REL_TOL


def test_set1_cli_defaults(set1):
    score, vcf, bed, _ = set1
    case = dict(restrict_to_covered=False, imp_locus="ps", imp_missing="homref", imp_sample="int_ps",
                maxmis=0.05, mincs=100, ignore_filter=False)
    scores, nloci, stats = gpu_set1(score, vcf, bed, case)
    assert nloci == 6
    assert np.allclose(scores, 0.1545, atol=1e-12)


# ------------------------------------------------------------------------------------------
# synthetic cohorts
def make_cohort(n, m, seed, rng, max_miss=0.1, force_missing_rows=True):
    eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
    miss = rng.uniform(0.0, max_miss, m)
    if force_missing_rows and m >= 8:
        miss[::7] = 0.3          # exceed any maxmis <= 0.3 on some rows
        miss[3] = 1.0 - 1e-9     # (almost) all missing
    beta = np.round(rng.normal(0, 0.02, m), 4)
    rie = (rng.uniform(size=m) < 0.25).astype(np.int32)
    th, tm, tmi = refcpu.hwe_thresholds(eaf, miss)
    codes = refcpu.synth_rows(n, 0, m, seed, th, tm, tmi)
    return dict(n=n, m=m, eaf=eaf, beta=beta, rie=rie, th=th, tm=tm, tmi=tmi, codes=codes, seed=seed)


def oracle_scores(co, params_kw, offset, kind=None):
    p = refcpu.make_params(**params_kw)
    kind = np.zeros(co["m"], np.int32) if kind is None else kind
    n_present = int((kind == 0).sum())
    return refcpu.score_packed(co["codes"][:n_present], co["n"], kind, co["rie"], co["beta"],
                               co["eaf"], p, offset)


def test_synth_matches_oracle():
    rng = np.random.default_rng(11)
    for n, m in [(1, 3), (15, 4), (16, 5), (17, 6), (1000, 33), (4097, 9)]:
        co = make_cohort(n, m, 1234 + n, rng)
        dev = capi.Cohort(n, m)
        dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
        got = dev.download(0, m)
        dev.close()
        assert np.array_equal(got, co["codes"][:, : (n + 15) // 16]), (n, m)


PARAM_GRID = [
    dict(imp_locus="ps", imp_missing="homref", imp_sample="int_ps", maxmis=0.05, mincs=100),
    dict(imp_locus="homref", imp_missing="ignore", imp_sample="ps", maxmis=0.2, mincs=0),
    dict(imp_locus="ignore", imp_missing="homref", imp_sample="homref", maxmis=0.05, mincs=100),
    dict(imp_locus="fail", imp_missing="homref", imp_sample="int_ps", maxmis=0.5, mincs=10),
    dict(imp_locus="ps", imp_missing="homref", imp_sample="fail", maxmis=1.0, mincs=100),
    dict(imp_locus="ps", imp_missing="homref", imp_sample="int_fail", maxmis=1.0, mincs=3000),
    dict(imp_locus="ps", imp_missing="homref", imp_sample="int_ps", maxmis=1.0, mincs=3000),
]


@pytest.mark.parametrize("pk", range(len(PARAM_GRID)))
@pytest.mark.parametrize("shape", [(1, 1), (6, 7), (16, 4), (17, 5), (255, 13), (2049, 37),
                                   (5000, 130)])
def test_streaming_packed_vs_oracle(pk, shape):
    n, m = shape
    rng = np.random.default_rng(100 * pk + n)
    co = make_cohort(n, m, 777 + pk, rng)
    kw = PARAM_GRID[pk]
    sc = capi.Scorer(n, capi.make_params(**kw))
    for j in range(m):
        sc.push_packed(co["codes"][j], co["rie"][j], co["beta"][j], co["eaf"][j])
    stats = sc.flush()
    scores, nloci = sc.finish(0.25)
    sc.close()
    ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 0.25)
    assert nloci == ref_nloci
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    assert rel_err(scores, ref_scores, co["beta"], max(nloci, 1)) <= REL_TOL


@pytest.mark.parametrize("pk", [0, 1, 3])
def test_streaming_gt_decode_vs_oracle(pk):
    """raw bcf_get_genotypes buffers (int32) decoded on the device, incl. phased, half-missing,
    haploid-padded and multi-allelic calls."""
    n, m = 777, 21
    rng = np.random.default_rng(5 + pk)
    kw = PARAM_GRID[pk]
    sc = capi.Scorer(n, capi.make_params(**kw))
    ref = refcpu.RefScorer(n, refcpu.make_params(**kw))
    betas = []
    for j in range(m):
        ploidy = 1 if j % 5 == 4 else 2
        n_alt = 1 + j % 3
        eaidx = int(rng.integers(0, n_alt + 1))
        alle = rng.integers(-1, n_alt + 1, size=(n, ploidy))            # -1 = missing allele
        alle[rng.uniform(size=n) < 0.03] = -1
        phased = rng.integers(0, 2, size=(n, ploidy))
        gts = ((alle + 1) << 1) | phased
        gts = gts.astype(np.int32)
        if ploidy == 2:
            hap = rng.uniform(size=n) < 0.05                              # haploid call, padded
            gts[hap, 1] = -2147483647
        beta, eaf = float(rng.normal(0, 0.1)), float(rng.uniform(0.05, 0.6))
        rie = eaidx == 0
        sc.push_gt(gts.ravel(), ploidy, eaidx, rie, beta, eaf)
        ref.row_gt(gts.ravel(), ploidy, eaidx, rie, beta, eaf)
        betas.append(beta)
        if j % 6 == 2:
            kind = [capi.ROW_UNCOVERED, capi.ROW_ABSENT, capi.ROW_FILTERED][j % 3]
            sc.push_locus(kind, rie, beta * 0.5, eaf)
            ref.row_locus(kind, rie, beta * 0.5, eaf)
            betas.append(beta * 0.5)
    stats = sc.flush()
    scores, nloci = sc.finish(-1.5)
    sc.close()
    ref_scores, ref_nloci = ref.finish(-1.5)
    assert nloci == ref_nloci
    assert_stats_equal(stats, ref.stats)
    assert rel_err(scores, ref_scores, betas, max(nloci, 1)) <= REL_TOL


@pytest.mark.parametrize("mode", [capi.MODE_TWOPASS, capi.MODE_FUSED, capi.MODE_AUTO])
@pytest.mark.parametrize("shape", [(33, 1), (100, 3), (1000, 64), (4096, 257), (20000, 1001),
                                   (16385, 47), (40000, 160), (1, 50), (70000, 33)])
def test_resident_cohort_vs_oracle(shape, mode):
    """two-pass kernels, the fused single-read persistent kernel (1, 2, 3 and 5 sample slices per
    team; ragged last batch; fewer batches than CUs) and AUTO all against the oracle"""
    n, m = shape
    rng = np.random.default_rng(n + m)
    co = make_cohort(n, m, 4242, rng)
    kw = PARAM_GRID[(n + m) % len(PARAM_GRID)]
    dev = capi.Cohort(n, m)
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    sc = capi.Scorer(n, capi.make_params(**kw))
    sc.score_cohort(dev, capi.row_descs(co["beta"], co["eaf"], None, co["rie"]), 0, mode)
    stats = sc.flush()
    scores, nloci = sc.finish(0.0)
    sc.close()
    dev.close()
    ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 0.0)
    assert nloci == ref_nloci
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    assert rel_err(scores, ref_scores, co["beta"], max(nloci, 1)) <= REL_TOL


@pytest.mark.parametrize("pk", range(len(PARAM_GRID)))
def test_fused_all_imputation_modes(pk):
    n, m = 3000, 200
    rng = np.random.default_rng(900 + pk)
    co = make_cohort(n, m, 31 + pk, rng)
    kw = PARAM_GRID[pk]
    dev = capi.Cohort(n, m)
    dev.upload(0, co["codes"])
    sc = capi.Scorer(n, capi.make_params(**kw))
    sc.score_cohort(dev, capi.row_descs(co["beta"], co["eaf"], None, co["rie"]), 0, capi.MODE_FUSED)
    stats = sc.flush()
    scores, nloci = sc.finish(0.125)
    sc.close()
    dev.close()
    ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 0.125)
    assert nloci == ref_nloci
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    assert rel_err(scores, ref_scores, co["beta"], max(nloci, 1)) <= REL_TOL


def test_fused_equals_twopass_large():
    """500 000 samples x 4096 rows (35 slices x 7 teams, the bench geometry): the two HIP paths
    must agree -- tallies bit for bit, scores to rounding -- at a size the oracle cannot reach."""
    n, m = 500_000, 4096
    rng = np.random.default_rng(123)
    eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
    miss = rng.uniform(0, 0.02, m)
    miss[::100] = 0.1
    beta = np.round(rng.normal(0, 0.02, m), 4)
    th, tm, tmi = refcpu.hwe_thresholds(eaf, miss)
    dev = capi.Cohort(n, m)
    dev.synth(0, 5, th, tm, tmi)
    res = []
    for mode in (capi.MODE_TWOPASS, capi.MODE_FUSED):
        sc = capi.Scorer(n, capi.make_params())
        sc.score_cohort(dev, capi.row_descs(beta, eaf), 0, mode)
        stats = sc.flush()
        scores, nloci = sc.finish(0.0)
        res.append((stats, scores, nloci))
        sc.close()
    dev.close()
    assert res[0][2] == res[1][2]
    assert np.array_equal(res[0][0], res[1][0])
    assert int((res[0][0]["reason"] == capi.REASON_MAXMIS).sum()) >= 30
    scale = np.sum(np.abs(beta)) / (2 * res[0][2])
    assert np.max(np.abs(res[0][1] - res[1][1])) <= 1e-12 * scale
    # spot-check 3 rows' tallies against the oracle's generator (full width, CPU seconds)
    for j in (0, 100, 4095):
        codes = refcpu.synth_rows(n, j, 1, 5, th[j:j + 1], tm[j:j + 1], tmi[j:j + 1])
        c = np.unpackbits(codes.view(np.uint8), bitorder="little").reshape(-1, 2)
        code = (c[:, 0] + 2 * c[:, 1])[:n]
        assert int((code == 2).sum()) == int(res[1][0]["nmissing"][j])       # NPS_CODE_MISSING
        assert int((code == 1).sum() + 2 * (code == 3).sum()) == int(res[1][0]["neffect"][j])


@pytest.mark.parametrize("mode", [capi.MODE_TWOPASS, capi.MODE_FUSED])
@pytest.mark.parametrize("shape", [(777, 50), (20011, 203), (40000, 64)])
def test_optimized_cohort_same_results(shape, mode):
    """nps_cohort_optimize puts the cohort into the parity layout (the high-bit plane of the first row of
    every group of 4 holds the XOR of the four planes: LDS bank selection): scores, per-row statistics,
    nloci and the downloaded rows are those of the plain cohort, for the whole cohort and for sub-ranges,
    also ones that end inside a group; an upload puts the plain layout back"""
    n, m = shape
    rng = np.random.default_rng(n + m)
    co = make_cohort(n, m, 4242 + n, rng)
    # skewed frequencies: rows with hardly any and rows with many dosage-2 / missing codes in one group
    eaf = np.where(np.arange(m) % 3 == 0, 0.02, co["eaf"])
    th, tm, tmi = refcpu.hwe_thresholds(eaf, rng.uniform(0, 0.08, m))
    descs = capi.row_descs(co["beta"], eaf, None, co["rie"])
    plain, opt = capi.Cohort(n, m), capi.Cohort(n, m)
    for dev in (plain, opt):
        dev.synth(0, co["seed"], th, tm, tmi)
    rows0 = plain.download(0, m)
    opt.optimize()
    assert np.array_equal(opt.download(0, m), rows0)
    assert np.array_equal(opt.download(4, min(9, m - 4)), rows0[4:4 + min(9, m - 4)])
    kw = PARAM_GRID[0]
    ranges = [(0, m)]
    if m >= 24:
        ranges += [(8, 16), (m // 8 * 4, m), (4, 10), (0, 1)]   # whole groups, the tail, ranges ending inside a group
    for r0, r1 in ranges:
        res = []
        for dev in (plain, opt):
            sc = capi.Scorer(n, capi.make_params(**kw))
            sc.score_cohort(dev, descs[r0:r1], r0, mode)
            stats = sc.flush()
            scores, nloci = sc.finish(0.25)
            sc.close()
            res.append((stats, scores, nloci))
        assert res[0][2] == res[1][2]
        assert np.array_equal(res[0][0], res[1][0])
        assert rel_err(res[1][1], res[0][1], co["beta"][r0:r1], max(res[0][2], 1)) <= 1e-12
    # an upload returns the cohort to the plain layout (and still holds the right rows)
    opt.upload(0, rows0[:4])
    assert np.array_equal(opt.download(0, m), rows0)
    plain.close()
    opt.close()


def test_resident_mixed_kinds_and_upload():
    """uploaded (not synthesised) rows, interleaved with rows that have no genotype data"""
    n, m = 999, 50
    rng = np.random.default_rng(9)
    co = make_cohort(n, m, 1, rng)
    kind = np.zeros(m, np.int32)
    kind[[2, 11, 12, 30, 49]] = [1, 2, 3, 2, 1]
    n_present = int((kind == 0).sum())
    kw = PARAM_GRID[1]
    dev = capi.Cohort(n, n_present)
    dev.upload(0, co["codes"][:n_present])
    sc = capi.Scorer(n, capi.make_params(**kw))
    sc.score_cohort(dev, capi.row_descs(co["beta"], co["eaf"], kind, co["rie"]))
    stats = sc.flush()
    scores, nloci = sc.finish(1.0)
    ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 1.0, kind)
    assert nloci == ref_nloci
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    assert rel_err(scores, ref_scores, co["beta"], max(nloci, 1)) <= REL_TOL
    sc.close()
    dev.close()


def test_batch_rollover_and_reset():
    """more rows than one device batch holds (batch cap 4096 rows at small N), then reuse"""
    n, m = 40, 4500
    rng = np.random.default_rng(3)
    co = make_cohort(n, m, 17, rng, force_missing_rows=False)
    kw = PARAM_GRID[0]
    sc = capi.Scorer(n, capi.make_params(**kw))
    for rep in range(2):
        for j in range(m):
            sc.push_packed(co["codes"][j], co["rie"][j], co["beta"][j], co["eaf"][j])
        stats = sc.flush()
        scores, nloci = sc.finish(0.5)
        ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 0.5)
        assert nloci == ref_nloci
        assert_stats_equal(stats, [tuple(s) for s in ref_stats])
        assert rel_err(scores, ref_scores, co["beta"], max(nloci, 1)) <= REL_TOL
        sc.reset()
    sc.close()


def test_empty_inputs():
    # no rows at all: nloci = 0 -> 0/0 = NaN for every sample (nimpress.nim:645)
    sc = capi.Scorer(5, capi.make_params())
    scores, nloci = sc.finish(0.1)
    assert nloci == 0 and np.isnan(scores).all()
    sc.close()
    # every row ignored
    sc = capi.Scorer(5, capi.make_params(imp_locus="ignore", imp_missing="ignore"))
    sc.push_locus(capi.ROW_ABSENT, False, 0.3, 0.2)
    sc.push_locus(capi.ROW_UNCOVERED, False, 0.3, 0.2)
    stats = sc.flush()
    scores, nloci = sc.finish(0.1)
    assert nloci == 0 and np.isnan(scores).all() and [int(s["used"]) for s in stats] == [0, 0]
    sc.close()


def test_nan_eaf_and_fail_propagation():
    n = 64
    rng = np.random.default_rng(21)
    co = make_cohort(n, 8, 5, rng, force_missing_rows=False)
    co["eaf"][2] = np.nan   # makescore.R:533 writes NaN eaf; ps then imputes NaN
    kw = dict(imp_locus="ps", imp_missing="homref", imp_sample="ps", maxmis=1.0, mincs=0)
    sc = capi.Scorer(n, capi.make_params(**kw))
    for j in range(8):
        sc.push_packed(co["codes"][j], co["rie"][j], co["beta"][j], co["eaf"][j])
    scores, nloci = sc.finish(0.0)
    sc.close()
    ref_scores, _, ref_nloci = oracle_scores(co, kw, 0.0)
    assert nloci == ref_nloci
    assert rel_err(scores, ref_scores, co["beta"], nloci) <= REL_TOL


def test_bad_arguments_are_errors_not_aborts():
    sc = capi.Scorer(4, capi.make_params())
    with pytest.raises(capi.NpsError) as ei:
        sc.push_gt(np.zeros(36, np.int32), 9, 1, False, 0.1, 0.1)     # ploidy 9
    assert ei.value.status == -6
    with pytest.raises(capi.NpsError):
        sc.push_locus(0, False, 0.1, 0.1)                              # PRESENT is not a no-data kind
    with pytest.raises(capi.NpsError):
        sc.push_gt(np.zeros(8, np.int32), 2, -1, False, 0.1, 0.1)     # eaidx < 0 (nimpress.nim:380)
    sc.close()
    with pytest.raises(capi.NpsError):
        capi.Scorer(4, capi.NpsParams(9, 0, 0, 0, 0.05, 100))


def test_linearity_property_large():
    """size-independent property at a size the oracle would not finish quickly:
    scores(beta1 + beta2) - offset == (scores(beta1) - offset) + (scores(beta2) - offset) when no
    imputation value depends on beta (it never does) -- checked on 100k samples x 4096 rows."""
    n, m = 100_000, 4096
    rng = np.random.default_rng(77)
    eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
    miss = rng.uniform(0, 0.04, m)
    th, tm, tmi = refcpu.hwe_thresholds(eaf, miss)
    dev = capi.Cohort(n, m)
    dev.synth(0, 99, th, tm, tmi)
    b1 = np.round(rng.normal(0, 0.02, m), 4)
    b2 = np.round(rng.normal(0, 0.02, m), 4)
    outs = []
    for b in (b1, b2, b1 + b2):
        sc = capi.Scorer(n, capi.make_params())
        sc.score_cohort(dev, capi.row_descs(b, eaf))
        s, nloci = sc.finish(0.0)
        assert nloci == m
        outs.append(s)
        sc.close()
    dev.close()
    scale = np.sum(np.abs(b1) + np.abs(b2)) / (2 * m)
    assert np.max(np.abs(outs[2] - (outs[0] + outs[1]))) <= 1e-12 * scale * 10


def test_full_size_config3_properties():
    """BASELINE.json configs[2] at its full size (500 000 samples x 1 000 000 rows, 125 GB of 2-bit
    codes resident in HBM): far beyond what the oracle can score, so parity goes through
    size-independent properties plus the oracle on what it CAN reach:
      * every row's decision (1000 rows over --maxmis, nloci = M); the whole-row tallies of 20 000 random
        rows recounted by the oracle over all 500 000 samples;
      * 2 800 samples taken from every one of the 35 slices of the persistent grid (and the last ragged
        word) scored over ALL rows by oracle/refcpu.c (ref_score_subset: the restated procs of
        nimpress.nim:367-391, 565-583, 639-649, fed with every row's whole-row tally);
      * fused single-read kernel == two-pass kernels for every sample;
      * scores(2 beta) == 2 scores(beta) bit for bit (scaling by 2 is exact in every step);
      * the two row halves scored separately add up to the whole."""
    import torch
    n, m, seed = 500_000, 1_000_000, 20250103
    from conftest import need_free_hbm
    need_free_hbm(150)
    rng = np.random.default_rng(seed)
    beta = np.round(rng.normal(0.0, 0.02, m), 4)
    eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
    miss = rng.uniform(0.0, 0.02, m)
    miss[::1000] = 0.10
    th, tm, tmi = refcpu.hwe_thresholds(eaf, miss)
    dev = capi.Cohort(n, m)
    for r0 in range(0, m, 1 << 15):
        r1 = min(m, r0 + (1 << 15))
        dev.synth(r0, seed, th[r0:r1], tm[r0:r1], tmi[r0:r1])
    descs = capi.row_descs(beta, eaf)

    def run(rows, row0=0, mode=capi.MODE_FUSED, want_stats=False):
        sc = capi.Scorer(n, capi.make_params())
        sc.score_cohort(dev, rows, row0, mode)
        stats = sc.flush() if want_stats else None
        part = torch.empty(n, dtype=torch.float64, device="cuda")
        nloci = sc.partial_device(part.data_ptr())
        sc.close()
        return part, nloci, stats

    whole, nloci, stats = run(descs, want_stats=True)
    assert nloci == m
    over = stats["reason"] == capi.REASON_MAXMIS
    assert int(over.sum()) == m // 1000 and bool(over[::1000].all())
    assert int(stats["used"].sum()) == m
    # whole-row tallies of 1000 random rows (+ first, last, an over-maxmis one) recounted by the oracle
    # over all 500 000 samples: decode nim:367-391 + tallyAlleles nim:32-47, bit for bit
    rows = np.unique(np.concatenate([np.random.default_rng(1).choice(m, 20000, replace=False),
                                     [0, 1000, 499_999, m - 1]])).astype(np.uint64)
    ri = rows.astype(np.int64)
    g, ms, ne = refcpu.tally_synth_rows(rows, n, seed, th[ri], tm[ri], tmi[ri])
    assert np.array_equal(g, stats["ngenotyped"][ri].astype(np.float64))
    assert np.array_equal(ms, stats["nmissing"][ri].astype(np.float64))
    assert np.array_equal(ne, stats["neffect"][ri])
    assert np.array_equal(stats["ngenotyped"] + stats["nmissing"], np.full(m, n, dtype=np.uint64))
    # samples from EVERY slice of the persistent grid (first / lane 31 / lane 63 / middle / last thread
    # of each of the 35 slices, and the last ragged word with sample 499 999) scored over ALL rows by
    # oracle/refcpu.c: the restated decode, maxmis decision, imputeLocus/SampleDosages and accumulation
    # in row order, fed with every row's whole-row tally
    sc0 = capi.Scorer(n, capi.make_params())
    slices, teams, sps = sc0.fused_geometry(m)
    sc0.close()
    assert slices * teams > 200 and sps % 16 == 0, (slices, teams, sps)  # the single-read kernel is in use
    units, per_slice = (n + 15) // 16, sps // 16
    cols = {units - 1}
    for p_ in range(slices):
        first, last = p_ * per_slice, min(units, (p_ + 1) * per_slice) - 1
        cols.update(c for c in (first, first + 31, first + 63, (first + last) // 2, last) if first <= c <= last)
    assert len({c // per_slice for c in cols}) == slices == (units + per_slice - 1) // per_slice
    samples = np.concatenate([np.arange(c * 16, min(n, (c + 1) * 16)) for c in sorted(cols)]).astype(np.uint64)
    assert samples[-1] == n - 1 and samples.size >= 64 * 16
    sums, ref_nloci = refcpu.score_subset(samples, n, 0, seed, th, tm, tmi, beta, eaf, 0,
                                          stats["ngenotyped"].astype(np.float64),
                                          stats["nmissing"].astype(np.float64), stats["neffect"],
                                          refcpu.make_params())
    assert ref_nloci == nloci == m
    expect = sums / (2.0 * ref_nloci)
    got = (whole / (2.0 * nloci)).cpu().numpy()[samples.astype(np.int64)]
    scale = float(np.sum(np.abs(beta))) / (2.0 * m)
    assert np.max(np.abs(got - expect)) <= 1e-12 * scale
    assert np.max(np.abs(got - expect) / np.maximum(np.abs(expect), 1e-12 * scale)) <= 1e-6
    # the two HIP paths agree for every sample
    twopass, nloci2, _ = run(descs, mode=capi.MODE_TWOPASS)
    assert nloci2 == m
    assert float((whole - twopass).abs().max()) <= 1e-12 * float(np.sum(np.abs(beta)))
    del twopass
    # exact scaling
    doubled, _, _ = run(capi.row_descs(2.0 * beta, eaf))
    assert bool(torch.equal(doubled, 2.0 * whole))
    del doubled
    # row halves
    h = m // 2
    lo, nlo, _ = run(descs[:h])
    hi, nhi, _ = run(descs[h:], row0=h)
    assert nlo + nhi == m
    assert float((lo + hi - whole).abs().max()) <= 1e-12 * float(np.sum(np.abs(beta)))
    dev.close()


# ------------------------------------------------------------------------------------------
# FORMAT/DS (float32 dosage) path -- build-defined extension, oracle = ref_row_ds
def make_ds_cohort(n, m, seed, rng):
    eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
    miss = rng.uniform(0.0, 0.10, m)           # mean 5 %: about half the rows exceed maxmis = 0.05
    beta = np.round(rng.normal(0, 0.02, m), 4)
    rie = (rng.uniform(size=m) < 0.3).astype(np.int32)
    th, tm, tmi = refcpu.hwe_thresholds(eaf, miss)
    ds = refcpu.synth_rows_ds(n, 0, m, seed, th, tm, tmi)
    return dict(n=n, m=m, eaf=eaf, beta=beta, rie=rie, th=th, tm=tm, tmi=tmi, ds=ds, seed=seed)


def oracle_ds(co, kw, offset):
    sc = refcpu.RefScorer(co["n"], refcpu.make_params(**kw))
    for j in range(co["m"]):
        sc.row_ds(co["ds"][j], bool(co["rie"][j]), co["beta"][j], co["eaf"][j])
    scores, nloci = sc.finish(offset)
    return scores, sc.stats, nloci


def assert_ds_stats(gpu_stats, ref_stats):
    assert len(gpu_stats) == len(ref_stats)
    for k, (g, r) in enumerate(zip(gpu_stats, ref_stats)):
        assert int(g["ngenotyped"]) == int(r[0]) and int(g["nmissing"]) == int(r[1]), (k, g, r)
        # neffect is a float64 sum here: fixed-shape tree on the GPU, sequential in the oracle
        assert abs(float(g["neffect"]) - float(r[2])) <= 1e-9 * max(1.0, abs(float(r[2]))), (k, g, r)
        assert int(g["used"]) == int(r[3]) and int(g["reason"]) == int(r[4]), (k, g, r)


@pytest.mark.parametrize("pk", [0, 1, 4])
@pytest.mark.parametrize("shape", [(1, 2), (63, 5), (1000, 40), (5003, 70)])
def test_ds_streaming_vs_oracle(shape, pk):
    n, m = shape
    rng = np.random.default_rng(n * 7 + pk)
    co = make_ds_cohort(n, m, 99, rng)
    kw = PARAM_GRID[pk]
    sc = capi.Scorer(n, capi.make_params(**kw))
    for j in range(m):
        sc.push_ds(co["ds"][j], co["rie"][j], co["beta"][j], co["eaf"][j])
    stats = sc.flush()
    scores, nloci = sc.finish(0.3)
    sc.close()
    ref_scores, ref_stats, ref_nloci = oracle_ds(co, kw, 0.3)
    assert nloci == ref_nloci
    assert_ds_stats(stats, ref_stats)
    assert rel_err(scores, ref_scores, co["beta"], max(nloci, 1)) <= REL_TOL


@pytest.mark.parametrize("mode", ["twopass", "fused"])
@pytest.mark.parametrize("shape", [(100, 3), (1, 1), (4099, 129), (7681, 11), (30000, 500), (15361, 2001)])
def test_ds_resident_vs_oracle(shape, mode):
    """resident FORMAT/DS cohort: the two-pass kernels and the single-read fused kernel (2 rows per
    batch, 7 680 samples per workgroup: the shapes cover ragged batches, an almost empty last slice
    and more row batches than teams)"""
    n, m = shape
    rng = np.random.default_rng(n + 3 * m)
    co = make_ds_cohort(n, m, 2025, rng)
    kw = dict(imp_locus="ps", imp_missing="homref", imp_sample="int_ps", maxmis=0.05, mincs=100)
    dev = capi.Cohort(n, m, fmt=capi.FMT_DS32)
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    got = dev.download(0, m)
    assert np.array_equal(got.view(np.uint32) & 0x7fffffff > 0x7f800000, np.isnan(co["ds"]))
    assert np.array_equal(np.nan_to_num(got, nan=-1.0), np.nan_to_num(co["ds"], nan=-1.0))
    sc = capi.Scorer(n, capi.make_params(**kw))
    sc.score_cohort(dev, capi.row_descs(co["beta"], co["eaf"], None, co["rie"]), 0,
                    capi.MODE_FUSED if mode == "fused" else capi.MODE_TWOPASS)
    stats = sc.flush()
    scores, nloci = sc.finish(0.0)
    sc.close()
    dev.close()
    ref_scores, ref_stats, ref_nloci = oracle_ds(co, kw, 0.0)
    assert nloci == ref_nloci
    if m >= 100:
        assert 0 < sum(1 for s in ref_stats if s[4] == 4) < m     # both branches exercised
    assert_ds_stats(stats, ref_stats)
    assert rel_err(scores, ref_scores, co["beta"], max(nloci, 1)) <= REL_TOL


# ---- NPS_FMT_DS16: the same dosages in 2 bytes per genotype (lossless for decimals with at most four places)
def make_ds16_cohort(n, m, seed, rng):
    co = make_ds_cohort(n, m, seed, rng)
    co["ds"] = refcpu.synth_rows_ds16(n, 0, m, seed, co["th"], co["tm"], co["tmi"])
    return co


def same_floats(a, b):
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a, nan=-1.0).view(np.uint32),
                                                                        np.nan_to_num(b, nan=-1.0).view(np.uint32))


@pytest.mark.parametrize("pk", range(len(PARAM_GRID)))
@pytest.mark.parametrize("shape", [(1, 1), (100, 3), (4099, 129), (7681, 11), (30000, 500), (15361, 2001)])
def test_ds16_resident_vs_oracle(shape, pk):
    """a NPS_FMT_DS16 cohort filled by the generator: the rows that come back are the oracle generator's float32 values bit
    for bit (the format is lossless), and the single-read kernel scores them as the oracle scores those floats -- same
    bars as the float32 cohort (statistics exact but for the float64 dosage sum, scores 1e-6 relative)"""
    n, m = shape
    if pk and shape not in ((4099, 129), (30000, 500)):
        pytest.skip("all imputation modes on two shapes")
    rng = np.random.default_rng(n + 5 * m + pk)
    co = make_ds16_cohort(n, m, 2026, rng)
    kw = PARAM_GRID[pk] if pk else dict(imp_locus="ps", imp_missing="homref", imp_sample="int_ps", maxmis=0.05, mincs=100)
    dev = capi.Cohort(n, m, fmt=capi.FMT_DS16)
    assert dev.fmt == capi.FMT_DS16 and dev.row_stride % 128 == 0 and dev.row_stride < 2 * n + 256
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    assert same_floats(dev.download(0, m), co["ds"])
    sc = capi.Scorer(n, capi.make_params(**kw))
    sc.score_cohort(dev, capi.row_descs(co["beta"], co["eaf"], None, co["rie"]), 0, capi.MODE_AUTO)
    stats = sc.flush()
    scores, nloci = sc.finish(0.25)
    sc.close()
    ref_scores, ref_stats, ref_nloci = oracle_ds(co, kw, 0.25)
    assert nloci == ref_nloci
    assert_ds_stats(stats, ref_stats)
    assert np.array_equal(np.isnan(scores), np.isnan(ref_scores))
    assert rel_err(scores, ref_scores, co["beta"], max(nloci, 1)) <= REL_TOL
    # the float32 cohort holding the very same values gives the same scores (same kernel, same arithmetic; the samples of a
    # thread are laid out differently, so the dosage sums -- and with them an imputed value -- may differ in the last bits)
    d32 = capi.Cohort(n, m, fmt=capi.FMT_DS32)
    d32.upload(0, co["ds"])
    sc = capi.Scorer(n, capi.make_params(**kw))
    sc.score_cohort(d32, capi.row_descs(co["beta"], co["eaf"], None, co["rie"]), 0, capi.MODE_FUSED)
    s32, n32 = sc.finish(0.25)
    sc.close()
    d32.close()
    dev.close()
    assert n32 == nloci
    ok = ~np.isnan(s32)
    assert np.allclose(scores[ok], s32[ok], rtol=1e-11, atol=1e-15)


def test_ds16_every_code_round_trips():
    """the device's value of code k (ds16_value: three float32 operations) is the parser's float32 of k / 10^4 for EVERY k:
    one row holding all 20 001 decimals goes up (the packer compares its own value of k with the float it was given, bit for
    bit, and would refuse the row) and comes back (the unpacker writes its value of k) unchanged; a cohort of that row is
    scored by the single-read kernel with beta = 1 and must give each sample its own dosage back"""
    n = 20001
    row = np.array(["%.4f" % (k / 10000.0) for k in range(n)], dtype=np.float32)[None, :]
    dev = capi.Cohort(n, 1, fmt=capi.FMT_DS16)
    dev.upload(0, row)
    assert same_floats(dev.download(0, 1), row)
    sc = capi.Scorer(n, capi.make_params())
    sc.score_cohort(dev, capi.row_descs(np.ones(1), np.full(1, 0.2)), 0, capi.MODE_FUSED)
    s, nl = sc.finish(0.0)
    sc.close()
    dev.close()
    assert nl == 1 and np.array_equal(s, row[0].astype(np.float64) / 2.0)


def test_ds16_upload_is_lossless_or_refused():
    """float32 rows of decimals with one to four places round-trip bit for bit; a row with any other value (a binary
    fraction that is no such decimal, a value above 2, a negative zero) is refused and named, never rounded"""
    n, m = 5000, 12
    rng = np.random.default_rng(16)
    rows = np.empty((m, n), dtype=np.float32)
    for j in range(m):
        d = 1 + j % 4
        k = rng.integers(0, 2 * 10 ** d + 1, n)
        rows[j] = np.array(["%.*f" % (d, v / 10 ** d) for v in k], dtype=np.float32)   # (what a VCF parser makes of the text)
    rows[rng.uniform(size=rows.shape) < 0.05] = np.nan
    dev = capi.Cohort(n, m, fmt=capi.FMT_DS16)
    dev.upload(0, rows)
    assert same_floats(dev.download(0, m), rows)
    for bad in (np.float32(1.0) / np.float32(3.0), np.float32(2.0001), np.float32(-0.0), np.float32(0.12345)):
        r2 = rows[3:5].copy()
        r2[1, 777] = bad
        with pytest.raises(capi.NpsError) as ei:
            dev.upload(3, r2)
        assert ei.value.status == capi.E_UNSUPPORTED and "row 4" in str(ei.value)
    dev.upload(3, rows[3:5])
    assert same_floats(dev.download(0, m), rows)
    # scoring such a cohort in two reads is refused, not done by another path
    sc = capi.Scorer(n, capi.make_params())
    with pytest.raises(capi.NpsError) as ei:
        sc.score_cohort(dev, capi.row_descs(np.ones(m), np.full(m, 0.2)), 0, capi.MODE_TWOPASS)
    assert ei.value.status == capi.E_UNSUPPORTED
    sc.close()
    dev.close()


def test_ds_and_gt_rows_mixed_in_one_score():
    n = 777
    rng = np.random.default_rng(4)
    gt = make_cohort(n, 12, 8, rng, force_missing_rows=False)
    dsc = make_ds_cohort(n, 9, 9, rng)
    kw = PARAM_GRID[0]
    sc = capi.Scorer(n, capi.make_params(**kw))
    ref = refcpu.RefScorer(n, refcpu.make_params(**kw))
    betas = []
    for j in range(12):
        sc.push_packed(gt["codes"][j], gt["rie"][j], gt["beta"][j], gt["eaf"][j])
        ref.row_gt(refcpu.codes_to_gt(gt["codes"][j], n), 2, 1, bool(gt["rie"][j]), gt["beta"][j], gt["eaf"][j])
        betas.append(gt["beta"][j])
        if j < 9:
            sc.push_ds(dsc["ds"][j], dsc["rie"][j], dsc["beta"][j], dsc["eaf"][j])
            ref.row_ds(dsc["ds"][j], bool(dsc["rie"][j]), dsc["beta"][j], dsc["eaf"][j])
            betas.append(dsc["beta"][j])
    stats = sc.flush()
    scores, nloci = sc.finish(0.0)
    sc.close()
    ref_scores, ref_nloci = ref.finish(0.0)
    assert nloci == ref_nloci and len(stats) == 21
    assert [int(s["nmissing"]) for s in stats] == [int(s[1]) for s in ref.stats]
    assert rel_err(scores, ref_scores, betas, max(nloci, 1)) <= REL_TOL


def test_polyploid_gt_rows():
    """ploidy 3 and 4: dosages up to the ploidy (nimpress.nim:385-390 counts every allele); decoded on
    the device into a float dosage row, mixed with diploid rows in one score"""
    n = 301
    rng = np.random.default_rng(33)
    kw = dict(imp_locus="ps", imp_missing="homref", imp_sample="int_ps", maxmis=0.2, mincs=10)
    sc = capi.Scorer(n, capi.make_params(**kw))
    ref = refcpu.RefScorer(n, refcpu.make_params(**kw))
    betas = []
    saw_high = False
    for j, ploidy in enumerate([3, 2, 4, 3, 2, 4, 1]):
        eaidx = int(rng.integers(0, 3))
        alle = rng.integers(-1, 3, size=(n, ploidy))
        saw_high |= bool(((alle == eaidx).sum(axis=1) > 2).any())
        alle[rng.uniform(size=n) < (0.5 if j == 3 else 0.04)] = -1
        gts = (((alle + 1) << 1) | rng.integers(0, 2, size=(n, ploidy))).astype(np.int32)
        beta, eaf = float(rng.normal(0, 0.1)), float(rng.uniform(0.1, 0.5))
        sc.push_gt(gts.ravel(), ploidy, eaidx, eaidx == 0, beta, eaf)
        ref.row_gt(gts.ravel(), ploidy, eaidx, eaidx == 0, beta, eaf)
        betas.append(beta)
    stats = sc.flush()
    scores, nloci = sc.finish(0.2)
    sc.close()
    ref_scores, ref_nloci = ref.finish(0.2)
    assert nloci == ref_nloci
    assert_stats_equal(stats, ref.stats)
    assert saw_high                                             # dosages above 2 occurred
    assert rel_err(scores, ref_scores, betas, max(nloci, 1)) <= REL_TOL


@pytest.mark.parametrize("pk", range(len(PARAM_GRID)))
def test_ds_fused_all_imputation_modes(pk):
    """every imputation combination through the single-read DS kernel, with a row offset"""
    n, m, row0 = 9000, 64, 6
    rng = np.random.default_rng(500 + pk)
    co = make_ds_cohort(n, m + row0, 4242, rng)
    kw = PARAM_GRID[pk]
    dev = capi.Cohort(n, m + row0, fmt=capi.FMT_DS32)
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    sc = capi.Scorer(n, capi.make_params(**kw))
    sc.score_cohort(dev, capi.row_descs(co["beta"][row0:], co["eaf"][row0:], None, co["rie"][row0:]),
                    row0, capi.MODE_FUSED)
    stats = sc.flush()
    scores, nloci = sc.finish(-0.5)
    sc.close()
    dev.close()
    sub = dict(co, m=m, ds=co["ds"][row0:], beta=co["beta"][row0:], eaf=co["eaf"][row0:],
               rie=co["rie"][row0:])
    ref_scores, ref_stats, ref_nloci = oracle_ds(sub, kw, -0.5)
    assert nloci == ref_nloci
    assert_ds_stats(stats, ref_stats)
    assert rel_err(scores, ref_scores, sub["beta"], max(nloci, 1)) <= REL_TOL


def test_ds_fused_range_and_small_sums():
    """The single-read DS kernel hands the slices' dosage sums over as fixed-point integers and needs 0 <= DS <= 2 (the
    FORMAT/DS convention).  nps_cohort_upload checks the rows it receives: while the cohort holds a value outside,
    NPS_MODE_FUSED is refused (NPS_E_INVAL) and NPS_MODE_AUTO / NPS_MODE_TWOPASS score it with the two-pass kernels,
    which take any value; uploading the row again with good values lifts the mark.  Rows whose whole dosage sum is
    tiny (one sample at 0.001, the rest 0; all samples at 2^-20) keep their sum to 1e-9 relative."""
    n, m = 20000, 8
    rng = np.random.default_rng(5)
    ds = np.round(rng.uniform(0.0, 2.0, size=(m, n)), 3).astype(np.float32)
    ds[rng.uniform(size=(m, n)) < 0.02] = np.nan
    ds[1, :] = 0.0
    ds[1, 777] = 0.001
    ds[2, :] = np.float32(2.0 ** -20)
    ds[3, :] = 2.0
    beta = np.round(rng.normal(0, 0.02, m), 4)
    eaf = np.full(m, 0.25)
    rie = np.zeros(m, np.int32)
    kw = dict(imp_locus="ps", imp_missing="homref", imp_sample="int_ps", maxmis=0.05, mincs=100)
    co = dict(n=n, m=m, ds=ds, beta=beta, eaf=eaf, rie=rie)
    dev = capi.Cohort(n, m, fmt=capi.FMT_DS32)
    dev.upload(0, ds)
    sc = capi.Scorer(n, capi.make_params(**kw))
    sc.score_cohort(dev, capi.row_descs(beta, eaf, None, rie), 0, capi.MODE_FUSED)
    stats = sc.flush()
    scores, nloci = sc.finish(0.0)
    ref_scores, ref_stats, ref_nloci = oracle_ds(co, kw, 0.0)
    assert nloci == ref_nloci
    assert_ds_stats(stats, ref_stats)
    assert rel_err(scores, ref_scores, beta, max(nloci, 1)) <= REL_TOL
    # one value out of range
    bad = ds.copy()
    bad[5, 12345] = 2.5
    dev.upload(0, bad)
    sc.reset()
    with pytest.raises(capi.NpsError) as ei:
        sc.score_cohort(dev, capi.row_descs(beta, eaf, None, rie), 0, capi.MODE_FUSED)
    assert ei.value.status == -1 and "[0, 2]" in str(ei.value)
    co_bad = dict(co, ds=bad)
    ref_bad, _, ref_nloci_bad = oracle_ds(co_bad, kw, 0.0)
    for mode in (capi.MODE_AUTO, capi.MODE_TWOPASS):    # the refused call left the context usable
        sc.reset()
        sc.score_cohort(dev, capi.row_descs(beta, eaf, None, rie), 0, mode)
        two, nloci2 = sc.finish(0.0)
        assert nloci2 == ref_nloci_bad
        assert rel_err(two, ref_bad, beta, max(nloci2, 1)) <= REL_TOL
    dev.upload(5, ds[5:6])                              # the row again, in range: single read again
    sc.reset()
    sc.score_cohort(dev, capi.row_descs(beta, eaf, None, rie), 0, capi.MODE_FUSED)
    again, nloci3 = sc.finish(0.0)
    assert nloci3 == ref_nloci and np.array_equal(again, scores, equal_nan=True)
    sc.close()
    dev.close()


def test_ds_large_fused_equals_twopass_and_scaling():
    """BASELINE.json configs[4] shape (200 000 samples, FORMAT/DS float32, missing rate U(0,0.10) so about
    half the rows exceed --maxmis) on 65 536 rows (52 GB): the single-read fused DS kernel and the
    two-pass DS kernels must agree for every sample and every row decision; scaling beta by 2 scales
    the sums exactly; one row's tally is recounted by the oracle's generator."""
    import torch
    n, m, seed = 200_000, 65_536, 20250105
    from conftest import need_free_hbm
    need_free_hbm(70)
    rng = np.random.default_rng(seed)
    beta = np.round(rng.normal(0.0, 0.02, m), 4)
    eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
    miss = rng.uniform(0.0, 0.10, m)
    rie = (rng.uniform(size=m) < 0.3).astype(np.int32)
    th, tm, tmi = refcpu.hwe_thresholds(eaf, miss)
    dev = capi.Cohort(n, m, fmt=capi.FMT_DS32)
    for r0 in range(0, m, 1 << 14):
        r1 = min(m, r0 + (1 << 14))
        dev.synth(r0, seed, th[r0:r1], tm[r0:r1], tmi[r0:r1])

    def run(b, mode):
        sc = capi.Scorer(n, capi.make_params(imp_locus="ps"))
        sc.score_cohort(dev, capi.row_descs(b, eaf, None, rie), 0, mode)
        stats = sc.flush()
        part = torch.empty(n, dtype=torch.float64, device="cuda")
        nloci = sc.partial_device(part.data_ptr())
        sc.close()
        return part, nloci, stats

    fused, nloci, st_f = run(beta, capi.MODE_FUSED)
    twop, nloci2, st_t = run(beta, capi.MODE_TWOPASS)
    assert nloci == nloci2 == m
    for k in ("ngenotyped", "nmissing", "used", "reason"):
        assert np.array_equal(st_f[k], st_t[k]), k
    assert np.max(np.abs(st_f["neffect"] - st_t["neffect"]) / np.maximum(1.0, np.abs(st_t["neffect"]))) <= 1e-9
    over = int((st_f["reason"] == capi.REASON_MAXMIS).sum())
    assert 0.35 * m < over < 0.65 * m
    scale = float(np.sum(np.abs(beta)))
    assert float((fused - twop).abs().max()) <= 1e-9 * scale
    doubled, _, _ = run(2.0 * beta, capi.MODE_FUSED)
    assert bool(torch.equal(doubled, 2.0 * fused))
    j = 4242
    row = refcpu.synth_rows_ds(n, j, 1, seed, th[j:j + 1], tm[j:j + 1], tmi[j:j + 1])[0]
    assert int(np.isnan(row).sum()) == int(st_f["nmissing"][j])
    dose = np.where(rie[j], 2.0 - row.astype(np.float64), row.astype(np.float64))
    assert abs(float(np.nansum(dose)) - float(st_f["neffect"][j])) <= 1e-9 * float(np.nansum(dose))
    dev.close()


def test_ds_config5_resident_chunks_vs_oracle_subset():
    """BASELINE.json configs[4] (200 000 samples, FORMAT/DS float32, 5 % mean missingness, --imp-locus=ps)
    at the largest size one GPU holds (300 000 rows = 240 GB; the full 2 000 000 rows are scored by
    bench.py in chunks of exactly this size):
      * samples from every slice of the DS kernel's persistent grid scored over ALL rows by
        oracle/refcpu.c (ref_score_subset), whole-row tallies of 300 random rows recounted by the oracle;
      * the rows scored as two resident chunks of half the size, each regenerated into the same buffer
        (nps_cohort_synth_rows) with the sums carried inside the context == the one resident run;
      * scores(2 beta) == 2 scores(beta) bit for bit on the chunked path."""
    import torch
    n, m, seed = 200_000, 300_000, 20250105
    from conftest import need_free_hbm
    need_free_hbm(250)
    rng = np.random.default_rng(seed)
    beta = np.round(rng.normal(0.0, 0.02, m), 4)
    eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
    miss = rng.uniform(0.0, 0.10, m)
    rie = (rng.uniform(size=m) < 0.3).astype(np.int32)
    th, tm, tmi = refcpu.hwe_thresholds(eaf, miss)
    prm = dict(imp_locus="ps")

    def fill(dev, a, b):
        for x in range(a, b, 1 << 14):
            y = min(b, x + (1 << 14))
            dev.synth_at(x - a, x, seed, th[x:y], tm[x:y], tmi[x:y])

    def chunked(bb, bounds, want_stats=False):
        dev = capi.Cohort(n, max(b - a for a, b in bounds), fmt=capi.FMT_DS32)
        sc = capi.Scorer(n, capi.make_params(**prm))
        stats = []
        for a, b in bounds:
            fill(dev, a, b)
            sc.score_cohort(dev, capi.row_descs(bb[a:b], eaf[a:b], None, rie[a:b]), 0, capi.MODE_AUTO)
            if want_stats:
                stats.append(sc.flush())
        part = torch.empty(n, dtype=torch.float64, device="cuda")
        nloci = sc.partial_device(part.data_ptr())
        geo = sc.fused_geometry(bounds[0][1] - bounds[0][0], capi.FMT_DS32)
        sc.close()
        dev.close()
        return part.cpu().numpy(), nloci, (np.concatenate(stats) if want_stats else None), geo

    whole, nloci, stats, geo = chunked(beta, [(0, m)], want_stats=True)
    assert nloci == m == int(stats["used"].sum())
    over = int((stats["reason"] == capi.REASON_MAXMIS).sum())
    assert 0.35 * m < over < 0.65 * m
    slices, teams, sps = geo
    assert slices * teams > 200 and sps % 8 == 0, geo   # the single-read DS kernel is in use
    # whole-row recount of 300 random rows (+ first and last)
    rows = np.unique(np.concatenate([np.random.default_rng(2).choice(m, 300, replace=False), [0, m - 1]])).astype(np.uint64)
    ri = rows.astype(np.int64)
    g, ms, ne = refcpu.tally_synth_rows(rows, n, seed, th[ri], tm[ri], tmi[ri], rie=rie[ri], is_ds=True)
    assert np.array_equal(g, stats["ngenotyped"][ri].astype(np.float64))
    assert np.array_equal(ms, stats["nmissing"][ri].astype(np.float64))
    assert np.allclose(ne, stats["neffect"][ri], rtol=1e-9, atol=0.0)
    # every slice: first / lane 31 / lane 63 / middle / last thread (8 samples each), and the last sample
    units, per_slice = (n + 7) // 8, sps // 8
    cols = {units - 1}
    for p_ in range(slices):
        first, last = p_ * per_slice, min(units, (p_ + 1) * per_slice) - 1
        cols.update(c for c in (first, first + 31, first + 63, (first + last) // 2, last) if first <= c <= last)
    assert len({c // per_slice for c in cols}) == slices
    samples = np.concatenate([np.arange(c * 8, min(n, (c + 1) * 8)) for c in sorted(cols)]).astype(np.uint64)
    sums, ref_nloci = refcpu.score_subset(samples, n, 0, seed, th, tm, tmi, beta, eaf, rie,
                                          stats["ngenotyped"].astype(np.float64),
                                          stats["nmissing"].astype(np.float64), stats["neffect"],
                                          refcpu.make_params(**prm), is_ds=True)
    assert ref_nloci == m
    scale = float(np.sum(np.abs(beta)))
    got = whole[samples.astype(np.int64)]
    # (the device's float64 dosage sums differ from the oracle's sequential sums in the last bits, and
    # with them the imputed value of a row: bar 1e-9 of sum|beta|, far inside the 1e-6 of the north star)
    assert np.max(np.abs(got - sums)) <= 1e-9 * scale
    assert np.max(np.abs(got - sums) / np.maximum(np.abs(sums), 1e-12 * scale)) <= 1e-6
    # two resident chunks == one resident run; exact scaling
    h = m // 2
    two, nloci2, _, _ = chunked(beta, [(0, h), (h, m)])
    assert nloci2 == m
    assert np.max(np.abs(two - whole)) <= 1e-12 * scale
    dbl, _, _, _ = chunked(2.0 * beta, [(0, h), (h, m)])
    assert np.array_equal(dbl, 2.0 * two)


def test_row_sharded_partial_sums_and_normalise():
    """one score evaluated as two row blocks (what two GPUs would hold), un-normalised sums added and
    normalised in place by the library == the unsharded evaluation"""
    import torch
    from nimpress_amd import multi
    n, m = 20011, 203
    rng = np.random.default_rng(77)
    co = make_cohort(n, m, 31337, rng)
    kw = PARAM_GRID[0]
    dev = capi.Cohort(n, m)
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    descs = capi.row_descs(co["beta"], co["eaf"], None, co["rie"])
    whole = capi.Scorer(n, capi.make_params(**kw))
    whole.score_cohort(dev, descs)
    ref_scores, ref_nloci = whole.finish(0.125)
    whole.close()
    total = torch.zeros(n, dtype=torch.float64, device="cuda")
    nloci = 0
    sc = capi.Scorer(n, capi.make_params(**kw))
    for rank in range(2):
        r0, r1 = multi.shard_rows(m, 2, rank)
        sc.reset()
        sc.score_cohort(dev, descs[r0:r1], r0)
        part = torch.empty(n, dtype=torch.float64, device="cuda")
        nloci += sc.partial_device(part.data_ptr())
        total += part
    torch.cuda.synchronize()
    sc.normalize_device(total.data_ptr(), nloci, 0.125)
    got = total.cpu().numpy()
    sc.close()
    dev.close()
    assert nloci == ref_nloci
    assert rel_err(got, ref_scores, co["beta"], max(nloci, 1)) <= 1e-12


@pytest.mark.parametrize("dtype", [np.int8, np.int16, np.int32])
def test_push_gt_raw_typed_vectors(dtype):
    """the typed GT vector of a BCF record (int8 / int16 with their own end-of-vector and missing
    values) gives the same rows as the widened bcf_get_genotypes buffer"""
    n = 1234
    rng = np.random.default_rng(8)
    kw = PARAM_GRID[0]
    info = np.iinfo(dtype)
    vend, vmiss = info.min + 1, info.min           # 0x81 / 0x8001 / 0x80000001 and 0x80 / ...
    sc = capi.Scorer(n, capi.make_params(**kw))
    ref = refcpu.RefScorer(n, refcpu.make_params(**kw))
    betas = []
    for j, ploidy in enumerate([2, 2, 1, 3, 2]):
        eaidx = int(rng.integers(0, 3))
        alle = rng.integers(-1, 3, size=(n, ploidy))
        typed = (((alle + 1) << 1) | rng.integers(0, 2, size=(n, ploidy))).astype(dtype)
        wide = typed.astype(np.int32)
        if ploidy >= 2:                             # haploid calls in a diploid record, typed missing
            hap = rng.uniform(size=n) < 0.1
            typed[hap, ploidy - 1] = vend
            wide[hap, ploidy - 1] = -2147483647     # bcf_int32_vector_end
            tm = rng.uniform(size=n) < 0.02
            typed[tm, 0] = vmiss
            wide[tm, 0] = -2147483648               # bcf_int32_missing
        beta, eaf = float(rng.normal(0, 0.1)), float(rng.uniform(0.1, 0.5))
        sc.push_gt_raw(typed.ravel(), ploidy, eaidx, eaidx == 0, beta, eaf)
        ref.row_gt(wide.ravel(), ploidy, eaidx, eaidx == 0, beta, eaf)
        betas.append(beta)
    stats = sc.flush()
    scores, nloci = sc.finish(0.0)
    sc.close()
    ref_scores, ref_nloci = ref.finish(0.0)
    assert nloci == ref_nloci
    assert_stats_equal(stats, ref.stats)
    assert rel_err(scores, ref_scores, betas, max(nloci, 1)) <= REL_TOL


# ------------------------------------------------------------------------------------------
# PLINK 1 .bed rows (values 0 hom A1, 1 missing, 2 het, 3 hom A2; 4 samples per byte)
def codes_to_bed(codes_row, n, effect_is_a1):
    """native 2-bit codes (dosage of the effect allele) -> the .bed bytes that encode the same calls"""
    c = (codes_row[np.arange(n) >> 4] >> ((np.arange(n) & 15) * 2)) & 3     # 0 d0, 1 d1, 3 d2, 2 missing
    if effect_is_a1:
        v = np.select([c == 0, c == 1, c == 3], [3, 2, 0], 1)              # dosage counts A1
    else:
        v = np.select([c == 0, c == 1, c == 3], [0, 2, 3], 1)              # dosage counts A2
    v = np.concatenate([v, np.zeros((-n) % 4, dtype=v.dtype)]).reshape(-1, 4)
    return (v[:, 0] | (v[:, 1] << 2) | (v[:, 2] << 4) | (v[:, 3] << 6)).astype(np.uint8)


def codes_to_pgen(codes_row, n, effect_is_ref):
    """native 2-bit codes (dosage of the effect allele) -> fixed-width .pgen record bytes (code = ALT count, 3 = missing)"""
    c = (codes_row[np.arange(n) >> 4] >> ((np.arange(n) & 15) * 2)) & 3     # 0 d0, 1 d1, 3 d2, 2 missing
    d = np.select([c == 0, c == 1, c == 3], [0, 1, 2], -1)
    v = np.where(d < 0, 3, (2 - d) if effect_is_ref else d)
    v = np.concatenate([v, np.zeros((-n) % 4, dtype=v.dtype)]).reshape(-1, 4)
    return (v[:, 0] | (v[:, 1] << 2) | (v[:, 2] << 4) | (v[:, 3] << 6)).astype(np.uint8)


@pytest.mark.parametrize("n", [1, 7, 16, 1001, 4097])
def test_pgen_code_maps_streamed_and_resident(n):
    """NPS_MAP_PGEN_ALT / NPS_MAP_PGEN_REF through nps_push_bed and nps_cohort_upload_bed: a fixed-width .pgen record
    (2-bit code = number of ALT alleles, 3 = missing) is the .bed path with another code map; rows come back as the
    native codes they encode and score like them"""
    m = 23
    rng = np.random.default_rng(n + 77)
    co = make_cohort(n, m, 556, rng)
    is_ref = rng.integers(0, 2, m).astype(np.uint8)
    rows = np.stack([codes_to_pgen(co["codes"][j], n, is_ref[j]) for j in range(m)])
    maps = (2 + is_ref).astype(np.uint8)                  # NPS_MAP_PGEN_ALT = 2, NPS_MAP_PGEN_REF = 3
    kw = PARAM_GRID[0]
    ref_scores, ref_stats, ref_nloci = refcpu.score_packed(
        co["codes"], n, np.zeros(m, np.int32), co["rie"], co["beta"], co["eaf"], refcpu.make_params(**kw), 0.5)
    sc = capi.Scorer(n, capi.make_params(**kw))
    for j in range(m):
        sc.push_bed(rows[j], int(maps[j]), co["rie"][j], co["beta"][j], co["eaf"][j])
    stats = sc.flush()
    scores, nloci = sc.finish(0.5)
    assert nloci == ref_nloci
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    assert rel_err(scores, ref_scores, co["beta"], max(nloci, 1)) <= REL_TOL
    dev = capi.Cohort(n, m)
    dev.upload_bed(0, rows, maps)
    assert np.array_equal(dev.download(0, m), co["codes"][:, : (n + 15) // 16])
    with pytest.raises(capi.NpsError):
        sc.push_bed(rows[0], 4, 0, 0.1, 0.2)              # not a code map
    sc.close()
    dev.close()


@pytest.mark.parametrize("n", [1, 7, 16, 1001, 4097])
def test_plink_bed_rows_streamed_and_resident(n):
    m = 23
    rng = np.random.default_rng(n)
    co = make_cohort(n, m, 555, rng)
    a1 = rng.integers(0, 2, m).astype(np.uint8)
    bed = np.stack([codes_to_bed(co["codes"][j], n, a1[j]) for j in range(m)])
    kw = PARAM_GRID[0]
    ref_scores, ref_stats, ref_nloci = refcpu.score_packed(
        co["codes"], n, np.zeros(m, np.int32), co["rie"], co["beta"], co["eaf"], refcpu.make_params(**kw), 0.5)
    # streamed
    sc = capi.Scorer(n, capi.make_params(**kw))
    for j in range(m):
        sc.push_bed(bed[j], a1[j], co["rie"][j], co["beta"][j], co["eaf"][j])
    stats = sc.flush()
    scores, nloci = sc.finish(0.5)
    assert nloci == ref_nloci
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    assert rel_err(scores, ref_scores, co["beta"], max(nloci, 1)) <= REL_TOL
    # resident: .bed rows -> cohort (recode + interleave on the device) == native upload
    dev = capi.Cohort(n, m)
    dev.upload_bed(0, bed, a1)
    assert np.array_equal(dev.download(0, m), co["codes"][:, : (n + 15) // 16])
    sc.reset()
    sc.score_cohort(dev, capi.row_descs(co["beta"], co["eaf"], None, co["rie"]))
    stats = sc.flush()
    scores, nloci = sc.finish(0.5)
    sc.close()
    dev.close()
    assert nloci == ref_nloci
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    assert rel_err(scores, ref_scores, co["beta"], max(nloci, 1)) <= REL_TOL
