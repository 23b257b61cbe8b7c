"""Pins the CPU oracle (oracle/refcpu.c + oracle/refcpu.py) to the reference's own golden vectors.

* 13 set1 cases       -- nimpress tests/test_set1.nim:36-190 (tol 1e-4 abs, NaN positions exact)
* 87 stats KATs       -- nimpress tests/test_stats.nim:21-139 (rel 1e-5 / abs 1e-9)
* PLINK cross-checks  -- tests/set1.plink190.result (pinned), set1.plink200.result (5 of 6; S3 is a
                         known fixture inconsistency)
* output text format  -- scores/*_nimpress_res.txt ("%.16g" + ".0")
"""
import json
import math
import os

import numpy as np
import pytest

from oracle import refcpu


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def set1(golden_dir):
    score = refcpu.read_score_file(os.path.join(golden_dir, "set1.score"))
    vcf = refcpu.read_vcf(os.path.join(golden_dir, "set1.vcf.gz"))
    bed = refcpu.read_bed(os.path.join(golden_dir, "set1.bed"))
    return score, vcf, bed


def check_floats(x, target, tol=1e-4):
    # tests/test_set1.nim:14-22
    assert len(x) == len(target)
    for xi, ti in zip(x, target):
        t_nan = ti is None or (isinstance(ti, float) and math.isnan(ti))
        assert t_nan == bool(np.isnan(xi)), (x, target)
        if not t_nan:
            assert abs(ti - xi) <= tol, (x, target)


def test_set1_fixture_shape(set1):
    score, vcf, bed = set1
    assert vcf.samples == ["S1", "S2", "S3", "S4", "S5", "S6"]
    assert len(vcf.records) == 7
    assert score.offset == 0.123 and len(score.entries) == 6
    assert [e.pos for e in score.entries] == [100, 150, 200, 300, 400, 500]
    assert sorted(bed) == ["1", "2", "3"]


@pytest.mark.parametrize("idx", range(13))
def test_set1_case(golden_dir, set1, idx):
    score, vcf, bed = set1
    case = _load(golden_dir, "set1_cases.json")["cases"][idx]
    scores, nloci, stats = refcpu.compute_polygenic_scores(
        score, vcf, case["restrict_to_covered"], bed, case["imp_locus"], case["imp_missing"],
        case["imp_sample"], case["maxmis"], case["mincs"], case["ignore_filter"])
    check_floats(scores, case["expected"])


def test_set1_plink190(golden_dir, set1):
    # tests/test_set1.nim:180-190: offset 0.123 + PLINK 1.90 SCORE column
    score, vcf, bed = set1
    plink = [float(l.split()[5]) for l in
             open(os.path.join(golden_dir, "set1.plink190.result")).read().splitlines()[1:]]
    scores, nloci, _ = refcpu.compute_polygenic_scores(score, vcf, False, bed, "ignore", "ignore",
                                                       "int_ps", 1.0, 0, True)
    assert nloci == 5
    check_floats(scores, [0.123 + p for p in plink])


@pytest.fixture(scope="module")
def set1_split(golden_dir):
    """tests/set1.plink.vcf.gz: the reference's own `bcftools norm -m-` split of set1 (two records at 1:300 with the
    same REF, ##contig lines, LF endings) -- the file set1.plink190.result was computed from
    (tests/set1.plink190.result.txt:1-5)"""
    score = refcpu.read_score_file(os.path.join(golden_dir, "set1.score"))
    vcf = refcpu.read_vcf(os.path.join(golden_dir, "set1.plink.vcf.gz"))
    bed = refcpu.read_bed(os.path.join(golden_dir, "set1.bed"))
    return score, vcf, bed


def test_split_fixture_shape_and_find_variant(set1_split):
    score, vcf, _ = set1_split
    assert vcf.samples == ["S1", "S2", "S3", "S4", "S5", "S6"] and len(vcf.records) == 8
    at300 = [r for r in vcf.records if r.pos == 300]
    assert [(r.ref, r.alts) for r in at300] == [("GA", ["T"]), ("GA", ["CT"])]
    # findVariant (nimpress.nim:359-364) skips the first REF-matching record, whose ALT lacks the effect allele
    e = [x for x in score.entries if x.pos == 300][0]
    assert (e.refseq, e.easeq) == ("GA", "CT")
    rec = refcpu.find_variant(vcf, e)
    assert rec is at300[1] and rec.gts.tolist() == [2, 2, 4, 4, 2, 2, 2, 2, 0, 0, 2, 2]


@pytest.mark.parametrize("idx", range(13))
def test_set1_case_on_split_records(golden_dir, set1_split, idx):
    """the 13 golden vectors hold on the split file too: the score row 1:300 GA/CT takes the GA>CT record"""
    score, vcf, bed = set1_split
    case = _load(golden_dir, "set1_cases.json")["cases"][idx]
    scores, nloci, stats = refcpu.compute_polygenic_scores(
        score, vcf, case["restrict_to_covered"], bed, case["imp_locus"], case["imp_missing"],
        case["imp_sample"], case["maxmis"], case["mincs"], case["ignore_filter"])
    check_floats(scores, case["expected"])


def test_set1_plink190_on_the_file_plink_read(golden_dir, set1_split):
    score, vcf, bed = set1_split
    plink = [float(l.split()[5]) for l in
             open(os.path.join(golden_dir, "set1.plink190.result")).read().splitlines()[1:]]
    scores, nloci, _ = refcpu.compute_polygenic_scores(score, vcf, False, bed, "ignore", "ignore",
                                                       "int_ps", 1.0, 0, True)
    assert nloci == 5
    check_floats(scores, [0.123 + p for p in plink])


def test_set1_plink200_five_of_six(golden_dir, set1):
    # tests/test_set1.nim:207-216 (commented out in the reference).  S3 differs by 0.018 because
    # set1.plink.freq lists 0.95 for the ALT allele of 1:100 while set1.score gives it to REF.
    score, vcf, bed = set1
    plink = [float(l.split()[3]) for l in
             open(os.path.join(golden_dir, "set1.plink200.result")).read().splitlines()[1:]]
    scores, _, _ = refcpu.compute_polygenic_scores(score, vcf, False, bed, "ps", "ignore", "ps",
                                                   1.0, 0, True)
    d = np.abs((scores - 0.123) - np.array(plink))
    assert (d < 1e-4).tolist() == [True, True, False, True, True, True]
    assert abs(d[2] - 0.018) < 1e-9


def test_set1_cli_defaults_derived(set1):
    # derived: CLI defaults -> every row locus-imputed, all scores 0.1545
    score, vcf, bed = set1
    scores, nloci, stats = refcpu.compute_polygenic_scores(score, vcf, False, bed, "ps", "homref",
                                                           "int_ps", 0.05, 100, False)
    assert nloci == 6
    assert np.allclose(scores, 0.1545, atol=1e-12)
    # per-row tallies (nmissing, neffect) of the genotyped rows
    got = [(int(s[1]), int(s[2])) for s in stats if s[4] in (0, 4)]
    assert got == [(1, 7), (1, 2), (5, 0), (1, 7)]


def test_set1_case13_trace(set1):
    # SURVEY.md appendix B worked trace: internal imputation values per row
    score, vcf, bed = set1
    _, nloci, stats = refcpu.compute_polygenic_scores(score, vcf, False, bed, "ignore", "ignore",
                                                      "int_ps", 1.0, 0, True)
    assert nloci == 5
    used = [s for s in stats if s[3]]
    assert [(int(s[0]), int(s[1]), int(s[2])) for s in used] == [
        (5, 1, 7), (3, 3, 3), (5, 1, 2), (1, 5, 0), (5, 1, 7)]
    assert [s[3] for s in stats] == [1, 1, 0, 1, 1, 1]


def test_stats_kats(golden_dir):
    d = _load(golden_dir, "stats_kats.json")
    fns = {"dbinom": refcpu.dbinom, "pbinom": refcpu.pbinom, "binom_test": refcpu.binom_test,
           "betai": refcpu.betai}
    n = 0
    for k in d["kats"]:
        val = fns[k["fn"]](*k["args"])
        tgt = k["expected"]
        if k["mode"] == "exact":
            assert val == tgt, k
        elif abs(tgt) < d["abs_tol"]:     # tests/test_stats.nim:10-17
            assert abs(val - tgt) < d["abs_tol"], k
        else:
            assert abs((val - tgt) / tgt) < d["rel_tol"], (k, val)
        n += 1
    assert n == 87


def test_output_format_pinned_by_reference_results(golden_dir):
    # every value in the reference's bundled result files re-prints identically
    rf = os.path.join(golden_dir, "result_format")
    n = 0
    for f in sorted(os.listdir(rf)):
        for line in open(os.path.join(rf, f)).read().splitlines():
            name, txt = line.split("\t")
            assert refcpu.format_score(float(txt)) == txt, (f, line)
            n += 1
    assert n == 3528


def test_raw_dosage_corner_cases():
    # nimpress.nim:385-390 with hts-nim value(): half-missing -> NaN, pad skipped, phase ignored
    L = refcpu.lib()
    import ctypes as C
    gts = np.array([2, 4,  5, 3,  0, 4,  4, 0,  4, -2147483647,  0, -2147483647,  4, 4], np.int32)
    raw = np.empty(7)
    L.ref_raw_dosages_gt(raw.ctypes.data_as(C.POINTER(C.c_double)),
                         gts.ctypes.data_as(C.POINTER(C.c_int32)), 7, 2, 1)
    assert raw[0] == 1 and raw[1] == 1 and np.isnan(raw[2]) and np.isnan(raw[3])
    assert raw[4] == 1 and np.isnan(raw[5]) and raw[6] == 2


def test_packed_matrix_path_equals_row_path():
    rng = np.random.default_rng(5)
    n, m = 37, 11
    eaf = rng.uniform(0.05, 0.5, m)
    miss = rng.uniform(0, 0.3, m)
    th, tm, tmi = refcpu.hwe_thresholds(eaf, miss)
    codes = refcpu.synth_rows(n, 0, m, 99, th, tm, tmi)
    beta = rng.normal(0, 0.1, m)
    kind = np.zeros(m, np.int32)
    rie = (rng.uniform(size=m) < 0.3).astype(np.int32)
    p = refcpu.make_params("ps", "homref", "int_ps", 0.2, 5)
    s1, st1, nl1 = refcpu.score_packed(codes, n, kind, rie, beta, eaf, p, 0.5)
    sc = refcpu.RefScorer(n, p)
    for j in range(m):
        sc.row_gt(refcpu.codes_to_gt(codes[j], n), 2, 1, bool(rie[j]), beta[j], eaf[j])
    s2, nl2 = sc.finish(0.5)
    assert nl1 == nl2
    assert np.array_equal(s1, s2, equal_nan=True)


# ------------------------------------------------------------------------------------------
# the full-size checker (ref_score_subset / ref_tally_synth_rows) against the literal whole-cohort path
@pytest.mark.parametrize("kw", [dict(), dict(imp_locus="ignore", imp_sample="int_fail", maxmis=0.03, mincs=10),
                                dict(imp_locus="homref", imp_sample="ps"), dict(imp_locus="fail", imp_sample="homref")])
def test_subset_scorer_equals_whole_cohort_path(kw):
    """Scoring a few samples with every row's whole-row tally handed in (how bench.py and the full-size GPU
    tests use the oracle at 500 000 x 1 000 000) gives bit for bit what the literal per-row path over the
    whole cohort gives for those samples."""
    n, m, seed = 3001, 257, 99
    rng = np.random.default_rng(5)
    eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
    miss = rng.uniform(0, 0.08, m)
    beta = np.round(rng.normal(0, 0.02, m), 4)
    th, tm, tmi = refcpu.hwe_thresholds(eaf, miss)
    prm = refcpu.make_params(**kw)
    codes = refcpu.synth_rows(n, 0, m, seed, th, tm, tmi)
    ref, st, nloci = refcpu.score_packed(codes, n, np.zeros(m, np.int32), np.zeros(m, np.int32), beta, eaf, prm, 0.0)
    g, ms, ne = refcpu.tally_synth_rows(np.arange(m), n, seed, th, tm, tmi)
    assert np.array_equal(g, st["ngenotyped"]) and np.array_equal(ms, st["nmissing"]) and np.array_equal(ne, st["neffect"])
    samples = np.array([0, 1, 15, 16, 17, 1500, 2999, 3000], dtype=np.uint64)
    sums, nl = refcpu.score_subset(samples, n, 0, seed, th, tm, tmi, beta, eaf, 0, g, ms, ne, prm)
    assert nl == nloci
    with np.errstate(invalid="ignore", divide="ignore"):
        got = sums / (2.0 * nl)
    assert np.array_equal(got, ref[samples.astype(int)], equal_nan=True)
    # FORMAT/DS rows, a third of them with the REF allele as effect allele
    rie = (np.arange(m) % 3 == 0).astype(np.int32)
    ds = refcpu.synth_rows_ds(n, 0, m, seed, th, tm, tmi)
    sc = refcpu.RefScorer(n, prm)
    for j in range(m):
        sc.row_ds(ds[j], bool(rie[j]), beta[j], eaf[j])
    ref, nloci = sc.finish(0.0)
    g, ms, ne = refcpu.tally_synth_rows(np.arange(m), n, seed, th, tm, tmi, rie=rie, is_ds=True)
    sums, nl = refcpu.score_subset(samples, n, 0, seed, th, tm, tmi, beta, eaf, rie, g, ms, ne, prm, is_ds=True)
    assert nl == nloci
    with np.errstate(invalid="ignore", divide="ignore"):
        got = sums / (2.0 * nl)
    assert np.array_equal(got, ref[samples.astype(int)], equal_nan=True)


def test_cpu_baseline_variants_agree():
    """the timed CPU-baseline loops of bench.py (with the binomTest call; split over all cores) compute the
    same scores as the plain literal loop"""
    n, rows, nd = 4000, 40, 4
    rng = np.random.default_rng(3)
    eaf = np.round(rng.uniform(0.05, 0.5, nd), 4)
    th, tm, tmi = refcpu.hwe_thresholds(eaf, np.array([0.0, 0.01, 0.02, 0.2]))
    codes = refcpu.synth_rows(n, 0, nd, 1, th, tm, tmi)
    gts = np.stack([refcpu.codes_to_gt(codes[j], n) for j in range(nd)])
    beta = np.round(rng.normal(0, 0.02, rows), 4)
    e = np.resize(eaf, rows)
    prm = refcpu.make_params()
    _, s0, n0 = refcpu.bench_gt(gts, n, rows, beta, e, prm)
    _, s1, n1, warned = refcpu.bench_gt_full(gts, n, rows, beta, e, prm, True, 0.001)
    _, s2, n2, threads = refcpu.bench_gt_allcores(gts, n, rows, beta, e, prm)
    assert n0 == n1 == n2 == rows and threads >= 1 and warned >= 0
    assert np.array_equal(s0, s1) and np.array_equal(s0, s2)
