"""GPU parity tests of the matrix-core single-read kernel (NPS_FMT_GT2X cohorts, nps_mx.hip) vs the CPU oracle.

Same bars as tests/test_gpu_parity.py: tallies / decisions / nloci bit-exact, scores within 1e-6 relative
(in practice ~1e-13: the row weights are 45-bit fixed point).  Everything goes through the C-ABI.
"""
import numpy as np
import pytest

from nimpress_amd import capi
from oracle import refcpu
from test_gpu_parity import PARAM_GRID, assert_stats_equal, make_cohort, oracle_scores, rel_err

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6
ESCAPES = []   # (samples that passed only through check_scores' 2^-50 escape, samples checked) per call


def check_scores(scores, ref_scores, beta, nloci):
    """the north star's bar -- 1e-6 relative, floored as in tests/test_gpu_parity.py -- or, for samples whose own
    terms cancel to (almost) nothing, an absolute difference below 2^-50 of the mean absolute weight: the row
    weights are 56-bit fixed point, i.e. as fine as the float64 rounding of the reference's own terms.  (Scores
    of beta values with four decimals cancel to exactly zero for a few samples in 10^4; the reference's result for
    those is its own rounding noise, ~1e-20, and no other summation reproduces that to six digits.)"""
    got, ref = np.asarray(scores), np.asarray(ref_scores)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), "NaN positions differ"
    ok = ~np.isnan(ref)
    if not ok.any():
        return 0
    sb = float(np.sum(np.abs(beta))) / max(2.0 * nloci, 1.0)
    d = np.abs(got[ok] - ref[ok])
    plain = REL_TOL * np.maximum(np.abs(ref[ok]), 1e-12 * sb)
    tol = np.maximum(plain, 2.0 ** -50 * sb)
    # how many samples needed the escape (a bar the builder wrote for itself is at least counted):
    # at most 1 in 1000 of the samples (and at most 2 in cohorts below 2000 samples)
    escaped = int(np.count_nonzero((d > plain) & (d <= tol)))
    ESCAPES.append((escaped, int(ok.sum())))
    assert escaped <= max(2, int(ok.sum()) // 1000), "%d of %d samples pass only through the 2^-50 escape" % (escaped, ok.sum())
    bad = np.nonzero(d > tol)[0]
    if bad.size:  # say where: strip / unit / parity of the samples that differ
        idx = np.nonzero(ok)[0][bad]
        raise AssertionError("%d samples differ (max |d| %.3g, max relative %.3g): strips %s units %s parity %s first %s" % (
            bad.size, d.max(), rel_err(got, ref, beta, max(nloci, 1)), np.unique(idx >> 11)[:20],
            np.unique((idx >> 5) & 63)[:64], np.unique(idx & 1), idx[:12]))
    return escaped


def score_gt2x(dev, n, kw, descs, offset, row0=0, mode=capi.MODE_AUTO):
    sc = capi.Scorer(n, capi.make_params(**kw))
    sc.score_cohort(dev, descs, row0, mode)
    stats = sc.flush()
    scores, nloci = sc.finish(offset)
    sc.close()
    return scores, nloci, stats


@pytest.mark.parametrize("shape", [(1, 1), (31, 3), (32, 128), (33, 129), (1000, 300), (2048, 5), (2049, 257),
                                   (4100, 40)])
def test_gt2x_fill_paths_agree(shape):
    """generator, upload of plain rows, conversion from a NPS_FMT_GT2 cohort: the same rows come back"""
    n, m = shape
    rng = np.random.default_rng(n * 7 + m)
    co = make_cohort(n, m, 99 + n, rng)
    words = (n + 15) // 16
    want = co["codes"][:, :words]
    a = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    a.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    assert np.array_equal(a.download(0, m), want)
    a.close()
    b = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    b.upload(0, co["codes"])
    assert np.array_equal(b.download(0, m), want)
    if m > 130:  # a sub-range that starts inside a superblock
        assert np.array_equal(b.download(129, m - 130), want[129:m - 1])
    b.close()
    src = capi.Cohort(n, m)
    src.upload(0, co["codes"])
    c = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    c.convert_from(src)
    assert np.array_equal(c.download(0, m), want)
    c.close()
    src.close()


@pytest.mark.parametrize("shape", [(1, 1), (33, 1), (100, 3), (1000, 64), (2047, 127), (2048, 128), (2049, 129),
                                   (4096, 257), (20000, 1001), (16385, 47), (40000, 385), (70000, 33), (5, 900),
                                   (70000, 2000), (3000, 40000), (250000, 700), (530000, 130), (1050000, 300)])
def test_gt2x_resident_vs_oracle(shape):
    """1 .. 513 strips (ragged last strip: 1 unit, 1 sample), 1 .. 313 superblocks (ragged last one), all decisions.
    Fewer strips than compute units: row teams (70 000 x 2 000: 35 strips x 7 teams of 2-3 superblocks; 3 000 x
    40 000: 2 strips x 128 teams; 250 000 x 700: 123 strips x 2 teams).  More strips than compute units (530 000 and
    1 050 000 samples: 259 and 513 strips, beyond the 255 x 2 048 samples of one resident grid): NPS_MODE_AUTO takes
    the tally pass + the accumulation with given tallies."""
    n, m = shape
    rng = np.random.default_rng(n + m)
    co = make_cohort(n, m, 4242, rng)
    kw = PARAM_GRID[(n + m) % len(PARAM_GRID)]
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    scores, nloci, stats = score_gt2x(dev, n, kw, capi.row_descs(co["beta"], co["eaf"], None, co["rie"]), 0.0)
    dev.close()
    ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 0.0)
    assert nloci == ref_nloci
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    check_scores(scores, ref_scores, co["beta"], nloci)


@pytest.mark.parametrize("shape", [(1, 1), (2049, 129), (20000, 1001), (70000, 2000), (3000, 5000)])
def test_gt2x_two_pass_mode_vs_oracle(shape):
    """NPS_MODE_TWOPASS on a strip cohort: the tally pass + the accumulation with given tallies (what shapes with more
    strips than compute units get under NPS_MODE_AUTO), here on shapes the single-read kernel also takes: equal to the
    oracle and bit-identical tallies; NPS_MODE_FUSED on a shape beyond the resident grid is refused, not mis-scored"""
    n, m = shape
    rng = np.random.default_rng(n * 3 + m)
    co = make_cohort(n, m, 777, rng)
    kw = PARAM_GRID[(n + m + 2) % len(PARAM_GRID)]
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    descs = capi.row_descs(co["beta"], co["eaf"], None, co["rie"])
    scores, nloci, stats = score_gt2x(dev, n, kw, descs, 0.5, mode=capi.MODE_TWOPASS)
    one, nloci1, stats1 = score_gt2x(dev, n, kw, descs, 0.5, mode=capi.MODE_FUSED)
    dev.close()
    ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 0.5)
    assert nloci == ref_nloci == nloci1
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    assert_stats_equal(stats1, [tuple(s) for s in ref_stats])
    check_scores(scores, ref_scores, co["beta"], nloci)
    # the same exact integer digit sums, folded in another grouping (teams): equal to a few ulps of the sum
    both = ~np.isnan(one)
    assert np.array_equal(np.isnan(one), np.isnan(scores))
    assert np.allclose(one[both], scores[both], rtol=1e-12, atol=1e-15)


def test_gt2x_given_tables_are_this_launch_s_own():
    """the accumulation with given tallies (nps_mxg.hip) fetches its operand tables by LDS-DMA and knows a table has landed by
    the superblock number every row carries.  LDS outlives a launch: the tables of the PREVIOUS launch carry the same numbers
    (round 5: 125 samples of one strip scored with the other score file's weights, once in many runs).  Two score files
    alternate over one cohort; every launch must reproduce its file's first result bit for bit."""
    n, m = 70000, 2000
    rng = np.random.default_rng(4242)
    co = make_cohort(n, m, 515, rng)
    kw = PARAM_GRID[0]
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    files = [capi.row_descs(co["beta"], co["eaf"], None, co["rie"]),
             capi.row_descs(co["beta"][::-1] * 3.0, co["eaf"], None, co["rie"])]
    first = [score_gt2x(dev, n, kw, d, 0.0, mode=capi.MODE_TWOPASS)[0] for d in files]
    assert not np.array_equal(first[0][~np.isnan(first[0])], first[1][~np.isnan(first[1])])
    for rep in range(6):
        for f, d in enumerate(files):
            again = score_gt2x(dev, n, kw, d, 0.0, mode=capi.MODE_TWOPASS)[0]
            assert np.array_equal(again.view(np.int64), first[f].view(np.int64)), "launch %d of file %d" % (rep, f)
    dev.close()


@pytest.mark.parametrize("shape,mode", [((70000, 4000), capi.MODE_FUSED), ((3000, 60000), capi.MODE_FUSED),
                                        ((70000, 4000), capi.MODE_TWOPASS)])
def test_gt2x_row_teams_bit_reproducible(shape, mode):
    """several row teams per strip, many rows over --maxmis in every team (their locus constants are float64 sums): 30
    passes give the same bits -- the teams' constants are added in slot order, not in the order the teams finish
    (a float atomicAdd per team was not reproducible: found by tools/soak.py in round 4)"""
    n, m = shape
    rng = np.random.default_rng(5 + n)
    co = make_cohort(n, m, 4243, rng)
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    sdef = capi.ScoreDef(capi.row_descs(co["beta"], co["eaf"], None, co["rie"]))
    sc = capi.Scorer(n, capi.make_params(**PARAM_GRID[0]))
    first = None
    for k in range(30):
        sc.reset()
        sc.score_cohort_def(dev, sdef, 0, mode)
        scores, nloci = sc.finish(0.25)
        if first is None:
            first = scores.copy()
            assert np.isfinite(first).all() and nloci == m
        else:
            assert np.array_equal(scores.view(np.int64), first.view(np.int64)), "pass %d differs" % k
    sc.close()
    sdef.close()
    dev.close()


def test_gt2x_fused_mode_refused_beyond_resident_grid():
    n, m = 530000, 130
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    sc = capi.Scorer(n, capi.make_params())
    with pytest.raises(capi.NpsError) as ei:
        sc.score_cohort(dev, capi.row_descs(np.ones(m), 0.3 * np.ones(m)), 0, capi.MODE_FUSED)
    assert ei.value.status == -6  # NPS_E_UNSUPPORTED
    sc.close()
    dev.close()


@pytest.mark.parametrize("pk", range(len(PARAM_GRID)))
def test_gt2x_all_imputation_modes(pk):
    n, m = 3000, 200
    rng = np.random.default_rng(900 + pk)
    co = make_cohort(n, m, 31 + pk, rng)
    kw = PARAM_GRID[pk]
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    dev.upload(0, co["codes"])
    scores, nloci, stats = score_gt2x(dev, n, kw, capi.row_descs(co["beta"], co["eaf"], None, co["rie"]), 0.125,
                                      mode=capi.MODE_FUSED)
    dev.close()
    ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 0.125)
    assert nloci == ref_nloci
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    check_scores(scores, ref_scores, co["beta"], nloci)


def test_gt2x_row_ranges_mixed_kinds_and_accumulation():
    """two calls over sub-ranges of one cohort (cohort_row0 = 0 and 256), rows without genotype data in between,
    NaN eaf, a large and a tiny beta in one definition"""
    n, m_present = 5000, 600
    rng = np.random.default_rng(77)
    # score rows: every 25th one has no genotype data; the cohort holds the PRESENT rows only
    kind_all = []
    for j in range(m_present):
        if j % 25 == 10:
            kind_all.append([capi.ROW_UNCOVERED, capi.ROW_ABSENT, capi.ROW_FILTERED][j % 3])
        kind_all.append(capi.ROW_PRESENT)
    kind_all = np.array(kind_all, np.int32)
    m = kind_all.size
    co = make_cohort(n, m, 555, rng)
    co["eaf"][5] = np.nan
    co["beta"][7] = 3.5
    co["beta"][9] = 1e-7
    kw = PARAM_GRID[0]
    dev = capi.Cohort(n, m_present, fmt=capi.FMT_GT2X)
    dev.upload(0, co["codes"][:m_present])
    sc = capi.Scorer(n, capi.make_params(**kw))
    present_seen = 0
    # first call: the score rows that consume cohort rows 0..255, second call: the rest
    cut = int(np.nonzero(np.cumsum(kind_all == capi.ROW_PRESENT) == 256)[0][0]) + 1
    for lo, hi in ((0, cut), (cut, m)):
        sc.score_cohort(dev, capi.row_descs(co["beta"][lo:hi], co["eaf"][lo:hi], kind_all[lo:hi], co["rie"][lo:hi]),
                        present_seen)
        present_seen += int((kind_all[lo:hi] == capi.ROW_PRESENT).sum())
    assert present_seen == m_present
    stats = sc.flush()
    scores, nloci = sc.finish(1.5)
    sc.close()
    dev.close()
    ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 1.5, kind_all)
    assert nloci == ref_nloci
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    check_scores(scores, ref_scores, co["beta"], nloci)


def test_gt2x_more_rows_than_one_flush():
    """300 000 rows x 70 samples: the float32 digit sums are flushed after 262 144 rows"""
    n, m = 70, 300_000
    rng = np.random.default_rng(3)
    co = make_cohort(n, m, 8, rng, force_missing_rows=False)
    kw = dict(imp_locus="ps", imp_missing="homref", imp_sample="int_ps", maxmis=0.05, mincs=10)
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    scores, nloci, stats = score_gt2x(dev, n, kw, capi.row_descs(co["beta"], co["eaf"], None, co["rie"]), 0.0)
    dev.close()
    ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 0.0)
    assert nloci == ref_nloci
    assert np.array_equal(stats["nmissing"], np.array([s[1] for s in ref_stats], dtype=np.uint64))
    check_scores(scores, ref_scores, co["beta"], nloci)


def test_gt2x_equals_table_kernel_large():
    """500 000 samples x 4096 rows (245 strips, the bench geometry): the matrix-core kernel on the strip layout and
    the table-lookup kernel on the row layout agree -- tallies and decisions bit for bit, scores to the weights'
    quantisation -- at a size the oracle cannot reach"""
    n, m = 500_000, 4096
    rng = np.random.default_rng(123)
    eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
    miss = rng.uniform(0, 0.02, m)
    miss[::100] = 0.1
    beta = np.round(rng.normal(0, 0.02, m), 4)
    th, tm, tmi = refcpu.hwe_thresholds(eaf, miss)
    res = []
    for fmt in (capi.FMT_GT2, capi.FMT_GT2X):
        dev = capi.Cohort(n, m, fmt=fmt)
        dev.synth(0, 5, th, tm, tmi)
        sc = capi.Scorer(n, capi.make_params())
        sc.score_cohort(dev, capi.row_descs(beta, eaf), 0, capi.MODE_FUSED)
        stats = sc.flush()
        scores, nloci = sc.finish(0.0)
        res.append((stats, scores, nloci))
        sc.close()
        dev.close()
    assert res[0][2] == res[1][2]
    assert np.array_equal(res[0][0], res[1][0])
    assert int((res[0][0]["reason"] == capi.REASON_MAXMIS).sum()) >= 30
    scale = np.sum(np.abs(beta)) / (2 * res[0][2])
    assert np.max(np.abs(res[0][1] - res[1][1])) <= 1e-10 * scale


def test_gt2x_full_size_config3_properties():
    """BASELINE.json configs[2] at its full size in the strip layout (500 000 samples x 1 000 000 rows, 125 GB
    resident): decisions of every row, whole-row tallies of 5 000 random rows recounted by the oracle over all
    samples, samples from EVERY strip (first / middle / last unit, both parities, the last ragged unit) scored
    over ALL rows by oracle/refcpu.c, exact scaling, row halves adding up."""
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
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    for r0 in range(0, m, 1 << 15):
        r1 = min(m, r0 + (1 << 15))
        dev.synth(r0, seed, th[r0:r1], tm[r0:r1], tmi[r0:r1])
    descs = capi.row_descs(beta, eaf)

    def run(rows, row0=0, want_stats=False):
        sc = capi.Scorer(n, capi.make_params())
        sc.score_cohort(dev, rows, row0, capi.MODE_FUSED)
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
    # (bench.py recounts 20 000 rows of this very cohort in every run; 5 000 here keep the suite's time down)
    rows = np.unique(np.concatenate([np.random.default_rng(1).choice(m, 5000, replace=False),
                                     [0, 1000, 499_999, m - 1]])).astype(np.uint64)
    ri = rows.astype(np.int64)
    g, ms, ne = refcpu.tally_synth_rows(rows, n, seed, th[ri], tm[ri], tmi[ri])
    assert np.array_equal(g, stats["ngenotyped"][ri].astype(np.float64))
    assert np.array_equal(ms, stats["nmissing"][ri].astype(np.float64))
    assert np.array_equal(ne, stats["neffect"][ri])
    assert np.array_equal(stats["ngenotyped"] + stats["nmissing"], np.full(m, n, dtype=np.uint64))
    sc0 = capi.Scorer(n, capi.make_params())
    strips, teams, sps = sc0.fused_geometry(m, capi.FMT_GT2X)
    sc0.close()
    # (one row team: the kernel cuts strips of 62 units from the unit sequence -- 253 workgroups; the LAYOUT has 245 strips of 64)
    assert strips == 253 and teams == 1 and sps == 62 * 32
    units = (n + 31) // 32
    us = {units - 1}
    for p_ in range(strips):   # of every strip of the kernel: its first unit, one behind a wave boundary (40: wave 4), a control
        first, last = p_ * 62, min(units, (p_ + 1) * 62) - 1   # wave's unit, the last
        us.update(u for u in (first, first + 40, first + 59, last) if first <= u <= last)
    us.update(range(64, units, 64 * 16))      # and units at boundaries of the layout's own 64-unit strips
    us.update(range(63, units, 64 * 16))
    samples = np.concatenate([np.arange(u * 32, min(n, (u + 1) * 32)) for u in sorted(us)]).astype(np.uint64)
    assert samples[-1] == n - 1
    sums, ref_nloci = refcpu.score_subset(samples, n, 0, seed, th, tm, tmi, beta, eaf, 0,
                                          stats["ngenotyped"].astype(np.float64),
                                          stats["nmissing"].astype(np.float64), stats["neffect"],
                                          refcpu.make_params())
    assert ref_nloci == nloci == m
    expect = sums / (2.0 * ref_nloci)
    got = (whole / (2.0 * nloci)).cpu().numpy()[samples.astype(np.int64)]
    check_scores(got, expect, beta, m)
    # exact scaling: doubling beta doubles every fixed-point weight (same digits one bit up)
    doubled, _, _ = run(capi.row_descs(2.0 * beta, eaf))
    assert bool(torch.equal(doubled, 2.0 * whole))
    del doubled
    # row halves (the second starts on a superblock boundary)
    h = 500_096
    lo, nlo, _ = run(descs[:h])
    hi, nhi, _ = run(descs[h:], row0=h)
    assert nlo + nhi == m
    assert float((lo + hi - whole).abs().max()) <= 1e-12 * float(np.sum(np.abs(beta)))
    dev.close()


def test_gt2x_every_row_of_a_65536_row_slice_recounted():
    """500 000 samples x 65 536 rows: EVERY row's whole-row tally (tallyAlleles, nimpress.nim:32-47) and decision
    recounted by the oracle over all samples (3.3e10 genotypes on the host's cores), not a sample of rows -- the
    scores of the full-size checks are fed with the device's tallies, so the tallies themselves are checked here in
    full for a slice; both single-read kernels"""
    n, m, seed = 500_000, 65_536, 20250103
    rng = np.random.default_rng(seed)
    beta = np.round(rng.normal(0.0, 0.02, m), 4)
    eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
    miss = rng.uniform(0.0, 0.02, m)
    miss[::1000] = 0.10
    th, tm, tmi = refcpu.hwe_thresholds(eaf, miss)
    rows = np.arange(m, dtype=np.uint64)
    g, ms, ne = refcpu.tally_synth_rows(rows, n, seed, th, tm, tmi)
    for fmt in (capi.FMT_GT2X, capi.FMT_GT2):
        dev = capi.Cohort(n, m, fmt=fmt)
        for r0 in range(0, m, 1 << 15):
            dev.synth(r0, seed, th[r0:r0 + (1 << 15)], tm[r0:r0 + (1 << 15)], tmi[r0:r0 + (1 << 15)])
        sc = capi.Scorer(n, capi.make_params())
        sc.score_cohort(dev, capi.row_descs(beta, eaf), 0, capi.MODE_FUSED)
        stats = sc.flush()
        _, nloci = sc.finish(0.0)
        sc.close()
        dev.close()
        assert nloci == m
        assert np.array_equal(g, stats["ngenotyped"].astype(np.float64))
        assert np.array_equal(ms, stats["nmissing"].astype(np.float64))
        assert np.array_equal(ne, stats["neffect"])
        assert np.array_equal(stats["reason"] == capi.REASON_MAXMIS, ms / n > 0.05)


@pytest.mark.parametrize("mode", [capi.MODE_AUTO, capi.MODE_TWOPASS])
def test_gt2x_beta_span_plain_relative_bar(mode):
    """beta from 1e-9 to 10 in ONE definition, and samples that carry only its small-beta rows: the north star's bar --
    1e-6 RELATIVE, plain, no absolute escape -- for every sample.  (56-bit fixed-point weights scaled to the largest
    |beta| alone would leave such a sample 2^-56 x 10 / 1e-9 = 1.4e-7 per term at best and nothing below 1e-17; the
    definition is scored in magnitude bands of 2^30, one pass each.)  All beta positive: no cancellation, the relative
    bar means what it says."""
    n, m = 6000, 512
    rng = np.random.default_rng(4711)
    mag = np.array([10.0, 1e-3, 1e-9, 3e-14])
    beta = np.concatenate([rng.uniform(0.1, 1.0, m // 4) * k for k in mag])
    eaf = np.round(rng.uniform(0.05, 0.5, m), 4)
    # quarter q of the samples has genotypes (dosage 0/1/2, 2 % missing) only in the rows of magnitudes >= q
    codes = np.zeros((m, (n + 15) // 16), np.uint32)
    g = rng.integers(0, 3, size=(m, n)).astype(np.uint32)
    g[g == 2] = 3                       # NPS_CODE_DOSAGE2
    miss = rng.uniform(size=(m, n)) < 0.02
    g[miss] = 2                         # NPS_CODE_MISSING
    for q in range(4):
        g[: q * (m // 4), q * (n // 4):(q + 1) * (n // 4)] = 0
    for k in range(16):
        cols = g[:, k::16]
        codes[:, :cols.shape[1]] |= cols << np.uint32(2 * k)
    co = dict(n=n, m=m, eaf=eaf, beta=beta, rie=np.zeros(m, np.int32), codes=codes)
    kw = dict(imp_locus="ps", imp_missing="homref", imp_sample="int_ps", maxmis=0.05, mincs=100)
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    dev.upload(0, codes)
    scores, nloci, stats = score_gt2x(dev, n, kw, capi.row_descs(beta, eaf), 0.0, mode=mode)
    dev.close()
    ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 0.0)
    assert nloci == ref_nloci == m
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    ref = np.asarray(ref_scores)
    assert not np.isnan(ref).any() and (ref > 0).all()
    rel = np.abs(np.asarray(scores) - ref) / ref
    assert rel.max() <= 1e-6, (rel.max(), int(rel.argmax()), ref[rel.argmax()])
    # the last quarter's scores are 1e-14 of the first quarter's
    assert ref[-1] < 1e-12 * ref[0]


@pytest.mark.parametrize("shape", [(300001, 300), (450123, 129), (505920, 130), (505889, 5), (262145, 257)])
def test_gt2x_strips_of_62_units_vs_oracle(shape):
    """Where a strip has one row team, the single-read kernel cuts the cohort's unit sequence into strips of 62 units instead
    of the layout's 64 (more strips = more compute units at work): a wave's units then cross a boundary of the LAYOUT's
    strips, the last layout strip is ragged, the last virtual strip too (505 920 samples: 255 strips of exactly 62 units;
    505 889: the last sample's unit is partial; 262 145: 129 layout strips, the last of one sample).  NPS_MODE_FUSED equals
    the oracle and, to the last bits, the two-read path, which works on the layout's own strips; and itself, bit for bit."""
    n, m = shape
    rng = np.random.default_rng(n + 7 * m)
    co = make_cohort(n, m, 6262, rng)
    kw = PARAM_GRID[(n + m) % len(PARAM_GRID)]
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    descs = capi.row_descs(co["beta"], co["eaf"], None, co["rie"])
    scores, nloci, stats = score_gt2x(dev, n, kw, descs, 0.0, mode=capi.MODE_FUSED)
    two, nloci2, _ = score_gt2x(dev, n, kw, descs, 0.0, mode=capi.MODE_TWOPASS)
    again, _, _ = score_gt2x(dev, n, kw, descs, 0.0, mode=capi.MODE_FUSED)
    dev.close()
    ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 0.0)
    assert nloci == ref_nloci == nloci2
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    check_scores(scores, ref_scores, co["beta"], nloci)
    assert np.array_equal(np.isnan(two), np.isnan(scores))
    ok = ~np.isnan(two)
    assert np.allclose(two[ok], scores[ok], rtol=1e-12, atol=1e-15)   # (exact digit sums; the locus constants are added in team order)
    assert np.array_equal(again.view(np.int64), scores.view(np.int64))


@pytest.mark.parametrize("n", [3000, 250000, 300000, 400000, 522240, 530000])
def test_gt_auto_layout_at_every_size(n):
    """NPS_FMT_GT_AUTO is the strip layout at every size (round 5); NPS_MODE_AUTO equals the oracle whether the run counts its
    tallies in the pass (the resident grid covers the chip) or is scored in two reads / with kept tallies (147 strips at
    300 000 samples; 259 strips at 530 000: more than compute units)"""
    m = 140
    rng = np.random.default_rng(n)
    co = make_cohort(n, m, 99, rng)
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT_AUTO)
    assert dev.fmt == capi.FMT_GT2X
    dev.upload(0, co["codes"])
    kw = PARAM_GRID[0]
    sc = capi.Scorer(n, capi.make_params(**kw))
    sc.score_cohort(dev, capi.row_descs(co["beta"], co["eaf"], None, co["rie"]), 0, capi.MODE_AUTO)
    stats = sc.flush()
    scores, nloci = sc.finish(0.0)
    sc.close()
    dev.close()
    ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 0.0)
    assert nloci == ref_nloci
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    check_scores(scores, ref_scores, co["beta"], nloci)


def test_gt2x_refusals():
    dev = capi.Cohort(100, 300, fmt=capi.FMT_GT2X)
    sc = capi.Scorer(100, capi.make_params())
    d = capi.row_descs(np.zeros(10), np.full(10, 0.1))
    with pytest.raises(capi.NpsError):
        sc.score_cohort(dev, d, 5)                       # cohort_row0 not a multiple of 128
    with pytest.raises(capi.NpsError):
        dev.upload(64, np.zeros((10, 7), np.uint32))     # row0 of an upload: whole superblocks
    sc.score_cohort(dev, d, 128)                         # refused calls left the context usable
    scores, nloci = sc.finish(0.0)
    assert nloci == 10 and np.all(scores == 0.0)
    sc.close()
    dev.close()


@pytest.mark.parametrize("shape", [(1, 1), (2049, 129), (20000, 1001), (70000, 2000), (250000, 700), (530000, 130)])
def test_gt2x_kept_tallies_vs_oracle(shape):
    """nps_cohort_keep_tallies (experiment B): the cohort carries tallyAlleles of every row, counted once;
    NPS_MODE_AUTO then scores with the tallies given (one read, no hand-over).  Equal to the oracle, row statistics
    bit-identical to the counting pass; the kept tallies equal the recount; rewriting rows drops them; a run that starts
    inside the cohort (row0 = 128) uses the right words."""
    n, m = shape
    rng = np.random.default_rng(n * 5 + m)
    co = make_cohort(n, m, 31337, rng)
    kw = PARAM_GRID[(n + m + 4) % len(PARAM_GRID)]
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    descs = capi.row_descs(co["beta"], co["eaf"], None, co["rie"])
    assert not dev.has_tallies()
    base, nloci0, stats0 = score_gt2x(dev, n, kw, descs, 0.25)
    dev.keep_tallies()
    assert dev.has_tallies()
    nm, ne = dev.row_tallies()
    scores, nloci, stats = score_gt2x(dev, n, kw, descs, 0.25)
    sc = capi.Scorer(n, capi.make_params(**kw))
    sc.profile_enable(True)
    sc.score_cohort(dev, descs, 0, capi.MODE_AUTO)
    p = sc.profile_get(reset=True)
    assert p.n_tally == 0 and p.n_fused == 0 and p.n_accumulate >= 1      # no tally pass, the given-tallies kernel
    sc.close()
    ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 0.25)
    assert nloci == ref_nloci == nloci0
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    assert_stats_equal(stats0, [tuple(s) for s in ref_stats])
    assert np.array_equal(nm, np.array([s[1] for s in ref_stats], dtype=np.uint64))
    assert np.array_equal(ne.astype(np.float64), np.array([s[2] for s in ref_stats]))
    check_scores(scores, ref_scores, co["beta"], nloci)
    if m > 200:   # a run over rows 128.. of the cohort
        sub, nl_sub, st_sub = score_gt2x(dev, n, kw, descs[128:], 0.0, row0=128)
        ref_sub = oracle_scores(dict(co, m=m - 128, codes=co["codes"][128:], beta=co["beta"][128:], eaf=co["eaf"][128:],
                                     rie=co["rie"][128:]), kw, 0.0)
        assert nl_sub == ref_sub[2]
        assert_stats_equal(st_sub, [tuple(s) for s in ref_sub[1]])
        check_scores(sub, ref_sub[0], co["beta"][128:], nl_sub)
    dev.synth(0, co["seed"] + 1, co["th"], co["tm"], co["tmi"])   # rows rewritten: the tallies are gone
    assert not dev.has_tallies()
    dev.close()


@pytest.mark.parametrize("n", [300_000, 530_000])
def test_auto_counts_tallies_once_where_the_resident_grid_does_not_cover_the_chip(n):
    """NPS_FMT_GT_AUTO is the strip layout at every size now.  300 000 samples (147 strips x 1 team: 147 of 256 compute
    units) and 530 000 (259 strips: more than compute units): the first NPS_MODE_AUTO run leaves the cohort's tallies with
    the cohort -- at 300 000 as a by-product of the single read that counts them anyway (round 6: one launch of the in-pass
    kernel, no tally pass), at 530 000 from a tally pass of their own -- later runs read the matrix once with the tallies
    given; equal to the oracle either way; rewriting rows drops the tallies and the next run counts again.  (300 000 x 16 384 -- the lazy rule wants that many rows -- is checked as
    bench.py checks the full size: a few hundred samples from all over the cohort scored over ALL rows by the oracle's
    subset path, fed with the device's row statistics, of which 200 rows are recounted over all samples; the whole-cohort
    oracle took 45 s of the suite for this one case.)"""
    big = n == 300_000
    m = 16384 if big else 640
    rng = np.random.default_rng(n)
    if big:   # (make_cohort's draws without its host copy of the matrix)
        eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
        miss = rng.uniform(0.0, 0.1, m)
        miss[::7] = 0.3
        miss[3] = 1.0 - 1e-9
        beta = np.round(rng.normal(0, 0.02, m), 4)
        rie = (rng.uniform(size=m) < 0.25).astype(np.int32)
        th, tm, tmi = refcpu.hwe_thresholds(eaf, miss)
        co = dict(n=n, m=m, eaf=eaf, beta=beta, rie=rie, th=th, tm=tm, tmi=tmi, seed=606)
    else:
        co = make_cohort(n, m, 606, rng)
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT_AUTO)
    assert dev.fmt == capi.FMT_GT2X
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    descs = capi.row_descs(co["beta"], co["eaf"], None, co["rie"])
    kw = PARAM_GRID[0]
    if not big:
        ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 0.0)
    assert not dev.has_tallies()
    for k in range(2):
        sc = capi.Scorer(n, capi.make_params(**kw))
        sc.profile_enable(True)
        sc.score_cohort(dev, descs, 0, capi.MODE_AUTO)
        p = sc.profile_get(reset=True)
        stats = sc.flush()
        scores, nloci = sc.finish(0.0)
        sc.close()
        if big and k == 0:
            assert dev.has_tallies() and p.n_fused >= 1 and p.n_tally == 0 and p.n_accumulate == 0    # in the pass, kept
        else:
            assert dev.has_tallies() and p.n_fused == 0 and p.n_tally == 0 and p.n_accumulate >= 1     # given
        if k == 0:
            first = scores.copy()
        else:   # exact integer digit sums in both kernels: the same scores
            ok = ~np.isnan(first)
            assert np.array_equal(np.isnan(scores), np.isnan(first))
            assert np.allclose(scores[ok], first[ok], rtol=1e-12, atol=1e-18)
        if big:
            pick = np.random.default_rng(5 + k)
            rows = np.unique(np.concatenate([pick.choice(m, 200, replace=False), [0, 3, 7, m - 1]])).astype(np.uint64)
            ri = rows.astype(np.int64)
            g, ms, ne = refcpu.tally_synth_rows(rows, n, co["seed"], co["th"][ri], co["tm"][ri], co["tmi"][ri], rie=co["rie"][ri])
            assert np.array_equal(g, stats["ngenotyped"][ri].astype(np.float64))
            assert np.array_equal(ms, stats["nmissing"][ri].astype(np.float64))
            assert np.array_equal(ne, stats["neffect"][ri])
            samples = np.unique(np.concatenate([pick.choice(n, 300, replace=False), [0, 1, 2047, 2048, n - 2, n - 1]])).astype(np.uint64)
            sums, ref_nloci = refcpu.score_subset(samples, n, 0, co["seed"], co["th"], co["tm"], co["tmi"], co["beta"], co["eaf"],
                                                  co["rie"], stats["ngenotyped"].astype(np.float64),
                                                  stats["nmissing"].astype(np.float64), stats["neffect"], refcpu.make_params(**kw))
            assert nloci == ref_nloci == int(stats["used"].sum())
            check_scores(scores[samples.astype(np.int64)], sums / (2.0 * ref_nloci), co["beta"], nloci)
        else:
            assert nloci == ref_nloci
            assert_stats_equal(stats, [tuple(s) for s in ref_stats])
            check_scores(scores, ref_scores, co["beta"], nloci)
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    assert not dev.has_tallies()
    dev.close()


def test_expect_passes_keeps_the_tallies_of_the_first_pass():
    """nps_cohort_expect_passes (round 6): at a size whose resident grid covers the chip NPS_MODE_AUTO counts the tallies in
    every pass -- unless the caller says the cohort will be scored again: then the first whole-cohort pass keeps what it
    counted (no extra read) and later passes run with the tallies given.  The kept tallies are the oracle's; a run over
    PART of the cohort keeps nothing; and the hint is ignored where it would not pay (several row teams per strip: at most
    262 144 samples -- the given-tallies kernel is no faster there than the pass that counts them)."""
    small = capi.Cohort(70_000, 1500, fmt=capi.FMT_GT2X)
    cs = make_cohort(70_000, 1500, 4321, np.random.default_rng(5))
    small.synth(0, cs["seed"], cs["th"], cs["tm"], cs["tmi"])
    small.expect_passes(8)
    for _ in range(2):
        sc = capi.Scorer(70_000, capi.make_params())
        sc.profile_enable(True)
        sc.score_cohort(small, capi.row_descs(cs["beta"], cs["eaf"], None, cs["rie"]), 0, capi.MODE_AUTO)
        p = sc.profile_get(reset=True)
        sc.finish(0.0)
        sc.close()
        assert p.n_fused >= 1 and p.n_accumulate == 0 and not small.has_tallies()
    small.close()
    n, m = 500_000, 2000
    rng = np.random.default_rng(77)
    co = make_cohort(n, m, 1234, rng)
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    descs = capi.row_descs(co["beta"], co["eaf"], None, co["rie"])
    kw = PARAM_GRID[0]
    ref_scores, ref_stats, ref_nloci = oracle_scores(co, kw, 0.0)

    def run(row0=0, d=descs):
        sc = capi.Scorer(n, capi.make_params(**kw))
        sc.profile_enable(True)
        sc.score_cohort(dev, d, row0, capi.MODE_AUTO)
        p = sc.profile_get(reset=True)
        stats = sc.flush()
        scores, nloci = sc.finish(0.0)
        sc.close()
        return p, stats, scores, nloci

    for _ in range(2):                                  # no hint: in the pass, every time
        p, stats, scores, nloci = run()
        assert p.n_fused >= 1 and p.n_accumulate == 0 and not dev.has_tallies()
    dev.expect_passes(8)
    p, _, _, _ = run(128, descs[128:])                  # a partial run keeps nothing
    assert p.n_fused >= 1 and not dev.has_tallies()
    p, stats, first, nloci = run()
    assert p.n_fused >= 1 and p.n_tally == 0 and p.n_accumulate == 0 and dev.has_tallies()
    assert nloci == ref_nloci
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    check_scores(first, ref_scores, co["beta"], nloci)
    nm, ne = dev.row_tallies(0, m)                      # what was kept is what tallyAlleles gives
    assert np.array_equal(nm, stats["nmissing"].astype(nm.dtype))          # (the statistics were just checked against the oracle's)
    assert np.array_equal(ne.astype(np.float64), stats["neffect"])
    p, stats, again, nloci = run()
    assert p.n_fused == 0 and p.n_accumulate >= 1      # given
    assert nloci == ref_nloci
    assert_stats_equal(stats, [tuple(s) for s in ref_stats])
    ok = ~np.isnan(first)
    assert np.array_equal(np.isnan(again), np.isnan(first)) and np.allclose(again[ok], first[ok], rtol=1e-12, atol=1e-18)
    dev.synth(0, co["seed"] + 1, co["th"], co["tm"], co["tmi"])
    assert not dev.has_tallies()
    # the rows were rewritten: the next pass counts and keeps the NEW rows' tallies, the one after runs with them given
    co2 = dict(co, seed=co["seed"] + 1, codes=refcpu.synth_rows(n, 0, m, co["seed"] + 1, co["th"], co["tm"], co["tmi"]))
    ref2_scores, ref2_stats, ref2_nloci = oracle_scores(co2, kw, 0.0)
    for k in range(2):
        p, stats, scores, nloci = run()
        assert dev.has_tallies() and ((p.n_fused >= 1 and p.n_accumulate == 0) if k == 0 else (p.n_fused == 0 and p.n_accumulate >= 1))
        assert nloci == ref2_nloci
        assert_stats_equal(stats, [tuple(s) for s in ref2_stats])
        check_scores(scores, ref2_scores, co["beta"], nloci)
    dev.close()


def test_auto_keeps_tallies_with_a_banded_definition():
    """A definition whose |beta| span needs magnitude bands (one pass per band, band 0 counts nloci and writes the statistics)
    on a cohort size where NPS_MODE_AUTO keeps the first pass's tallies (280 000 samples: 137 strips): the tallies are taken
    from band 0's pass, later runs score every band with them given; both agree with the explicit single-read mode."""
    n, m = 280_000, 1536
    rng = np.random.default_rng(1601)
    eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
    miss = rng.uniform(0.0, 0.1, m)
    miss[::7] = 0.3
    beta = np.where(np.arange(m) % 2 == 0, rng.uniform(0.1, 1.0, m) * 10.0, rng.uniform(0.1, 1.0, m) * 1e-12)
    th, tm, tmi = refcpu.hwe_thresholds(eaf, miss)
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT_AUTO)
    dev.synth(0, 77, th, tm, tmi)
    descs = capi.row_descs(beta, eaf)
    kw = PARAM_GRID[0]
    fused, nl_f, st_f = score_gt2x(dev, n, kw, descs, 0.0, mode=capi.MODE_FUSED)
    assert not dev.has_tallies()
    runs = []
    for k in range(3):
        sc = capi.Scorer(n, capi.make_params(**kw))
        sc.profile_enable(True)
        sc.score_cohort(dev, descs, 0, capi.MODE_AUTO)
        p = sc.profile_get(reset=True)
        stats = sc.flush()
        scores, nloci = sc.finish(0.0)
        sc.close()
        assert dev.has_tallies()
        assert (p.n_fused >= 2 and p.n_accumulate == 0) if k == 0 else (p.n_fused == 0 and p.n_accumulate >= 2)   # (two bands)
        assert nloci == nl_f
        for key in ("ngenotyped", "nmissing", "neffect", "used", "reason"):
            assert np.array_equal(stats[key], st_f[key]), key
        runs.append(scores)
    assert np.array_equal(runs[0].view(np.int64), fused.view(np.int64))          # the same kernel, the same bits
    ok = ~np.isnan(fused)
    assert np.array_equal(np.isnan(runs[1]), ~ok) and np.array_equal(runs[1].view(np.int64), runs[2].view(np.int64))
    rel = np.abs(runs[1][ok] - fused[ok]) / np.maximum(np.abs(fused[ok]), 1e-300)
    assert rel.max() <= 1e-11, rel.max()
    nm, ne = dev.row_tallies(0, m)
    assert np.array_equal(nm, st_f["nmissing"].astype(nm.dtype)) and np.array_equal(ne.astype(np.float64), st_f["neffect"])
    dev.close()


def test_two_threads_score_one_auto_cohort():
    """two contexts on two threads score ONE cohort under NPS_MODE_AUTO at a size where the first run
    attaches kept tallies to the cohort (280 000 samples: 137 strips).  Whoever comes first counts and publishes them (an
    atomic flag, release / acquire); both threads get the single-threaded result, bit for bit among the given-tallies runs."""
    import threading
    n, m = 280_000, 2048
    rng = np.random.default_rng(99)
    eaf = np.round(rng.uniform(0.01, 0.5, m), 4)
    miss = rng.uniform(0.0, 0.1, m)
    miss[::7] = 0.3
    beta = np.round(rng.normal(0, 0.02, m), 4)
    th, tm, tmi = refcpu.hwe_thresholds(eaf, miss)
    descs = capi.row_descs(beta, eaf)
    out, errs = {}, []

    def work(tag, dev, passes):
        try:
            sc = capi.Scorer(n, capi.make_params())
            res = []
            for _ in range(passes):
                sc.reset()
                sc.score_cohort(dev, descs, 0, capi.MODE_AUTO)
                res.append(sc.finish(0.0))
            sc.close()
            out[tag] = res
        except Exception as e:     # noqa: BLE001
            errs.append((tag, repr(e)))

    ref_dev = capi.Cohort(n, m, fmt=capi.FMT_GT_AUTO)
    ref_dev.synth(0, 31, th, tm, tmi)
    work("ref", ref_dev, 2)
    ref_dev.close()
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT_AUTO)
    dev.synth(0, 31, th, tm, tmi)
    assert not dev.has_tallies()
    ts = [threading.Thread(target=work, args=(k, dev, 3)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert dev.has_tallies()
    ref_first, ref_given = out["ref"]
    ok = ~np.isnan(ref_first[0])
    for k in range(2):
        for scores, nloci in out[k]:
            assert nloci == ref_first[1] == ref_given[1]
            assert np.array_equal(np.isnan(scores), ~ok)
            assert np.allclose(scores[ok], ref_first[0][ok], rtol=1e-12, atol=1e-18)
        assert np.array_equal(out[k][-1][0].view(np.int64), ref_given[0].view(np.int64))   # (both with the tallies given)
    dev.close()


def test_zz_escape_count_of_this_module():
    """runs last in this file: over every check_scores call of the module, the samples that passed only through the
    2^-50 escape are fewer than 1 in 1000 (printed with -s)"""
    esc, tot = sum(e for e, _ in ESCAPES), sum(t for _, t in ESCAPES)
    print("check_scores: %d of %d samples passed through the 2^-50 escape only (%d calls)" % (esc, tot, len(ESCAPES)))
    assert tot > 0 and esc * 1000 <= tot
