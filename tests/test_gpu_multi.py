"""GPU parity tests of the multi-score path: S score definitions in one pass over
a NPS_FMT_GT2M cohort on the matrix cores (nps_score_cohort_multi), every score against the oracle run
once per score (the reference's loop, nimpress.nim:634-641, S times) and against the single-score HIP path.

Bars: nloci and the row tallies the cohort carries bit-exact; scores within 1e-6 relative (floored),
in practice ~1e-13: the per-row weights are quantised to 2^-49 of the largest one, everything after
that is exact integer arithmetic."""
import numpy as np
import pytest

from nimpress_amd import capi
from oracle import refcpu

pytestmark = pytest.mark.gpu

# the north star's bar is 1e-6 relative (floored); the quantised weights give ~1e-12 for typical samples and up
# to ~1e-9 where a sample's own terms cancel to a small score
REL_TOL = 1e-7

PARAM_GRID = [
    dict(),
    dict(imp_locus="homref", imp_missing="ignore", imp_sample="ps", maxmis=0.03, mincs=10),
    dict(imp_locus="ignore", imp_missing="homref", imp_sample="homref", maxmis=0.02, mincs=0),
    dict(imp_locus="fail", imp_missing="homref", imp_sample="int_fail", maxmis=0.06, mincs=100),
    dict(imp_locus="ps", imp_missing="homref", imp_sample="fail", maxmis=1.0, mincs=100),
    dict(imp_locus="ps", imp_missing="ignore", imp_sample="int_ps", maxmis=0.05, mincs=10 ** 9),
]


def rel_err(got, ref, beta_abs_sum, nloci):
    assert np.array_equal(np.isnan(got), np.isnan(ref)), "NaN positions differ"
    ok = ~np.isnan(ref)
    if not ok.any():
        return 0.0
    floor = 1e-12 * beta_abs_sum / max(2.0 * nloci, 1.0)
    return float(np.max(np.abs(got[ok] - ref[ok]) / np.maximum(np.abs(ref[ok]), max(floor, 1e-300))))


def make_case(n, m, S, seed, with_kinds=True):
    rng = np.random.default_rng(seed)
    eaf_c = np.round(rng.uniform(0.01, 0.5, m), 4)               # the cohort's allele frequencies
    miss = rng.uniform(0.0, 0.08, m)
    th, tm, tmi = refcpu.hwe_thresholds(eaf_c, miss)
    descs = np.zeros((S, m), dtype=capi.ROW_DESC_DTYPE)
    for s in range(S):
        descs[s]["beta"] = np.round(rng.normal(0, 0.02 * (1 + 10 * s), m), 4)
        descs[s]["eaf"] = np.where(rng.uniform(size=m) < 0.02, np.nan, np.round(rng.uniform(0.01, 0.9, m), 4))
        descs[s]["ref_is_effect"] = (rng.uniform(size=m) < 0.3).astype(np.int32)
        kind = np.zeros(m, dtype=np.int32)
        if with_kinds:
            u = rng.uniform(size=m)
            kind[u < 0.10] = capi.ROW_NOT_IN_SCORE
            kind[(u >= 0.10) & (u < 0.13)] = capi.ROW_ABSENT
            kind[(u >= 0.13) & (u < 0.15)] = capi.ROW_UNCOVERED
            kind[(u >= 0.15) & (u < 0.17)] = capi.ROW_FILTERED
        descs[s]["kind"] = kind
    return eaf_c, th, tm, tmi, descs


def oracle_scores(codes, n, descs, params_kw, offsets):
    """the reference's loop once per score; NOT_IN_SCORE rows simply are not rows of that score"""
    out, nl = [], []
    for s in range(descs.shape[0]):
        d = descs[s]
        keep = d["kind"] != capi.ROW_NOT_IN_SCORE
        present = keep & (d["kind"] == capi.ROW_PRESENT)
        sc, _, nloci = refcpu.score_packed(codes[present], n, d["kind"][keep], d["ref_is_effect"][keep],
                                           d["beta"][keep], d["eaf"][keep], refcpu.make_params(**params_kw),
                                           float(offsets[s]))
        out.append(sc)
        nl.append(nloci)
    return np.stack(out), np.array(nl)


@pytest.mark.parametrize("wbits", [0, 41])
@pytest.mark.parametrize("bits", [56, 40, 32])
@pytest.mark.parametrize("pk", range(len(PARAM_GRID)))
@pytest.mark.parametrize("shape", [(1, 1, 1), (17, 33, 2), (6, 7, 3), (33, 129, 4), (257, 300, 5), (65, 130, 6),
                                   (40, 260, 7), (1000, 1025, 8), (4099, 64, 8)])
def test_multi_vs_oracle_converted_cohort(shape, pk, bits, wbits):
    """a 2-bit cohort uploaded row-major, repacked on the device (nps_cohort_convert), S definitions with
    their own beta / eaf / effect allele, rows a score does not list, absent / uncovered / FILTER rows; with
    full-width and with 32-bit weights for the imputed value of a missing genotype (the NaN cases of PARAM_GRID 3
    and 4 must stay exact in both), with six and with seven digits per weight"""
    n, m, S = shape
    kw = PARAM_GRID[pk]
    eaf_c, th, tm, tmi, descs = make_case(n, m, S, 1000 * pk + n + m)
    codes = refcpu.synth_rows(n, 0, m, 31, th, tm, tmi)
    src = capi.Cohort(n, m)
    src.upload(0, codes)
    co = capi.Cohort(n, m, fmt=capi.FMT_GT2M)
    co.convert_from(src)
    src.close()
    # the tallies the cohort carries == tallyAlleles of every row
    nm, ne = co.row_tallies()
    g, ms, neff = refcpu.tally_synth_rows(np.arange(m), n, 31, th, tm, tmi)
    assert np.array_equal(nm, ms.astype(np.uint64)) and np.array_equal(ne, neff.astype(np.uint64))
    offsets = np.linspace(-0.5, 0.5, S)
    msc = capi.MultiScorer(n, capi.make_params(**kw), S)
    msc.set_missing_weight_bits(bits)
    mdef = capi.MultiDef(descs, weight_bits=wbits)   # (49-bit weights, 41 on request: every score count 1..8 has its own
    # number of column tiles in both)
    msc.score_cohort(co, mdef)
    got, nloci = msc.finish(offsets)
    ref, ref_nloci = oracle_scores(codes, n, descs, kw, offsets)
    assert np.array_equal(nloci.astype(np.int64), ref_nloci)
    for s in range(S):
        keep = descs[s]["kind"] != capi.ROW_NOT_IN_SCORE
        if bits in (32, 40) or wbits == 41:
            # the documented bounds (include/nps.h), after the division by 2 nloci: 32-bit is-missing weights:
            # (missing genotypes of the sample) x 2^-24 x B; 41-bit weights: every term within 3 x 2^-41 of the
            # largest weight the fixed point holds (< 2 B)
            d = descs[s][keep]
            B = float(np.max(np.abs(d["beta"])) * (3.0 + max(2.0, 2.0 * float(np.max(np.abs(np.nan_to_num(d["eaf"])))))))
            bound = np.zeros(n)
            if bits in (32, 40):
                # (NPS_CODE_MISSING = 2; 16 codes per word); five digits instead of four: 2^-32 instead of 2^-24
                plain = ((np.asarray(codes)[keep][:, :, None] >> (2 * np.arange(16, dtype=np.uint32))) & 3).reshape(keep.sum(), -1)
                bound += (plain[:, :n] == 2).sum(axis=0) * 2.0 ** (-24 if bits == 32 else -32) * B
            if wbits == 41:
                bound += keep.sum() * 6.0 * 2.0 ** -40 * B
            bound /= max(2.0 * int(ref_nloci[s]), 1)
            ok = ~np.isnan(ref[s])
            assert np.array_equal(np.isnan(got[s]), np.isnan(ref[s]))
            assert np.all(np.abs(got[s][ok] - ref[s][ok]) <= bound[ok] + REL_TOL * np.abs(ref[s][ok])), s
            continue
        assert rel_err(got[s], ref[s], float(np.sum(np.abs(descs[s]["beta"][keep]))), int(ref_nloci[s])) <= REL_TOL, s
    # a second call on the same context after a reset gives the same bits
    msc.reset()
    msc.score_cohort(co, mdef)
    again, nloci2 = msc.finish(offsets)
    assert np.array_equal(again, got, equal_nan=True) and np.array_equal(nloci, nloci2)
    msc.close()
    mdef.close()
    co.close()


def test_multi_synth_cohort_and_row_chunks():
    """a cohort generated on the device in the NPS_FMT_GT2M layout == the oracle's generator; the rows
    scored as two calls (chunks) add up to the one-call result"""
    n, m, S, seed = 5000, 640, 8, 77
    eaf_c, th, tm, tmi, descs = make_case(n, m, S, 5, with_kinds=False)
    co = capi.Cohort(n, m, fmt=capi.FMT_GT2M)
    co.synth(0, seed, th, tm, tmi)
    nm, ne = co.row_tallies()
    g, ms, neff = refcpu.tally_synth_rows(np.arange(m), n, seed, th, tm, tmi)
    assert np.array_equal(nm, ms.astype(np.uint64)) and np.array_equal(ne, neff.astype(np.uint64))
    codes = refcpu.synth_rows(n, 0, m, seed, th, tm, tmi)
    offsets = np.zeros(S)
    msc = capi.MultiScorer(n, capi.make_params(), S)
    mdef = capi.MultiDef(descs)
    msc.score_cohort(co, mdef)
    got, nloci = msc.finish(offsets)
    ref, ref_nloci = oracle_scores(codes, n, descs, {}, offsets)
    assert np.array_equal(nloci.astype(np.int64), ref_nloci)
    for s in range(S):
        assert rel_err(got[s], ref[s], float(np.sum(np.abs(descs[s]["beta"]))), m) <= REL_TOL
    # two chunks (the second starts on a superblock boundary)
    msc.reset()
    a = capi.MultiDef(descs[:, :384])
    b = capi.MultiDef(descs[:, 384:])
    msc.score_cohort(co, a, 0)
    msc.score_cohort(co, b, 384)
    two, nloci2 = msc.finish(offsets)
    assert np.array_equal(nloci2, nloci)
    for s in range(S):
        assert rel_err(two[s], got[s], float(np.sum(np.abs(descs[s]["beta"]))), m) <= 1e-12
    for x in (a, b, mdef):
        x.close()
    msc.close()
    co.close()


@pytest.mark.parametrize("where", ["none", "first_half", "one_row"])
def test_multi_superblocks_without_missing(where):
    """superblocks (128 rows) in which no sample is missing skip the is-missing matrix: a cohort with no missing
    genotype at all, one whose first two superblocks have none, one with a single missing-bearing row"""
    n, m, S = 700, 512, 8
    rng = np.random.default_rng(41)
    eaf_c = np.round(rng.uniform(0.01, 0.5, m), 4)
    miss = np.zeros(m)
    if where == "first_half":
        miss[256:] = rng.uniform(0.0, 0.08, 256)
    elif where == "one_row":
        miss[300] = 0.3
    th, tm, tmi = refcpu.hwe_thresholds(eaf_c, miss)
    _, _, _, _, descs = make_case(n, m, S, 3)
    codes = refcpu.synth_rows(n, 0, m, 5, th, tm, tmi)
    co = capi.Cohort(n, m, fmt=capi.FMT_GT2M)
    co.synth(0, 5, th, tm, tmi)
    nm, _ = co.row_tallies()
    assert (nm[:256] == 0).all() and ((nm > 0).any() == (where != "none"))
    for kw in (PARAM_GRID[0], PARAM_GRID[4]):
        msc = capi.MultiScorer(n, capi.make_params(**kw), S)
        mdef = capi.MultiDef(descs)
        msc.score_cohort(co, mdef)
        got, nloci = msc.finish(np.zeros(S))
        ref, ref_nloci = oracle_scores(codes, n, descs, kw, np.zeros(S))
        assert np.array_equal(nloci.astype(np.int64), ref_nloci)
        for s in range(S):
            keep = descs[s]["kind"] != capi.ROW_NOT_IN_SCORE
            assert rel_err(got[s], ref[s], float(np.sum(np.abs(descs[s]["beta"][keep]))), int(ref_nloci[s])) <= REL_TOL, s
        msc.close()
        mdef.close()
    co.close()


def test_multi_equals_single_score_path_medium():
    """8 definitions in one pass == the single-score fused kernel run 8 times, 60 000 samples x 8 192 rows"""
    n, m, S, seed = 60_000, 8192, 8, 20250104
    eaf_c, th, tm, tmi, descs = make_case(n, m, S, 9, with_kinds=False)
    gt = capi.Cohort(n, m)
    for r0 in range(0, m, 1 << 12):
        gt.synth(r0, seed, th[r0:r0 + (1 << 12)], tm[r0:r0 + (1 << 12)], tmi[r0:r0 + (1 << 12)])
    co = capi.Cohort(n, m, fmt=capi.FMT_GT2M)
    co.synth(0, seed, th, tm, tmi)
    msc = capi.MultiScorer(n, capi.make_params(), S)
    mdef = capi.MultiDef(descs)
    msc.score_cohort(co, mdef)
    got, nloci = msc.finish(np.zeros(S))
    msc.reset()
    msc.set_missing_weight_bits(32)
    msc.score_cohort(co, mdef)
    got32, nloci32 = msc.finish(np.zeros(S))
    msc.set_missing_weight_bits(56)
    assert np.array_equal(nloci, nloci32)
    for s in range(S):
        sc = capi.Scorer(n, capi.make_params())
        sc.score_cohort(gt, descs[s])
        one, nl = sc.finish(0.0)
        sc.close()
        assert nl == int(nloci[s])
        assert rel_err(got[s], one, float(np.sum(np.abs(descs[s]["beta"]))), nl) <= REL_TOL, s
        # 32-bit weights for the ~4 % missing genotypes of this cohort: still far inside the 1e-6 bar
        assert rel_err(got32[s], one, float(np.sum(np.abs(descs[s]["beta"]))), nl) <= 1e-6, s
    # converting the row-major cohort gives the same units as generating them directly
    co2 = capi.Cohort(n, m, fmt=capi.FMT_GT2M)
    co2.convert_from(gt)
    msc.reset()
    msc.score_cohort(co2, mdef)
    got2, _ = msc.finish(np.zeros(S))
    assert np.array_equal(got, got2, equal_nan=True)
    assert all(np.array_equal(a, b) for a, b in zip(co.row_tallies(), co2.row_tallies()))
    for x in (msc, mdef, co, co2, gt):
        x.close()


def test_multi_bad_arguments():
    co = capi.Cohort(64, 128, fmt=capi.FMT_GT2M)
    gt = capi.Cohort(64, 128)
    msc = capi.MultiScorer(64, capi.make_params(), 2)
    d = np.zeros((2, 128), dtype=capi.ROW_DESC_DTYPE)
    mdef = capi.MultiDef(d)
    with pytest.raises(capi.NpsError):
        msc.score_cohort(gt, mdef)                    # wrong cohort format
    with pytest.raises(capi.NpsError):
        msc.score_cohort(co, mdef, 64)                # not on a superblock boundary
    with pytest.raises(capi.NpsError):
        capi.MultiDef(np.zeros((9, 4), dtype=capi.ROW_DESC_DTYPE))   # more than 8 scores
    with pytest.raises(capi.NpsError):
        capi.MultiDef(d, weight_bits=40)              # 41 or 49 (nps_multidef_create_bits)
    bad = d.copy()
    bad["beta"][0, 0] = np.inf
    with pytest.raises(capi.NpsError):
        capi.MultiDef(bad)
    sc = capi.Scorer(64, capi.make_params())
    with pytest.raises(capi.NpsError):
        sc.score_cohort(co, d[0])                     # the single-score kernels need FMT_GT2
    with pytest.raises(capi.NpsError):
        co.upload(0, np.zeros((128, 4), dtype=np.uint32))
    for x in (sc, msc, mdef, co, gt):
        x.close()


def test_multi_partial_sums_of_row_blocks_add_up():
    """rows sharded x all S scores per GPU: two contexts score the two 128-aligned row blocks of
    the cohort; the un-normalised sums and nloci (nps_multi_partial_device) added up and normalised
    (multi.normalize_matrix) equal the one-context run"""
    import torch
    from nimpress_amd import multi
    n, m, S = 20_000, 640, 5
    eaf_c, th, tm, tmi, descs = make_case(n, m, S, 4321, with_kinds=True)
    co = capi.Cohort(n, m, fmt=capi.FMT_GT2M)
    co.synth(0, 4321, th, tm, tmi)
    offsets = np.arange(S) * 0.25
    whole = capi.MultiScorer(n, capi.make_params(), S)
    mdef = capi.MultiDef(descs)
    whole.score_cohort(co, mdef)
    ref, ref_nloci = whole.finish(offsets)
    whole.close()
    mdef.close()
    sums = torch.zeros((S, n), dtype=torch.float64, device="cuda")
    total = np.zeros(S, np.int64)
    for r0, r1 in ((0, 384), (384, m)):
        part = capi.MultiScorer(n, capi.make_params(), S)
        d = capi.MultiDef(np.ascontiguousarray(descs[:, r0:r1]))
        part.score_cohort(co, d, r0)
        buf = torch.empty((S, n), dtype=torch.float64, device="cuda")
        total += part.partial_device(buf.data_ptr()).astype(np.int64)
        sums += buf
        part.close()
        d.close()
    got = multi.normalize_matrix(sums, torch.from_numpy(total), offsets).cpu().numpy()
    assert np.array_equal(total, ref_nloci.astype(np.int64))
    assert np.allclose(got, ref, rtol=0, atol=1e-12 * float(np.abs(descs["beta"]).sum()), equal_nan=True)
    co.close()


def test_multi_beta_span_is_refused_not_mis_scored():
    """the multi-score weights have ONE fixed-point scale per score; a definition whose |beta| span more
    than 2^25 (2^17 with 41-bit weights) would silently lose the samples that carry only its small-beta rows.
    nps_multidef_create refuses it and names the single-score path (which bands such a definition and holds the plain
    1e-6 bar on exactly this span: tests/test_gpu_mx.py::test_gt2x_beta_span_plain_relative_bar); the same eight
    definitions with the wide one narrowed are accepted, and rows with beta = 0 do not count towards the span."""
    m, S = 256, 8
    rng = np.random.default_rng(99)
    d = np.zeros((S, m), dtype=capi.ROW_DESC_DTYPE)
    for s in range(S):
        d[s]["beta"] = np.round(rng.normal(0, 0.02, m), 4) + 1e-4
        d[s]["eaf"] = 0.3
    wide = d.copy()
    wide["beta"][5] = np.concatenate([rng.uniform(0.1, 1.0, m // 2) * 10.0, rng.uniform(0.1, 1.0, m // 2) * 1e-9])
    with pytest.raises(capi.NpsError) as e:
        capi.MultiDef(wide)
    assert e.value.status == capi.E_UNSUPPORTED and "score 5" in str(e.value) and "single-score path" in str(e.value)
    ok = d.copy()
    ok["beta"][5, : m // 2] = 0.0                       # zeros are not part of the span
    capi.MultiDef(ok).close()
    edge = d.copy()
    edge["beta"][2] = 1.0
    edge["beta"][2, 0] = 2.0 ** -25                     # exactly at the limit: accepted
    capi.MultiDef(edge).close()
    edge["beta"][2, 0] = 2.0 ** -26
    with pytest.raises(capi.NpsError):
        capi.MultiDef(edge)
    edge["beta"][2, 0] = 2.0 ** -17                     # six digits: 2^17
    capi.MultiDef(edge, weight_bits=41).close()
    edge["beta"][2, 0] = 2.0 ** -18
    with pytest.raises(capi.NpsError):
        capi.MultiDef(edge, weight_bits=41)
