"""End-to-end runs of the `nimpress` command line (C++ host + libnps on the GPU) on the reference's
own fixtures: the 13 golden cases of tests/test_set1.nim through the real flags, CLI defaults, and
the warning text."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import refcpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(ROOT, "nimpress_amd", "nimpress")


def run_cli(*flags, env=None, vcf="set1.vcf.gz"):
    r = subprocess.run([CLI, *flags, os.path.join(G, "set1.score"), os.path.join(G, vcf)],
                       capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr
    warns = [l for l in r.stdout.splitlines() if l.startswith("WARN ")]
    rows = [l.split("\t") for l in r.stdout.splitlines() if not l.startswith(("WARN ", "FATAL "))]
    return [x[0] for x in rows], [float(x[1]) for x in rows], [x[1] for x in rows], warns


@pytest.mark.parametrize("idx", range(13))
def test_cli_set1_golden(idx):
    case = json.load(open(os.path.join(G, "set1_cases.json")))["cases"][idx]
    flags = ["--imp-locus=" + case["imp_locus"], "--imp-missing=" + case["imp_missing"],
             "--imp-sample=" + case["imp_sample"], "--maxmis=%r" % case["maxmis"],
             "--mincs=%d" % case["mincs"], "--afmisp=%r" % case["afmisp"]]
    if case["restrict_to_covered"]:
        flags.append("--cov=" + os.path.join(G, "set1.bed"))
    if case["ignore_filter"]:
        flags.append("--ignorefilt")
    names, vals, texts, warns = run_cli(*flags)
    assert names == ["S1", "S2", "S3", "S4", "S5", "S6"]
    for got, exp in zip(vals, case["expected"]):
        assert (exp is None) == bool(np.isnan(got))
        if exp is not None:
            assert abs(got - exp) <= 1e-4
    # text format = the oracle's rendering of the oracle's numbers (they agree to the last digit here)
    score = refcpu.read_score_file(os.path.join(G, "set1.score"))
    vcf = refcpu.read_vcf(os.path.join(G, "set1.vcf.gz"))
    bed = refcpu.read_bed(os.path.join(G, "set1.bed"))
    ref, _, _ = refcpu.compute_polygenic_scores(score, vcf, case["restrict_to_covered"], bed,
                                                case["imp_locus"], case["imp_missing"],
                                                case["imp_sample"], case["maxmis"], case["mincs"],
                                                case["ignore_filter"])
    for t, r in zip(texts, ref):
        assert t == refcpu.format_score(r) or abs(float(t) - r) <= 1e-12


@pytest.mark.parametrize("idx", range(13))
def test_cli_set1_golden_on_split_records(idx):
    """the reference's own split of set1 (tests/set1.plink.vcf.gz + .tbi, written by bcftools / tabix; the file
    set1.plink190.result was computed from): the 13 golden vectors through the command line, with tabix random
    access and by whole-file scan.  The score row 1:300 GA/CT must take the second record at that position."""
    case = json.load(open(os.path.join(G, "set1_cases.json")))["cases"][idx]
    flags = ["--imp-locus=" + case["imp_locus"], "--imp-missing=" + case["imp_missing"],
             "--imp-sample=" + case["imp_sample"], "--maxmis=%r" % case["maxmis"],
             "--mincs=%d" % case["mincs"], "--afmisp=%r" % case["afmisp"]]
    if case["restrict_to_covered"]:
        flags.append("--cov=" + os.path.join(G, "set1.bed"))
    if case["ignore_filter"]:
        flags.append("--ignorefilt")
    base = run_cli(*flags)
    for env in ({}, {"NIMPRESS_NO_INDEX": "1"}):
        names, vals, texts, warns = run_cli(*flags, env=env, vcf="set1.plink.vcf.gz")
        assert names == ["S1", "S2", "S3", "S4", "S5", "S6"]
        for got, exp in zip(vals, case["expected"]):
            assert (exp is None) == bool(np.isnan(got))
            if exp is not None:
                assert abs(got - exp) <= 1e-4
        assert texts == base[2]      # the same numbers, to the last printed digit, as on the unsplit file


def test_cli_plink190_on_the_file_plink_read():
    plink = [float(l.split()[5]) for l in open(os.path.join(G, "set1.plink190.result")).read().splitlines()[1:]]
    _, vals, _, _ = run_cli("--imp-locus=ignore", "--imp-missing=ignore", "--imp-sample=int_ps", "--maxmis=1.0",
                            "--mincs=0", "--afmisp=1.0", "--ignorefilt", vcf="set1.plink.vcf.gz")
    assert np.allclose(vals, [0.123 + p for p in plink], atol=1e-4)


def test_cli_streaming_windows_and_whole_file_agree():
    """the command line streams indexed files window by window of score rows (here: 2 rows a window, 3
    fetch threads); the output must equal the run that reads the whole file (no index)"""
    flags = ["--imp-locus=ps", "--imp-sample=int_ps", "--maxmis=1.0", "--mincs=3"]
    a = run_cli(*flags, env={"NIMPRESS_WINDOW": "2", "NIMPRESS_THREADS": "3"})
    b = run_cli(*flags, env={"NIMPRESS_NO_INDEX": "1"})
    c = run_cli(*flags)
    assert a[2] == b[2] == c[2] and a[3] == b[3] == c[3]


def test_cli_defaults_and_warnings():
    names, vals, texts, warns = run_cli()
    assert np.allclose(vals, 0.1545, atol=1e-12)
    # every row is locus-imputed under the defaults; the messages are the reference's (nim:554-570)
    assert any('has a FILTER flag set (value "FAIL")' in w for w in warns)
    assert any("1:100-100 has 16.66666666666667% of samples missing a genotype" in w or
               "1:100-100 has 16.666666666666" in w for w in warns)
    names, vals, texts, warns = run_cli("--cov", os.path.join(G, "set1.bed"), "--maxmis", "1.0",
                                        "--imp-sample", "ps", "--afmisp", "1.0")
    assert np.allclose(vals, [0.081, 0.081, 0.081, 0.1545, 0.006, 0.006], atol=1e-4)
    assert sum("is not covered by the sequence coverage BED" in w for w in warns) == 3


@pytest.mark.parametrize("gt_dtype", [np.int8, np.int16])
def test_cli_on_bcf_equals_cli_on_vcf_bcf_parity_unpinned(tmp_path, gt_dtype):
    """the fixture re-written as BCF2 (+CSI) by tests/bcfwriter.py: the typed GT vectors go to the
    device as they stand in the file (nps_push_gt_raw); output text identical to the vcf.gz run"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bcfwriter
    vcf = refcpu.read_vcf(os.path.join(G, "set1.vcf.gz"))
    contigs, recs = bcfwriter.records_from_oracle_vcf(vcf)
    path = str(tmp_path / "set1.bcf")
    bcfwriter.write_bcf(path, contigs, vcf.samples, recs, gt_dtype=gt_dtype)
    for flags in (["--imp-locus=ps"], ["--imp-locus=ignore", "--imp-missing=ignore", "--imp-sample=int_ps",
                                      "--maxmis=1.0", "--mincs=0", "--ignorefilt"],
                  ["--cov=" + os.path.join(G, "set1.bed"), "--imp-sample=ps", "--maxmis=1.0"]):
        a = subprocess.run([CLI, *flags, os.path.join(G, "set1.score"), path], capture_output=True, text=True)
        b = subprocess.run([CLI, *flags, os.path.join(G, "set1.score"), os.path.join(G, "set1.vcf.gz")],
                           capture_output=True, text=True)
        assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
        assert a.stdout == b.stdout


def test_cli_on_plink_filesets_equals_cli_on_vcf_pgen_parity_unpinned(tmp_path):
    """the same biallelic calls as text VCF, as a PLINK 1 fileset (.bed/.bim/.fam, A1 = ALT, A2 = REF, and for every
    third variant the other way round) and as a PLINK 2 fixed-width .pgen (+ .pvar with and without header line,
    .psam with and without FID): identical output.  The .pgen reader's conformance is PARITY UNPINNED (see below)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    rng = np.random.default_rng(4)
    n, m = 203, 57
    names = ["I%03d" % i for i in range(n)]
    vcf = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names)]
    bim, bed = [], bytearray(b"\x6c\x1b\x01")
    pvars, alt_counts, missing = [], [], []
    score = ["t", "", "", "x", "0.05"]
    pos = 100
    for j in range(m):
        pos += int(rng.integers(1, 500))
        contig = "1" if j < 30 else "7"
        ref, alt = ("A", "G") if j % 5 else ("AT", "A")
        nalt = rng.integers(0, 3, n)
        miss = rng.uniform(size=n) < (0.3 if j % 11 == 0 else 0.03)
        gt = ["./." if miss[i] else ("0/0", "0/1", "1/1")[nalt[i]] for i in range(n)]
        vcf.append("%s\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t%s" % (contig, pos, ref, alt, "\t".join(gt)))
        swap = j % 3 == 0                      # .bim lists the alleles the other way round
        a1, a2 = (ref, alt) if swap else (alt, ref)
        n_a1 = (2 - nalt) if swap else nalt
        code = np.where(miss, 1, np.select([n_a1 == 2, n_a1 == 1], [0, 2], 3))
        code = np.concatenate([code, np.zeros((-n) % 4, dtype=code.dtype)]).reshape(-1, 4)
        bed += bytes((code[:, 0] | (code[:, 1] << 2) | (code[:, 2] << 4) | (code[:, 3] << 6)).astype(np.uint8))
        bim.append("%s\trs%d\t0\t%d\t%s\t%s" % (contig, j, pos, a1, a2))
        pvars.append((contig, pos, "rs%d" % j, ref, alt))
        alt_counts.append(nalt)
        missing.append(miss)
        if j % 4 != 3:                         # every 4th variant is not in the score
            ea = alt if j % 2 else ref         # effect allele = ALT or REF
            score.append("%s\t%d\t%s\t%s\t%.4f\t%.4f" % (contig, pos, ref, ea, rng.normal(0, 0.1),
                                                          rng.uniform(0.05, 0.5)))
    score.append("9\t5\tC\tT\t0.1\t0.2")        # absent locus
    (tmp_path / "c.vcf").write_text("\n".join(vcf) + "\n")
    (tmp_path / "c.bed").write_bytes(bytes(bed))
    (tmp_path / "c.bim").write_text("\n".join(bim) + "\n")
    (tmp_path / "c.fam").write_text("".join("F%d %s 0 0 0 -9\n" % (i, nm) for i, nm in enumerate(names)))
    (tmp_path / "s.score").write_text("\n".join(score))
    # ... and as a PLINK 2 .pgen (storage mode 0x02, fixed-width hard calls) + .pvar + .psam.  PARITY UNPINNED: the
    # reader is pinned by the build's own writer (tests/pgenwriter.py) alone -- no plink2, no specification and no
    # .pgen fixture of the reference's exist in this image; what this shows is that the three inputs agree.
    import pgenwriter
    pgenwriter.write_pgen(str(tmp_path / "p"), names, pvars, np.array(alt_counts), np.array(missing))
    pgenwriter.write_pgen(str(tmp_path / "q"), names, pvars, np.array(alt_counts), np.array(missing), pvar_header=False,
                          psam_fid=True)
    for flags in ([], ["--imp-locus=homref", "--imp-sample=int_fail", "--maxmis=0.1", "--mincs=10"],
                  ["--imp-locus=ignore", "--imp-missing=ignore", "--imp-sample=ps"]):
        a = subprocess.run([CLI, *flags, str(tmp_path / "s.score"), str(tmp_path / "c.bed")],
                           capture_output=True, text=True)
        b = subprocess.run([CLI, *flags, str(tmp_path / "s.score"), str(tmp_path / "c.vcf")],
                           capture_output=True, text=True)
        assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
        assert a.stdout == b.stdout
        assert len(a.stdout.splitlines()) >= n
        for pg in ("p.pgen", "q.pgen"):
            c = subprocess.run([CLI, *flags, str(tmp_path / "s.score"), str(tmp_path / pg)], capture_output=True, text=True)
            assert c.returncode == 0, c.stderr
            assert c.stdout == b.stdout, pg


def test_config2_wood_height_on_100k_sample_bcf(tmp_path):
    """BASELINE.json configs[1] at its full size: scores/wood-25282103-height.scores (697 loci, 188 of
    them with the REF allele as effect allele) on a synthetic 100 000-sample BCF2 (+CSI) written by
    tests/bcfwriter.py -- HWE genotypes at the row's eaf, per-row missing rate U(0, 0.02), a few loci
    absent or FILTER-failed -- through the `nimpress` command line (CSI random access per locus, the
    record's int8 GT vector decoded on the device), against the oracle fed with the same arrays.
    Prints the end-to-end wall time of the command (host decode dominates; not asserted)."""
    import sys
    import time
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import config2
    wood = os.path.join(G, "scores", "wood-25282103-height.scores")
    score = refcpu.read_score_file(wood)
    path, n_rec, n, samples, truth_by_row = config2.write_cohort(tmp_path, wood)
    truth = [(e,) + truth_by_row[k] for k, e in enumerate(score.entries)]
    t0 = time.perf_counter()
    r = subprocess.run([CLI, "--afmisp=0", wood, path], capture_output=True, text=True)
    wall = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l.split("\t") for l in r.stdout.splitlines() if not l.startswith(("WARN ", "FATAL "))]
    assert [x[0] for x in rows] == samples
    got = np.array([float(x[1]) for x in rows])
    # the oracle, in score-file order, with the reference's early returns
    by_entry = {id(e): (g, f) for e, g, f in truth}
    sc = refcpu.RefScorer(n, refcpu.make_params())        # CLI defaults
    for e in score.entries:
        g, f = by_entry[id(e)]
        rie = e.easeq == e.refseq
        if g is None:
            sc.row_locus(2, rie, e.beta, e.eaf)           # absent
        elif f == ["FAIL"]:
            sc.row_locus(3, rie, e.beta, e.eaf)           # FILTER
        else:
            sc.row_gt(g.reshape(-1), 2, 0 if rie else 1, rie, e.beta, e.eaf)
    ref, nloci = sc.finish(score.offset)
    assert nloci == len(score.entries)
    from test_gpu_parity import rel_err   # per sample: 1e-6 relative, |ref| floored as SURVEY.md 8(d) prescribes
    assert rel_err(got, ref, np.array([e.beta for e in score.entries]), nloci) <= 1e-6
    print("\n[config 2] nimpress on a 100000-sample BCF, %d loci (%d records, %.0f MB of int8 GT): %.2f s end to end "
          "= %.3g genotypes/s" % (len(score.entries), n_rec, n_rec * n * 2 / 1e6, wall, n_rec * n / wall))


def test_af_mismatch_warnings_at_default_afmisp_100k_samples(tmp_path):
    """SURVEY.md section 8 f3, end to end: the command line at its DEFAULT --afmisp=0.001 on a
    100 000-sample BCF whose cohort allele counts sit at chosen distances from the score file's eaf.  The
    rows warned about, and the text of every warning (nimpress.nim:538-541, 575-579), must be exactly what
    the oracle's literal binomTest (nim:155-188, the O(n) enumeration) decides -- including
      * both sides of the 0.001 threshold on both sides of the mean,
      * a row near the mean where betacf runs out of its 100 iterations (nim:117): the p-value is NaN,
        `NaN < afmisp` is false, no warning,
      * a row exactly at the mean (p = 1), a row whose eaf is NaN (test skipped, nim:573),
      * absent variants (binomTest(0, 2N, eaf), nim:537) with a plausible and an implausible eaf,
      * a row over --maxmis (locus-imputed BEFORE the test is reached, nim:565-571) and a FILTER-failed one,
      * a row with the REF allele as effect allele, and one with missing samples (2 * ngenotyped trials)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bcfwriter
    n = 100_000
    # (pos, ref, ea, eaf, ALT allele count, missing samples, FILTER, in file?)
    rows = [
        (1000, "A", "C", 0.30, 60000, 0, [], True),        # exactly the expectation: p = 1
        (1100, "A", "C", 0.30, 60050, 0, ["PASS"], True),  # near the mean: betacf -> NaN -> no warning
        (1200, "A", "C", 0.30, 60600, 0, [], True),        # p = 0.0034
        (1300, "A", "C", 0.30, 60700, 0, [], True),        # p = 0.00065 -> warned
        (1400, "A", "C", 0.30, 59300, 0, [], True),        # p = 0.00063 -> warned (x below the mean)
        (1500, "A", "C", 0.30, 59400, 0, [], True),        # p = 0.0034
        (1600, "G", "C", 0.10, 40000, 0, [], True),        # far off: p = 0 -> warned
        (1700, "A", "A", 0.70, 60700, 0, [], True),        # effect allele = REF, 139 300 REF alleles: warned
        (1800, "A", "C", 0.30, 0, 0, [], False),           # absent, eaf 0.3: "cohort EAF is 0" warned
        (1900, "A", "C", 1e-7, 0, 0, [], False),           # absent, eaf 1e-7: p = 1
        (2000, "A", "C", float("nan"), 90000, 0, [], True),  # eaf NaN: no test
        (2100, "A", "C", 0.30, 10000, 10000, [], True),    # 10 % missing: over --maxmis, no AF test
        (2200, "A", "C", 0.30, 60120, 1000, [], True),     # 1 % missing: 198 000 trials, p = 0.00042 -> warned
        (2300, "A", "C", 0.30, 60000, 1000, [], True),     # 1 % missing: p = 0.0033
        (2400, "A", "C", 0.30, 10000, 0, ["FAIL"], True),  # FILTER: locus-imputed, no AF test
        (2500, "A", "G", 0.30, 60700, 0, [], True),        # ea not among the ALT alleles -> absent -> warned
    ]
    lines = ["af-mismatch", "", "", "GRCh37", "0.5"]
    recs = []
    for pos, ref, ea, eaf, x, nmiss, filt, present in rows:
        lines.append("7\t%d\t%s\t%s\t0.01\t%s" % (pos, ref, ea, "NaN" if eaf != eaf else repr(eaf)))
        if not present:
            continue
        a = np.zeros(2 * n, dtype=np.int64)
        a[:x] = 1                                          # x ALT alleles, at the front
        gts = ((a + 1) << 1).reshape(n, 2)
        if nmiss:
            gts[n - nmiss:] = 0                            # missing samples at the back (REF/REF before)
        recs.append(dict(contig="7", pos=pos, id=".", ref=ref, alts=["C"], filters=filt, gts=gts))
    (tmp_path / "af.score").write_text("\n".join(lines))
    path = str(tmp_path / "af.bcf")
    samples = ["S%06d" % i for i in range(n)]
    bcfwriter.write_bcf(path, ["7"], samples, recs, gt_dtype=np.int8)
    r = subprocess.run([CLI, str(tmp_path / "af.score"), path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    warns = [l[5:] for l in r.stdout.splitlines() if l.startswith("WARN ")]
    assert sum(1 for l in r.stdout.splitlines() if not l.startswith("WARN ")) == n

    # the oracle's decisions and texts, in score-file order
    fmt = refcpu.format_score
    expected, n_nan = [], 0
    for pos, ref, ea, eaf, x, nmiss, filt, present in rows:
        var = "7:%d:%s:%s" % (pos, ref, ea)
        if not present or (ea != ref and ea != "C"):       # findVariant returns nil (nim:536-541)
            if eaf == eaf and refcpu.binom_test(0, 2 * n, eaf) < 0.001:
                expected.append("Variant %s cohort EAF is 0 in %d samples.  This is highly unlikely given "
                                "polygenic score EAF of %s" % (var, n, fmt(eaf)))
            continue
        if filt == ["FAIL"]:
            expected.append('Variant %s has a FILTER flag set (value "FAIL").  Imputing all dosages at this locus.' % var)
            continue
        if nmiss / n > 0.05:
            expected.append("Locus 7:%d-%d has %s%% of samples missing a genotype. This exceeds the missingness "
                            "threshold; imputing all dosages at this locus." % (pos, pos, fmt(nmiss / n * 100)))
            continue
        neff = (2 * (n - nmiss) - x) if ea == ref else x   # tallyAlleles counts the EFFECT allele
        nobs = (n - nmiss) * 2
        if eaf == eaf:
            p = refcpu.binom_test(neff, nobs, eaf)
            n_nan += int(p != p)
            if p < 0.001:
                expected.append("Variant %s cohort EAF is %s in %d samples.  This is highly unlikely given "
                                "polygenic score EAF of %s" % (var, fmt(neff / nobs), n, fmt(eaf)))
    assert n_nan == 1                                      # the MAXIT case really is in the fixture
    assert sum("cohort EAF is" in w for w in expected) == 7
    assert warns == expected


# ------------------------------------------------------------------------------------------
# FORMAT/DS through the host reader and the command line (build-defined extension; oracle = ref_row_ds)
def _ds_cohort(n, m, seed):
    """m score rows / records of float32 ALT dosages (HWE genotype + N(0, 0.05), clipped to [0, 2], 3 decimals),
    per-row missing rate U(0, 0.1) (about half the rows over --maxmis 0.05), some rows with the REF allele as the
    effect allele, with eaf = NaN, absent from the file or FILTER-failed"""
    rng = np.random.default_rng(seed)
    entries, recs = [], []
    for j in range(m):
        pos = 1000 + 37 * j
        eaf = float(np.round(rng.uniform(0.02, 0.5), 4))
        beta = float(np.round(rng.normal(0, 0.05), 4))
        rie = j % 5 == 2
        ref, alt = "A", "G"
        entries.append(refcpu.ScoreEntry("7", pos, ref, ref if rie else alt, beta, float("nan") if j % 9 == 4 else eaf))
        if j % 13 == 6:
            continue                                   # absent
        g = (rng.random(n) < eaf).astype(np.float32) + (rng.random(n) < eaf).astype(np.float32)
        d = np.round(np.clip(g + rng.normal(0, 0.05, n), 0, 2), 3).astype(np.float32)
        nan_eaf = j % 9 == 4    # (such a row must stay genotyped: locus imputation with ps would make every score NaN)
        d[rng.random(n) < (0.01 if nan_eaf else rng.uniform(0, 0.1))] = np.nan
        recs.append(refcpu.VcfRecord("7", pos, ref, [alt], "FAIL" if (j % 17 == 3 and not nan_eaf) else ("PASS" if j % 2 else "."),
                                     None, 0, d.reshape(n, 1)))
    return entries, recs


def _write_ds_files(tmp_path, n, entries, recs, text):
    import bcfwriter
    samples = ["D%06d" % i for i in range(n)]
    spath = str(tmp_path / "ds.score")
    with open(spath, "w") as f:
        f.write("ds test\n\n\nGRCh37\n0.25\n")
        f.write("\n".join("%s\t%d\t%s\t%s\t%r\t%s" % (e.contig, e.pos, e.refseq, e.easeq, e.beta,
                                                     "NaN" if np.isnan(e.eaf) else repr(e.eaf)) for e in entries))
    if text:
        import gzip
        path = str(tmp_path / "ds.vcf.gz")
        with gzip.open(path, "wt") as f:
            f.write("##fileformat=VCFv4.2\n##contig=<ID=7>\n"
                    '##FORMAT=<ID=DS,Number=A,Type=Float,Description="ALT allele dosage">\n')
            f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples) + "\n")
            for r in recs:
                vals = ["." if np.isnan(x) else ("%.3f" % x).rstrip("0").rstrip(".") for x in r.ds[:, 0]]
                f.write("%s\t%d\t.\t%s\t%s\t.\t%s\t.\tDS\t%s\n" % (r.contig, r.pos, r.ref, ",".join(r.alts), r.filt,
                                                                   "\t".join(vals)))
    else:
        path = str(tmp_path / "ds.bcf")
        bcfwriter.LEVEL = 1
        bcfwriter.write_bcf(path, ["7"], samples,
                            [dict(contig=r.contig, pos=r.pos, id=".", ref=r.ref, alts=r.alts,
                                  filters=[] if r.filt == "." else [r.filt], gts=None, ds=r.ds) for r in recs])
        bcfwriter.LEVEL = 6
    return spath, path, samples


@pytest.mark.parametrize("kind,n", [("bcf", 100_000), ("text", 20_000)])
def test_cli_format_ds_equals_oracle(tmp_path, kind, n):
    """a 100 000-sample BCF2 whose records carry FORMAT/DS only (typed float vectors, missing = 0x7F800001), and a
    text vcf.gz with the same kind of rows: the command line scores them through nps_push_ds; scores, nloci (through
    the scores) and the warnings' rows equal the oracle's ref_row_ds on the same rows -- rows over --maxmis, NaN
    eaf, REF-effect rows (2 - DS), absent and FILTER-failed records included"""
    m = 60
    entries, recs = _ds_cohort(n, m, 77 + n)
    spath, path, samples = _write_ds_files(tmp_path, n, entries, recs, kind == "text")
    score = refcpu.ScoreFile("ds test", "", "", "GRCh37", 0.25, entries)
    vcf = refcpu.Vcf(samples=samples, records=recs)
    for flags, args in ((["--imp-locus=ps", "--afmisp=0"], ("ps", "homref", "int_ps", 0.05, 100, False)),
                        (["--imp-locus=homref", "--imp-sample=ps", "--maxmis=1.0", "--afmisp=0", "--ignorefilt"],
                         ("homref", "homref", "ps", 1.0, 100, True))):
        r = subprocess.run([CLI, *flags, spath, path], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        rows = [l.split("\t") for l in r.stdout.splitlines() if not l.startswith(("WARN ", "FATAL "))]
        assert [x[0] for x in rows] == samples
        got = np.array([float(x[1]) for x in rows])
        ref, nloci, stats = refcpu.compute_polygenic_scores(score, vcf, False, None, *args)
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        ok = ~np.isnan(ref)
        assert ok.sum() > n // 2          # (samples missing at a NaN-eaf row are NaN under --imp-sample=ps)
        scale = sum(abs(e.beta) for e in entries) / (2.0 * max(nloci, 1))
        assert np.max(np.abs(got[ok] - ref[ok])) <= 1e-9 * scale, (kind, flags)
        # the FILTER / maxmis warnings name the same rows as the oracle's decisions
        over = sum(1 for s_ in stats if s_[4] == 4)
        assert sum("missingness" in l or "missing" in l.lower() for l in r.stdout.splitlines() if l.startswith("WARN ")) >= (1 if over else 0)
