"""BASELINE.json configs[1] as a fixture: scores/wood-25282103-height.scores (697 loci, 188 of them with
the REF allele as effect allele) on a synthetic 100 000-sample BCF2 (+CSI) written by tests/bcfwriter.py
-- HWE genotypes at the row's eaf, per-row missing rate U(0, 0.02), a few loci absent or FILTER-failed.
Used by tests/test_gpu_cli.py (checked against the oracle) and by bench.py (end-to-end wall time of the
command line).  Test infrastructure: builds inputs, computes nothing that is shipped."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def parse_score_entries(path):
    """(contig, pos, ref, ea, beta, eaf) of every row (nimpress.nim:247-254: 5 header lines, 6 columns)"""
    out = []
    with open(path) as f:
        lines = f.read().split("\n")
    for line in lines[5:]:
        line = line.rstrip()
        if not line:
            continue
        c = line.split("\t")
        out.append((c[0], int(c[1]), c[2], c[3], float(c[4]), float(c[5])))
    return out


def write_cohort(dirname, score_path, n=100_000, seed=20250102, level=6):
    """Writes cohort.bcf (+ .csi) for the score file; returns (path, number of records, n, samples, truth)
    where truth[j] = (score row index, int32 GT array [n, 2] or None if absent, FILTER list)."""
    import bcfwriter
    bcfwriter.LEVEL = level
    entries = parse_score_entries(score_path)
    rng = np.random.default_rng(seed)
    samples = ["S%06d" % i for i in range(n)]
    contigs, recs, truth = [], [], {}

    def ckey(k):
        e = entries[k]
        return ((0, int(e[0])) if e[0].isdigit() else (1, e[0])), e[1]
    for j, k in enumerate(sorted(range(len(entries)), key=ckey)):
        contig, pos, ref, ea, _beta, eaf = entries[k]
        if contig not in contigs:
            contigs.append(contig)
        if j % 41 == 7:
            truth[k] = (None, None)                       # absent from the file
            continue
        rie = ea == ref
        alt = ea if not rie else next(b for b in "ACGT" if b != ref[0])
        p_alt = (1.0 - eaf) if rie else eaf               # eaf is the frequency of the EFFECT allele
        p_alt = min(max(p_alt, 0.0), 1.0)
        a = (rng.uniform(size=(n, 2)) < p_alt).astype(np.int64)
        gts = (a + 1) << 1                                # bcf GT encoding, unphased
        gts[rng.uniform(size=n) < rng.uniform(0.0, 0.02)] = 0   # both alleles missing
        filt = ["FAIL"] if j % 53 == 11 else (["PASS"] if j % 2 else [])
        recs.append(dict(contig=contig, pos=pos, id=".", ref=ref, alts=[alt], filters=filt, gts=gts))
        truth[k] = (gts.astype(np.int32), filt)
    path = os.path.join(str(dirname), "cohort.bcf")
    bcfwriter.write_bcf(path, contigs, samples, recs, gt_dtype=np.int8)
    bcfwriter.LEVEL = 6
    return path, len(recs), n, samples, truth


def write_union_cohort(dirname, score_paths, n, seed=20250104, level=1):
    """BASELINE.json configs[3]: ONE cohort holding the union of the loci of several score files (one record
    per (contig, pos, ref); ALT = the effect alleles the files name there), HWE genotypes, per-row missing
    rate U(0, 0.02), a few loci absent / FILTER-failed / with a second ALT allele, as int8 BCF2 + CSI.
    Returns (path, samples, records) with records = [dict(contig, pos, ref, alts, filt, gts int8 [n, 2])] in
    file order (what the oracle's findVariant scans)."""
    import bcfwriter
    bcfwriter.LEVEL = level
    rng = np.random.default_rng(seed)
    loci = {}
    for sp in score_paths:
        for contig, pos, ref, ea, _beta, eaf in parse_score_entries(sp):
            loci.setdefault((contig, pos, ref), []).append((ea, eaf))
    samples = ["P%06d" % i for i in range(n)]

    def ckey(k):
        return ((0, int(k[0])) if k[0].isdigit() else (1, k[0])), k[1]
    contigs, recs = [], []
    for j, key in enumerate(sorted(loci, key=ckey)):
        contig, pos, ref = key
        if contig not in contigs:
            contigs.append(contig)
        if j % 23 == 5:
            continue                                      # absent from the file
        alts = sorted({ea for ea, _ in loci[key] if ea != ref})
        if not alts:
            alts = [next(b for b in "ACGT" if b != ref[0])]
        if j % 17 == 3:
            alts = alts + [next(b for b in "ACGT" if b not in alts and b != ref[0]) + "T"]
        ea0, eaf0 = loci[key][0]
        p_alt = eaf0 if (ea0 != ref and not np.isnan(eaf0)) else 0.2
        p_alt = min(max(p_alt, 0.01), 0.99)
        a = (rng.random((n, 2), dtype=np.float32) < p_alt).astype(np.int8)
        gts = ((a + 1) << 1).astype(np.int8)              # bcf GT encoding, unphased: 0/0, 0/1, 1/1 of ALT 1
        gts[rng.random(n, dtype=np.float32) < (0.3 if j % 11 == 0 else rng.uniform(0.0, 0.02))] = 0
        filt = ["FAIL"] if j % 29 == 7 else (["PASS"] if j % 2 else [])
        recs.append(dict(contig=contig, pos=pos, id=".", ref=ref, alts=alts, filters=filt, gts=gts))
    path = os.path.join(str(dirname), "union.bcf")
    bcfwriter.write_bcf(path, contigs, samples, recs, gt_dtype=np.int8)
    bcfwriter.LEVEL = 6
    return path, samples, recs
