"""BASELINE.json configs[1] and [3] at test size: the bundled score definitions (wood height, ~700
loci, and the other 7 score-format files of the reference tree) evaluated on a synthetic cohort
written as a BGZF-compatible vcf.gz, through the C++ host + libnps, against the oracle's driver.
Multi-score evaluation uses nimpress_amd.multi (sharding + gather; world size 1 on this box)."""
import gzip
import os

import numpy as np
import pytest
import torch

from nimpress_amd import host, multi
from oracle import refcpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
SCORES = sorted(os.path.join(G, "scores", f) for f in os.listdir(os.path.join(G, "scores"))) + \
    [os.path.join(G, "set1.score")]
BASES = "ACGT"


def write_cohort_vcf(path, score_files, n, seed):
    """Union of the loci of all score files; HWE genotypes at the row's eaf; a few loci dropped,
    filtered, multi-allelic, phased or partly missing."""
    rng = np.random.default_rng(seed)
    loci = {}
    for sf in score_files:
        for e in refcpu.read_score_file(sf).entries:
            loci.setdefault((e.contig, e.pos, e.refseq), []).append(e)
    samples = ["P%05d" % i for i in range(n)]
    lines = ["##fileformat=VCFv4.2", '##FILTER=<ID=LowQual,Description="x">',
             '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
             "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples)]
    def ckey(k):
        c = k[0]
        return (0, int(c)) if c.isdigit() else (1, c), k[1]
    for j, key in enumerate(sorted(loci, key=ckey)):
        contig, pos, ref = key
        es = loci[key]
        if j % 23 == 5:
            continue                                   # absent from the VCF
        alts = sorted({e.easeq for e in es if e.easeq != ref})
        if not alts:
            alts = [next(b for b in BASES if b != ref[0])]
        if j % 17 == 3:
            alts = alts + [next(b for b in BASES if b not in alts and b != ref[0]) + "T"]
        eaf = es[0].eaf if es[0].easeq != ref and not np.isnan(es[0].eaf) else 0.2
        eaf = min(max(eaf, 0.01), 0.99)
        a1 = (rng.uniform(size=n) < eaf).astype(int)
        a2 = (rng.uniform(size=n) < eaf).astype(int)
        miss = rng.uniform(size=n) < (0.3 if j % 11 == 0 else 0.01)
        sep = "|" if j % 5 == 0 else "/"
        gts = np.where(miss, "./.", np.char.add(np.char.add(a1.astype(str), sep), a2.astype(str)))
        filt = "LowQual" if j % 29 == 7 else ("." if j % 2 else "PASS")
        lines.append("%s\t%d\t.\t%s\t%s\t.\t%s\t.\tGT\t%s" % (contig, pos, ref, ",".join(alts), filt,
                                                              "\t".join(gts.tolist())))
    with gzip.open(path, "wt") as fh:
        fh.write("\n".join(lines) + "\n")


@pytest.fixture(scope="module")
def cohort_vcf(tmp_path_factory):
    p = str(tmp_path_factory.mktemp("cohort") / "cohort.vcf.gz")
    write_cohort_vcf(p, SCORES[:-1], 1500, 20250102)
    return p


def assert_within_bar(got, ref, score_path, nloci, what=""):
    """north_star's bar PER SAMPLE: |got - ref| <= 1e-6 |ref|, |ref| floored at 1e-12 x sum|beta| / (2 nloci) as SURVEY.md
    8(d) prescribes (these comparisons used 1e-6 x max|ref|, an absolute bar); NaN positions equal"""
    from test_gpu_parity import rel_err
    beta = np.array([e.beta for e in refcpu.read_score_file(score_path).entries])
    r = rel_err(np.asarray(got), np.asarray(ref), beta, max(int(nloci), 1))
    assert r <= 1e-6, (what or score_path, r)


def oracle_run(score_path, vcf, **kw):
    score = refcpu.read_score_file(score_path)
    return refcpu.compute_polygenic_scores(score, vcf, False, {}, kw.get("imp_locus", "ps"),
                                           kw.get("imp_missing", "homref"),
                                           kw.get("imp_sample", "int_ps"), kw.get("maxmis", 0.05),
                                           kw.get("mincs", 100), kw.get("ignorefilt", False))


def test_wood_height_on_synthetic_cohort(cohort_vcf):
    wood = [s for s in SCORES if "wood-25282103" in s][0]
    vcf = refcpu.read_vcf(cohort_vcf)
    for kw in (dict(), dict(imp_locus="homref", imp_missing="ignore", imp_sample="ps", maxmis=0.5,
                            ignorefilt=True)):
        scores, nloci, log = host.compute_polygenic_scores(wood, cohort_vcf, afmisp=0.0, **kw)
        ref, ref_nloci, ref_stats = oracle_run(wood, vcf, **kw)
        assert nloci == ref_nloci and len(scores) == 1500
        assert_within_bar(scores, ref, wood, ref_nloci)
        assert sum(1 for s in ref_stats if s[4] == 2) > 10      # absent loci exercised
        if not kw:
            assert sum(1 for s in ref_stats if s[4] == 4) > 10  # over --maxmis (0.05) exercised
            assert sum(1 for s in ref_stats if s[4] == 3) > 5   # FILTER-failed loci exercised


def test_eight_scores_sharded_and_gathered(cohort_vcf):
    """configs[3]: the 8 score-format files of the reference tree, sharded over ranks (one here),
    gathered into the samples x scores matrix."""
    vcf = refcpu.read_vcf(cohort_vcf)
    files = SCORES[:-1] + [SCORES[0]]          # 7 bundled + one repeated = 8 definitions
    assert len(files) == 8

    def score_fn(i, out_row):
        s, _, _ = host.compute_polygenic_scores(files[i], cohort_vcf, afmisp=0.0)
        out_row.copy_(torch.from_numpy(s))

    full = multi.evaluate_sharded(len(files), len(vcf.samples), score_fn, torch.device("cpu"))
    for i, f in enumerate(files):
        ref, ref_nloci, _ = oracle_run(f, vcf)
        assert_within_bar(full[i].numpy(), ref, f, ref_nloci)


def test_one_pass_multi_equals_file_by_file(cohort_vcf):
    """computePolygenicScoresMulti (union of the loci decoded once into a resident cohort, the 8 definitions applied
    together on the matrix cores) gives, per file, the scores, nloci and warnings of the reference's loop run file
    by file -- default flags and a second set, AF-mismatch warnings on"""
    vcf = refcpu.read_vcf(cohort_vcf)
    files = SCORES[:-1]
    for kw in (dict(afmisp=0.001), dict(imp_locus="homref", imp_missing="ignore", imp_sample="ps", maxmis=0.5,
                                        ignorefilt=True, afmisp=0.001)):
        got, nloci, logs = host.compute_polygenic_scores_multi(files, cohort_vcf, **kw)
        assert got.shape == (len(files), 1500)
        for i, f in enumerate(files):
            s1, n1, log1 = host.compute_polygenic_scores(f, cohort_vcf, **kw)
            assert nloci[i] == n1, f
            assert logs[i] == log1, (f, logs[i][:3], log1[:3])
            okw = {k: v for k, v in kw.items() if k != "afmisp"}
            ref, ref_nloci, _ = oracle_run(f, vcf, **okw)
            assert n1 == ref_nloci
            assert_within_bar(got[i], ref, f, ref_nloci)
            assert_within_bar(s1, ref, f, ref_nloci, "file by file: " + f)


def test_results_left_in_device_memory_equal_the_host_copies(cohort_vcf):
    """what the multi-GPU exchange reads (tools/score_many.py with RCCL): nh_compute_dev / nh_compute_multi_dev leave
    the scores -- or, rows sharded, a block's un-normalised sums -- in DEVICE memory (nps_finish_device,
    nps_multi_finish_device, nps_multi_partial_device): bit-identical to what the host-buffer entry points return,
    with the same nloci and log lines; and the per-stage timings of a call add up to something sane"""
    files = SCORES[:-1]
    n = 1500
    kw = dict(afmisp=0.001)
    d_row = torch.full((n,), -7.0, dtype=torch.float64, device="cuda")
    for f in files[:3]:
        want, nloci, log = host.compute_polygenic_scores(f, cohort_vcf, **kw)
        none, nloci_d, log_d = host.compute_polygenic_scores(f, cohort_vcf, d_out=d_row.data_ptr(), **kw)
        assert none is None and nloci_d == nloci and log_d == log
        assert np.array_equal(d_row.cpu().numpy().view(np.int64), want.view(np.int64)), f
    t = host.last_timings()
    assert set(t) == set(host.TIMING_KEYS) and all(v >= 0.0 for v in t.values())
    # (a file without an index is inflated and parsed by open(): counted there; an indexed one under inflate_parse_s)
    assert t["open_s"] + t["inflate_parse_s"] > 0.0 and t["hip_init_wait_s"] <= t["hip_init_s"] + 1e-3 and sum(t.values()) < 60.0
    d_mat = torch.full((len(files), n), -7.0, dtype=torch.float64, device="cuda")
    want, nloci, logs = host.compute_polygenic_scores_multi(files, cohort_vcf, **kw)
    none, nloci_d, logs_d = host.compute_polygenic_scores_multi(files, cohort_vcf, d_out=d_mat.data_ptr(), **kw)
    assert none is None and np.array_equal(nloci_d, nloci) and logs_d == logs
    assert np.array_equal(d_mat.cpu().numpy().view(np.int64), want.view(np.int64))
    # rows sharded in two blocks: the blocks' sums, left on the device, add up and normalise to the one-pass scores
    total = torch.zeros((len(files), n), dtype=torch.float64, device="cuda")
    cnt = np.zeros(len(files), np.int64)
    for shard in range(2):
        d_mat.fill_(-7.0)
        want_s, nl_s, offs, lg = host.compute_polygenic_scores_multi_partial(files, cohort_vcf, shard, 2, **kw)
        none, nl_d, offs_d, lg_d = host.compute_polygenic_scores_multi_partial(files, cohort_vcf, shard, 2,
                                                                             d_out=d_mat.data_ptr(), **kw)
        assert none is None and np.array_equal(nl_d, nl_s) and np.array_equal(offs_d, offs) and lg_d == lg
        assert np.array_equal(d_mat.cpu().numpy().view(np.int64), want_s.view(np.int64))
        total += d_mat
        cnt += nl_d
    assert np.array_equal(cnt, nloci)
    got = multi.normalize_matrix(total, torch.from_numpy(cnt), offs).cpu().numpy()
    ok = ~np.isnan(want)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    scale = 1e-12 + np.max(np.abs(want[ok]))
    assert np.max(np.abs(got[ok] - want[ok])) <= 1e-9 * scale


@pytest.mark.parametrize("extra", [[], ["--one-pass"], ["--shard", "rows"]])
def test_score_many_device_result_paths_on_one_gpu(cohort_vcf, tmp_path, extra):
    """the code score_many.py runs on 8 GPUs -- results left in device memory, handed to the exchange from there -- taken
    with ONE rank (NIMPRESS_DEVICE_RESULTS=1: the gather / all-reduce of a single rank is the identity): the matrix
    equals the host-buffer run's, text for text"""
    import subprocess
    import sys
    files = SCORES[:-1]
    outs = []
    for env_extra in ({}, {"NIMPRESS_DEVICE_RESULTS": "1"}):
        out = str(tmp_path / ("m%d.tsv" % len(outs)))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "score_many.py"), "--gpus", "1", "--afmisp=0",
                            "--out", out] + extra + files + [cohort_vcf], capture_output=True, text=True,
                           env=dict(os.environ, **env_extra), timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(open(out).read())
    assert outs[0] == outs[1] and len(outs[0].splitlines()) == 1500


def test_log_longer_than_the_callers_buffer_comes_back_whole(cohort_vcf, monkeypatch):
    """the warnings of eight files do not fit a small buffer: the buffer ends at a line end with the truncation mark,
    nh_last_log has all of it, and host.py hands every file its own lines"""
    import ctypes as C
    files = SCORES[:-1]
    want, nloci, logs = host.compute_polygenic_scores_multi(files, cohort_vcf, afmisp=0.5)
    n_lines = sum(len(l) for l in logs)
    assert n_lines > 50
    real = C.create_string_buffer

    def small(size_or_init, *a):
        return real(2048 if size_or_init == (4 << 20) else size_or_init, *a)
    monkeypatch.setattr(host.C, "create_string_buffer", small)
    got, nloci2, logs2 = host.compute_polygenic_scores_multi(files, cohort_vcf, afmisp=0.5)
    assert logs2 == logs and np.array_equal(nloci2, nloci)
    # and the short buffer itself: whole lines, then the mark
    L = host.load()
    buf = real(2048)
    nl = np.zeros(len(files), np.uint64)
    sc = np.empty((len(files), 1500))
    from nimpress_amd import capi
    n = L.nh_compute_multi("\n".join(files).encode(), cohort_vcf.encode(), None, capi.LOCUS["ps"], capi.MISSING["homref"],
                           capi.SAMPLE["int_ps"], 0.05, 0.5, 100, 0, 0, sc.ctypes.data, 1500, nl.ctypes.data, buf, 2048)
    assert n == 1500
    text = buf.value.decode()
    assert text.endswith("... log truncated\n") and L.nh_last_log_size() > 2048
    assert all(l.partition("\t")[0].isdigit() for l in text.split("\n")[:-2])


def test_vectorised_float_format_equals_scalar():
    x = np.concatenate([np.random.default_rng(1).normal(0, 1, 2000), [0.0, 1.0, -2.0, 1e22, 1e-7, np.nan, np.inf, 123456789.0]])
    assert host.format_scores(x) == [host.format_score(float(v)) for v in x]


@pytest.mark.parametrize("one_pass", [False, True])
def test_score_many_eight_files_on_500k_sample_bcf(tmp_path, one_pass):
    """BASELINE.json configs[3] at its cohort size: the 8 score-format files of the reference tree on ONE
    500 000-sample BCF2 (+CSI) holding the union of their loci, through tools/score_many.py (one score
    definition per rank at a time; a single rank here, the gather is exercised with 2 ranks on the CPU in
    test_multi_gloo.py) -- every score against the oracle's own driver (findVariant + getImputedDosages +
    accumulate over the same records), all 500 000 samples."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import config2
    n = 500_000
    files = SCORES                     # 4 scores/*.scores + 3 makescore examples + tests/set1.score
    assert len(files) == 8
    path, samples, recs = config2.write_union_cohort(tmp_path, files, n)
    out = str(tmp_path / "matrix.tsv")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "score_many.py"), "--gpus", "1",
                        "--afmisp=0", "--out", out] + (["--one-pass"] if one_pass else []) + files + [path],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert ("one pass over the genotypes" in r.stderr) == one_pass, r.stderr[-500:]
    rows = [l.rstrip("\n").split("\t") for l in open(out)]
    assert [x[0] for x in rows] == samples
    got = np.array([[float(v) for v in x[1:]] for x in rows]).T           # [scores, samples]
    assert got.shape == (8, n)
    vcf = refcpu.Vcf(samples=samples, records=[
        refcpu.VcfRecord(contig=q["contig"], pos=q["pos"], ref=q["ref"], alts=q["alts"],
                         filt=";".join(q["filters"]) if q["filters"] else ".",
                         gts=q["gts"].reshape(-1), ploidy=2) for q in recs])
    for i, f in enumerate(files):
        ref, nloci, stats = oracle_run(f, vcf)
        assert_within_bar(got[i], ref, f, nloci)


@pytest.mark.parametrize("world", [1, 3])
def test_score_many_rows_sharded_over_ranks_all_files_per_rank(tmp_path, world):
    """The rows-sharded x all-scores layout from FILES (tools/score_many.py --shard rows): every
    rank locates, decodes and scores only its block of the union of the 8 files' loci -- for all 8 files, one
    matrix-core pass -- and one sum all-reduce of the [files, samples] sums and of the per-file locus counts follows.
    Three ranks share this box's one GPU here (exchange over gloo on CPU tensors; on a node it is RCCL, one GPU per
    rank); the matrix is checked against the oracle's own driver for every file and sample, the warnings against
    the single-rank run."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import config2
    n = 30_000
    files = SCORES
    path, samples, recs = config2.write_union_cohort(tmp_path, files, n)
    out = str(tmp_path / "matrix.tsv")
    env = dict(os.environ, NIMPRESS_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "score_many.py"), "--gpus", str(world), "--shard", "rows",
                        "--out", out] + files + [path], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "rows sharded over the GPUs" in r.stderr, r.stderr[-500:]
    rows = [l.rstrip("\n").split("\t") for l in open(out)]
    assert [x[0] for x in rows] == samples
    got = np.array([[float(v) for v in x[1:]] for x in rows]).T
    assert got.shape == (8, n)
    vcf = refcpu.Vcf(samples=samples, records=[
        refcpu.VcfRecord(contig=q["contig"], pos=q["pos"], ref=q["ref"], alts=q["alts"],
                         filt=";".join(q["filters"]) if q["filters"] else ".",
                         gts=q["gts"].reshape(-1), ploidy=2) for q in recs])
    for i, f in enumerate(files):
        ref, nloci, stats = oracle_run(f, vcf)
        assert_within_bar(got[i], ref, f, nloci)
    # the warnings (default --afmisp): the same set as the one-pass run of a single process
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "score_many.py"), "--gpus", "1", "--one-pass",
                         "--out", str(tmp_path / "m1.tsv")] + files + [path], capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-3000:]

    def warnings(text):
        return sorted(l for l in text.splitlines() if l.startswith("[") and "] WARN " in l)
    assert warnings(r.stderr) == warnings(r1.stderr)
    assert len(warnings(r.stderr)) > 0


def test_score_many_auto_shard_falls_back_to_files_on_format_ds(tmp_path):
    """`--shard auto` picks the rows x all-files layout whenever there are several ranks and the
    genotype file has an index -- but that layout's one-pass multi-score path does not take FORMAT/DS records.  Under
    `auto` every rank falls back to `--shard files` (agreed through an all-reduce of a flag) and the run equals the
    single-process one; an explicit `--shard rows` stays a hard error."""
    import shutil
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_cli as tc
    n, m = 3000, 40
    entries, recs = tc._ds_cohort(n, m, 4242)
    spath, path, samples = tc._write_ds_files(tmp_path, n, entries, recs, False)     # BCF2 + CSI, FORMAT/DS only
    assert os.path.exists(path + ".csi")
    spath2 = str(tmp_path / "ds_b.score")
    shutil.copy(spath, spath2)
    env = dict(os.environ, NIMPRESS_DIST_BACKEND="gloo")
    exe = [sys.executable, os.path.join(ROOT, "tools", "score_many.py")]
    out1, out2 = str(tmp_path / "m1.tsv"), str(tmp_path / "m2.tsv")
    r1 = subprocess.run(exe + ["--gpus", "1", "--afmisp=0", "--out", out1, spath, spath2, path], capture_output=True, text=True,
                        timeout=600)
    assert r1.returncode == 0, r1.stderr[-3000:]
    r2 = subprocess.run(exe + ["--gpus", "2", "--afmisp=0", "--out", out2, spath, spath2, path], capture_output=True, text=True,
                        env=env, timeout=600)
    assert r2.returncode == 0, r2.stderr[-3000:]
    assert "sharding the score files instead" in r2.stderr and "rows sharded over the GPUs" not in r2.stderr, r2.stderr[-800:]
    assert open(out1).read() == open(out2).read()
    r3 = subprocess.run(exe + ["--gpus", "2", "--shard", "rows", "--afmisp=0", "--out", str(tmp_path / "m3.tsv"), spath, spath2,
                               path], capture_output=True, text=True, env=env, timeout=600)
    assert r3.returncode != 0 and "one-pass" in r3.stderr, r3.stderr[-1500:]
