"""CPU tests of the host side above the C-ABI (nimpress_amd/csrc/host, C++): .scores parsing, BED
coverage, the text-VCF/BGZF reader, the findVariant rule, the stats used for warnings, float
formatting and the CLI surface.  No GPU compute here; the end-to-end CLI runs are in
tests/test_gpu_cli.py."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import refcpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
LIB = os.path.join(ROOT, "nimpress_amd", "libnimpress_host.so")
CLI = os.path.join(ROOT, "nimpress_amd", "nimpress")


@pytest.fixture(scope="module")
def host():
    if not os.path.exists(LIB):
        from nimpress_amd import build
        build.build_host()
    L = C.CDLL(LIB)
    L.nh_last_error.restype = C.c_char_p
    L.nh_score_parse.restype = C.c_long
    L.nh_score_parse.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.c_long, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_char_p, C.c_long]
    L.nh_bed_covered.restype = C.c_long
    L.nh_bed_covered.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_long]
    L.nh_vcf_open.restype = C.c_void_p
    L.nh_vcf_open.argtypes = [C.c_char_p, C.c_char_p]
    L.nh_vcf_open_streaming.restype = C.c_void_p
    L.nh_vcf_open_streaming.argtypes = [C.c_char_p, C.c_char_p, C.c_long]
    L.nh_vcf_close.argtypes = [C.c_void_p]
    L.nh_vcf_n_samples.restype = C.c_long
    L.nh_vcf_n_samples.argtypes = [C.c_void_p]
    L.nh_vcf_indexed.restype = C.c_int
    L.nh_vcf_indexed.argtypes = [C.c_void_p]
    L.nh_vcf_n_records.restype = C.c_long
    L.nh_vcf_n_records.argtypes = [C.c_void_p]
    L.nh_vcf_sample.restype = C.c_char_p
    L.nh_vcf_sample.argtypes = [C.c_void_p, C.c_long]
    L.nh_vcf_find.restype = C.c_long
    L.nh_vcf_find.argtypes = [C.c_void_p, C.c_char_p, C.c_long, C.c_char_p, C.c_char_p,
                              C.POINTER(C.c_long), C.POINTER(C.c_int), C.c_char_p, C.c_long,
                              C.c_void_p, C.c_long]
    L.nh_vcf_find_ds.restype = C.c_long
    L.nh_vcf_find_ds.argtypes = [C.c_void_p, C.c_char_p, C.c_long, C.c_char_p, C.c_char_p, C.c_void_p, C.c_long]
    for f in ("nh_dbinom", "nh_pbinom", "nh_binom_test", "nh_binom_test_fast"):
        getattr(L, f).restype = C.c_double
        getattr(L, f).argtypes = [C.c_long, C.c_long, C.c_double]
    L.nh_betai.restype = C.c_double
    L.nh_betai.argtypes = [C.c_double] * 3
    L.nh_format_float.argtypes = [C.c_double, C.c_char_p, C.c_long]
    return L


SCORE_FILES = sorted(os.listdir(os.path.join(G, "scores"))) + ["../set1.score"]


@pytest.mark.parametrize("name", SCORE_FILES)
def test_score_parser_matches_oracle(host, name):
    path = os.path.join(G, "scores", name)
    ref = refcpu.read_score_file(path)
    cap = len(ref.entries) + 8
    pos = np.zeros(cap, np.int64)
    beta = np.zeros(cap)
    eaf = np.zeros(cap)
    rie = np.zeros(cap, np.int32)
    text = C.create_string_buffer(1 << 20)
    off = C.c_double()
    n = host.nh_score_parse(path.encode(), C.byref(off), cap, pos.ctypes.data, beta.ctypes.data,
                            eaf.ctypes.data, rie.ctypes.data, text, len(text))
    assert n == len(ref.entries), host.nh_last_error()
    assert off.value == ref.offset
    assert pos[:n].tolist() == [e.pos for e in ref.entries]
    assert beta[:n].tolist() == [e.beta for e in ref.entries]
    assert np.array_equal(eaf[:n], np.array([e.eaf for e in ref.entries]), equal_nan=True)
    assert rie[:n].tolist() == [int(e.refseq == e.easeq) for e in ref.entries]
    lines = text.value.decode().split("\n")
    assert lines[:4] == [ref.name, ref.desc, ref.cite, ref.genomever]
    assert lines[4:4 + n] == ["%s\t%s\t%s" % (e.contig, e.refseq, e.easeq) for e in ref.entries]


def test_score_parser_rejects_short_rows(host, tmp_path):
    p = tmp_path / "bad.score"
    p.write_text("n\nd\nc\ng\n0.0\n1\t10\tA\tC\t0.1\n")           # 5 fields: doAssert nim:252
    assert host.nh_score_parse(str(p).encode(), None, 0, None, None, None, None, None, 0) == -2
    p.write_text("n\nd\nc\ng\n0.0\n1\t10\tA\tC\t0.1\t0.2\n\n")      # blank trailing line asserts too
    assert host.nh_score_parse(str(p).encode(), None, 0, None, None, None, None, None, 0) == -2
    p.write_text("n\nd\nc\ng\n0.0\n1\t10\tA\tC\t0.1\tNaN")         # NaN eaf, no final newline: fine
    assert host.nh_score_parse(str(p).encode(), None, 0, None, None, None, None, None, 0) == 1
    assert host.nh_score_parse(b"/nonexistent", None, 0, None, None, None, None, None, 0) == -1


def test_bed_coverage_off_by_one_traps(host):
    out = np.zeros(8, np.int32)
    n = host.nh_bed_covered(os.path.join(G, "set1.score").encode(), os.path.join(G, "set1.bed").encode(),
                            out.ctypes.data, 8)
    assert n == 6
    # 1:100 no, 1:150 yes, 1:200 yes, 1:300 GA needs end >= 301: no, 2:400 start 400 is not < 400: no
    assert out[:6].tolist() == [0, 1, 1, 0, 0, 1]
    score = refcpu.read_score_file(os.path.join(G, "set1.score"))
    bed = refcpu.read_bed(os.path.join(G, "set1.bed"))
    assert out[:6].tolist() == [int(refcpu.is_variant_covered(e, bed)) for e in score.entries]


def test_vcf_reader_and_find_variant_match_oracle(host):
    path = os.path.join(G, "set1.vcf.gz")
    score = refcpu.read_score_file(os.path.join(G, "set1.score"))
    ref_vcf = refcpu.read_vcf(path)
    for keep in (None, os.path.join(G, "set1.score").encode()):
        h = host.nh_vcf_open(path.encode(), keep)
        assert h, host.nh_last_error()
        assert [host.nh_vcf_sample(h, i).decode() for i in range(host.nh_vcf_n_samples(h))] == ref_vcf.samples
        assert host.nh_vcf_n_records(h) == (7 if keep is None else 6)   # 1:50 is not a score locus
        # with the score loci known, the reference's own .tbi (written by htslib's tabix) is used
        assert host.nh_vcf_indexed(h) == (0 if keep is None else 1)
        for e in score.entries:
            rec = refcpu.find_variant(ref_vcf, e)
            pos, ploidy = C.c_long(), C.c_int()
            filt = C.create_string_buffer(64)
            gts = np.zeros(64, np.int32)
            idx = host.nh_vcf_find(h, e.contig.encode(), e.pos, e.refseq.encode(), e.easeq.encode(),
                                   C.byref(pos), C.byref(ploidy), filt, 64, gts.ctypes.data, 64)
            if rec is None:
                assert idx == -1
            else:
                assert idx >= 0 and pos.value == rec.pos and ploidy.value == rec.ploidy
                assert filt.value.decode() == rec.filt
                assert gts[: rec.gts.size].tolist() == rec.gts.tolist()
        host.nh_vcf_close(h)


def test_vcf_reader_on_reference_split_records(host):
    """tests/set1.plink.vcf.gz (+ .tbi), the other htslib-written file the reference holds: two records at 1:300
    with the same REF.  Whole-file scan and tabix random access both give the oracle's record for every score row;
    1:300 GA/CT must skip GA>T (nimpress.nim:359-364)."""
    path = os.path.join(G, "set1.plink.vcf.gz")
    score = refcpu.read_score_file(os.path.join(G, "set1.score"))
    ref_vcf = refcpu.read_vcf(path)
    for keep in (None, os.path.join(G, "set1.score").encode()):
        h = host.nh_vcf_open(path.encode(), keep)
        assert h, host.nh_last_error()
        assert host.nh_vcf_indexed(h) == (0 if keep is None else 1)
        assert host.nh_vcf_n_records(h) == (8 if keep is None else 7)
        for e in score.entries:
            rec = refcpu.find_variant(ref_vcf, e)
            pos, ploidy = C.c_long(), C.c_int()
            filt = C.create_string_buffer(64)
            gts = np.zeros(64, np.int32)
            idx = host.nh_vcf_find(h, e.contig.encode(), e.pos, e.refseq.encode(), e.easeq.encode(),
                                   C.byref(pos), C.byref(ploidy), filt, 64, gts.ctypes.data, 64)
            if rec is None:
                assert idx == -1
            else:
                assert idx >= 0 and pos.value == rec.pos and filt.value.decode() == rec.filt
                assert gts[: rec.gts.size].tolist() == rec.gts.tolist()
                if e.pos == 300:
                    assert rec.alts == ["CT"]
        host.nh_vcf_close(h)


DS_VCF = """##fileformat=VCFv4.2
##contig=<ID=1>
##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">
##FORMAT=<ID=DS,Number=A,Type=Float,Description="ALT allele dosage">
#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\tC\tD
1\t100\t.\tA\tC\t.\tPASS\t.\tGT:DS\t0/1:1.004\t1/1:1.75\t./.:.\t0/0:0
1\t200\t.\tG\tT\t.\t.\t.\tDS\t0.25\t.\t2\t1.5
1\t300\t.\tGA\tT,CT\t.\tPASS\t.\tDS:GT\t0.5,1:0/2\t0,0:0/0\t.,.:./.\t1.25:1/1
1\t400\t.\tT\tC\t.\tFAIL\t.\tGT\t0/1\t1/1\t./.\t0/0
"""
DS_SCORE = "t\n\n\nhg19\n0.5\n" + "\n".join([
    "1\t100\tA\tC\t0.1\t0.3", "1\t100\tA\tA\t0.2\t0.7", "1\t200\tG\tT\t-0.3\tNaN",
    "1\t300\tGA\tCT\t0.4\t0.2", "1\t300\tGA\tGA\t0.05\t0.6", "1\t300\tGA\tT\t0.15\t0.2",
    "1\t400\tT\tC\t0.25\t0.1"])


@pytest.mark.parametrize("prefer", [False, True])
@pytest.mark.parametrize("kind", ["text", "bcf"])
def test_format_ds_rows_match_oracle(host, tmp_path, monkeypatch, prefer, kind):
    """FORMAT/DS (build-defined: the reference reads GT only): the dosage row handed to nps_push_ds for every score
    row -- records without GT, records with both under NIMPRESS_FORMAT=DS, multi-allelic Number=A vectors (effect
    allele = second ALT, = REF: the sum of the ALT dosages), missing values, a vector shorter than the ALT count --
    from the text reader and from the BCF2 typed float vectors equals the oracle's reading."""
    import bcfwriter
    vpath, spath = str(tmp_path / "ds.vcf"), str(tmp_path / "ds.score")
    open(vpath, "w").write(DS_VCF)
    open(spath, "w").write(DS_SCORE)
    ref_vcf = refcpu.read_vcf(vpath, prefer_ds=prefer)
    assert [r.ds is not None for r in ref_vcf.records] == ([True, True, True, False] if prefer else [False, True, False, False])
    path = vpath
    if kind == "bcf":
        both = refcpu.read_vcf(vpath, prefer_ds=True), refcpu.read_vcf(vpath, prefer_ds=False)
        recs = []
        for rd, rg in zip(both[0].records, both[1].records):
            recs.append(dict(contig=rd.contig, pos=rd.pos, id=".", ref=rd.ref, alts=rd.alts,
                             filters=[] if rd.filt == "." else rd.filt.split(";"),
                             gts=None if rg.gts is None else np.asarray(rg.gts).reshape(4, rg.ploidy),
                             ds=rd.ds))
        path = str(tmp_path / "ds.bcf")
        bcfwriter.write_bcf(path, ["1"], ["A", "B", "C", "D"], recs, gt_dtype=np.int8)
    if prefer:
        monkeypatch.setenv("NIMPRESS_FORMAT", "DS")
    else:
        monkeypatch.delenv("NIMPRESS_FORMAT", raising=False)
    score = refcpu.read_score_file(spath)
    for keep in (None, spath.encode()) if kind == "bcf" else (None,):
        h = host.nh_vcf_open(path.encode(), keep)
        assert h, host.nh_last_error()
        for e in score.entries:
            rec = refcpu.find_variant(ref_vcf, e)
            out = np.zeros(4, np.float32)
            k = host.nh_vcf_find_ds(h, e.contig.encode(), e.pos, e.refseq.encode(), e.easeq.encode(), out.ctypes.data, 4)
            if rec.ds is None:
                assert k == 0
                continue
            eaidx = 0 if e.refseq == e.easeq else rec.alts.index(e.easeq) + 1
            want = refcpu.ds_row(rec, eaidx)
            assert k == 4 and np.array_equal(np.isnan(out), np.isnan(want)), (e, out, want)
            assert np.array_equal(out[~np.isnan(want)], want[~np.isnan(want)]), (e, out, want)
        host.nh_vcf_close(h)


@pytest.mark.parametrize("kind", ["text", "bcf"])
def test_format_ds_outside_the_dosage_range_is_read_and_infinities_are_refused(host, tmp_path, kind):
    """a dosage is 0 <= DS <= 2, but the readers do not police it (the streamed nps_push_ds rows they feed
    take any finite value; nps_cohort_upload marks rows outside the range for the resident single-read kernel itself): a
    record scored from DS that holds 2.5 opens and is handed out as it is.  Only a value no kernel can score -- an
    infinity -- is refused when it is read, naming the record; a DS that is NOT used (the record has GT, NIMPRESS_FORMAT
    unset) is not even looked at"""
    import bcfwriter
    wide = DS_VCF.replace("0.25\t.\t2\t1.5", "0.25\t.\t2.5\t1.5")            # the DS-only record 1:200
    bad = DS_VCF.replace("0.25\t.\t2\t1.5", "0.25\t.\tinf\t1.5")
    unused = DS_VCF.replace("0/1:1.004", "0/1:inf")                              # 1:100 has GT: its DS is not used
    for name, text, ok in (("wide", wide, True), ("bad", bad, False), ("unused", unused, True)):
        vpath = str(tmp_path / (name + ".vcf"))
        open(vpath, "w").write(text)
        path = vpath
        if kind == "bcf":
            # (the oracle's reader is the fixture writer's source here; it does not police the range)
            rd, rg = refcpu.read_vcf(vpath, prefer_ds=True), refcpu.read_vcf(vpath, prefer_ds=False)
            recs = [dict(contig=a.contig, pos=a.pos, id=".", ref=a.ref, alts=a.alts,
                         filters=[] if a.filt == "." else a.filt.split(";"),
                         gts=None if b.gts is None else np.asarray(b.gts).reshape(4, b.ploidy), ds=a.ds)
                    for a, b in zip(rd.records, rg.records)]
            path = str(tmp_path / (name + ".bcf"))
            bcfwriter.write_bcf(path, ["1"], ["A", "B", "C", "D"], recs, gt_dtype=np.int8)
        h = host.nh_vcf_open(path.encode(), None)
        if ok:
            assert h, host.nh_last_error()
            host.nh_vcf_close(h)
        else:
            assert not h
            msg = host.nh_last_error().decode()
            assert "not finite" in msg and "1:200" in msg, msg


def test_vcf_reader_plain_text_phased_haploid(host, tmp_path):
    p = tmp_path / "t.vcf"
    p.write_text("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\tC\n"
                 "X\t7\t.\tG\tT,C\t.\tq10;s50\t.\tDP:GT\t3:0|2\t1:1\t.:./.\n")
    h = host.nh_vcf_open(str(p).encode(), None)
    assert h, host.nh_last_error()
    pos, ploidy = C.c_long(), C.c_int()
    filt = C.create_string_buffer(64)
    gts = np.zeros(16, np.int32)
    idx = host.nh_vcf_find(h, b"X", 7, b"G", b"C", C.byref(pos), C.byref(ploidy), filt, 64,
                           gts.ctypes.data, 16)
    assert idx == 0 and ploidy.value == 2 and filt.value == b"q10;s50"
    # 0|2 -> (0+1)<<1, (2+1)<<1|1 ; haploid 1 -> 4, vector-end pad ; ./. -> 0,0
    assert gts[:6].tolist() == [2, 7, 4, -2147483647, 0, 0]
    assert host.nh_vcf_find(h, b"X", 7, b"A", b"C", None, None, None, 0, None, 0) == -1   # REF differs
    assert host.nh_vcf_find(h, b"X", 7, b"G", b"G", None, None, None, 0, None, 0) == 0    # ea == ref
    host.nh_vcf_close(h)


def test_host_stats_kats(host):
    d = json.load(open(os.path.join(G, "stats_kats.json")))
    fns = {"dbinom": lambda x, n, p: host.nh_dbinom(int(x), int(n), p),
           "pbinom": lambda x, n, p: host.nh_pbinom(int(x), int(n), p),
           "binom_test": lambda x, n, p: host.nh_binom_test(int(x), int(n), p),
           "betai": lambda a, b, x: host.nh_betai(a, b, x)}
    for k in d["kats"]:
        val, tgt = fns[k["fn"]](*k["args"]), k["expected"]
        if k["mode"] == "exact":
            assert val == tgt, k
        elif abs(tgt) < d["abs_tol"]:
            assert abs(val - tgt) < d["abs_tol"], k
        else:
            assert abs((val - tgt) / tgt) < d["rel_tol"], (k, val)


def test_host_float_format(host):
    buf = C.create_string_buffer(64)
    n = 0
    rf = os.path.join(G, "result_format")
    for f in sorted(os.listdir(rf)):
        for line in open(os.path.join(rf, f)).read().splitlines():
            txt = line.split("\t")[1]
            host.nh_format_float(float(txt), buf, 64)
            assert buf.value.decode() == txt
            n += 1
    assert n == 3528
    host.nh_format_float(float("nan"), buf, 64)
    assert buf.value == b"nan"


def test_write_matrix_tsv_equals_the_reference_format(tmp_path):
    """nh_write_matrix_tsv (threaded C++ writer of the samples x scores matrix, tools/score_many.py): every value
    as the reference prints a score -- the 3 528 values of scores/*_nimpress_res.txt re-print identically -- plus
    nan / inf / integers, with enough samples for several worker threads; equal to the per-value formatter."""
    from nimpress_amd import host as H
    vals = []
    rf = os.path.join(G, "result_format")
    for f in sorted(os.listdir(rf)):
        vals += [line.split("\t")[1] for line in open(os.path.join(rf, f)).read().splitlines()]
    assert len(vals) == 3528
    rng = np.random.default_rng(5)
    n = 3 * 3528
    m = np.empty((3, n))
    m[0] = np.tile(np.array([float(v) for v in vals]), 3)
    m[1] = rng.normal(0, 1e-3, n)
    m[2] = rng.integers(-5, 5, n).astype(np.float64)
    m[1, 7], m[1, 8], m[1, 9], m[2, 0] = np.nan, np.inf, -np.inf, 1e22
    names = ["S%05d" % i for i in range(n)]
    out = tmp_path / "m.tsv"
    H.write_matrix_tsv(str(out), names, m)
    lines = out.read_text().split("\n")
    assert lines[-1] == "" and len(lines) == n + 1
    for j in (0, 1, 7, 8, 9, 3527, 3528, n - 1):
        f = lines[j].split("\t")
        assert f[0] == names[j] and f[1:] == [H.format_score(m[k, j]) for k in range(3)], (j, f)
    assert [ln.split("\t")[1] for ln in lines[:3528]] == vals
    cols = [H.format_scores(m[k]) for k in range(3)]
    assert lines[:-1] == [names[j] + "\t" + "\t".join(c[j] for c in cols) for j in range(n)]
    with pytest.raises(ValueError):
        H.write_matrix_tsv(str(out), names[:-1], m)


def test_cli_surface():
    if not os.path.exists(CLI):
        from nimpress_amd import build
        build.build_host()
    r = subprocess.run([CLI, "--version"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == "nimpress 1.0.0\n"
    r = subprocess.run([CLI, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--imp-locus=<m>" in r.stdout and "[default: int_ps]" in r.stdout
    r = subprocess.run([CLI, os.path.join(G, "set1.score"), "/nonexistent.vcf.gz"], capture_output=True, text=True)
    assert r.returncode == 255 and r.stdout.startswith("FATAL Could not open input VCF file")
    r = subprocess.run([CLI, "/nonexistent.score", os.path.join(G, "set1.vcf.gz")], capture_output=True, text=True)
    assert r.returncode == 255 and "FATAL Could not open polygenic score file" in r.stdout
    r = subprocess.run([CLI, "--imp-locus=bogus", os.path.join(G, "set1.score"), os.path.join(G, "set1.vcf.gz")],
                       capture_output=True, text=True)
    assert r.returncode == 1
    r = subprocess.run([CLI, "onlyone"], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout.startswith("Usage:")


# ------------------------------------------------------------------------------------------
# BGZF + tabix writer (test-side) to exercise multi-block random access
import struct
import zlib


def _bgzf_block(data: bytes) -> bytes:
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize)
            + comp + struct.pack("<II", zlib.crc32(data), len(data)))


def _reg2bin(beg, end):
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


def write_bgzf_vcf_with_tbi(path, header_lines, records):
    """records: list of (contig, pos, ref, line) sorted by contig order of appearance, pos."""
    out = bytearray()
    buf = bytearray()
    voffs = []

    def flush():
        nonlocal buf
        if buf:
            out.extend(_bgzf_block(bytes(buf)))
            buf = bytearray()

    def add_line(txt):
        nonlocal buf
        data = txt.encode() + b"\n"
        start = (len(out) << 16) | len(buf)
        i = 0
        while i < len(data):
            room = 0xff00 - len(buf)
            if room == 0:
                flush()
                room = 0xff00
            buf.extend(data[i:i + room])
            i += room
        if len(buf) >= 0xff00:
            flush()
        end = (len(out) << 16) | len(buf)
        return start, end

    for h in header_lines:
        add_line(h)
    contigs = []
    index = {}
    for contig, pos, ref, line in records:
        vs, ve = add_line(line)
        if contig not in index:
            contigs.append(contig)
            index[contig] = ({}, {})
        bins, lin = index[contig]
        beg, end = pos - 1, pos - 1 + len(ref)
        bins.setdefault(_reg2bin(beg, end), []).append((vs, ve))
        for w in range(beg >> 14, ((end - 1) >> 14) + 1):
            lin[w] = min(lin.get(w, vs), vs)
    flush()
    out.extend(_bgzf_block(b""))
    open(path, "wb").write(bytes(out))
    names = b"".join(c.encode() + b"\0" for c in contigs)
    t = bytearray(b"TBI\1" + struct.pack("<8i", len(contigs), 2, 1, 2, 0, ord("#"), 0, len(names)) + names)
    for c in contigs:
        bins, lin = index[c]
        t += struct.pack("<i", len(bins))
        for b, chunks in bins.items():
            merged = [(chunks[0][0], chunks[-1][1])]
            t += struct.pack("<Ii", b, len(merged))
            for vs, ve in merged:
                t += struct.pack("<QQ", vs, ve)
        nw = max(lin) + 1 if lin else 0
        t += struct.pack("<i", nw)
        last = 0
        for w in range(nw):
            last = lin.get(w, last)
            t += struct.pack("<Q", last)
    tb = bytearray()
    for i in range(0, len(t), 0xff00):
        tb += _bgzf_block(bytes(t[i:i + 0xff00]))
    tb += _bgzf_block(b"")
    open(path + ".tbi", "wb").write(bytes(tb))


def test_tabix_random_access_multi_block(host, tmp_path):
    rng = np.random.default_rng(12)
    n = 3000                       # ~12 kB per line, several lines per 64 kB block, lines span blocks
    samples = ["Q%d" % i for i in range(n)]
    header = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples)]
    recs = []
    for contig in ("1", "2", "X"):
        pos = 1000
        for _ in range(120):
            pos += int(rng.integers(1, 60000))
            ref = "A" if rng.uniform() < 0.8 else "ACG"
            g = rng.integers(0, 2, size=(n, 2))
            gts = "\t".join("%d/%d" % (a, b) for a, b in g)
            recs.append((contig, pos, ref, "%s\t%d\t.\t%s\tG\t.\tPASS\t.\tGT\t%s" % (contig, pos, ref, gts)))
    vcf_path = str(tmp_path / "big.vcf.gz")
    write_bgzf_vcf_with_tbi(vcf_path, header, recs)
    # a score file hitting 40 random records (+ 5 loci that are absent)
    pick = rng.choice(len(recs), 40, replace=False)
    lines = ["t", "", "", "x", "0.0"]
    for k in pick:
        c, pos, ref, _ = recs[k]
        lines.append("%s\t%d\t%s\tG\t0.1\t0.2" % (c, pos, ref))
    for j in range(5):
        lines.append("2\t%d\tA\tG\t0.1\t0.2" % (7 + j))
    score_path = str(tmp_path / "s.score")
    open(score_path, "w").write("\n".join(lines))
    results = []
    # whole-file scan, index (all loci at once, several threads), and the streaming driver's windowed
    # fetch (windows of 7 score rows; one thread and several)
    for no_index, window, threads in ((False, 0, None), (True, 0, None), (False, 7, "1"), (False, 7, "5")):
        if no_index:
            os.environ["NIMPRESS_NO_INDEX"] = "1"
        if threads:
            os.environ["NIMPRESS_THREADS"] = threads
        try:
            if window:
                h = host.nh_vcf_open_streaming(vcf_path.encode(), score_path.encode(), window)
            else:
                h = host.nh_vcf_open(vcf_path.encode(), score_path.encode())
            assert h, host.nh_last_error()
            assert host.nh_vcf_indexed(h) == (0 if no_index else 1)
            assert host.nh_vcf_n_samples(h) == n
            got = []
            gts = np.zeros(2 * n, np.int32)
            for ln in lines[5:]:
                c, pos, ref, ea, _, _ = ln.split("\t")
                rp, pl = C.c_long(), C.c_int()
                idx = host.nh_vcf_find(h, c.encode(), int(pos), ref.encode(), ea.encode(), C.byref(rp),
                                       C.byref(pl), None, 0, gts.ctypes.data, 2 * n)
                got.append((idx >= 0, rp.value if idx >= 0 else -1, gts.copy() if idx >= 0 else None))
            results.append((host.nh_vcf_n_records(h), got))
            host.nh_vcf_close(h)
        finally:
            os.environ.pop("NIMPRESS_NO_INDEX", None)
            os.environ.pop("NIMPRESS_THREADS", None)
    assert all(r[0] == 40 for r in results)
    for other in results[1:]:
        for a, b in zip(results[0][1], other[1]):
            assert a[0] == b[0] and a[1] == b[1]
            if a[0]:
                assert np.array_equal(a[2], b[2])
    assert sum(1 for a in results[0][1] if a[0]) == 40


def test_fast_binom_test_equals_enumeration(host):
    """the O(log n) two-sided binomial test used for the AF-mismatch warnings must give the p-value of
    the reference's O(n) enumeration (oracle: ref_binom_test), incl. large cohorts and both tails"""
    rng = np.random.default_rng(2024)
    cases = [(0, 12, 0.3), (12, 12, 0.3), (0, 2 * 500_000, 0.01), (7, 10, 0.95), (3, 6, 0.5)]
    for _ in range(1500):
        n = int(rng.choice([6, 12, 50, 200, 2000, 20000, 200000, 1000000]))
        p = float(rng.choice([rng.uniform(0.001, 0.999), rng.uniform(0.0005, 0.05), 0.5]))
        x = int(np.clip(rng.binomial(n, p) + rng.integers(-3, 4) * max(1, int(np.sqrt(n * p * (1 - p)))), 0, n))
        cases.append((x, n, p))
    for x, n, p in cases:
        ref = refcpu.binom_test(x, n, p)
        got = host.nh_binom_test_fast(x, n, p)
        if np.isnan(ref):
            assert np.isnan(got), (x, n, p)
        else:
            assert got == ref or abs(got - ref) <= 1e-12 * max(abs(ref), 1e-300), (x, n, p, got, ref)


# ---- BCF2 + CSI (fixtures written by tests/bcfwriter.py; no htslib here: parity unpinned) -------------
@pytest.mark.parametrize("gt_dtype", [np.int8, np.int16, np.int32])
def test_bcf_reader_matches_vcf_reader_on_set1_bcf_parity_unpinned(host, tmp_path, gt_dtype):
    """the reference's own fixture re-written as BCF: same samples, records, FILTER strings, alleles
    and (widened) GT values as the text reader gives; with the CSI index and by whole-file scan"""
    import bcfwriter
    vcf = refcpu.read_vcf(os.path.join(G, "set1.vcf.gz"))
    contigs, recs = bcfwriter.records_from_oracle_vcf(vcf)
    path = str(tmp_path / "set1.bcf")
    bcfwriter.write_bcf(path, contigs, vcf.samples, recs, gt_dtype=gt_dtype)
    score = refcpu.read_score_file(os.path.join(G, "set1.score"))
    spath = os.path.join(G, "set1.score").encode()
    for keep, no_index in ((None, False), (spath, False), (spath, True)):
        if no_index:
            os.environ["NIMPRESS_NO_INDEX"] = "1"
        try:
            hb = host.nh_vcf_open(path.encode(), keep)
            ht = host.nh_vcf_open(os.path.join(G, "set1.vcf.gz").encode(), keep)
            assert hb and ht, host.nh_last_error()
            assert host.nh_vcf_indexed(hb) == (1 if keep and not no_index else 0)
            n = host.nh_vcf_n_samples(hb)
            assert [host.nh_vcf_sample(hb, i).decode() for i in range(n)] == vcf.samples
            assert host.nh_vcf_n_records(hb) == host.nh_vcf_n_records(ht)
            for e in score.entries:
                out = []
                for h in (hb, ht):
                    rp, pl = C.c_long(), C.c_int()
                    filt = C.create_string_buffer(64)
                    gts = np.full(8 * n, 7, np.int32)
                    idx = host.nh_vcf_find(h, e.contig.encode(), e.pos, e.refseq.encode(), e.easeq.encode(),
                                           C.byref(rp), C.byref(pl), filt, 64, gts.ctypes.data, 8 * n)
                    out.append((idx, rp.value, pl.value, filt.value, gts.tolist()) if idx >= 0 else (idx,))
                assert out[0] == out[1], (e, out)
            host.nh_vcf_close(hb)
            host.nh_vcf_close(ht)
        finally:
            os.environ.pop("NIMPRESS_NO_INDEX", None)


def test_bcf_random_access_multi_block_csi_parity_unpinned(host, tmp_path):
    """records larger than a BGZF block, three contigs, IDX= in the header, a long allele (length
    descriptor 15), multi-filter records: CSI access == whole-file scan"""
    import bcfwriter
    rng = np.random.default_rng(21)
    n = 40000                                   # 80 kB of int8 GT per record: spans BGZF blocks
    samples = ["Q%d" % i for i in range(n)]
    contigs = ["1", "2", "X"]
    recs = []
    for contig in contigs:
        pos = 1000
        for k in range(25):
            pos += int(rng.integers(1, 90000))
            ref = "A" if rng.uniform() < 0.7 else "ACGTACGTACGTACGTACGT"
            g = ((rng.integers(-1, 3, size=(n, 2)) + 1) << 1) | rng.integers(0, 2, size=(n, 2))
            g[rng.uniform(size=n) < 0.05, 1] = bcfwriter.INT32_END
            recs.append(dict(contig=contig, pos=pos, id="rs%d" % k, ref=ref, alts=["G", "T"],
                             filters=[[], ["PASS"], ["FAIL"], ["FAIL", "q10"]][k % 4], gts=g))
    path = str(tmp_path / "big.bcf")
    bcfwriter.write_bcf(path, contigs, samples, recs, gt_dtype=np.int8, filters=("PASS", "FAIL", "q10"),
                        extra_header=['##INFO=<ID=DP,Number=1,Type=Integer,Description="d">'])
    pick = rng.choice(len(recs), 20, replace=False)
    lines = ["t", "", "", "x", "0.0"]
    for k in pick:
        r = recs[k]
        lines.append("%s\t%d\t%s\tG\t0.1\t0.2" % (r["contig"], r["pos"], r["ref"]))
    lines.append("2\t7\tA\tG\t0.1\t0.2")
    score_path = str(tmp_path / "s.score")
    open(score_path, "w").write("\n".join(lines))
    results = []
    # CSI access (all loci, several threads), whole-file scan, windowed streaming fetch (3 rows a window)
    for no_index, window in ((False, 0), (True, 0), (False, 3)):
        if no_index:
            os.environ["NIMPRESS_NO_INDEX"] = "1"
        try:
            if window:
                h = host.nh_vcf_open_streaming(path.encode(), score_path.encode(), window)
            else:
                h = host.nh_vcf_open(path.encode(), score_path.encode())
            assert h, host.nh_last_error()
            assert host.nh_vcf_indexed(h) == (0 if no_index else 1)
            got = []
            gts = np.zeros(2 * n, np.int32)
            for ln in lines[5:]:
                c, pos, ref, ea, _, _ = ln.split("\t")
                rp, pl = C.c_long(), C.c_int()
                filt = C.create_string_buffer(64)
                idx = host.nh_vcf_find(h, c.encode(), int(pos), ref.encode(), ea.encode(), C.byref(rp),
                                       C.byref(pl), filt, 64, gts.ctypes.data, 2 * n)
                got.append((idx >= 0, rp.value, filt.value, gts.copy()) if idx >= 0 else (False,))
            results.append((host.nh_vcf_n_records(h), got))
            host.nh_vcf_close(h)
        finally:
            os.environ.pop("NIMPRESS_NO_INDEX", None)
    assert results[0][0] == results[1][0] == results[2][0] == 20
    for other in results[1:]:
        for (a, b), k in zip(zip(results[0][1][:-1], other[1][:-1]), pick):
            assert a[0] and b[0] and a[1] == b[1] == recs[k]["pos"]
            exp_f = ";".join(recs[k]["filters"]) or "."
            assert a[2].decode() == b[2].decode() == exp_f
            assert np.array_equal(a[3], b[3]) and np.array_equal(a[3], recs[k]["gts"].ravel().astype(np.int32))
        assert results[0][1][-1] == other[1][-1] == (False,)


@pytest.mark.parametrize("kind", ["bcf", "vcf.gz"])
def test_indexed_fetch_dense_bin_resumes_correctly(host, tmp_path, kind):
    """Many score rows inside ONE 16 kb index bin, in file order that goes back and forth, next to a record
    whose long REF allele spans the following loci: the indexed fetch resumes its scan of the bin where the
    previous row stopped only when no earlier record can overlap the new row; results == whole-file scan,
    with one fetching thread (everything in one resume map) and several."""
    import bcfwriter
    rng = np.random.default_rng(5)
    n = 50
    samples = ["Q%d" % i for i in range(n)]
    recs, vlines = [], []
    pos_list = list(range(16400, 16400 + 60 * 37, 37))          # 60 loci inside bin [16384, 32768)
    for j, pos in enumerate(pos_list):
        ref = "A" * 120 if j == 10 else ("AC" if j % 7 == 3 else "A")   # j == 10 spans the next three loci
        g = ((rng.integers(0, 2, size=(n, 2)) + 1) << 1)
        recs.append(dict(contig="1", pos=pos, id=".", ref=ref, alts=["G"], filters=[], gts=g))
        gts = "\t".join("%d/%d" % ((a >> 1) - 1, (b >> 1) - 1) for a, b in g)
        vlines.append(("1", pos, ref, "1\t%d\t.\t%s\tG\t.\tPASS\t.\tGT\t%s" % (pos, ref, gts)))
    if kind == "bcf":
        path = str(tmp_path / "dense.bcf")
        bcfwriter.write_bcf(path, ["1"], samples, recs, gt_dtype=np.int8)
    else:
        path = str(tmp_path / "dense.vcf.gz")
        header = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples)]
        write_bgzf_vcf_with_tbi(path, header, vlines)
    order = list(rng.permutation(len(recs)))
    lines = ["t", "", "", "x", "0.0"]
    for k in order:
        r = recs[k]
        lines.append("1\t%d\t%s\tG\t0.1\t0.2" % (r["pos"], r["ref"]))
    # rows that only the long record overlaps: REF differs there, so findVariant must walk past it
    lines.append("1\t%d\tA\tG\t0.1\t0.2" % (pos_list[11]))
    lines.append("1\t%d\tC\tG\t0.1\t0.2" % (pos_list[10] + 50))
    score_path = str(tmp_path / "s.score")
    open(score_path, "w").write("\n".join(lines))
    results = []
    for no_index, window, threads in ((True, 0, None), (False, 0, "1"), (False, 0, "4"), (False, 5, "1"), (False, 5, "3")):
        if no_index:
            os.environ["NIMPRESS_NO_INDEX"] = "1"
        if threads:
            os.environ["NIMPRESS_THREADS"] = threads
        try:
            h = (host.nh_vcf_open_streaming(path.encode(), score_path.encode(), window) if window
                 else host.nh_vcf_open(path.encode(), score_path.encode()))
            assert h, host.nh_last_error()
            got = []
            gts = np.zeros(2 * n, np.int32)
            for ln in lines[5:]:
                c, pos, ref, ea, _, _ = ln.split("\t")
                rp, pl = C.c_long(), C.c_int()
                idx = host.nh_vcf_find(h, c.encode(), int(pos), ref.encode(), ea.encode(), C.byref(rp),
                                       C.byref(pl), None, 0, gts.ctypes.data, 2 * n)
                got.append((idx >= 0, rp.value if idx >= 0 else -1, gts.copy() if idx >= 0 else None))
            results.append(got)
            host.nh_vcf_close(h)
        finally:
            os.environ.pop("NIMPRESS_NO_INDEX", None)
            os.environ.pop("NIMPRESS_THREADS", None)
    assert sum(1 for a in results[0] if a[0]) >= len(recs)
    for other in results[1:]:
        for a, b in zip(results[0], other):
            assert a[0] == b[0] and a[1] == b[1]
            if a[0]:
                assert np.array_equal(a[2], b[2])


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_ingest_under_sanitizers(tmp_path, san):
    """the host ingest (BGZF inflate, tabix / CSI random access, BCF decode, the per-contig record index,
    the multi-threaded window fetch) built with -fsanitize=address,undefined and -fsanitize=thread (CPU
    build only) on a multi-block vcf.gz + .tbi and a multi-block BCF + .csi: no report, and the
    streaming windows pick the same records as the all-at-once load"""
    import shutil
    import subprocess
    import bcfwriter
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "ingest")
    cmd = [gxx, "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-o", exe,
           os.path.join(ROOT, "tests", "native", "ingest_driver.cpp"),
           os.path.join(ROOT, "nimpress_amd", "csrc", "host", "nimpress_host.cpp"),
           "-L" + os.path.join(ROOT, "nimpress_amd"), "-lnps", "-lz",
           "-Wl,-rpath," + os.path.join(ROOT, "nimpress_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not installed: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    rng = np.random.default_rng(5)
    n = 9000
    samples = ["Q%d" % i for i in range(n)]
    contigs = ["1", "2", "X"]
    recs, vlines = [], []
    for contig in contigs:
        pos = 500
        for k in range(30):
            pos += int(rng.integers(1, 70000))
            ref = "A" if rng.uniform() < 0.7 else "ACGTAC"
            a = rng.integers(0, 3, size=(n, 2))
            g = (a + 1) << 1
            g[rng.uniform(size=n) < 0.03] = 0
            recs.append(dict(contig=contig, pos=pos, id=".", ref=ref, alts=["G", "T"],
                             filters=[[], ["PASS"], ["FAIL"]][k % 3], gts=g))
            txt = np.where(g[:, 0] == 0, "./.", np.char.add(np.char.add((a[:, 0]).astype(str), "/"), a[:, 1].astype(str)))
            vlines.append((contig, pos, ref, "%s\t%d\t.\t%s\tG,T\t.\t%s\t.\tGT\t%s" % (
                contig, pos, ref, ";".join(recs[-1]["filters"]) or ".", "\t".join(txt.tolist()))))
    bcf = str(tmp_path / "c.bcf")
    bcfwriter.write_bcf(bcf, contigs, samples, recs, gt_dtype=np.int16)
    vcf = str(tmp_path / "c.vcf.gz")
    write_bgzf_vcf_with_tbi(vcf, ["##fileformat=VCFv4.2",
                                  "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples)], vlines)
    lines = ["t", "", "", "x", "0.0"]
    for k in rng.permutation(len(recs))[:50]:
        lines.append("%s\t%d\t%s\t%s\t0.1\t0.2" % (recs[k]["contig"], recs[k]["pos"], recs[k]["ref"], "GT"[k % 2]))
    lines.append("2\t3\tA\tG\t0.1\t0.2")
    score = str(tmp_path / "s.score")
    open(score, "w").write("\n".join(lines))
    env = dict(os.environ, NIMPRESS_THREADS="4", ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="halt_on_error=1")
    outs = []
    for path in (bcf, vcf):
        r = subprocess.run([exe, score, path, "7"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
        for bad in ("Sanitizer", "runtime error"):
            assert bad not in r.stderr, r.stderr[-3000:]
        out = r.stdout.splitlines()
        assert len(out) == 2 * 51 and out[:51] == out[51:]        # whole load == streaming windows
        assert sum("absent" in l for l in out[:51]) == 1
        outs.append(out)
    # BCF (int16 vectors) and text VCF give the same widened GT values
    assert outs[0] == outs[1]
    # round 4: the PLINK 2 fixed-width .pgen reader and the in-place FORMAT/DS text parser under the same sanitizers, on
    # good files and on files they must refuse (truncated records, a header that lies about the samples, a .pvar line with
    # too few columns, an infinite dosage, an over-long number): a message and status 5, never a report
    import pgenwriter
    m = 40
    variants = [("1", 1000 + 37 * j, "v%d" % j, "A", "G") for j in range(m)]
    alt = rng.integers(0, 3, size=(m, n))
    miss = rng.uniform(size=(m, n)) < 0.05
    pgenwriter.write_pgen(str(tmp_path / "p"), samples, variants, alt, miss)
    pscore = str(tmp_path / "p.score")
    open(pscore, "w").write("\n".join(["t", "", "", "x", "0.0"] + ["1\t%d\tA\t%s\t0.1\t0.2" % (v[1], "GA"[j % 2])
                                                                   for j, v in enumerate(variants)] + ["1\t5\tA\tG\t0.1\t0.2"]))

    def drive(path, want_rc):
        r = subprocess.run([exe, pscore, path, "7"], capture_output=True, text=True, env=env)
        assert r.returncode == want_rc, (path, r.returncode, r.stdout[-300:], r.stderr[-2000:])
        for bad in ("Sanitizer", "runtime error"):
            assert bad not in r.stderr, r.stderr[-3000:]
        return r.stdout

    out = drive(str(tmp_path / "p.pgen"), 0)
    assert sum("absent" in l for l in out.splitlines()) == 1 and "no index" in out
    raw = open(str(tmp_path / "p.pgen"), "rb").read()
    for name, data in (("trunc", raw[:len(raw) // 2]), ("hdr", raw[:7] + np.uint32(n + 3).tobytes() + raw[11:]),
                       ("short", raw[:9])):
        for ext in (".pvar", ".psam"):
            shutil.copy(str(tmp_path / ("p" + ext)), str(tmp_path / (name + ext)))
        open(str(tmp_path / (name + ".pgen")), "wb").write(data)
        assert "refused" in drive(str(tmp_path / (name + ".pgen")), 5)
    shutil.copy(str(tmp_path / "p.pgen"), str(tmp_path / "cols.pgen"))
    shutil.copy(str(tmp_path / "p.psam"), str(tmp_path / "cols.psam"))
    open(str(tmp_path / "cols.pvar"), "w").write("#CHROM\tPOS\tID\tREF\tALT\n1\t1000\tv0\n")
    assert "refused" in drive(str(tmp_path / "cols.pgen"), 5)
    # a .pvar that does not belong to the .pgen (one record fewer / one more than the header's count), or a
    # .pgen with bytes behind its last record, is refused instead of being scored with another variant's genotypes; a
    # comment line in the .pvar other than '##...' and '#CHROM' is not a record
    pv = open(str(tmp_path / "p.pvar")).read().splitlines()
    for name, lines, data, rc in (("fewer", pv[:-1], raw, 5), ("more", pv + [pv[-1].replace("v39", "v40")], raw, 5),
                                  ("tail", pv, raw + b"\0", 5), ("note", pv[:1] + ["#a note\tnot\ta\trecord\tat\tall"] + pv[1:], raw, 0)):
        shutil.copy(str(tmp_path / "p.psam"), str(tmp_path / (name + ".psam")))
        open(str(tmp_path / (name + ".pvar")), "w").write("\n".join(lines) + "\n")
        open(str(tmp_path / (name + ".pgen")), "wb").write(data)
        o = drive(str(tmp_path / (name + ".pgen")), rc)
        assert ("refused" in o) == (rc == 5), (name, o[-300:])
    head = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples[:50])]

    def ds_vcf(name, cell):
        ds = np.round(rng.uniform(0, 2, size=(m, 50)), 3)
        lines = list(head)
        for j, v in enumerate(variants):
            cells = ["%g" % x for x in ds[j]]
            cells[j % 50] = "." if j % 7 == 0 else cells[j % 50]
            if j == 3 and cell is not None:
                cells[5] = cell
            lines.append("1\t%d\t.\tA\tG\t.\tPASS\t.\tDS\t%s" % (v[1], "\t".join(cells)))
        p = str(tmp_path / (name + ".vcf"))
        open(p, "w").write("\n".join(lines) + "\n")
        return p
    assert "refused" not in drive(ds_vcf("ds_ok", None), 0)
    assert "refused" not in drive(ds_vcf("ds_range", "2.25"), 0)        # outside [0, 2]: read (the streamed rows take it)
    assert "not finite" in drive(ds_vcf("ds_inf", "inf"), 5)
    assert "refused" in drive(ds_vcf("ds_junk", "1.5x"), 5)
    assert "refused" in drive(ds_vcf("ds_long", "0." + "1" * 80), 5)


def test_pgen_fixed_width_reader_vs_writer_parity_unpinned(host, tmp_path):
    """PLINK 2 .pgen, storage mode 0x02 (fixed-width hard calls) + .pvar + .psam: samples, records and every genotype
    come back as written (tests/pgenwriter.py); findVariant applies the VCF rule (REF must match, effect allele = REF
    or ALT); other storage modes and multi-allelic .pvar records are refused with a message.  PARITY UNPINNED: reader
    and writer are both the build's own (no plink2 / specification / reference fixture in this image)."""
    import pgenwriter
    rng = np.random.default_rng(12)
    n, m = 37, 9
    names = ["S%02d" % i for i in range(n)]
    variants = [("2", 100 + 10 * j, "v%d" % j, "AC"[j % 2], "GT"[j % 2]) for j in range(m)]
    alt = rng.integers(0, 3, size=(m, n))
    miss = rng.uniform(size=(m, n)) < 0.1
    for header, fid in ((True, False), (False, True)):
        prefix = str(tmp_path / ("g%d" % header))
        pgenwriter.write_pgen(prefix, names, variants, alt, miss, pvar_header=header, psam_fid=fid)
        h = host.nh_vcf_open((prefix + ".pgen").encode(), None)
        assert h, host.nh_last_error()
        assert host.nh_vcf_n_samples(h) == n and host.nh_vcf_n_records(h) == m
        pos, ploidy = C.c_long(), C.c_int()
        filt = C.create_string_buffer(16)
        gts = np.zeros(2 * n, np.int32)
        for j, (c, p, _, r, a) in enumerate(variants):
            for ea in (a, r):
                idx = host.nh_vcf_find(h, c.encode(), p, r.encode(), ea.encode(), C.byref(pos), C.byref(ploidy), filt, 16,
                                       gts.ctypes.data, 2 * n)
                assert idx == j and ploidy.value == 2 and filt.value == b"."
                want = np.zeros((n, 2), np.int32)               # bcf_get_genotypes layout: (allele + 1) << 1, 0 = missing
                for i in range(n):
                    if not miss[j, i]:
                        want[i] = [4 if k < alt[j, i] else 2 for k in range(2)]
                assert gts.reshape(n, 2).tolist() == want.tolist(), (j, ea)
            assert host.nh_vcf_find(h, c.encode(), p, a.encode(), r.encode(), None, None, None, 0, None, 0) == -1  # REF differs
        host.nh_vcf_close(h)
    # a storage mode the reader does not know
    raw = bytearray(open(str(tmp_path / "g1.pgen"), "rb").read())
    raw[2] = 0x10
    (tmp_path / "x.pgen").write_bytes(bytes(raw))
    for ext in (".pvar", ".psam"):
        (tmp_path / ("x" + ext)).write_bytes(open(str(tmp_path / ("g1" + ext)), "rb").read())
    assert not host.nh_vcf_open(str(tmp_path / "x.pgen").encode(), None)
    assert "storage mode 0x10" in host.nh_last_error().decode()
    # a multi-allelic .pvar line
    pgenwriter.write_pgen(str(tmp_path / "m"), names, [("2", 5, "a", "A", "C,G")], alt[:1], miss[:1])
    assert not host.nh_vcf_open(str(tmp_path / "m.pgen").encode(), None)
    assert "multi-allelic" in host.nh_last_error().decode()


def test_score_many_shards_rows_by_default_when_the_file_has_a_locus_index(tmp_path):
    """tools/score_many.py --shard auto (the 8-GPU default must not read the cohort file 8 times):
    an indexed genotype file -> rows (decided before any GPU is touched; every rank sees the same files), else files"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("score_many", os.path.join(ROOT, "tools", "score_many.py"))
    sm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sm)
    a = sm.parse_args(["x.scores", "y.scores", "c.bcf"])
    assert a.shard == "auto" and a.gpus == 1
    p = str(tmp_path / "c.bcf")
    open(p, "w").close()
    assert not sm.has_locus_index(p)
    open(p + ".csi", "w").close()
    assert sm.has_locus_index(p)
    v = str(tmp_path / "c.vcf.gz")
    open(v + ".tbi", "w").close()
    assert sm.has_locus_index(v)
    b = str(tmp_path / "p.bed")
    assert not sm.has_locus_index(b)
    open(str(tmp_path / "p.bim"), "w").close()
    assert sm.has_locus_index(b)


def test_ds16_codes_are_the_parser_s_float32_for_every_decimal():
    """NPS_FMT_DS16 (include/nps.h): code k stands for float32(double(k) * 1e-4).  That is the float32 a text parser makes
    of the decimal k / 10^4 written with one to four places -- for EVERY k in 0 .. 20 000 and every such spelling: the
    format loses nothing on FORMAT/DS values as imputation tools print them."""
    from oracle import refcpu
    for k in range(20001):
        v = refcpu.ds16_value(k)
        assert v == np.float32("%.4f" % (k / 10000.0)), k
        for d in (1, 2, 3):
            if k % 10 ** (4 - d) == 0:
                assert v == np.float32("%.*f" % (d, k / 10000.0)), (k, d)
    assert np.isnan(refcpu.ds16_value(0xffff))
