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
    L.nh_vcf_close.argtypes = [C.c_void_p]
    L.nh_vcf_n_samples.restype = C.c_long
    L.nh_vcf_n_samples.argtypes = [C.c_void_p]
    L.nh_vcf_n_records.restype = C.c_long
    L.nh_vcf_n_records.argtypes = [C.c_void_p]
    L.nh_vcf_sample.restype = C.c_char_p
    L.nh_vcf_sample.argtypes = [C.c_void_p, C.c_long]
    L.nh_vcf_find.restype = C.c_long
    L.nh_vcf_find.argtypes = [C.c_void_p, C.c_char_p, C.c_long, C.c_char_p, C.c_char_p,
                              C.POINTER(C.c_long), C.POINTER(C.c_int), C.c_char_p, C.c_long,
                              C.c_void_p, C.c_long]
    for f in ("nh_dbinom", "nh_pbinom", "nh_binom_test"):
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
