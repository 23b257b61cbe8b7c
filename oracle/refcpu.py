"""oracle/refcpu.py -- loader + small-case driver for the CPU oracle (TEST INFRASTRUCTURE).

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module.  The product (``nimpress_amd`` / ``libnps.so``) never does.

The numeric loops live in ``oracle/refcpu.c`` (a literal restatement of
``/root/reference/src/nimpress.nim``); this file adds the file handling the reference gets
from hts-nim / the Nim stdlib, in plain Python, for the small fixture-sized cases:

* ``.scores`` parsing            -- nimpress.nim:195-254
* BED loading + containment      -- nimpress.nim:262-345
* VCF (text, gz/BGZF) scan + the findVariant rule -- nimpress.nim:353-364
* the whole computePolygenicScores driver          -- nimpress.nim:592-649
"""
from __future__ import annotations

import ctypes as C
import gzip
import math
import os
import subprocess
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librefcpu.so")

LOCUS = {"ps": 0, "homref": 1, "fail": 2, "ignore": 3}       # nimpress.nim:412
MISSING = {"homref": 0, "ignore": 1}                          # nimpress.nim:413
SAMPLE = {"ps": 0, "homref": 1, "fail": 2, "int_ps": 3, "int_fail": 4}  # nimpress.nim:414

ROW_PRESENT, ROW_UNCOVERED, ROW_ABSENT, ROW_FILTERED = 0, 1, 2, 3
REASON_NAMES = ["genotyped", "uncovered", "absent", "filtered", "maxmis"]


class RefParams(C.Structure):
    _fields_ = [("imp_locus", C.c_int32), ("imp_missing", C.c_int32), ("imp_sample", C.c_int32),
                ("_pad", C.c_int32), ("max_missing_rate", C.c_double), ("min_cs", C.c_int64)]


class RefLocusStat(C.Structure):
    _fields_ = [("ngenotyped", C.c_double), ("nmissing", C.c_double), ("neffect", C.c_double),
                ("used", C.c_int32), ("reason", C.c_int32)]


STAT_DTYPE = np.dtype([("ngenotyped", "<f8"), ("nmissing", "<f8"), ("neffect", "<f8"),
                       ("used", "<i4"), ("reason", "<i4")])


def build(force: bool = False) -> str:
    """Compile oracle/refcpu.c -> oracle/librefcpu.so (gcc; seconds)."""
    src = os.path.join(_HERE, "refcpu.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "librefcpu.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    dp, i32p, u32p, fp = (C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint32),
                          C.POINTER(C.c_float))
    L.ref_tally_alleles.argtypes = [dp, C.c_size_t, dp, dp, dp]
    L.ref_tally_alleles.restype = None
    L.ref_raw_dosages_gt.argtypes = [dp, i32p, C.c_size_t, C.c_int, C.c_int]
    L.ref_raw_dosages_gt.restype = None
    L.ref_begin.argtypes = [C.c_size_t, C.POINTER(RefParams)]
    L.ref_begin.restype = C.c_void_p
    L.ref_row_gt.argtypes = [C.c_void_p, i32p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                             C.POINTER(RefLocusStat)]
    L.ref_row_gt.restype = None
    L.ref_row_ds.argtypes = [C.c_void_p, fp, C.c_int, C.c_double, C.c_double,
                             C.POINTER(RefLocusStat)]
    L.ref_row_ds.restype = None
    L.ref_row_raw.argtypes = [C.c_void_p, dp, C.c_int, C.c_double, C.c_double,
                              C.POINTER(RefLocusStat)]
    L.ref_row_raw.restype = None
    L.ref_row_locus.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double,
                                C.POINTER(RefLocusStat)]
    L.ref_row_locus.restype = None
    L.ref_finish.argtypes = [C.c_void_p, C.c_double, dp, C.POINTER(C.c_int64)]
    L.ref_finish.restype = None
    L.ref_partial.argtypes = [C.c_void_p, dp, C.POINTER(C.c_int64)]
    L.ref_partial.restype = None
    L.ref_score_packed.argtypes = [u32p, C.c_size_t, C.c_size_t, C.c_size_t, i32p, i32p, dp, dp,
                                   C.POINTER(RefParams), C.c_double, dp, C.c_void_p,
                                   C.POINTER(C.c_int64)]
    L.ref_score_packed.restype = None
    for name in ("ref_dbinom", "ref_pbinom", "ref_binom_test"):
        f = getattr(L, name)
        f.argtypes = [C.c_int64, C.c_int64, C.c_double]
        f.restype = C.c_double
    L.ref_betai.argtypes = [C.c_double, C.c_double, C.c_double]
    L.ref_betai.restype = C.c_double
    L.ref_interval_contains.argtypes = [C.c_int64] * 4
    L.ref_interval_contains.restype = C.c_int
    L.ref_synth_rows.argtypes = [u32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                 C.c_uint64, u32p, u32p, u32p]
    L.ref_synth_rows.restype = None
    L.ref_synth_rows_ds.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                    C.c_uint64, u32p, u32p, u32p]
    L.ref_synth_rows_ds.restype = None
    L.ref_synth_rows_ds16.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                    C.c_uint64, u32p, u32p, u32p]
    L.ref_synth_rows_ds16.restype = None
    L.ref_ds16_value.argtypes = [C.c_uint32]
    L.ref_ds16_value.restype = C.c_float
    L.ref_codes_to_gt.argtypes = [u32p, C.c_size_t, i32p]
    L.ref_codes_to_gt.restype = None
    L.ref_bench_gt.argtypes = [i32p, C.c_size_t, C.c_size_t, C.c_size_t, dp, dp,
                               C.POINTER(RefParams), dp, C.POINTER(C.c_int64)]
    L.ref_bench_gt.restype = C.c_double
    u64p = C.POINTER(C.c_uint64)
    L.ref_score_subset.argtypes = [C.c_int, u64p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                   C.c_uint64, u32p, u32p, u32p, dp, dp, i32p, dp, dp, dp,
                                   C.POINTER(RefParams), C.c_int, dp, C.POINTER(C.c_int64)]
    L.ref_score_subset.restype = None
    L.ref_tally_synth_rows.argtypes = [C.c_int, u64p, C.c_size_t, C.c_size_t, C.c_uint64, u32p, u32p,
                                       u32p, i32p, C.c_int, dp, dp, dp]
    L.ref_tally_synth_rows.restype = None
    L.ref_bench_gt_full.argtypes = [i32p, C.c_size_t, C.c_size_t, C.c_size_t, dp, dp,
                                    C.POINTER(RefParams), C.c_int, C.c_double, dp,
                                    C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.ref_bench_gt_full.restype = C.c_double
    L.ref_bench_gt_allcores.argtypes = [i32p, C.c_size_t, C.c_size_t, C.c_size_t, dp, dp,
                                        C.POINTER(RefParams), C.c_int, dp, C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int)]
    L.ref_bench_gt_allcores.restype = C.c_double
    _lib = L
    return L


def host_threads(cap: int = 64) -> int:
    """threads for the checker's OpenMP helpers: the CPUs this process may use, at most `cap` (the CPU baseline of
    bench.py asks for 16, the share of a GPU box one GPU's job gets; the full-size checks take what is there)"""
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return max(1, min(usable, cap))


def make_params(imp_locus="ps", imp_missing="homref", imp_sample="int_ps", maxmis=0.05,
                mincs=100) -> RefParams:
    return RefParams(LOCUS[imp_locus], MISSING[imp_missing], SAMPLE[imp_sample], 0,
                     float(maxmis), int(mincs))


def _p(a: np.ndarray, ty):
    return a.ctypes.data_as(C.POINTER(ty))


# ----------------------------------------------------------------------------------------
# stats (nimpress.nim:50-188)
def dbinom(x, n, p): return lib().ref_dbinom(int(x), int(n), float(p))
def pbinom(x, n, p): return lib().ref_pbinom(int(x), int(n), float(p))
def binom_test(x, n, p): return lib().ref_binom_test(int(x), int(n), float(p))
def betai(a, b, x): return lib().ref_betai(float(a), float(b), float(x))


# ----------------------------------------------------------------------------------------
# streaming scorer over in-memory rows
class RefScorer:
    """computePolygenicScores (nimpress.nim:592-649) fed one score row at a time."""

    def __init__(self, n_samples: int, params: RefParams):
        self.n = int(n_samples)
        self._p = params
        self._s = lib().ref_begin(self.n, C.byref(params))
        self.stats: List[Tuple[float, float, float, int, int]] = []

    def _push_stat(self, st: RefLocusStat):
        self.stats.append((st.ngenotyped, st.nmissing, st.neffect, st.used, st.reason))

    def row_gt(self, gts: np.ndarray, ploidy: int, eaidx: int, ref_is_effect: bool, beta: float,
               eaf: float):
        gts = np.ascontiguousarray(gts, dtype=np.int32)
        assert gts.size == self.n * ploidy
        st = RefLocusStat()
        lib().ref_row_gt(self._s, _p(gts, C.c_int32), ploidy, eaidx, int(ref_is_effect),
                         float(beta), float(eaf), C.byref(st))
        self._push_stat(st)

    def row_ds(self, ds: np.ndarray, ref_is_effect: bool, beta: float, eaf: float):
        ds = np.ascontiguousarray(ds, dtype=np.float32)
        assert ds.size == self.n
        st = RefLocusStat()
        lib().ref_row_ds(self._s, _p(ds, C.c_float), int(ref_is_effect), float(beta), float(eaf),
                         C.byref(st))
        self._push_stat(st)

    def row_locus(self, status: int, ref_is_effect: bool, beta: float, eaf: float):
        st = RefLocusStat()
        lib().ref_row_locus(self._s, status, int(ref_is_effect), float(beta), float(eaf),
                            C.byref(st))
        self._push_stat(st)

    def partial(self) -> Tuple[np.ndarray, int]:
        """un-normalised sums and nloci so far (the state is kept)"""
        out = np.empty(max(self.n, 1), dtype=np.float64)
        nloci = C.c_int64(0)
        lib().ref_partial(self._s, _p(out, C.c_double), C.byref(nloci))
        return out[: self.n], int(nloci.value)

    def finish(self, offset: float) -> Tuple[np.ndarray, int]:
        out = np.empty(max(self.n, 1), dtype=np.float64)
        nloci = C.c_int64(0)
        lib().ref_finish(self._s, float(offset), _p(out, C.c_double), C.byref(nloci))
        self._s = None
        return out[: self.n], int(nloci.value)


def score_packed(codes: np.ndarray, n: int, kind, ref_is_effect, beta, eaf, params: RefParams,
                 offset: float):
    """Whole packed matrix (2-bit codes, [rows, stride_words] uint32) through the literal path."""
    codes = np.ascontiguousarray(codes, dtype=np.uint32)
    kind = np.ascontiguousarray(kind, dtype=np.int32)
    rie = np.ascontiguousarray(ref_is_effect, dtype=np.int32)
    beta = np.ascontiguousarray(beta, dtype=np.float64)
    eaf = np.ascontiguousarray(eaf, dtype=np.float64)
    m = kind.size
    stride = codes.shape[1] if codes.ndim == 2 and codes.size else max((n + 15) // 16, 1)
    scores = np.empty(max(n, 1), dtype=np.float64)
    stats = np.zeros(max(m, 1), dtype=STAT_DTYPE)
    nloci = C.c_int64(0)
    lib().ref_score_packed(_p(codes, C.c_uint32) if codes.size else None, stride, n, m,
                           _p(kind, C.c_int32), _p(rie, C.c_int32), _p(beta, C.c_double),
                           _p(eaf, C.c_double), C.byref(params), float(offset),
                           _p(scores, C.c_double), stats.ctypes.data, C.byref(nloci))
    return scores[:n], stats[:m], int(nloci.value)


# ----------------------------------------------------------------------------------------
# synthetic cohorts (DESIGN.md "Synthetic cohorts") -- thresholds are shared integers
def hwe_thresholds(eaf: np.ndarray, miss: np.ndarray):
    """uint32 thresholds: g < t_hom -> 2, g < t_het -> 1, else 0; ms < t_miss -> missing."""
    eaf = np.asarray(eaf, dtype=np.float64)
    miss = np.asarray(miss, dtype=np.float64)
    p_hom = eaf * eaf
    p_het = 2.0 * eaf * (1.0 - eaf)
    scale = 4294967296.0
    t_hom = np.minimum(np.floor(p_hom * scale), 4294967295.0).astype(np.uint32)
    t_het = np.minimum(np.floor((p_hom + p_het) * scale), 4294967295.0).astype(np.uint32)
    t_miss = np.minimum(np.floor(miss * scale), 4294967295.0).astype(np.uint32)
    return t_het, t_hom, t_miss


def synth_rows(n: int, row0: int, nrows: int, seed: int, t_het, t_hom, t_miss,
               stride_words: Optional[int] = None) -> np.ndarray:
    stride = stride_words or max((n + 15) // 16, 1)
    out = np.zeros((nrows, stride), dtype=np.uint32)
    th, tm, tmi = (np.ascontiguousarray(a, dtype=np.uint32) for a in (t_het, t_hom, t_miss))
    if nrows:
        lib().ref_synth_rows(_p(out, C.c_uint32), stride, n, row0, nrows, seed,
                             _p(th, C.c_uint32), _p(tm, C.c_uint32), _p(tmi, C.c_uint32))
    return out


def synth_rows_ds(n: int, row0: int, nrows: int, seed: int, t_het, t_hom, t_miss) -> np.ndarray:
    out = np.zeros((nrows, max(n, 1)), dtype=np.float32)
    th, tm, tmi = (np.ascontiguousarray(a, dtype=np.uint32) for a in (t_het, t_hom, t_miss))
    if nrows and n:
        lib().ref_synth_rows_ds(_p(out, C.c_float), max(n, 1), n, row0, nrows, seed,
                                _p(th, C.c_uint32), _p(tm, C.c_uint32), _p(tmi, C.c_uint32))
    return out[:, :n]


def synth_rows_ds16(n: int, row0: int, nrows: int, seed: int, t_het, t_hom, t_miss) -> np.ndarray:
    """the generator of NPS_FMT_DS16 cohorts: three-decimal dosages as the float32 a parser makes of them (NaN = missing)"""
    out = np.zeros((nrows, max(n, 1)), dtype=np.float32)
    th, tm, tmi = (np.ascontiguousarray(a, dtype=np.uint32) for a in (t_het, t_hom, t_miss))
    if nrows and n:
        lib().ref_synth_rows_ds16(_p(out, C.c_float), max(n, 1), n, row0, nrows, seed,
                                  _p(th, C.c_uint32), _p(tm, C.c_uint32), _p(tmi, C.c_uint32))
    return out[:, :n]


def ds16_value(k: int) -> np.float32:
    """the float32 value of code k of a NPS_FMT_DS16 cohort (0xFFFF: NaN)"""
    return np.float32(lib().ref_ds16_value(int(k)))


def score_subset(samples, n_total: int, row0: int, seed: int, t_het, t_hom, t_miss, beta, eaf, rie,
                 row_ngen, row_nmiss, row_neff, params: RefParams, is_ds=False):  # is_ds: False / True / 2 (the NPS_FMT_DS16 generator)
    """The chosen samples of a synthetic cohort scored over rows [row0, row0 + m) with the restated procs,
    given every row's whole-row tally (see refcpu.c "Full-size checks").  Returns (un-normalised sums,
    nloci): the state of the reference's loop at nimpress.nim:641."""
    samples = np.ascontiguousarray(samples, dtype=np.uint64)
    th, tm, tmi = (np.ascontiguousarray(a, dtype=np.uint32) for a in (t_het, t_hom, t_miss))
    m = th.size
    beta, eaf, g, ms, ne = (np.ascontiguousarray(a, dtype=np.float64)
                            for a in (beta, eaf, row_ngen, row_nmiss, row_neff))
    rie = np.ascontiguousarray(np.broadcast_to(np.asarray(rie, dtype=np.int32), (m,)))
    assert beta.size == eaf.size == g.size == ms.size == ne.size == m
    sums = np.zeros(max(samples.size, 1), dtype=np.float64)
    nloci = C.c_int64(0)
    lib().ref_score_subset(int(is_ds), _p(samples, C.c_uint64), samples.size, n_total, row0, m, seed,
                           _p(th, C.c_uint32), _p(tm, C.c_uint32), _p(tmi, C.c_uint32),
                           _p(beta, C.c_double), _p(eaf, C.c_double), _p(rie, C.c_int32),
                           _p(g, C.c_double), _p(ms, C.c_double), _p(ne, C.c_double), C.byref(params),
                           host_threads(), _p(sums, C.c_double), C.byref(nloci))
    return sums[: samples.size], int(nloci.value)


def tally_synth_rows(rows, n: int, seed: int, t_het, t_hom, t_miss, rie=0, is_ds: bool = False):
    """Literal whole-row recount (decode + tallyAlleles over all n samples) of the given rows; the
    threshold arrays hold one entry per SELECTED row.  Returns (ngenotyped, nmissing, neffect)."""
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    th, tm, tmi = (np.ascontiguousarray(a, dtype=np.uint32) for a in (t_het, t_hom, t_miss))
    assert th.size == tm.size == tmi.size == rows.size
    rie = np.ascontiguousarray(np.broadcast_to(np.asarray(rie, dtype=np.int32), (rows.size,)))
    out = [np.zeros(max(rows.size, 1), dtype=np.float64) for _ in range(3)]
    lib().ref_tally_synth_rows(int(is_ds), _p(rows, C.c_uint64), rows.size, n, seed,
                               _p(th, C.c_uint32), _p(tm, C.c_uint32), _p(tmi, C.c_uint32),
                               _p(rie, C.c_int32), host_threads(), *(_p(o, C.c_double) for o in out))
    return tuple(o[: rows.size] for o in out)


def codes_to_gt(row: np.ndarray, n: int) -> np.ndarray:
    row = np.ascontiguousarray(row, dtype=np.uint32)
    out = np.empty(2 * max(n, 1), dtype=np.int32)
    lib().ref_codes_to_gt(_p(row, C.c_uint32), n, _p(out, C.c_int32))
    return out[: 2 * n]


def bench_gt(gts_rows: np.ndarray, n: int, m: int, beta, eaf, params: RefParams):
    """CPU baseline leg of bench.py: seconds for m rows of the literal per-row path (1 thread)."""
    gts_rows = np.ascontiguousarray(gts_rows, dtype=np.int32)
    n_distinct = gts_rows.shape[0]
    assert gts_rows.shape[1] == 2 * n
    beta = np.ascontiguousarray(beta, dtype=np.float64)
    eaf = np.ascontiguousarray(eaf, dtype=np.float64)
    scores = np.empty(max(n, 1), dtype=np.float64)
    nloci = C.c_int64(0)
    secs = lib().ref_bench_gt(_p(gts_rows, C.c_int32), n_distinct, n, m, _p(beta, C.c_double),
                              _p(eaf, C.c_double), C.byref(params), _p(scores, C.c_double),
                              C.byref(nloci))
    return float(secs), scores[:n], int(nloci.value)


def bench_gt_full(gts_rows: np.ndarray, n: int, m: int, beta, eaf, params: RefParams,
                  with_binomtest: bool, afmisp: float = 0.001):
    """As bench_gt, optionally with the binomTest call of nimpress.nim:573 in its place.
    Returns (seconds, scores, nloci, rows that would have been warned about)."""
    gts_rows = np.ascontiguousarray(gts_rows, dtype=np.int32)
    beta = np.ascontiguousarray(beta, dtype=np.float64)
    eaf = np.ascontiguousarray(eaf, dtype=np.float64)
    scores = np.empty(max(n, 1), dtype=np.float64)
    nloci, warned = C.c_int64(0), C.c_int64(0)
    secs = lib().ref_bench_gt_full(_p(gts_rows, C.c_int32), gts_rows.shape[0], n, m,
                                   _p(beta, C.c_double), _p(eaf, C.c_double), C.byref(params),
                                   int(bool(with_binomtest)), float(afmisp), _p(scores, C.c_double),
                                   C.byref(nloci), C.byref(warned))
    return float(secs), scores[:n], int(nloci.value), int(warned.value)


def bench_gt_allcores(gts_rows: np.ndarray, n: int, m: int, beta, eaf, params: RefParams,
                      threads: Optional[int] = None):
    """The same passes split over the samples with OpenMP on all host cores (NOT a restatement: the
    reference is single-threaded).  Returns (seconds, scores, nloci, threads)."""
    gts_rows = np.ascontiguousarray(gts_rows, dtype=np.int32)
    beta = np.ascontiguousarray(beta, dtype=np.float64)
    eaf = np.ascontiguousarray(eaf, dtype=np.float64)
    scores = np.empty(max(n, 1), dtype=np.float64)
    nloci, used = C.c_int64(0), C.c_int(0)
    secs = lib().ref_bench_gt_allcores(_p(gts_rows, C.c_int32), gts_rows.shape[0], n, m,
                                       _p(beta, C.c_double), _p(eaf, C.c_double), C.byref(params),
                                       int(threads or host_threads()), _p(scores, C.c_double),
                                       C.byref(nloci), C.byref(used))
    return float(secs), scores[:n], int(nloci.value), int(used.value)


# ----------------------------------------------------------------------------------------
# file handling for fixture-sized cases (pure Python)
@dataclass
class ScoreEntry:  # nimpress.nim:221-231
    contig: str
    pos: int
    refseq: str
    easeq: str
    beta: float
    eaf: float

    @property
    def stop(self) -> int:
        return self.pos + len(self.refseq) - 1


@dataclass
class ScoreFile:  # nimpress.nim:195-254
    name: str
    desc: str
    cite: str
    genomever: str
    offset: float
    entries: List[ScoreEntry]


def nim_parse_float(s: str) -> float:
    """Nim parseFloat accepts nan/inf spellings case-insensitively; so does Python float()."""
    return float(s)


def read_score_file(path: str) -> ScoreFile:
    with open(path, "r", newline="") as fh:
        text = fh.read()
    # Nim readLine splits on \n, \r\n or \r and a final unterminated line is still a line
    lines = text.replace("\r\n", "\n").replace("\r", "\n").split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    hdr = [ln.rstrip() for ln in lines[:5]]      # strip(leading=false), :238-243
    if len(hdr) < 5:
        raise ValueError("score file has fewer than 5 header lines")
    entries = []
    for ln in lines[5:]:
        parts = ln.rstrip().split("\t")
        if len(parts) != 6:                        # doAssert :252
            raise AssertionError("score row does not have 6 tab-separated fields: %r" % ln)
        entries.append(ScoreEntry(parts[0], int(parts[1]), parts[2], parts[3],
                                  nim_parse_float(parts[4]), nim_parse_float(parts[5])))
    return ScoreFile(hdr[0], hdr[1], hdr[2], hdr[3], nim_parse_float(hdr[4]), entries)


def read_bed(path: str) -> Dict[str, List[Tuple[int, int]]]:  # nimpress.nim:278-308
    ivals: Dict[str, List[Tuple[int, int]]] = {}
    with open(path) as fh:
        for ln in fh.read().splitlines():
            parts = ln.rstrip().split("\t")
            assert len(parts) >= 3
            ivals.setdefault(parts[0], []).append((int(parts[1]), int(parts[2])))
    return ivals


def is_variant_covered(e: ScoreEntry, ivals: Dict[str, List[Tuple[int, int]]]) -> bool:
    # nimpress.nim:313-345; lapper only pre-selects overlapping intervals, the decision is :310-311
    for (s, t) in ivals.get(e.contig, ()):
        if lib().ref_interval_contains(s, t, e.pos, e.stop):
            return True
    return False


@dataclass
class VcfRecord:
    contig: str
    pos: int
    ref: str
    alts: List[str]
    filt: str
    gts: np.ndarray  # int32 [n_samples * ploidy], bcf_get_genotypes layout (None: the record is scored from FORMAT/DS)
    ploidy: int
    ds: Optional[np.ndarray] = None  # float32 [n_samples, values per sample] ALT dosages, NaN = missing (build-defined)


@dataclass
class Vcf:
    samples: List[str]
    records: List[VcfRecord] = field(default_factory=list)


def _encode_gt(field_: str) -> List[int]:
    out = []
    tok = ""
    phased = 0
    first = True
    alleles = []
    seps = []
    for ch in field_:
        if ch in "/|":
            alleles.append(tok)
            seps.append(ch)
            tok = ""
        else:
            tok += ch
    alleles.append(tok)
    for k, a in enumerate(alleles):
        ph = 1 if (k > 0 and seps[k - 1] == "|") else 0
        if a == "." or a == "":
            out.append(0 | ph)
        else:
            out.append(((int(a) + 1) << 1) | ph)
    return out


def ds_row(rec: "VcfRecord", eaidx: int) -> np.ndarray:
    """The float32 dosage row scored for effect allele index eaidx (build-defined FORMAT/DS extension): ALT[k-1]'s
    column, or for the REF allele the float32 sum of the ALT dosages (ref_row_ds then takes 2 - sum)."""
    d = rec.ds
    if eaidx >= 1:
        return d[:, eaidx - 1].copy() if eaidx <= d.shape[1] else np.full(d.shape[0], np.nan, np.float32)
    out = np.zeros(d.shape[0], dtype=np.float32)
    for k in range(d.shape[1]):          # float32, column after column, as the host does
        col = d[:, k]
        pad = np.isnan(col) & (col.view(np.uint32) == 0x7F800002)   # end-of-vector padding is skipped
        out = (out + np.where(pad, np.float32(0), col)).astype(np.float32)
    return out


def read_vcf(path: str, prefer_ds: bool = False) -> Vcf:
    """Text VCF (plain or gzip/BGZF).  CRLF tolerant (tests/set1.vcf.gz has CRLF endings).  A record is read from
    FORMAT/DS instead of FORMAT/GT when it has no GT, or when prefer_ds is set and it has DS (build-defined)."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt", newline="") as fh:
        text = fh.read()
    vcf = Vcf(samples=[])
    for ln in text.split("\n"):
        ln = ln.rstrip("\r")
        if not ln:
            continue
        if ln.startswith("##"):
            continue
        if ln.startswith("#CHROM"):
            vcf.samples = ln.split("\t")[9:]
            continue
        f = ln.split("\t")
        fmt = f[8].split(":")
        if "DS" in fmt and ("GT" not in fmt or prefer_ds):
            di = fmt.index("DS")
            vals = [[np.float32(np.nan) if (x in (".", "")) else np.float32(x) for x in
                     (s.split(":")[di] if len(s.split(":")) > di else ".").split(",")] for s in f[9:]]
            per_s = max(len(v) for v in vals) if vals else 1
            eov = np.array([0x7F800002], dtype=np.uint32).view(np.float32)[0]
            ds = np.full((len(vals), per_s), eov, dtype=np.float32)
            for i, v in enumerate(vals):
                ds[i, :len(v)] = v
            vcf.records.append(VcfRecord(f[0], int(f[1]), f[3], f[4].split(",") if f[4] != "." else [],
                                         f[6], None, 0, ds))
            continue
        gi = fmt.index("GT")
        per = [_encode_gt(s.split(":")[gi]) for s in f[9:]]
        ploidy = max(len(x) for x in per) if per else 2
        gts = np.empty(len(per) * ploidy, dtype=np.int32)
        for i, x in enumerate(per):
            x = x + [-2147483647] * (ploidy - len(x))   # bcf_int32_vector_end pad
            gts[i * ploidy:(i + 1) * ploidy] = x
        vcf.records.append(VcfRecord(f[0], int(f[1]), f[3], f[4].split(",") if f[4] != "." else [],
                                     f[6], gts, ploidy))
    return vcf


def find_variant(vcf: Vcf, e: ScoreEntry) -> Optional[VcfRecord]:
    """nimpress.nim:353-364 -- records overlapping contig:pos-stop in file order; first with
    REF == ref and (ea == ref or ea in ALT).  POS itself is never compared."""
    for r in vcf.records:
        if r.contig != e.contig:
            continue
        r_end = r.pos + len(r.ref) - 1
        if r.pos <= e.stop and r_end >= e.pos:   # htslib region overlap, 1-based inclusive
            if r.ref == e.refseq:
                if e.easeq == e.refseq:
                    return r
                if e.easeq in r.alts:
                    return r
    return None


def compute_polygenic_scores(score: ScoreFile, vcf: Vcf, restrict_to_covered: bool, ivals,
                             imp_locus: str, imp_missing: str, imp_sample: str, maxmis: float,
                             mincs: int, ignore_filter: bool):
    """nimpress.nim:592-649 + 484-585 over in-memory files.  Returns (scores, nloci, stats)."""
    sc = RefScorer(len(vcf.samples), make_params(imp_locus, imp_missing, imp_sample, maxmis, mincs))
    for e in score.entries:
        rie = e.refseq == e.easeq
        if restrict_to_covered and not is_variant_covered(e, ivals):        # :526-531
            sc.row_locus(ROW_UNCOVERED, rie, e.beta, e.eaf)
            continue
        rec = find_variant(vcf, e)                                           # :533
        if rec is None:                                                      # :536-551
            sc.row_locus(ROW_ABSENT, rie, e.beta, e.eaf)
            continue
        if not ignore_filter and rec.filt != "." and rec.filt != "PASS":    # :553-558
            sc.row_locus(ROW_FILTERED, rie, e.beta, e.eaf)
            continue
        eaidx = 0 if rie else rec.alts.index(e.easeq) + 1                    # :375-379
        if rec.ds is not None:                                               # build-defined FORMAT/DS row
            sc.row_ds(ds_row(rec, eaidx), rie, e.beta, e.eaf)
        else:
            sc.row_gt(rec.gts, rec.ploidy, eaidx, rie, e.beta, e.eaf)
    scores, nloci = sc.finish(score.offset)
    return scores, nloci, sc.stats


def format_score(x: float) -> str:
    """Nim `$float` as the reference prints it (pinned by scores/*_nimpress_res.txt):
    C "%.16g", plus ".0" when the result has no '.', 'e', 'n' or 'i'."""
    if math.isnan(x):
        return "nan"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    s = "%.16g" % x
    if not any(c in s for c in ".en"):
        s += ".0"
    return s
