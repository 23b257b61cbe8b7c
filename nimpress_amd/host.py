"""ctypes binding of the C++ host layer (libnimpress_host.so): the reference's
computePolygenicScores over real files (.scores + VCF/vcf.gz [+ BED]), with the row loop on the GPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import numpy as np

from . import capi

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libnimpress_host.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise capi.NpsError(-2, "libnimpress_host.so not built (%s)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.nh_last_error.restype = C.c_char_p
        L.nh_compute.restype = C.c_long
        L.nh_compute.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                 C.c_double, C.c_double, C.c_long, C.c_int, C.c_int, C.c_void_p,
                                 C.c_long, C.POINTER(C.c_ulonglong), C.c_char_p, C.c_long]
        L.nh_compute_multi.restype = C.c_long
        L.nh_compute_multi.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double,
                                       C.c_double, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p,
                                       C.c_char_p, C.c_long]
        L.nh_compute_multi_partial.restype = C.c_long
        L.nh_compute_multi_partial.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double,
                                               C.c_double, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                               C.c_long, C.c_void_p, C.c_void_p, C.c_char_p, C.c_long]
        L.nh_compute_dev.restype = C.c_long
        L.nh_compute_dev.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                     C.c_long, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_ulonglong), C.c_char_p, C.c_long]
        L.nh_compute_multi_dev.restype = C.c_long
        L.nh_compute_multi_dev.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double,
                                           C.c_double, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_char_p, C.c_long]
        L.nh_last_log_size.restype = C.c_long
        L.nh_last_log.restype = C.c_long
        L.nh_last_log.argtypes = [C.c_char_p, C.c_long]
        L.nh_last_timings.restype = None
        L.nh_last_timings.argtypes = [C.c_void_p]
        L.nh_vcf_open.restype = C.c_void_p
        L.nh_vcf_open.argtypes = [C.c_char_p, C.c_char_p]
        L.nh_vcf_close.argtypes = [C.c_void_p]
        L.nh_vcf_n_samples.restype = C.c_long
        L.nh_vcf_n_samples.argtypes = [C.c_void_p]
        L.nh_vcf_open_header.restype = C.c_void_p
        L.nh_vcf_open_header.argtypes = [C.c_char_p]
        L.nh_vcf_sample.restype = C.c_char_p
        L.nh_vcf_sample.argtypes = [C.c_void_p, C.c_long]
        L.nh_vcf_samples_joined.restype = C.c_long
        L.nh_vcf_samples_joined.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
        L.nh_format_float.restype = None
        L.nh_format_float.argtypes = [C.c_double, C.c_char_p, C.c_long]
        L.nh_write_matrix_tsv.restype = C.c_long
        L.nh_write_matrix_tsv.argtypes = [C.c_char_p, C.c_char_p, C.c_long, C.c_void_p, C.c_long, C.c_long]
        _lib = L
    return _lib


def _full_log(L, buf) -> str:
    """the log text of the call that just returned: the caller's buffer, or -- when that was too small and ends in the
    library's "... log truncated" line -- the complete text the library kept (nh_last_log)"""
    size = L.nh_last_log_size()
    if size < len(buf):
        return buf.value.decode("utf-8", "replace")
    big = C.create_string_buffer(size + 1)
    L.nh_last_log(big, len(big))
    return big.value.decode("utf-8", "replace")


TIMING_KEYS = ("hip_init_s", "hip_init_wait_s", "open_s", "inflate_parse_s", "push_s", "kernel_s", "warnings_s")


def last_timings() -> dict:
    """where the last compute_polygenic_scores* call of this thread spent its time (seconds; hip_init_s ran on a thread
    of its own beside open + inflate + parse, hip_init_wait_s is what of it the run had to wait for)"""
    t = np.zeros(7, dtype=np.float64)
    load().nh_last_timings(t.ctypes.data)
    return dict(zip(TIMING_KEYS, (float(x) for x in t)))


def compute_polygenic_scores(score_path: str, vcf_path: str, cov: Optional[str] = None,
                             imp_locus: str = "ps", imp_missing: str = "homref",
                             imp_sample: str = "int_ps", maxmis: float = 0.05, mincs: int = 100,
                             afmisp: float = 0.001, ignorefilt: bool = False, device: int = 0,
                             max_samples: int = 1 << 22, d_out: Optional[int] = None) -> Tuple[np.ndarray, int, List[str]]:
    """nimpress's main() minus the printing (CLI defaults).  Returns (scores, nloci, log lines).  d_out: a device
    pointer to n_samples doubles -- the scores are left there (nps_finish_device) and None is returned for them."""
    L = load()
    nloci = C.c_ulonglong(0)
    log = C.create_string_buffer(1 << 20)
    args = (score_path.encode(), vcf_path.encode(), cov.encode() if cov else None,
            capi.LOCUS[imp_locus], capi.MISSING[imp_missing], capi.SAMPLE[imp_sample],
            float(maxmis), float(afmisp), int(mincs), int(ignorefilt), device)
    if d_out is not None:
        scores = None
        n = L.nh_compute_dev(*args, C.c_void_p(int(d_out)), C.byref(nloci), log, len(log))
    else:
        scores = np.empty(max_samples, dtype=np.float64)
        n = L.nh_compute(*args, scores.ctypes.data, max_samples, C.byref(nloci), log, len(log))
    if n < 0:
        raise capi.NpsError(-3 if n == -2 else -1, L.nh_last_error().decode("utf-8", "replace"))
    return (None if scores is None else scores[:n].copy()), int(nloci.value), [l for l in _full_log(L, log).split("\n") if l]


def compute_polygenic_scores_multi(score_paths, vcf_path: str, cov: Optional[str] = None, imp_locus: str = "ps",
                                   imp_missing: str = "homref", imp_sample: str = "int_ps", maxmis: float = 0.05,
                                   mincs: int = 100, afmisp: float = 0.001, ignorefilt: bool = False, device: int = 0,
                                   max_samples: int = 1 << 22, d_out: Optional[int] = None):
    """Several score files on one genotype file in ONE pass over the genotypes (computePolygenicScoresMulti: the union
    of the files' loci decoded once into a resident cohort, all definitions applied together on the matrix cores).
    Returns (scores [files, samples], nloci [files], log lines per file).  d_out: a device pointer to
    [files, samples] doubles (samples = the cohort's) -- the scores are left there (nps_multi_finish_device) and None is
    returned for them."""
    L = load()
    S = len(score_paths)
    nloci = np.zeros(S, dtype=np.uint64)
    log = C.create_string_buffer(4 << 20)
    args = ("\n".join(score_paths).encode(), vcf_path.encode(), cov.encode() if cov else None,
            capi.LOCUS[imp_locus], capi.MISSING[imp_missing], capi.SAMPLE[imp_sample], float(maxmis),
            float(afmisp), int(mincs), int(ignorefilt), device)
    if d_out is not None:
        scores = None
        n = L.nh_compute_multi_dev(*args, 0, 1, 0, C.c_void_p(int(d_out)), nloci.ctypes.data, None, log, len(log))
    else:
        scores = np.empty((S, max_samples), dtype=np.float64)
        n = L.nh_compute_multi(*args, scores.ctypes.data, max_samples, nloci.ctypes.data, log, len(log))
    if n < 0:
        raise capi.NpsError(-3 if n == -2 else -1, L.nh_last_error().decode("utf-8", "replace"))
    return (None if scores is None else scores[:, :n].copy()), nloci.astype(np.int64), _split_logs(_full_log(L, log), S)


def _split_logs(text: str, S: int):
    """"<file index> TAB line" -> the lines of every file (a line without a valid index -- there is none unless the
    text was cut -- is dropped rather than attached to the wrong file)"""
    logs = [[] for _ in range(S)]
    for l in text.split("\n"):
        k, tab, rest = l.partition("\t")
        if tab and k.isdigit() and int(k) < S:
            logs[int(k)].append(rest)
    return logs


def compute_polygenic_scores_multi_partial(score_paths, vcf_path: str, shard: int, n_shards: int, cov: Optional[str] = None,
                                           imp_locus: str = "ps", imp_missing: str = "homref",
                                           imp_sample: str = "int_ps", maxmis: float = 0.05, mincs: int = 100,
                                           afmisp: float = 0.001, ignorefilt: bool = False, device: int = 0,
                                           max_samples: int = 1 << 22, d_out: Optional[int] = None):
    """Rows sharded over several GPUs x ALL score files on each: block `shard` of `n_shards` of
    the union of the files' loci is located, decoded and scored on this GPU -- 1 / n_shards of the ingest and of the
    cohort.  Returns (sums [files, samples] BEFORE the normalisation, nloci [files] of the block, offsets [files],
    log lines per file for the block's rows); the caller sum-all-reduces sums and nloci over the shards
    (multi.all_reduce_partial_matrix) and applies sums / (2 nloci) + offset (multi.normalize_matrix)."""
    L = load()
    S = len(score_paths)
    nloci = np.zeros(S, dtype=np.uint64)
    offsets = np.zeros(S, dtype=np.float64)
    log = C.create_string_buffer(4 << 20)
    args = ("\n".join(score_paths).encode(), vcf_path.encode(), cov.encode() if cov else None,
            capi.LOCUS[imp_locus], capi.MISSING[imp_missing], capi.SAMPLE[imp_sample],
            float(maxmis), float(afmisp), int(mincs), int(ignorefilt), device, int(shard), int(n_shards))
    if d_out is not None:   # the block's sums stay on the device for the all-reduce (nps_multi_partial_device)
        sums = None
        n = L.nh_compute_multi_dev(*args, 1, C.c_void_p(int(d_out)), nloci.ctypes.data, offsets.ctypes.data, log, len(log))
    else:
        sums = np.zeros((S, max_samples), dtype=np.float64)
        n = L.nh_compute_multi_partial(*args, sums.ctypes.data, max_samples, nloci.ctypes.data, offsets.ctypes.data,
                                       log, len(log))
    if n < 0:
        raise capi.NpsError(-3 if n == -2 else -1, L.nh_last_error().decode("utf-8", "replace"))
    return (None if sums is None else sums[:, :n].copy()), nloci.astype(np.int64), offsets, _split_logs(_full_log(L, log), S)


def format_scores(x: np.ndarray) -> List[str]:
    """format_score over an array, without a C call per value: "%.16g" with ".0" appended where the text has no
    '.', 'e', 'n' or 'i' (nimpress.nim:753; pinned by scores/*_nimpress_res.txt)"""
    out = np.char.mod("%.16g", np.asarray(x, dtype=np.float64))
    plain = ~(np.char.find(out, ".") >= 0) & ~(np.char.find(out, "e") >= 0) & ~(np.char.find(out, "n") >= 0) & \
        ~(np.char.find(out, "i") >= 0)
    return np.where(plain, np.char.add(out, ".0"), out).tolist()


def write_matrix_tsv(path: str, names: List[str], scores: np.ndarray) -> None:
    """scores [n_scores, n_samples] -> one line per sample, `name TAB score 1 TAB score 2 ...`, every value in the
    reference's float format (nimpress.nim:752-753); formatted by up to 16 threads of the host library
    (nh_write_matrix_tsv).  path "-" writes to stdout."""
    m = np.ascontiguousarray(scores, dtype=np.float64)
    if m.ndim != 2 or m.shape[1] != len(names):
        raise ValueError("scores must be [n_scores, %d]" % len(names))
    if any("\n" in s for s in names):
        raise ValueError("a sample name contains a newline")
    if path == "-":
        import sys
        sys.stdout.flush()
    rc = load().nh_write_matrix_tsv(path.encode(), "\n".join(names).encode(), len(names), m.ctypes.data, m.shape[0],
                                    m.shape[1])
    if rc != 0:
        raise capi.NpsError(-1, load().nh_last_error().decode("utf-8", "replace"))


def sample_names(vcf_path: str) -> List[str]:
    """samples(vcf) of the reference (nimpress.nim:753): the header's sample names, in order"""
    L = load()
    h = L.nh_vcf_open_header(vcf_path.encode())
    if not h:
        raise capi.NpsError(-1, "cannot open %s: %s" % (vcf_path, L.nh_last_error().decode("utf-8", "replace")))
    try:
        if L.nh_vcf_n_samples(h) == 0:
            return []
        need = L.nh_vcf_samples_joined(h, None, 0)
        buf = C.create_string_buffer(need + 1)
        L.nh_vcf_samples_joined(h, buf, need + 1)
        return buf.raw[:need].decode().split("\n")
    finally:
        L.nh_vcf_close(h)


def format_score(x: float) -> str:
    """Nim's `$float` as the reference prints it (nimpress.nim:753; "%.16g", ".0" for integers, "nan")"""
    buf = C.create_string_buffer(64)
    load().nh_format_float(float(x), buf, len(buf))
    return buf.value.decode()
