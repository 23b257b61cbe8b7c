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
        L.nh_vcf_open.restype = C.c_void_p
        L.nh_vcf_open.argtypes = [C.c_char_p, C.c_char_p]
        L.nh_vcf_close.argtypes = [C.c_void_p]
        L.nh_vcf_n_samples.restype = C.c_long
        L.nh_vcf_n_samples.argtypes = [C.c_void_p]
        L.nh_vcf_open_header.restype = C.c_void_p
        L.nh_vcf_open_header.argtypes = [C.c_char_p]
        L.nh_vcf_sample.restype = C.c_char_p
        L.nh_vcf_sample.argtypes = [C.c_void_p, C.c_long]
        L.nh_format_float.restype = None
        L.nh_format_float.argtypes = [C.c_double, C.c_char_p, C.c_long]
        L.nh_write_matrix_tsv.restype = C.c_long
        L.nh_write_matrix_tsv.argtypes = [C.c_char_p, C.c_char_p, C.c_long, C.c_void_p, C.c_long, C.c_long]
        _lib = L
    return _lib


def compute_polygenic_scores(score_path: str, vcf_path: str, cov: Optional[str] = None,
                             imp_locus: str = "ps", imp_missing: str = "homref",
                             imp_sample: str = "int_ps", maxmis: float = 0.05, mincs: int = 100,
                             afmisp: float = 0.001, ignorefilt: bool = False, device: int = 0,
                             max_samples: int = 1 << 22) -> Tuple[np.ndarray, int, List[str]]:
    """nimpress's main() minus the printing (CLI defaults).  Returns (scores, nloci, log lines)."""
    L = load()
    scores = np.empty(max_samples, dtype=np.float64)
    nloci = C.c_ulonglong(0)
    log = C.create_string_buffer(1 << 20)
    n = L.nh_compute(score_path.encode(), vcf_path.encode(), cov.encode() if cov else None,
                     capi.LOCUS[imp_locus], capi.MISSING[imp_missing], capi.SAMPLE[imp_sample],
                     float(maxmis), float(afmisp), int(mincs), int(ignorefilt), device,
                     scores.ctypes.data, max_samples, C.byref(nloci), log, len(log))
    if n < 0:
        raise capi.NpsError(-3 if n == -2 else -1, L.nh_last_error().decode("utf-8", "replace"))
    return scores[:n].copy(), int(nloci.value), [l for l in log.value.decode().split("\n") if l]


def compute_polygenic_scores_multi(score_paths, vcf_path: str, cov: Optional[str] = None, imp_locus: str = "ps",
                                   imp_missing: str = "homref", imp_sample: str = "int_ps", maxmis: float = 0.05,
                                   mincs: int = 100, afmisp: float = 0.001, ignorefilt: bool = False, device: int = 0,
                                   max_samples: int = 1 << 22):
    """Several score files on one genotype file in ONE pass over the genotypes (computePolygenicScoresMulti: the union
    of the files' loci decoded once into a resident cohort, all definitions applied together on the matrix cores).
    Returns (scores [files, samples], nloci [files], log lines per file)."""
    L = load()
    S = len(score_paths)
    scores = np.empty((S, max_samples), dtype=np.float64)
    nloci = np.zeros(S, dtype=np.uint64)
    log = C.create_string_buffer(4 << 20)
    n = L.nh_compute_multi("\n".join(score_paths).encode(), vcf_path.encode(), cov.encode() if cov else None,
                           capi.LOCUS[imp_locus], capi.MISSING[imp_missing], capi.SAMPLE[imp_sample], float(maxmis),
                           float(afmisp), int(mincs), int(ignorefilt), device, scores.ctypes.data, max_samples,
                           nloci.ctypes.data, log, len(log))
    if n < 0:
        raise capi.NpsError(-3 if n == -2 else -1, L.nh_last_error().decode("utf-8", "replace"))
    logs = [[] for _ in range(S)]
    for l in log.value.decode().split("\n"):
        if l:
            k, _, text = l.partition("\t")
            logs[int(k)].append(text)
    return scores[:, :n].copy(), nloci.astype(np.int64), logs


def compute_polygenic_scores_multi_partial(score_paths, vcf_path: str, shard: int, n_shards: int, cov: Optional[str] = None,
                                           imp_locus: str = "ps", imp_missing: str = "homref",
                                           imp_sample: str = "int_ps", maxmis: float = 0.05, mincs: int = 100,
                                           afmisp: float = 0.001, ignorefilt: bool = False, device: int = 0,
                                           max_samples: int = 1 << 22):
    """Rows sharded over several GPUs x ALL score files on each: block `shard` of `n_shards` of
    the union of the files' loci is located, decoded and scored on this GPU -- 1 / n_shards of the ingest and of the
    cohort.  Returns (sums [files, samples] BEFORE the normalisation, nloci [files] of the block, offsets [files],
    log lines per file for the block's rows); the caller sum-all-reduces sums and nloci over the shards
    (multi.all_reduce_partial_matrix) and applies sums / (2 nloci) + offset (multi.normalize_matrix)."""
    L = load()
    S = len(score_paths)
    sums = np.zeros((S, max_samples), dtype=np.float64)
    nloci = np.zeros(S, dtype=np.uint64)
    offsets = np.zeros(S, dtype=np.float64)
    log = C.create_string_buffer(4 << 20)
    n = L.nh_compute_multi_partial("\n".join(score_paths).encode(), vcf_path.encode(), cov.encode() if cov else None,
                                   capi.LOCUS[imp_locus], capi.MISSING[imp_missing], capi.SAMPLE[imp_sample],
                                   float(maxmis), float(afmisp), int(mincs), int(ignorefilt), device, int(shard),
                                   int(n_shards), sums.ctypes.data, max_samples, nloci.ctypes.data, offsets.ctypes.data,
                                   log, len(log))
    if n < 0:
        raise capi.NpsError(-3 if n == -2 else -1, L.nh_last_error().decode("utf-8", "replace"))
    logs = [[] for _ in range(S)]
    for l in log.value.decode().split("\n"):
        if l:
            k, _, text = l.partition("\t")
            logs[int(k)].append(text)
    return sums[:, :n].copy(), nloci.astype(np.int64), offsets, logs


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
        return [L.nh_vcf_sample(h, i).decode() for i in range(L.nh_vcf_n_samples(h))]
    finally:
        L.nh_vcf_close(h)


def format_score(x: float) -> str:
    """Nim's `$float` as the reference prints it (nimpress.nim:753; "%.16g", ".0" for integers, "nan")"""
    buf = C.create_string_buffer(64)
    load().nh_format_float(float(x), buf, len(buf))
    return buf.value.decode()
