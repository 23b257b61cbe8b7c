"""ctypes binding of the libnps C-ABI (include/nps.h).

This is plumbing for tests, bench.py and Python callers; the product is libnps.so.  There is no
fallback: if libnps.so is missing or no MI355X is visible, calls raise ``NpsError``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libnps.so")

# enums (nimpress.nim:412-414 declaration order)
LOCUS = {"ps": 0, "homref": 1, "fail": 2, "ignore": 3}
MISSING = {"homref": 0, "ignore": 1}
SAMPLE = {"ps": 0, "homref": 1, "fail": 2, "int_ps": 3, "int_fail": 4}
ROW_PRESENT, ROW_UNCOVERED, ROW_ABSENT, ROW_FILTERED = 0, 1, 2, 3
REASON_GENOTYPED, REASON_UNCOVERED, REASON_ABSENT, REASON_FILTERED, REASON_MAXMIS = range(5)
FMT_GT2, FMT_DS32, FMT_GT2M, FMT_GT2X, FMT_GT_AUTO, FMT_DS16 = 0, 1, 2, 3, 4, 5
ROW_NOT_IN_SCORE = 4
MULTI_MAX_SCORES = 8
MODE_AUTO, MODE_TWOPASS, MODE_FUSED = 0, 1, 2

NPS_OK = 0
E_INVAL, E_NODEVICE, E_HIP, E_NOMEM, E_STATE, E_UNSUPPORTED, E_TIMEOUT = -1, -2, -3, -4, -5, -6, -7
STATUS_NAMES = {0: "NPS_OK", -1: "NPS_E_INVAL", -2: "NPS_E_NODEVICE", -3: "NPS_E_HIP",
                -4: "NPS_E_NOMEM", -5: "NPS_E_STATE", -6: "NPS_E_UNSUPPORTED", -7: "NPS_E_TIMEOUT"}


class NpsParams(C.Structure):
    _fields_ = [("imp_locus", C.c_int32), ("imp_missing", C.c_int32), ("imp_sample", C.c_int32),
                ("reserved", C.c_int32), ("max_missing_rate", C.c_double), ("min_cs", C.c_int64)]


class NpsProfile(C.Structure):
    _fields_ = [("ms_decode", C.c_double), ("ms_tally", C.c_double), ("ms_params", C.c_double),
                ("ms_accumulate", C.c_double), ("ms_fused", C.c_double), ("ms_reduce", C.c_double),
                ("n_decode", C.c_uint64), ("n_tally", C.c_uint64), ("n_params", C.c_uint64),
                ("n_accumulate", C.c_uint64), ("n_fused", C.c_uint64), ("n_reduce", C.c_uint64)]


STAT_DTYPE = np.dtype([("ngenotyped", "<u8"), ("nmissing", "<u8"), ("neffect", "<f8"),
                       ("used", "<i4"), ("reason", "<i4")])
ROW_DESC_DTYPE = np.dtype([("beta", "<f8"), ("eaf", "<f8"), ("kind", "<i4"),
                           ("ref_is_effect", "<i4")])

# every symbol include/nps.h declares (tests/test_capi_symbols.py checks the two lists agree)
SYMBOLS = [
    "nps_abi_version", "nps_last_error", "nps_device_count", "nps_warmup", "nps_create", "nps_n_samples", "nps_device", "nps_push_gt",
    "nps_push_ds", "nps_push_packed", "nps_push_locus", "nps_flush", "nps_finish",
    "nps_push_gt_raw", "nps_push_bed", "nps_cohort_upload_bed", "nps_finish_device", "nps_partial_device", "nps_normalize_device", "nps_reset", "nps_scoredef_create", "nps_scoredef_n_present",
    "nps_scoredef_destroy", "nps_score_cohort_def",
    "nps_destroy", "nps_cohort_create", "nps_cohort_row_stride", "nps_cohort_n_rows", "nps_cohort_format",
    "nps_cohort_upload", "nps_cohort_download", "nps_cohort_synth", "nps_cohort_synth_rows",
    "nps_cohort_optimize",
    "nps_cohort_destroy",
    "nps_score_cohort", "nps_profile_enable", "nps_profile_get", "nps_stream", "nps_fused_geometry",
    "nps_multidef_create", "nps_multidef_create_bits", "nps_multidef_destroy", "nps_multi_create", "nps_score_cohort_multi",
    "nps_multi_finish", "nps_multi_finish_device", "nps_multi_reset", "nps_multi_destroy", "nps_multi_timing",
    "nps_cohort_convert", "nps_cohort_row_tallies", "nps_cohort_keep_tallies", "nps_cohort_has_tallies", "nps_multi_set_missing_weight_bits",
    "nps_cohort_push_gt_raw", "nps_cohort_push_bed", "nps_multi_partial_device", "nps_multi_partial",
    "nps_multi_n_scores", "nps_multi_n_samples", "nps_multi_device", "nps_cohort_expect_passes",
]


class NpsError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, str(status)), msg))
        self.status = status


_lib = None


def _hip_runtimes_mapped():
    """paths of the libamdhip64 copies mapped into this process"""
    seen = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.split(None, 5)[-1].strip() if line.count("/") else ""
                if "libamdhip64" in path and path not in seen:
                    seen.append(path)
    except OSError:
        pass
    return seen


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  libnps.so needs `libamdhip64.so.7`; the PyTorch wheel ships its own
    copy of that library (same SONAME) next to libtorch_hip.so.  If libnps.so were loaded first it would
    pull in /opt/rocm's copy, a later `import torch` would map the wheel's copy as well, and device
    pointers handed from torch to libnps (nps_finish_device into a tensor, the RCCL path) would cross two
    runtimes.  So when torch is installed it is imported FIRST, here: the loader then resolves
    libnps.so's NEEDED entry to the copy torch has mapped, whatever order the caller imports things in.
    (Loading the wheel's libamdhip64.so alone by path works too, but changes the order in which the
    libraries are torn down at exit, which rocprofv3's tool library does not survive.)  Without torch
    (the C++ command line, C callers) /opt/rocm's copy is the only one."""
    import sys
    if "torch" in sys.modules or _hip_runtimes_mapped():
        return
    try:
        import importlib.util
        if importlib.util.find_spec("torch") is None:
            return
        import torch  # noqa: F401
    except (ImportError, ValueError):
        return


def load(with_torch: bool = True):
    """dlopen libnps.so (raises if it has not been built -- there is no fallback).  with_torch=False: the process
    will never import torch (a one-GPU tool run), so the half second its import takes is not spent on keeping the
    two libraries on one HIP runtime; loading torch afterwards is then an error of the caller's."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NpsError(-2, "libnps.so not built (%s); run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` -- the HIP library is the only compute path" % LIB_PATH)
    if with_torch:
        _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    rts = _hip_runtimes_mapped()
    if len(rts) > 1:
        raise NpsError(-5, "two HIP runtimes are mapped into this process (%s): device pointers cannot "
                           "be shared between them; import nimpress_amd.capi (or torch) before anything "
                           "else loads a libamdhip64" % ", ".join(rts))
    vp, u64, i32, dbl = C.c_void_p, C.c_uint64, C.c_int, C.c_double
    L.nps_abi_version.restype = C.c_int
    L.nps_last_error.restype = C.c_char_p
    L.nps_device_count.restype = C.c_int
    L.nps_create.argtypes = [C.POINTER(vp), i32, u64, C.POINTER(NpsParams)]
    L.nps_n_samples.argtypes = [vp]
    L.nps_n_samples.restype = u64
    L.nps_device.argtypes = [vp]
    L.nps_push_gt.argtypes = [vp, vp, i32, i32, i32, dbl, dbl]
    L.nps_push_gt_raw.argtypes = [vp, vp, i32, i32, i32, i32, dbl, dbl]
    L.nps_push_bed.argtypes = [vp, vp, i32, i32, dbl, dbl]
    L.nps_cohort_upload_bed.argtypes = [vp, u64, u64, vp, C.c_size_t, vp]
    L.nps_push_ds.argtypes = [vp, vp, i32, dbl, dbl]
    L.nps_push_packed.argtypes = [vp, vp, i32, dbl, dbl]
    L.nps_push_locus.argtypes = [vp, i32, i32, dbl, dbl]
    L.nps_flush.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.nps_finish.argtypes = [vp, dbl, vp, C.POINTER(u64)]
    L.nps_finish_device.argtypes = [vp, dbl, vp, C.POINTER(u64)]
    L.nps_partial_device.argtypes = [vp, vp, C.POINTER(u64)]
    L.nps_normalize_device.argtypes = [vp, vp, u64, dbl]
    L.nps_scoredef_create.argtypes = [C.POINTER(vp), i32, vp, u64]
    L.nps_scoredef_n_present.argtypes = [vp]
    L.nps_scoredef_n_present.restype = u64
    L.nps_scoredef_destroy.argtypes = [vp]
    L.nps_scoredef_destroy.restype = None
    L.nps_score_cohort_def.argtypes = [vp, vp, u64, vp, i32]
    L.nps_reset.argtypes = [vp, C.POINTER(NpsParams)]
    L.nps_destroy.argtypes = [vp]
    L.nps_destroy.restype = None
    L.nps_cohort_create.argtypes = [C.POINTER(vp), i32, u64, u64, i32]
    L.nps_cohort_row_stride.argtypes = [vp]
    L.nps_cohort_row_stride.restype = u64
    L.nps_cohort_format.argtypes = [vp]
    L.nps_cohort_format.restype = C.c_int
    L.nps_cohort_n_rows.argtypes = [vp]
    L.nps_cohort_n_rows.restype = u64
    L.nps_cohort_upload.argtypes = [vp, u64, u64, vp, C.c_size_t]
    L.nps_cohort_download.argtypes = [vp, u64, u64, vp, C.c_size_t]
    L.nps_cohort_synth.argtypes = [vp, u64, u64, u64, vp, vp, vp]
    L.nps_cohort_synth_rows.argtypes = [vp, u64, u64, u64, u64, vp, vp, vp]
    L.nps_cohort_optimize.argtypes = [vp]
    L.nps_cohort_destroy.argtypes = [vp]
    L.nps_cohort_destroy.restype = None
    L.nps_score_cohort.argtypes = [vp, vp, u64, vp, u64, i32]
    L.nps_profile_enable.argtypes = [vp, i32]
    L.nps_profile_get.argtypes = [vp, C.POINTER(NpsProfile), i32]
    L.nps_stream.argtypes = [vp]
    L.nps_stream.restype = vp
    u32p = C.POINTER(C.c_uint32)
    L.nps_fused_geometry.argtypes = [vp, i32, u64, u32p, u32p, u32p]
    L.nps_multidef_create.argtypes = [C.POINTER(vp), i32, vp, i32, u64]
    L.nps_multidef_create_bits.argtypes = [C.POINTER(vp), i32, vp, i32, u64, i32]
    L.nps_multidef_destroy.argtypes = [vp]
    L.nps_multidef_destroy.restype = None
    L.nps_multi_create.argtypes = [C.POINTER(vp), i32, u64, C.POINTER(NpsParams), i32]
    L.nps_score_cohort_multi.argtypes = [vp, vp, u64, vp]
    L.nps_multi_finish.argtypes = [vp, vp, vp, vp]
    L.nps_multi_finish_device.argtypes = [vp, vp, vp, vp]
    L.nps_multi_partial_device.argtypes = [vp, vp, vp]
    L.nps_multi_n_scores.argtypes = [vp]
    L.nps_multi_n_samples.argtypes = [vp]
    L.nps_multi_n_samples.restype = u64
    L.nps_multi_device.argtypes = [vp]
    L.nps_multi_reset.argtypes = [vp, C.POINTER(NpsParams)]
    L.nps_multi_destroy.argtypes = [vp]
    L.nps_multi_destroy.restype = None
    dp = C.POINTER(C.c_double)
    L.nps_multi_timing.argtypes = [vp, dp, dp, dp]
    L.nps_multi_set_missing_weight_bits.argtypes = [vp, i32]
    L.nps_cohort_convert.argtypes = [vp, vp]
    L.nps_cohort_row_tallies.argtypes = [vp, u64, u64, vp, vp]
    L.nps_cohort_keep_tallies.argtypes = [vp]
    L.nps_cohort_expect_passes.argtypes = [vp, C.c_uint32]
    L.nps_cohort_has_tallies.argtypes = [vp]
    _lib = L
    return L


def _check(rc: int):
    if rc != NPS_OK:
        raise NpsError(rc, load().nps_last_error().decode("utf-8", "replace"))


def make_params(imp_locus="ps", imp_missing="homref", imp_sample="int_ps", maxmis=0.05,
                mincs=100) -> NpsParams:
    """Defaults are the CLI defaults, nimpress.nim:670-684."""
    return NpsParams(LOCUS[imp_locus], MISSING[imp_missing], SAMPLE[imp_sample], 0, float(maxmis),
                     int(mincs))


def device_count() -> int:
    return load().nps_device_count()


class Cohort:
    """A genotype matrix resident in HBM, variant-major / sample-minor: 2-bit GT codes (FMT_GT2) or
    float32 dosages with NaN = missing (FMT_DS32; FMT_DS16 holds them as 16-bit decimals and looks the same from here)."""

    def __init__(self, n_samples: int, n_rows: int, device: int = 0, fmt: int = FMT_GT2):
        self._h = C.c_void_p()
        self.n_samples, self.n_rows, self.device, self.fmt = int(n_samples), int(n_rows), device, fmt
        _check(load().nps_cohort_create(C.byref(self._h), device, self.n_samples, self.n_rows, fmt))
        self.fmt = int(load().nps_cohort_format(self._h))   # (FMT_GT_AUTO has become FMT_GT2X or FMT_GT2)

    @property
    def row_stride(self) -> int:
        return int(load().nps_cohort_row_stride(self._h))

    def upload(self, row0: int, rows: np.ndarray):
        rows = np.ascontiguousarray(rows)
        assert rows.ndim == 2
        if self.fmt in (FMT_DS32, FMT_DS16):
            assert rows.dtype == np.float32
        # (a leading axis of length 1 made by rows[None, :] has stride 0: the row's own width is what is meant)
        stride = rows.strides[0] if rows.shape[0] > 1 else rows.shape[1] * rows.itemsize
        _check(load().nps_cohort_upload(self._h, row0, rows.shape[0], rows.ctypes.data, stride))

    def upload_bed(self, row0: int, bed_rows: np.ndarray, effect_is_a1):
        """rows of a PLINK .bed file ([k, ceil(n/4)] uint8, e.g. a view of the mmap'ed file)"""
        assert bed_rows.ndim == 2 and bed_rows.dtype == np.uint8 and bed_rows.strides[1] == 1
        flags = np.ascontiguousarray(effect_is_a1, dtype=np.uint8)
        assert flags.size == bed_rows.shape[0]
        _check(load().nps_cohort_upload_bed(self._h, row0, bed_rows.shape[0], bed_rows.ctypes.data,
                                            bed_rows.strides[0], flags.ctypes.data))

    def download(self, row0: int, nrows: int) -> np.ndarray:
        if self.fmt in (FMT_DS32, FMT_DS16):
            width, dtype = self.n_samples, np.float32
        else:
            width, dtype = (self.n_samples + 15) // 16, np.uint32
        out = np.zeros((nrows, max(width, 1)), dtype=dtype)
        _check(load().nps_cohort_download(self._h, row0, nrows, out.ctypes.data, out.strides[0]))
        return out[:, :width]

    def synth(self, row0: int, seed: int, t_het, t_hom, t_miss):
        th, tm, tmi = (np.ascontiguousarray(a, dtype=np.uint32) for a in (t_het, t_hom, t_miss))
        assert th.size == tm.size == tmi.size
        _check(load().nps_cohort_synth(self._h, row0, th.size, seed, th.ctypes.data, tm.ctypes.data,
                                       tmi.ctypes.data))

    def synth_at(self, row0: int, gen_row0: int, seed: int, t_het, t_hom, t_miss):
        """cohort rows row0.. receive the generator's rows gen_row0.. (a block / chunk of a larger matrix)"""
        th, tm, tmi = (np.ascontiguousarray(a, dtype=np.uint32) for a in (t_het, t_hom, t_miss))
        assert th.size == tm.size == tmi.size
        _check(load().nps_cohort_synth_rows(self._h, row0, th.size, gen_row0, seed, th.ctypes.data,
                                            tm.ctypes.data, tmi.ctypes.data))

    def convert_from(self, src: "Cohort"):
        """fill this FMT_GT2M cohort (and its row tallies) from a FMT_GT2 cohort of the same shape"""
        _check(load().nps_cohort_convert(self._h, src._h))

    def row_tallies(self, row0: int = 0, nrows: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
        """(nmissing, neffect) per row of a FMT_GT2M cohort"""
        nrows = self.n_rows - row0 if nrows is None else nrows
        nm, ne = np.zeros(max(nrows, 1), dtype=np.uint64), np.zeros(max(nrows, 1), dtype=np.uint64)
        _check(load().nps_cohort_row_tallies(self._h, row0, nrows, nm.ctypes.data, ne.ctypes.data))
        return nm[:nrows], ne[:nrows]

    def keep_tallies(self):
        """count every row's tallyAlleles once and keep them with this FMT_GT2X cohort: MODE_AUTO then scores it with the
        tallies given (nps_cohort_keep_tallies)"""
        _check(load().nps_cohort_keep_tallies(self._h))

    def expect_passes(self, n: int):
        """hint: this cohort will be scored n times (nps_cohort_expect_passes): with n >= 2 the first whole-cohort MODE_AUTO
        run keeps the tallies it counts anyway, later runs score with them given"""
        _check(load().nps_cohort_expect_passes(self._h, int(n)))

    def has_tallies(self) -> bool:
        return bool(load().nps_cohort_has_tallies(self._h))

    def optimize(self):
        """one-time layout change (nps_cohort_optimize): parity layout of the high-bit planes, fewer LDS bank
        conflicts in the accumulation kernels; results unchanged, its own inverse"""
        _check(load().nps_cohort_optimize(self._h))

    def close(self):
        if self._h:
            load().nps_cohort_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ScoreDef:
    """The rows of one score file (+ what the host found for each), resident on the device."""

    def __init__(self, rows: np.ndarray, device: int = 0):
        rows = np.ascontiguousarray(rows, dtype=ROW_DESC_DTYPE)
        self._h = C.c_void_p()
        self.n_desc = rows.size
        _check(load().nps_scoredef_create(C.byref(self._h), device, rows.ctypes.data, rows.size))

    @property
    def n_present(self) -> int:
        return int(load().nps_scoredef_n_present(self._h))

    def close(self):
        if self._h:
            load().nps_scoredef_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scorer:
    """computePolygenicScores (nimpress.nim:592-649) with the row loop on the GPU."""

    def __init__(self, n_samples: int, params: NpsParams, device: int = 0):
        self._h = C.c_void_p()
        self.n = int(n_samples)
        self.device = device
        _check(load().nps_create(C.byref(self._h), device, self.n, C.byref(params)))

    # -- one call per score row, in score-file order
    def push_gt(self, gts: np.ndarray, ploidy: int, eaidx: int, ref_is_effect, beta: float, eaf: float):
        gts = np.ascontiguousarray(gts, dtype=np.int32)
        if gts.size != self.n * ploidy:
            raise ValueError("gts has %d values, expected %d" % (gts.size, self.n * ploidy))
        _check(load().nps_push_gt(self._h, gts.ctypes.data, ploidy, eaidx, int(bool(ref_is_effect)),
                                  float(beta), float(eaf)))

    def push_gt_raw(self, gt: np.ndarray, ploidy: int, eaidx: int, ref_is_effect, beta: float,
                    eaf: float):
        """typed GT vector as a BCF record stores it: int8 / int16 / int32 array of n*ploidy values"""
        gt = np.ascontiguousarray(gt)
        if gt.dtype not in (np.int8, np.int16, np.int32):
            raise ValueError("gt dtype must be int8, int16 or int32")
        if gt.size != self.n * ploidy:
            raise ValueError("gt has %d values, expected %d" % (gt.size, self.n * ploidy))
        _check(load().nps_push_gt_raw(self._h, gt.ctypes.data, gt.dtype.itemsize, ploidy, eaidx,
                                      int(bool(ref_is_effect)), float(beta), float(eaf)))

    def push_bed(self, bed_row: np.ndarray, effect_is_a1, ref_is_effect, beta: float, eaf: float):
        """one variant of a PLINK .bed file: ceil(n/4) bytes"""
        bed_row = np.ascontiguousarray(bed_row, dtype=np.uint8)
        if bed_row.size != (self.n + 3) // 4:
            raise ValueError("bed row has %d bytes, expected %d" % (bed_row.size, (self.n + 3) // 4))
        # effect_is_a1 is the row's code map (NPS_MAP_*: 0 / 1 .bed effect A2 / A1, 2 / 3 .pgen effect ALT / REF)
        _check(load().nps_push_bed(self._h, bed_row.ctypes.data, int(effect_is_a1),
                                   int(bool(ref_is_effect)), float(beta), float(eaf)))

    def push_ds(self, ds: np.ndarray, ref_is_effect, beta: float, eaf: float):
        ds = np.ascontiguousarray(ds, dtype=np.float32)
        if ds.size != self.n:
            raise ValueError("ds has %d values, expected %d" % (ds.size, self.n))
        _check(load().nps_push_ds(self._h, ds.ctypes.data, int(bool(ref_is_effect)), float(beta),
                                  float(eaf)))

    def push_packed(self, row: np.ndarray, ref_is_effect, beta: float, eaf: float):
        row = np.ascontiguousarray(row, dtype=np.uint32)
        if row.size < (self.n + 15) // 16:
            raise ValueError("packed row too short")
        _check(load().nps_push_packed(self._h, row.ctypes.data, int(bool(ref_is_effect)),
                                      float(beta), float(eaf)))

    def push_locus(self, kind: int, ref_is_effect, beta: float, eaf: float):
        _check(load().nps_push_locus(self._h, kind, int(bool(ref_is_effect)), float(beta), float(eaf)))

    def score_cohort(self, cohort: Cohort, rows: np.ndarray, cohort_row0: int = 0,
                     mode: int = MODE_AUTO):
        rows = np.ascontiguousarray(rows, dtype=ROW_DESC_DTYPE)
        _check(load().nps_score_cohort(self._h, cohort._h, cohort_row0, rows.ctypes.data, rows.size,
                                       mode))

    def score_cohort_def(self, cohort: Cohort, sdef: ScoreDef, cohort_row0: int = 0,
                         mode: int = MODE_AUTO):
        _check(load().nps_score_cohort_def(self._h, cohort._h, cohort_row0, sdef._h, mode))

    def finish_device(self, offset: float, d_scores_ptr: int) -> int:
        """scores -> caller-allocated device buffer (n_samples float64); returns nloci."""
        nloci = C.c_uint64(0)
        _check(load().nps_finish_device(self._h, float(offset), C.c_void_p(d_scores_ptr),
                                        C.byref(nloci)))
        return int(nloci.value)

    def partial_device(self, d_sums_ptr: int) -> int:
        """un-normalised sums of this context's rows -> device buffer; returns its nloci."""
        nloci = C.c_uint64(0)
        _check(load().nps_partial_device(self._h, C.c_void_p(d_sums_ptr), C.byref(nloci)))
        return int(nloci.value)

    def normalize_device(self, d_sums_ptr: int, nloci: int, offset: float):
        """in place: d[i] = d[i] / (2 nloci) + offset (nimpress.nim:643-649)."""
        _check(load().nps_normalize_device(self._h, C.c_void_p(d_sums_ptr), int(nloci), float(offset)))

    def flush(self, max_rows: Optional[int] = None) -> np.ndarray:
        """Completes pushed rows; returns their stats (structured array, push order)."""
        out = []
        while True:
            cap = 65536 if max_rows is None else max_rows
            buf = np.zeros(cap, dtype=STAT_DTYPE)
            n = C.c_size_t(0)
            _check(load().nps_flush(self._h, buf.ctypes.data, cap, C.byref(n)))
            out.append(buf[: n.value])
            if n.value < cap or max_rows is not None:
                break
        return np.concatenate(out) if out else np.zeros(0, dtype=STAT_DTYPE)

    def sync(self):
        n = C.c_size_t(0)
        _check(load().nps_flush(self._h, None, 0, C.byref(n)))

    def finish(self, offset: float) -> Tuple[np.ndarray, int]:
        scores = np.empty(max(self.n, 1), dtype=np.float64)
        nloci = C.c_uint64(0)
        _check(load().nps_finish(self._h, float(offset), scores.ctypes.data, C.byref(nloci)))
        return scores[: self.n], int(nloci.value)

    def reset(self, params: Optional[NpsParams] = None):
        _check(load().nps_reset(self._h, C.byref(params) if params is not None else None))

    def profile_enable(self, on: bool = True):
        _check(load().nps_profile_enable(self._h, int(on)))

    def profile_get(self, reset: bool = False) -> NpsProfile:
        p = NpsProfile()
        _check(load().nps_profile_get(self._h, C.byref(p), int(reset)))
        return p

    @property
    def stream(self) -> int:
        return int(load().nps_stream(self._h) or 0)

    def fused_geometry(self, n_rows: int, fmt: int = FMT_GT2) -> Tuple[int, int, int]:
        """(slices, teams, samples per slice) of the persistent grid for n_rows rows; zeros = two-pass"""
        a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        _check(load().nps_fused_geometry(self._h, fmt, int(n_rows), C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def close(self):
        if self._h:
            load().nps_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiDef:
    """S score definitions over the same cohort rows ([S, n_desc] row descriptors), resident on the device"""

    def __init__(self, rows: np.ndarray, device: int = 0, weight_bits: int = 0):
        """weight_bits: 49 (default: seven digits per weight) or 41 (six) -- include/nps.h"""
        rows = np.ascontiguousarray(rows, dtype=ROW_DESC_DTYPE)
        assert rows.ndim == 2
        self._h = C.c_void_p()
        self.n_scores, self.n_desc = rows.shape
        _check(load().nps_multidef_create_bits(C.byref(self._h), device, rows.ctypes.data, self.n_scores, self.n_desc,
                                               weight_bits))

    def close(self):
        if self._h:
            load().nps_multidef_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiScorer:
    """computePolygenicScores for S score definitions in one pass over a FMT_GT2M cohort (matrix cores)"""

    def __init__(self, n_samples: int, params: NpsParams, n_scores: int, device: int = 0):
        self._h = C.c_void_p()
        self.n, self.n_scores = int(n_samples), int(n_scores)
        _check(load().nps_multi_create(C.byref(self._h), device, self.n, C.byref(params), self.n_scores))

    def set_missing_weight_bits(self, bits: int):
        """56 (default): full-width weights for the imputed value of a missing genotype; 32: rounded to 2^-32 of the
        score's largest weight, a quarter fewer matrix instructions with more than 4 scores (include/nps.h)"""
        _check(load().nps_multi_set_missing_weight_bits(self._h, int(bits)))

    def score_cohort(self, cohort: Cohort, mdef: MultiDef, cohort_row0: int = 0):
        _check(load().nps_score_cohort_multi(self._h, cohort._h, cohort_row0, mdef._h))

    def finish(self, offsets) -> Tuple[np.ndarray, np.ndarray]:
        off = np.ascontiguousarray(offsets, dtype=np.float64)
        assert off.size == self.n_scores
        scores = np.empty((self.n_scores, max(self.n, 1)), dtype=np.float64)
        nloci = np.zeros(self.n_scores, dtype=np.uint64)
        _check(load().nps_multi_finish(self._h, off.ctypes.data, scores.ctypes.data, nloci.ctypes.data))
        return scores[:, : self.n], nloci

    def partial_device(self, d_sums_ptr: int) -> np.ndarray:
        """un-normalised sums [n_scores, n_samples] of this context's rows -> device buffer; returns nloci per score"""
        nloci = np.zeros(self.n_scores, dtype=np.uint64)
        _check(load().nps_multi_partial_device(self._h, C.c_void_p(d_sums_ptr), nloci.ctypes.data))
        return nloci

    def finish_device(self, offsets, d_scores_ptr: int) -> np.ndarray:
        off = np.ascontiguousarray(offsets, dtype=np.float64)
        nloci = np.zeros(self.n_scores, dtype=np.uint64)
        _check(load().nps_multi_finish_device(self._h, off.ctypes.data, C.c_void_p(d_scores_ptr),
                                              nloci.ctypes.data))
        return nloci

    def reset(self, params: Optional[NpsParams] = None):
        _check(load().nps_multi_reset(self._h, C.byref(params) if params is not None else None))

    def timing(self) -> Tuple[float, float, float]:
        a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
        _check(load().nps_multi_timing(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def close(self):
        if self._h:
            load().nps_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def row_descs(beta, eaf, kind=None, ref_is_effect=None) -> np.ndarray:
    beta = np.asarray(beta, dtype=np.float64)
    out = np.zeros(beta.size, dtype=ROW_DESC_DTYPE)
    out["beta"] = beta
    out["eaf"] = np.asarray(eaf, dtype=np.float64)
    if kind is not None:
        out["kind"] = np.asarray(kind, dtype=np.int32)
    if ref_is_effect is not None:
        out["ref_is_effect"] = np.asarray(ref_is_effect, dtype=np.int32)
    return out


# ---- libnps_rccl.so (include/nps_comm.h): the exchange step for single-process hosts ---------------------------------
COMM_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnps_rccl.so")
COMM_SYMBOLS = ["nps_comm_last_error", "nps_comm_init_all", "nps_comm_size", "nps_comm_device", "nps_comm_destroy",
                "nps_comm_allgather_scores", "nps_comm_allreduce_partial", "nps_comm_allreduce_partial_multi"]
_comm_lib = None


def load_comm():
    """libnps_rccl.so, loaded on first use (it brings in RCCL; libnps.so does not)"""
    global _comm_lib
    if _comm_lib is None:
        load()
        if not os.path.exists(COMM_LIB_PATH):
            raise RuntimeError("libnps_rccl.so is not built (python -m nimpress_amd.build)")
        L = C.CDLL(COMM_LIB_PATH, mode=C.RTLD_GLOBAL)
        vp, u64 = C.c_void_p, C.c_uint64
        L.nps_comm_last_error.restype = C.c_char_p
        L.nps_comm_init_all.argtypes = [C.POINTER(vp), C.c_int, vp]
        L.nps_comm_size.argtypes = [vp]
        L.nps_comm_device.argtypes = [vp, C.c_int]
        L.nps_comm_destroy.argtypes = [vp]
        L.nps_comm_destroy.restype = None
        L.nps_comm_allgather_scores.argtypes = [vp, vp, vp, vp, vp]
        L.nps_comm_allreduce_partial.argtypes = [vp, vp, C.c_double, vp, vp]
        L.nps_comm_allreduce_partial_multi.argtypes = [vp, vp, C.c_int, u64, vp, vp, vp]
        _comm_lib = L
    return _comm_lib


class Comm:
    """One process, several GPUs of one node: ncclCommInitAll + the two collectives of the sharded layouts."""

    def __init__(self, n_devices: int, devices=None):
        L = load_comm()
        h = C.c_void_p()
        dv = None if devices is None else (C.c_int * n_devices)(*devices)
        self._check(L.nps_comm_init_all(C.byref(h), n_devices, dv))
        self._h = h
        self.size = L.nps_comm_size(h)

    @staticmethod
    def _check(rc):
        if rc != 0:
            raise NpsError(rc, (load_comm().nps_comm_last_error() or b"").decode())

    def device(self, rank: int) -> int:
        return load_comm().nps_comm_device(self._h, rank)

    def allgather_scores(self, scorers, offsets, d_matrix_ptrs) -> np.ndarray:
        n = self.size
        ctxs = (C.c_void_p * n)(*[s._h for s in scorers])
        offs = (C.c_double * n)(*[float(o) for o in offsets])
        ptrs = (C.c_void_p * n)(*d_matrix_ptrs)
        nl = (C.c_uint64 * n)()
        self._check(load_comm().nps_comm_allgather_scores(self._h, ctxs, offs, ptrs, nl))
        return np.array(list(nl), dtype=np.uint64)

    def allreduce_partial(self, scorers, offset: float, d_scores_ptrs) -> int:
        n = self.size
        ctxs = (C.c_void_p * n)(*[s._h for s in scorers])
        ptrs = (C.c_void_p * n)(*d_scores_ptrs)
        nl = C.c_uint64(0)
        self._check(load_comm().nps_comm_allreduce_partial(self._h, ctxs, float(offset), ptrs, C.byref(nl)))
        return int(nl.value)

    def allreduce_partial_multi(self, multi_scorers, n_scores: int, n_samples: int, offsets, d_matrix_ptrs) -> np.ndarray:
        n = self.size
        ms = (C.c_void_p * n)(*[m._h for m in multi_scorers])
        offs = (C.c_double * n_scores)(*[float(o) for o in offsets])
        ptrs = (C.c_void_p * n)(*d_matrix_ptrs)
        nl = (C.c_uint64 * n_scores)()
        self._check(load_comm().nps_comm_allreduce_partial_multi(self._h, ms, n_scores, n_samples, offs, ptrs, nl))
        return np.array(list(nl), dtype=np.uint64)

    def close(self):
        if self._h:
            load_comm().nps_comm_destroy(self._h)
            self._h = None
