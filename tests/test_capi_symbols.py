"""CPU-side checks of the C-ABI: the library loads, exports every symbol include/nps.h declares,
and refuses to compute without a GPU (no CPU fallback).  No compute calls are made here."""
import ctypes
import os
import re

import pytest

from nimpress_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(name="nps.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nps_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), "libnps.so does not export %s" % name


def test_abi_version_matches_header():
    text = open(os.path.join(ROOT, "include", "nps.h")).read()
    ver = int(re.search(r"#define NPS_ABI_VERSION (\d+)", text).group(1))
    assert capi.load().nps_abi_version() == ver


def test_struct_layouts_match_header():
    assert ctypes.sizeof(capi.NpsParams) == 32
    assert capi.STAT_DTYPE.itemsize == 32
    assert capi.ROW_DESC_DTYPE.itemsize == 24
    assert ctypes.sizeof(capi.NpsProfile) == 96


def test_no_cpu_fallback_without_device():
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible; the refusal path is covered on CPU-only hosts")
    with pytest.raises(capi.NpsError) as ei:
        capi.Scorer(6, capi.make_params())
    assert ei.value.status == -2  # NPS_E_NODEVICE
    with pytest.raises(capi.NpsError):
        capi.Cohort(6, 4)


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: nothing under nimpress_amd/ may reference it
    pkg = os.path.join(ROOT, "nimpress_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                src = open(os.path.join(root, f), errors="replace").read()
                for bad in ("import oracle", "from oracle", "librefcpu", "refcpu.h"):
                    assert bad not in src, (f, bad)


def test_comm_library_exports_every_symbol_of_nps_comm_h():
    """libnps_rccl.so (the RCCL exchange for single-process hosts) loads without a GPU, exports what include/nps_comm.h
    declares, and libnps.so itself has no RCCL dependency"""
    import subprocess
    assert header_symbols("nps_comm.h") == sorted(capi.COMM_SYMBOLS)
    lib = capi.load_comm()
    for name in capi.COMM_SYMBOLS:
        assert hasattr(lib, name), name
    ldd = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in ldd
    ldd = subprocess.run(["ldd", capi.COMM_LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" in ldd and "libnps.so" in ldd
    if capi.device_count() == 0:
        with pytest.raises(capi.NpsError) as ei:
            capi.Comm(1)
        assert ei.value.status == capi.E_NODEVICE


def test_release_libraries_read_no_environment_variable():
    """the release build has no run-time switches: neither library imports getenv (diagnostics builds -- tools/mkexp.sh
    -DNPS_DIAGNOSTICS -- do, and are never what the package loads)"""
    import shutil
    import subprocess
    nm = shutil.which("nm")
    if not nm:
        pytest.skip("no nm")
    for path in (capi.LIB_PATH, capi.COMM_LIB_PATH):
        if not os.path.exists(path):
            pytest.skip("library not built")
        out = subprocess.run([nm, "-D", "--undefined-only", path], capture_output=True, text=True).stdout
        assert "getenv" not in out, path
