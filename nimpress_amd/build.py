"""Build helpers: compile the HIP sources of this package for gfx950 into in-tree shared objects.

hipcc cross-compiles without a GPU.  The .so files are git-ignored but travel with the working
tree (gpurun snapshot), so the GPU box loads exactly what was built here.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from typing import List

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIBNPS = os.path.join(PKG_DIR, "libnps.so")

HIPCC_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
               # float64 score arithmetic must not be contracted into FMAs the reference lacks
               "-ffp-contract=off", "-Wall", "-Wextra", "-Wno-unused-parameter"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libnps.so cannot be built (there is no CPU fallback)")
    return exe


def libnps_sources() -> List[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target: str, deps: List[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


OBJ_DIR = os.path.join(os.path.dirname(PKG_DIR), "build", "obj")


def build_libnps(force: bool = False, verbose: bool = False) -> str:
    """libnps.so from csrc/*.hip: one object per source (rebuilt when the source, a csrc header or include/nps.h is newer),
    compiled in parallel, then linked.  NPS_HIPCC_EXTRA: experiment builds (e.g. -DNPS_DIAGNOSTICS) -- never set by the
    driver or the tests; the objects of such a build live in their own directory."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = libnps_sources()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(PKG_DIR), "include", "nps.h"))
    extra = os.environ.get("NPS_HIPCC_EXTRA", "").split()
    import hashlib
    odir = OBJ_DIR + ("_" + hashlib.sha1(" ".join(extra).encode()).hexdigest()[:8] if extra else "")
    os.makedirs(odir, exist_ok=True)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    jobs = []
    objs = []
    for src in srcs:
        obj = os.path.join(odir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([_hipcc()] + flags + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIBNPS, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBNPS] + objs)
    return LIBNPS


HOST_DIR = os.path.join(CSRC, "host")
LIBHOST = os.path.join(PKG_DIR, "libnimpress_host.so")
CLI = os.path.join(PKG_DIR, "nimpress")
GXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-pthread", "-Wall", "-Wextra", "-ffp-contract=off"]


def build_host(force: bool = False, verbose: bool = False) -> List[str]:
    """The host side above the C-ABI (C++): libnimpress_host.so + the `nimpress` CLI.  Plain g++,
    links libnps.so (rpath $ORIGIN) and zlib."""
    build_libnps(force=False, verbose=verbose)
    gxx = shutil.which("g++") or "g++"
    srcs = [os.path.join(HOST_DIR, "nimpress_host.cpp")]
    deps = srcs + [os.path.join(HOST_DIR, "nimpress_host.hpp"), LIBNPS,
                   os.path.join(os.path.dirname(PKG_DIR), "include", "nps.h")]
    link = ["-L" + PKG_DIR, "-lnps", "-lz", "-Wl,-rpath,$ORIGIN"]
    if force or _stale(LIBHOST, deps):
        cmd = [gxx] + GXX_FLAGS + ["-shared", "-o", LIBHOST] + srcs + link
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    main_src = os.path.join(HOST_DIR, "main.cpp")
    if force or _stale(CLI, deps + [main_src]):
        cmd = [gxx] + GXX_FLAGS + ["-o", CLI, main_src] + srcs + link
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return [LIBHOST, CLI]


COMM_DIR = os.path.join(CSRC, "comm")
LIBCOMM = os.path.join(PKG_DIR, "libnps_rccl.so")


def build_comm(force: bool = False, verbose: bool = False) -> str:
    """libnps_rccl.so (include/nps_comm.h): the RCCL exchange step for single-process hosts.  Its own library, so that
    libnps.so keeps no RCCL dependency; links libnps.so (rpath $ORIGIN) and librccl."""
    build_libnps(force=False, verbose=verbose)
    src = os.path.join(COMM_DIR, "nps_comm.hip")
    inc = os.path.join(os.path.dirname(PKG_DIR), "include")
    deps = [src, os.path.join(inc, "nps_comm.h"), os.path.join(inc, "nps.h"), LIBNPS]
    if force or _stale(LIBCOMM, deps):
        cmd = [_hipcc()] + HIPCC_FLAGS + ["-I" + inc, "-o", LIBCOMM, src, "-L" + PKG_DIR, "-lnps", "-lrccl",
                                          "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIBCOMM


def build_all(force: bool = False, verbose: bool = False) -> List[str]:
    """Everything native in this package: libnps.so (HIP), libnimpress_host.so + nimpress (C++), libnps_rccl.so (HIP + RCCL)."""
    return ([build_libnps(force=force, verbose=verbose)] + build_host(force=force, verbose=verbose)
            + [build_comm(force=force, verbose=verbose)])


if __name__ == "__main__":
    print(build_all(force=True, verbose=True))
