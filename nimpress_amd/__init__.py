"""nimpress_amd -- MI355X-native engine for the nimpress polygenic-score inner loop.

The product is ``libnps.so`` (hand-written HIP for gfx950 behind the C-ABI in ``include/nps.h``).
This package holds its sources (``csrc/``), the build helper and a thin ctypes binding
(``capi``).  Nothing here computes on the CPU and nothing here imports ``oracle/``.
"""
from . import capi  # noqa: F401
from .capi import (Cohort, NpsError, ScoreDef, Scorer, make_params, row_descs)  # noqa: F401

__all__ = ["capi", "Cohort", "NpsError", "ScoreDef", "Scorer", "make_params", "row_descs"]
