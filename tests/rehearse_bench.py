"""One RANK of a CPU rehearsal of bench.py's N > 1 code (tests/test_bench_host.py starts two of these).

bench.py itself has no CPU path and gets none: this script stands a fake `nimpress_amd.capi` in (every score of
rank r is the constant r + 1, every partial sum 1.0 with nloci = rows of the shard), sends the "cuda" tensors
to the CPU and the "nccl" process group to gloo, and then runs bench.main() unchanged -- so the launcher
environment, the sharding, the exchange calls (multi.gather_scores / all_reduce_partial), the max-over-ranks
timing and the JSON line of the code that the driver runs on 8 GPUs are executed once before it gets there.
The printed line says "rehearsal": true and measures nothing."""
import ctypes
import json
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nimpress_amd  # noqa: E402
from nimpress_amd import capi as real  # noqa: E402  (constants and pure-numpy helpers only; load() is never called)

RANK = int(os.environ.get("RANK", "0"))
fake = types.ModuleType("nimpress_amd.capi")
for name in ("MODE_AUTO", "MODE_TWOPASS", "MODE_FUSED", "FMT_GT2", "FMT_DS32", "FMT_GT2M", "FMT_GT2X", "ROW_DESC_DTYPE",
             "row_descs", "make_params"):
    setattr(fake, name, getattr(real, name))
fake.load = lambda: None


class Cohort:
    def __init__(self, n, m, fmt=0, device=0):
        self.n, self.m = n, m

    def synth_at(self, *a):
        pass

    def optimize(self):
        pass

    def close(self):
        pass


class ScoreDef:
    def __init__(self, rows, device=0):
        self.rows = rows

    def close(self):
        pass


class Prof:
    ms_tally = ms_params = ms_accumulate = ms_reduce = 0.0
    ms_fused = 1.0
    n_tally = n_params = n_accumulate = 0
    n_fused = 1


class Scorer:
    def __init__(self, n, params, device=0):
        self.n, self.rows = n, 0

    def reset(self):
        self.rows = 0

    def score_cohort_def(self, cohort, sdef, row0, mode):
        self.rows += len(sdef.rows)

    def _fill(self, ptr, value):
        (ctypes.c_double * self.n).from_address(ptr)[:] = [value] * self.n

    def finish_device(self, offset, ptr):
        self._fill(ptr, RANK + 1.0)
        return self.rows

    def partial_device(self, ptr):
        self._fill(ptr, 1.0)
        return self.rows

    def normalize_device(self, ptr, nloci, offset):
        a = np.ctypeslib.as_array((ctypes.c_double * self.n).from_address(ptr))
        a /= 2.0 * nloci
        Scorer.last = (float(a[0]), int(nloci))

    def fused_geometry(self, m, fmt=0):
        return (3, 2, 1024)

    def profile_enable(self, on):
        pass

    def profile_get(self, reset=False):
        return Prof()

    def sync(self):
        pass

    def close(self):
        pass


fake.Cohort, fake.ScoreDef, fake.Scorer = Cohort, ScoreDef, Scorer
sys.modules["nimpress_amd.capi"] = fake
nimpress_amd.capi = fake

# "cuda" -> cpu, "nccl" -> gloo
torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a: None
torch.cuda.empty_cache = lambda: None
torch.cuda.current_stream = lambda *a: types.SimpleNamespace(synchronize=lambda: None)
for fname in ("empty", "ones", "zeros", "tensor"):
    orig = getattr(torch, fname)
    setattr(torch, fname, (lambda f: lambda *a, **k: f(*a, **dict(k, device="cpu") if "device" in k else k))(orig))
_init = dist.init_process_group
dist.init_process_group = lambda backend, device_id=None, **k: _init("gloo", **k)

import bench  # noqa: E402
from nimpress_amd import multi  # noqa: E402

_gather = multi.gather_scores


def gather_checked(local, n_scores, group=None):
    out = _gather(local, n_scores, group)
    expect = torch.arange(1, n_scores + 1, dtype=out.dtype).view(-1, 1).expand_as(out)
    assert torch.equal(out, expect), "gathered matrix is not [score of rank r = r + 1]"
    return out


multi.gather_scores = gather_checked
_print = print


def tagged_print(*a, **k):
    if len(a) == 1 and isinstance(a[0], str) and a[0].startswith("{"):
        d = json.loads(a[0])
        d["rehearsal"] = True
        if hasattr(Scorer, "last"):
            d["rehearsal_normalised"] = Scorer.last
        a = (json.dumps(d),)
    _print(*a, **k)


bench.print = tagged_print
sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
