"""libnps_rccl.so on the GPU box (one device: n = 1 is legal, the collectives become copies through RCCL): the two exchange
steps of SURVEY.md 8(e) through the C-ABI of include/nps_comm.h, from Python (ctypes) and from a plain C host
(tests/native/comm_driver.c, what a Nim host would do).  No run on more than one GPU is possible here; with N visible
GPUs `comm_driver N` covers them."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from nimpress_amd import capi
from test_gpu_parity import make_cohort

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_comm_one_device_equals_finish():
    import torch
    n, m = 4000, 640
    rng = np.random.default_rng(8)
    co = make_cohort(n, m, 515, rng)
    dev = capi.Cohort(n, m, fmt=capi.FMT_GT2X)
    dev.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    descs = capi.row_descs(co["beta"], co["eaf"], None, co["rie"])
    sc = capi.Scorer(n, capi.make_params())
    sc.score_cohort(dev, descs, 0, capi.MODE_AUTO)
    want, nloci = sc.finish(0.375)
    comm = capi.Comm(1)
    assert comm.size == 1 and comm.device(0) == 0
    d = torch.full((1, n), -1.0, dtype=torch.float64, device="cuda")
    sc.reset()
    sc.score_cohort(dev, descs, 0, capi.MODE_AUTO)
    nl = comm.allgather_scores([sc], [0.375], [d.data_ptr()])
    assert int(nl[0]) == nloci and np.array_equal(d.cpu().numpy()[0].view(np.int64), want.view(np.int64))
    sc.reset()
    sc.score_cohort(dev, descs, 0, capi.MODE_AUTO)
    d.fill_(-1.0)
    total = comm.allreduce_partial([sc], 0.375, [d.data_ptr()])
    assert total == nloci and np.array_equal(d.cpu().numpy()[0].view(np.int64), want.view(np.int64))
    # rows sharded x all scores: the multi-score partial sums through the same exchange
    S = 3
    md = np.zeros((S, m), dtype=capi.ROW_DESC_DTYPE)
    for s in range(S):
        md[s]["beta"] = np.round(rng.normal(0, 0.02, m), 4)
        md[s]["eaf"] = co["eaf"]
    src = capi.Cohort(n, m)
    src.synth(0, co["seed"], co["th"], co["tm"], co["tmi"])
    gm = capi.Cohort(n, m, fmt=capi.FMT_GT2M)
    gm.convert_from(src)
    msc = capi.MultiScorer(n, capi.make_params(), S)
    mdef = capi.MultiDef(md)
    msc.score_cohort(gm, mdef)
    offs = np.array([0.0, 0.5, -0.25])
    want_m, nl_m = msc.finish(offs)
    msc.reset()
    msc.score_cohort(gm, mdef)
    dm = torch.full((S, n), -1.0, dtype=torch.float64, device="cuda")
    got_nl = comm.allreduce_partial_multi([msc], S, n, offs, [dm.data_ptr()])
    assert np.array_equal(got_nl, np.asarray(nl_m, dtype=np.uint64))
    got = dm.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want_m))
    ok = ~np.isnan(want_m)
    assert np.max(np.abs(got[ok] - want_m[ok]) / np.maximum(np.abs(want_m[ok]), 1e-12)) <= 1e-12
    # refusals: what the call says must be what the scorers hold (a mismatch would be a device buffer overrun)
    msc.reset()
    msc.score_cohort(gm, mdef)
    with pytest.raises(capi.NpsError):
        comm.allreduce_partial_multi([msc], S - 1, n, offs[:S - 1], [dm.data_ptr()])
    with pytest.raises(capi.NpsError):
        comm.allreduce_partial_multi([msc], S, n - 1, offs, [dm.data_ptr()])
    # refusals: a context on another device / differing sample counts / too many devices
    with pytest.raises(capi.NpsError):
        capi.Comm(capi.device_count() + 1)
    with pytest.raises(capi.NpsError):
        capi.Comm(1, devices=[capi.device_count()])
    for x in (comm, msc, mdef, gm, src, sc, dev):
        x.close()


def test_comm_from_a_plain_c_host(tmp_path):
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    exe = str(tmp_path / "comm_driver")
    pkg = os.path.join(ROOT, "nimpress_amd")
    r = subprocess.run([gcc, "-std=c11", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                        os.path.join(ROOT, "tests", "native", "comm_driver.c"), "-L" + pkg, "-lnps_rccl", "-lnps",
                        "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "comm_driver ok: 1 device(s)" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    # the per-rank lines a first run on N GPUs is read by: one per rank and layout, no MISMATCH
    assert r.stdout.count("comm_driver rank 0 device 0:") == 2 and "MISMATCH" not in r.stdout, r.stdout[-800:]
