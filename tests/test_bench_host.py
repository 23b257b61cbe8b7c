"""bench.py's host-side pieces that need no GPU: the rank launcher and the choice of checked samples."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cover_columns_touch_every_slice():
    import bench
    for k, n, sps in ((16, 500_000, 896 * 16), (16, 500_000, 960 * 16), (8, 200_000, 896 * 8), (16, 6, 0)):
        samples, n_slices, n_cols = bench.cover_columns(k, n, sps if sps else 960 * 16)
        per = (sps if sps else 960 * 16)
        assert samples.max() == n - 1 and samples.min() == 0
        assert len(np.unique(samples)) == samples.size
        assert set((samples // per).tolist()) == set(range(n_slices))
        assert n_cols >= min(n_slices, 3)


def test_bench_spawns_its_ranks_without_a_launcher():
    """`python bench.py --gpus 2` under plain python (no torch.distributed.run): the parent starts two rank
    processes with RANK / WORLD_SIZE / MASTER_* set and passes their exit code on.  There is no GPU here, so
    each rank stops with the "needs an MI355X" message -- which proves they were started as ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("needs an MI355X") == 2 or "NPS_E_NODEVICE" in r.stderr, r.stderr[-2000:]
    assert "must be launched" not in r.stderr


def _rehearse(world, extra, expect_rc=0, full=None):
    """N ranks of tests/rehearse_bench.py (a fake libnps, CPU tensors, gloo): bench.main() of every rank, unchanged.
    Returns the LAST stdout line of rank 0 (the compact object the driver parses); `full` receives the full object."""
    import json
    import socket
    import tempfile
    tmp = tempfile.mkdtemp()
    extra = list(extra) + ["--full-out", os.path.join(tmp, "full.json")]
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rehearse_bench.py"), "--gpus",
                                       str(world), "--steps", "2", "--warmup", "1", "--samples", "4000", "--variants",
                                       "640", "--no-extras", "--no-cpu-baseline", "--ds-samples", "3000", "--ds-variants",
                                       "1000", "--ds-chunk-rows", "256"] + extra,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=300) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        # (after rank 0 has given up, the others end with the watchdog's status or with a broken collective)
        assert p.returncode == expect_rc or (expect_rc and r and p.returncode), se[-3000:]
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not any(ln.startswith("{") for so, _ in outs[1:] for ln in so.splitlines())
    # the driver parses the LAST stdout line: it is the compact object, well under 4 KiB (round 4's 20 KB line was not parsed)
    assert outs[0][0].rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 4096, len(lines[0])
    if full is not None:
        full.update(json.load(open(os.path.join(tmp, "full.json"))))
        # ... and the same full object is on stderr
        assert any(ln.startswith("bench_full {") for ln in outs[0][1].splitlines())
    return json.loads(lines[0])


def test_bench_two_ranks_weak_scaling_rehearsal():
    full = {}
    d = _rehearse(2, [], full=full)
    assert d["rehearsal"] and d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak"
    assert d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["ms_per_step"] > 0
    assert d["config"]["parallelism"].startswith("score-sharded x2") and d["config"]["nloci"] == 640
    assert "cpu_baseline" not in d and "secondary" not in d and d["vs_baseline"] is None
    # the two strong-scaling legs that follow the headline on N > 1 (the other two north-star curves, same command):
    # configs[2] with the rows of one GT score sharded over the GPUs, configs[4] (FORMAT/DS) likewise, in chunks
    legs = d["multi_gpu"]                      # (summarised in the compact line, complete in the full object)
    gt, ds = legs["configs2_gt_rows_sharded"], legs["configs4_ds_rows_sharded"]
    assert gt["scaling"] == ds["scaling"] == "strong" and gt["n_gpus"] == ds["n_gpus"] == 2
    assert gt["nloci"] == 640 and ds["nloci"] == 1000 and gt["value"] > 0 and ds["value"] > 0
    assert "workload" not in ds and d["full_object"]
    fds = full["multi_gpu"]["configs4_ds_rows_sharded"]
    assert "2 resident chunk(s)" in fds["workload"] and "blocks of 512 rows" in fds["workload"]
    assert "roofline" in d and d["roofline"]["frac"] > 0 and "kernel_ms_per_launch" in d["roofline"]
    # every rank contributed partial sums of 1.0: 2.0 / (2 x nloci) after the all-reduce and the normalisation
    assert abs(d["rehearsal_normalised"][0] - 2.0 / 2000.0) < 1e-15 and d["rehearsal_normalised"][1] == 1000


def test_bench_three_ranks_strong_scaling_rehearsal():
    d = _rehearse(3, ["--scaling", "strong"])
    assert d["n_gpus"] == 3 and d["rccl_ranks"] == 3 and d["scaling"] == "strong"
    assert d["config"]["nloci"] == 640                      # the shards' nloci, all-reduced
    # every rank contributed partial sums of 1.0: 3.0 / (2 x 640) after the all-reduce and the normalisation
    assert abs(d["rehearsal_normalised"][0] - 3.0 / 1280.0) < 1e-15 and d["rehearsal_normalised"][1] == 640
    assert d["config"]["parallelism"].startswith("one score, rows sharded x3")


def test_bench_eight_ranks_weak_scaling_rehearsal():
    """The driver's N = 8 command has never met an 8-GPU node (SCALE_rNN: skipped): its whole host path -- eight ranks, the
    all-gather of eight score rows, both strong-scaling legs, the compact line -- runs here over gloo.  The line must stay
    under 4 KiB with `multi_gpu` summarised, and report the eight ranks it really had."""
    full = {}
    d = _rehearse(8, [], full=full)
    assert d["rehearsal"] and d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["scaling"] == "weak"
    assert d["config"]["parallelism"].startswith("score-sharded x8") and d["config"]["nloci"] == 640
    legs = d["multi_gpu"]
    gt, ds = legs["configs2_gt_rows_sharded"], legs["configs4_ds_rows_sharded"]
    assert gt["scaling"] == ds["scaling"] == "strong" and gt["n_gpus"] == ds["n_gpus"] == 8
    assert gt["nloci"] == 640 and ds["nloci"] == 1000 and gt["value"] > 0 and ds["value"] > 0
    assert "workload" not in ds and "workload" in full["multi_gpu"]["configs4_ds_rows_sharded"]
    # (a rank whose block of rows is empty still takes part in the all-reduce: eight partial sums of 1.0 each)
    assert abs(d["rehearsal_normalised"][0] - 8.0 / 2000.0) < 1e-15 and d["rehearsal_normalised"][1] == 1000


def test_bench_eight_ranks_strong_scaling_rehearsal():
    d = _rehearse(8, ["--scaling", "strong"])
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["scaling"] == "strong"
    assert d["config"]["nloci"] == 640 and d["config"]["parallelism"].startswith("one score, rows sharded x8")
    assert abs(d["rehearsal_normalised"][0] - 8.0 / 1280.0) < 1e-15 and d["rehearsal_normalised"][1] == 640


def test_bench_legs_watchdog_keeps_the_headline_line():
    """N > 1: if the strong-scaling legs do not finish in time (a collective some rank never reaches), rank 0 still
    prints the headline line -- without the legs -- and ends with status 3 (the other ranks: non-zero)"""
    d = _rehearse(2, ["--multi-legs-timeout", "0"], expect_rc=3)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert "did not finish" in d["multi_gpu"]["error"]


def test_compact_line_stays_under_4k_whatever_the_legs_return():
    """bench.compact_line on a full object shaped like round 4's (20 KB of secondary legs): contract keys, roofline,
    cpu_baseline and the parity flags survive; the line is under 4 KiB; an over-long text field is shed, never a number"""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_full.json")))
    assert len(json.dumps(full)) > 15000
    c = bench.compact_line(full)
    line = json.dumps(c)
    assert len(line) < 4096, len(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "score_delta_vs_reference",
              "secondary_summary"):
        assert k in c, k
    assert c["roofline"]["bound"] == "hbm" and 0.5 < c["roofline"]["frac"] < 1.0 and c["roofline"]["traffic"]
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] == 1 and c["cpu_baseline"]["sample"]
    assert c["score_delta_vs_reference"]["within_1e-6_relative"] is True
    assert c["config"]["workload"] and c["config"]["samples"] == 500000
    assert len(c["secondary_summary"]) <= 10 and c["secondary_summary"]["config4_e2e_s"] > 0
    # a pathological leg cannot push the line over the limit
    full["config"]["workload"] = "x" * 5000
    full["cpu_baseline"]["sample"] = "y" * 5000
    c = bench.compact_line(full)
    assert len(json.dumps(c)) < 4096 and c["value"] == bench._r(full["value"]) and "roofline" in c


def test_bench_one_rank_rehearsal_has_roofline_and_full_object():
    full = {}
    d = _rehearse(1, [], full=full)
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and "multi_gpu" not in d
    assert d["roofline"]["achieved"] > 0 and d["config"]["workload"]
    assert full["roofline"]["launches_per_step"]["fused"] > 0   # (the fake library reports one launch, whatever the steps)
