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
