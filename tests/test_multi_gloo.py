"""world_size-2 gloo test of the N>1 path (CPU): score definitions sharded over ranks, scores
gathered with one all-gather.  The per-score numbers come from the oracle here (no GPU in this
container); on a GPU box the same code path runs with libnps and backend nccl (RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nimpress_amd import multi
from oracle import refcpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cohort(n, m, seed):
    rng = np.random.default_rng(seed)
    eaf = np.round(rng.uniform(0.05, 0.5, m), 4)
    miss = rng.uniform(0, 0.1, m)
    th, tm, tmi = refcpu.hwe_thresholds(eaf, miss)
    return eaf, refcpu.synth_rows(n, 0, m, seed, th, tm, tmi)


def _score_all(n, m, n_scores):
    eaf, codes = _cohort(n, m, 7)
    out = []
    for i in range(n_scores):
        beta = np.round(np.random.default_rng(100 + i).normal(0, 0.05, m), 4)
        s, _, _ = refcpu.score_packed(codes, n, np.zeros(m, np.int32), np.zeros(m, np.int32), beta,
                                      eaf, refcpu.make_params("ps", "homref", "int_ps", 0.05, 10),
                                      0.01 * i)
        out.append(s)
    return np.stack(out)


def _worker(rank, world, port, n, m, n_scores, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eaf, codes = _cohort(n, m, 7)

        def score_fn(i, out_row):
            beta = np.round(np.random.default_rng(100 + i).normal(0, 0.05, m), 4)
            s, _, _ = refcpu.score_packed(codes, n, np.zeros(m, np.int32), np.zeros(m, np.int32),
                                          beta, eaf,
                                          refcpu.make_params("ps", "homref", "int_ps", 0.05, 10),
                                          0.01 * i)
            out_row.copy_(torch.from_numpy(s))

        full = multi.evaluate_sharded(n_scores, n, score_fn, torch.device("cpu"))
        ret[rank] = full.numpy().copy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_scores", [2, 5, 1])
def test_sharded_scores_gathered_in_order(n_scores):
    world, n, m = 2, 257, 40
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n, m, n_scores, ret), nprocs=world, join=True)
    ref = _score_all(n, m, n_scores)
    for r in range(world):
        assert np.array_equal(ret[r], ref, equal_nan=True), r


def test_shard_indices_cover_everything():
    for n_scores in (0, 1, 7, 8, 9):
        for world in (1, 2, 4, 8):
            seen = sorted(i for r in range(world) for i in multi.shard_indices(n_scores, world, r))
            assert seen == list(range(n_scores))
