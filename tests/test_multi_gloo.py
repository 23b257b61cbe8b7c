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


# ---- one score, rows sharded over ranks: all-reduce of the un-normalised sums ----
def _row_worker(rank, world, port, n, m, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eaf, codes = _cohort(n, m, 11)
        beta = np.round(np.random.default_rng(5).normal(0, 0.05, m), 4)
        r0, r1 = multi.shard_rows(m, world, rank)
        sc = refcpu.RefScorer(n, refcpu.make_params("ps", "homref", "int_ps", 0.05, 10))
        for j in range(r0, r1):
            sc.row_gt(refcpu.codes_to_gt(codes[j], n), 2, 1, False, beta[j], eaf[j])
        sums, nloci = sc.partial()
        sc.finish(0.0)
        t, total = multi.all_reduce_partial(torch.from_numpy(sums.copy()), nloci)
        ret[rank] = (t.numpy() / (2.0 * total) + 0.25, total)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("m", [40, 3, 1])
def test_row_sharded_all_reduce(m):
    world, n = 2, 131
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_row_worker, args=(world, port, n, m, ret), nprocs=world, join=True)
    eaf, codes = _cohort(n, m, 11)
    beta = np.round(np.random.default_rng(5).normal(0, 0.05, m), 4)
    ref, _, ref_nloci = refcpu.score_packed(codes, n, np.zeros(m, np.int32), np.zeros(m, np.int32), beta,
                                            eaf, refcpu.make_params("ps", "homref", "int_ps", 0.05, 10),
                                            0.25)
    for r in range(world):
        got, total = ret[r]
        assert total == ref_nloci
        # blocked summation order: a few ulps of the largest partial sum
        assert np.allclose(got, ref, rtol=0, atol=1e-13 * max(1.0, np.abs(beta).sum()))


# ---- rows sharded over ranks x ALL S scores on every rank: one all-reduce of the [S, N] sums ----
def _row_matrix_worker(rank, world, port, n, m, S, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eaf, codes = _cohort(n, m, 13)
        r0, r1 = multi.shard_rows(m, world, rank, align=128)   # (the multi-score cohort layout has 128-row superblocks)
        sums = np.zeros((S, n))
        nloci = np.zeros(S, np.int64)
        for s_ in range(S):      # stand-in for ONE multi-score pass over this rank's block (nps_multi_partial_device)
            beta = np.round(np.random.default_rng(300 + s_).normal(0, 0.05, m), 4)
            sc = refcpu.RefScorer(n, refcpu.make_params("ps", "homref", "int_ps", 0.05, 10))
            for j in range(r0, r1):
                sc.row_gt(refcpu.codes_to_gt(codes[j], n), 2, 1, False, beta[j], eaf[j])
            sums[s_], nloci[s_] = sc.partial()
            sc.finish(0.0)
        local = (r0, r1, float(np.abs(sums).max()) if sums.size else 0.0, nloci.copy())
        t, total = multi.all_reduce_partial_matrix(torch.from_numpy(sums), nloci)
        out = multi.normalize_matrix(t, total, [0.1 * s_ for s_ in range(S)])
        ret[rank] = (out.numpy().copy(), total.numpy().copy(), local)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("m", [300, 129, 5])
def test_row_sharded_multi_score_all_reduce(m):
    world, n, S = 2, 97, 3
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_row_matrix_worker, args=(world, port, n, m, S, ret), nprocs=world, join=True)
    eaf, codes = _cohort(n, m, 13)
    for s_ in range(S):
        beta = np.round(np.random.default_rng(300 + s_).normal(0, 0.05, m), 4)
        ref, _, ref_nloci = refcpu.score_packed(codes, n, np.zeros(m, np.int32), np.zeros(m, np.int32), beta, eaf,
                                                refcpu.make_params("ps", "homref", "int_ps", 0.05, 10), 0.1 * s_)
        for r in range(world):
            got, total, _ = ret[r]
            assert total[s_] == ref_nloci
            assert np.allclose(got[s_], ref, rtol=0, atol=1e-13 * max(1.0, np.abs(beta).sum()))


# ---- world size 8 (no 8-GPU node has ever been available, so the N = 8 shapes are rehearsed here) ----
@pytest.mark.parametrize("n_scores", [8, 11, 5])
def test_sharded_scores_gathered_in_order_world8(n_scores):
    """configs[3] in miniature: 8 (or more, or fewer) score definitions over 8 ranks, one all-gather (nimpress.nim:747-753 is
    the caller that sits on top: one score file per invocation).  With 5 scores three ranks have nothing to score and send
    padding only; with 11 the gather carries two rows per rank."""
    world, n, m = 8, 131, 24
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n, m, n_scores, ret), nprocs=world, join=True)
    ref = _score_all(n, m, n_scores)
    for r in range(world):
        assert np.array_equal(ret[r], ref, equal_nan=True), r


def test_row_sharded_multi_score_world8_empty_blocks():
    """configs[3]'s 708-row union over 8 ranks in 128-aligned blocks (the multi-score layout's superblocks): six ranks hold
    rows, two hold EMPTY blocks -- their sums are zero, their nloci 0, and the all-reduce + normalisation still equal the
    single-rank result on every rank."""
    world, n, m, S = 8, 67, 708, 2
    blocks = [multi.shard_rows(m, world, r, align=128) for r in range(world)]
    assert [b for b in blocks if b[1] > b[0]] == [(0, 128), (128, 256), (256, 384), (384, 512), (512, 640), (640, 708)]
    assert blocks[6] == (708, 708) and blocks[7] == (708, 708)
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_row_matrix_worker, args=(world, port, n, m, S, ret), nprocs=world, join=True)
    eaf, codes = _cohort(n, m, 13)
    for r in (6, 7):
        r0, r1, absmax, nloci = ret[r][2]
        assert (r0, r1) == (708, 708) and absmax == 0.0 and not nloci.any()
    assert sum(int(ret[r][2][3][0]) for r in range(world)) == int(ret[0][1][0])
    for s_ in range(S):
        beta = np.round(np.random.default_rng(300 + s_).normal(0, 0.05, m), 4)
        ref, _, ref_nloci = refcpu.score_packed(codes, n, np.zeros(m, np.int32), np.zeros(m, np.int32), beta, eaf,
                                                refcpu.make_params("ps", "homref", "int_ps", 0.05, 10), 0.1 * s_)
        for r in range(world):
            got, total, _ = ret[r]
            assert total[s_] == ref_nloci
            assert np.allclose(got[s_], ref, rtol=0, atol=1e-13 * max(1.0, np.abs(beta).sum()))
            assert np.array_equal(got[s_], ret[0][0][s_])      # every rank ends with the same bits


def test_shard_rows_partition():
    for n_rows in (0, 1, 3, 4, 5, 17, 1000, 1_000_003):
        for world in (1, 2, 3, 8):
            blocks = [multi.shard_rows(n_rows, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n_rows
            for (a0, a1), (b0, b1) in zip(blocks, blocks[1:]):
                assert a1 == b0 and a0 <= a1
            assert all(b[0] % 4 == 0 or b[0] == n_rows for b in blocks)
