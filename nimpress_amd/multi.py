"""Multi-GPU evaluation: score definitions sharded over ranks, one RCCL all-gather at the end
(or, for ONE long score, its rows sharded over ranks and one RCCL all-reduce before the
normalisation: shard_rows / all_reduce_partial).

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).
The path shards by score file (BASELINE.json north_star): every rank holds the cohort, evaluates
its own scores with libnps, and the only exchange is the gather of the samples x scores matrix.
Nothing here computes scores: `score_fn` does (libnps on a GPU box; the tests pass the oracle).
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def shard_indices(n_scores: int, world: int, rank: int) -> List[int]:
    """Round-robin assignment: score i goes to rank i % world (balances unequal score sizes a
    little better than contiguous blocks when files are listed by size)."""
    return list(range(rank, n_scores, world))


def gather_scores(local: torch.Tensor, n_scores: int, group=None) -> torch.Tensor:
    """local: [k_local, N] float64 scores of this rank's shard (rows in shard order).
    Returns the full [n_scores, N] matrix on every rank (score order restored)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        assert local.shape[0] == n_scores
        return local
    n = local.shape[1]
    k_max = (n_scores + world - 1) // world
    send = torch.zeros((k_max, n), dtype=local.dtype, device=local.device)
    send[: local.shape[0]] = local
    recv = torch.empty((world, k_max, n), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv.view(world * k_max, n), send, group=group)
    out = torch.empty((n_scores, n), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard_indices(n_scores, world, r)
        if idx:
            out[idx] = recv[r, : len(idx)]
    assert len(shard_indices(n_scores, world, rank)) == local.shape[0]
    return out


def evaluate_sharded(n_scores: int, n_samples: int,
                     score_fn: Callable[[int, torch.Tensor], None],
                     device: torch.device, group=None) -> torch.Tensor:
    """score_fn(i, out_row) writes the N scores of score definition i into out_row (a float64
    view on `device`).  Returns the gathered [n_scores, N] matrix."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = shard_indices(n_scores, world, rank)
    local = torch.empty((len(mine), n_samples), dtype=torch.float64, device=device)
    for j, i in enumerate(mine):
        score_fn(i, local[j])
    return gather_scores(local, n_scores, group)


def shard_rows(n_rows: int, world: int, rank: int, align: int = 4) -> tuple:
    """Contiguous block [r0, r1) of a score's rows for `rank` (second layout:
    every GPU holds all samples of its rows, so tallies stay local and exact).  Block starts are
    multiples of `align` (the resident 2-bit cohort stores rows in groups of 4)."""
    per = (n_rows + world - 1) // world
    per = (per + align - 1) // align * align
    r0 = min(n_rows, rank * per)
    r1 = min(n_rows, r0 + per)
    return r0, r1


def all_reduce_partial(sums: torch.Tensor, nloci: int, group=None):
    """The one exchange of the row-sharded layout: sum all-reduce of the un-normalised score sums
    (float64 [N], from nps_partial_device) and of nloci.  Returns (sums, nloci_total); the caller
    normalises with nps_normalize_device (sums / (2 nloci) + offset, nimpress.nim:643-649)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return sums, int(nloci)
    cnt = torch.tensor([int(nloci)], dtype=torch.int64, device=sums.device)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    return sums, int(cnt.item())


def all_reduce_partial_matrix(sums: torch.Tensor, nloci, group=None):
    """Rows sharded over the GPUs AND all S scores on every GPU (one multi-score pass per GPU over its block of the
    cohort's rows: 1 / world of the matrix per GPU instead of a replica): the one exchange is a sum all-reduce of
    the un-normalised [S, N] sums (from nps_multi_partial_device) and of the S nloci counts.  Returns
    (sums, nloci_total [S] int64 tensor on the CPU)."""
    cnt = torch.as_tensor([int(x) for x in nloci], dtype=torch.int64, device=sums.device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    return sums, cnt.cpu()


def normalize_matrix(sums: torch.Tensor, nloci: torch.Tensor, offsets) -> torch.Tensor:
    """nimpress.nim:643-649 per score, in place: sums[s] / (nloci[s] * 2.0) + offset[s] (float64; nloci = 0 gives NaN
    as in the reference)."""
    d = (nloci.to(torch.float64) * 2.0).to(sums.device).view(-1, 1)
    off = torch.as_tensor(list(offsets), dtype=torch.float64, device=sums.device).view(-1, 1)
    sums.div_(d).add_(off)
    return sums
