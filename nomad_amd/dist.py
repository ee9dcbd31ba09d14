"""Multi-GPU NOMAD scoring: clips shard across ranks, one all-gather of embeddings, local slabs.

The reference has no distributed code at all.  The path shards naturally
(section 8e): every clip runs through the backbone independently, and the only cross-clip step is
the N_deg x N_ref distance matrix.  So:

  rank r embeds deg[r-th slice] and ref[r-th slice]                  (no communication)
  ONE all-gather (RCCL over xGMI; gloo in CPU tests) of the ref embeddings -> every rank has (N_ref,256)
  rank r computes its (N_deg/W x N_ref) slab + row means             (no communication)
  optional all-gather of the row means for output

One process per GPU, ``torch.distributed`` (backend "nccl" is RCCL on ROCm).  The compute
callables are injected so the partition/collective logic is testable on CPU with gloo.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def partition(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced slice [start, stop) of ``n_items`` for ``rank``; first ranks get the remainder."""
    base, rem = divmod(n_items, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_gather_rows(x: torch.Tensor, group=None, equal: bool = False, force_collective: bool = False) -> torch.Tensor:
    """Concatenate per-rank (n_r, ...) tensors along dim 0 (n_r may differ between ranks).

    equal=True promises that every rank contributes the same number of rows: one collective, no size exchange and
    no host synchronisation (the steady-state path of bench.py).
    force_collective=True issues the collective(s) even in a group of ONE rank, where the result is the input: that is
    how the RCCL path (communicator set-up, the all-gather launch, its ordering against the engine's stream) is
    exercised on a single-GPU box (tests/test_gpu_rccl.py, bench.py under the distributed launcher)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force_collective):
        return x
    world = dist.get_world_size(group)
    if equal:
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
        return out
    counts = torch.zeros(world, dtype=torch.int64, device=x.device)
    mine = torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts_l: List[int] = [int(c) for c in counts.tolist()]
    m = max(counts_l)
    if all(c == m for c in counts_l):  # common case: one collective, no padding
        out = torch.empty((world * m,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
        return out
    pad = torch.zeros((m,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[:x.shape[0]] = x
    buf = torch.empty((world * m,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * m:r * m + counts_l[r]] for r in range(world)], dim=0)


class ShardedScorer:
    """NOMAD scores for this rank's degraded clips against the GLOBAL reference set.

    embed_fn(wav (B,N)) -> (B,256) fp32;  pairwise_fn(deg (Nd,256), ref (Nr,256), want_matrix) ->
    (dist (Nd,Nr) float64 | None, mean (Nd,) float64).  On the GPU these are ``Engine.embed`` and
    ``Engine.pairwise``.
    """

    def __init__(self, embed_fn: Callable, pairwise_fn: Callable, group=None, equal_shards: bool = False,
                 force_collective: bool = False):
        self.embed_fn = embed_fn
        self.pairwise_fn = pairwise_fn
        self.group = group
        self.equal_shards = equal_shards  # every rank holds the same number of reference clips
        self.force_collective = force_collective  # run the all-gather even at world size 1 (see all_gather_rows)

    def score(self, deg_wav: torch.Tensor, ref_wav: Optional[torch.Tensor], want_matrix: bool = False,
              ref_emb_local: Optional[torch.Tensor] = None):
        """-> (mean_local (Nd_local,), dist_local or None, ref_emb_all (N_ref,256)).

        deg_wav/ref_wav are THIS rank's slices.  When both have the same clip length they go through
        the backbone as one batch (one launch sequence)."""
        if ref_emb_local is None:
            if ref_wav is not None and ref_wav.shape[-1] == deg_wav.shape[-1]:
                emb = self.embed_fn(torch.cat([deg_wav, ref_wav], dim=0))
                deg_emb, ref_emb_local = emb[:deg_wav.shape[0]], emb[deg_wav.shape[0]:]
            else:
                deg_emb = self.embed_fn(deg_wav)
                ref_emb_local = self.embed_fn(ref_wav)
        else:
            deg_emb = self.embed_fn(deg_wav)
        ref_all = all_gather_rows(ref_emb_local.contiguous(), self.group, equal=self.equal_shards,
                                  force_collective=self.force_collective)
        d, mean = self.pairwise_fn(deg_emb.contiguous(), ref_all, want_matrix)
        return mean, d, ref_all

    def gather_scores(self, mean_local: torch.Tensor) -> torch.Tensor:
        """All ranks' row means in global deg order (for output on any rank)."""
        return all_gather_rows(mean_local, self.group, force_collective=self.force_collective)
