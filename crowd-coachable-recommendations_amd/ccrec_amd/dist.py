"""Multi-GPU search: one process per GPU, corpus row-sharded, queries replicated, per-shard fused
top-k, then ONE exchange step -- an all-gather of the [Q,k] (score, id) lists over RCCL/xGMI --
and a merge on every rank.  (The reference scores on GPU 0 only: SURVEY 2a/8e.)

shard_bounds() gives contiguous row blocks; ids are global (local row + global_row_offset), and
scores are canonical (shard-independent), so the merged result is identical to a single-GPU search."""
import torch
import torch.distributed as dist

from . import ops


def shard_bounds(n_rows, world_size, rank):
    """Contiguous, balanced row block of `rank`: [lo, hi)."""
    base, rem = divmod(n_rows, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_topk(scores, ids, group=None):
    """[Q,k] per rank -> ([R,Q,k], [R,Q,k]) on every rank (two all_gather_into_tensor calls)."""
    world = dist.get_world_size(group)
    gs = torch.empty((world,) + tuple(scores.shape), dtype=scores.dtype, device=scores.device)
    gi = torch.empty((world,) + tuple(ids.shape), dtype=ids.dtype, device=ids.device)
    # concatenated-along-dim-0 form (works for both the RCCL and the gloo backend)
    dist.all_gather_into_tensor(gs.view(-1, scores.shape[-1]), scores.contiguous(), group=group)
    dist.all_gather_into_tensor(gi.view(-1, ids.shape[-1]), ids.contiguous(), group=group)
    return gs, gi


def merge_gathered(gs, gi, merge_fn=None):
    """Merge gathered per-shard lists.  merge_fn defaults to the HIP kernel (ops.merge_topk)."""
    return (merge_fn or ops.merge_topk)(gs, gi)


def sharded_search(index, queries_bf16, k, group=None, merge_fn=None, search_fn=None):
    """index: this rank's CorpusIndex (built with global_row_offset = its shard's first row)."""
    k_local = min(k, index.n_rows)
    scores, ids = (search_fn or index.search)(queries_bf16, k_local)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return scores, ids
    if k_local < k:  # tiny shard: pad so that every rank gathers the same shape
        pad = k - k_local
        scores = torch.cat([scores, torch.full((scores.shape[0], pad), -float("inf"), device=scores.device)], 1)
        ids = torch.cat([ids, torch.full((ids.shape[0], pad), torch.iinfo(torch.int64).max, dtype=torch.int64,
                                         device=ids.device)], 1)
    gs, gi = all_gather_topk(scores, ids, group)
    return merge_gathered(gs, gi, merge_fn)
