"""Multi-GPU search: one process per GPU, corpus row-sharded, queries replicated, per-shard fused
top-k, then ONE exchange step -- a single all-gather of the packed [Q,k] (score, id) message over RCCL/xGMI --
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


class TopkMessage:
    """One rank's packed exchange message {scores [n_q,k] fp32 | ids [n_q,k] int64} and the gathered copy of all
    ranks' messages: ONE all_gather_into_tensor moves both arrays (xGMI collectives are latency-bound at this size).
    `scores` / `ids` are views of the send buffer (search writes them directly); after gather(), `all_scores` /
    `all_ids` are rank-strided [R, n_q, k] views of the receive buffer, merged in place by ops.merge_topk."""

    def __init__(self, n_q, k, device, world):
        self.n_q, self.k, self.world = int(n_q), int(k), int(world)
        n = self.n_q * self.k
        self.ids_at = (n * 4 + 15) // 16 * 16            # byte offset of the id block (int64-aligned)
        self.nbytes = (self.ids_at + n * 8 + 15) // 16 * 16
        self.send = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.recv = torch.empty(self.world * self.nbytes, dtype=torch.uint8, device=device)
        self.scores = self.send[:n * 4].view(torch.float32).view(self.n_q, self.k)
        self.ids = self.send[self.ids_at:self.ids_at + n * 8].view(torch.int64).view(self.n_q, self.k)
        per_rank = self.recv.view(self.world, self.nbytes)
        self.all_scores = per_rank[:, :n * 4].view(torch.float32).view(self.world, self.n_q, self.k)
        self.all_ids = per_rank[:, self.ids_at:self.ids_at + n * 8].view(torch.int64).view(self.world, self.n_q, self.k)

    def gather(self, group=None):
        dist.all_gather_into_tensor(self.recv, self.send, group=group)
        return self.all_scores, self.all_ids

    def gather_async(self, group=None):
        """Start the all-gather on the communication stream and return its work handle: kernels enqueued afterwards on
        the compute stream (the next step's pack and search) overlap with it; call work.wait() before merging
        `all_scores` / `all_ids`.  The message must not be rewritten before that wait (double-buffer it)."""
        return dist.all_gather_into_tensor(self.recv, self.send, group=group, async_op=True)


def all_gather_topk(scores, ids, group=None, message=None):
    """[Q,k] per rank -> rank-strided ([R,Q,k], [R,Q,k]) views on every rank, one collective.
    message: a reusable TopkMessage; when `scores` / `ids` already ARE its views nothing is copied."""
    world = dist.get_world_size(group)
    n_q, k = scores.shape
    m = message if message is not None else TopkMessage(n_q, k, scores.device, world)
    assert (m.n_q, m.k, m.world) == (n_q, k, world)
    if scores.data_ptr() != m.scores.data_ptr():
        m.scores.copy_(scores)
    if ids.data_ptr() != m.ids.data_ptr():
        m.ids.copy_(ids)
    return m.gather(group)


def merge_gathered(gs, gi, merge_fn=None):
    """Merge gathered per-shard lists.  merge_fn defaults to the HIP kernel (ops.merge_topk)."""
    return (merge_fn or ops.merge_topk)(gs, gi)


def sharded_search(index, queries_bf16, k, group=None, merge_fn=None, search_fn=None, message=None, block=None, n_total=None):
    """index: this rank's CorpusIndex (built with global_row_offset = its shard's first row).
    message: optional reusable TopkMessage(n_q, k, device, world) -- the search then writes straight into it.
    block: (ptr, idx) CSR of per-query blocked GLOBAL row ids (the same on every rank; each shard applies its own part).
    n_total: rows of the whole corpus; k is clamped to it (a corpus smaller than k cannot fill k ranks)."""
    if n_total is not None:
        k = min(k, int(n_total))
    k_local = min(k, index.n_rows)
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    if block is not None and search_fn is None:
        scores, ids = index.search_blocked(queries_bf16, k_local, block[0], block[1])
    elif multi and search_fn is None and k_local == k:   # the kernel writes the exchange message itself
        if message is None:
            message = TopkMessage(queries_bf16.shape[0], k, queries_bf16.device, dist.get_world_size(group))
        scores, ids = index.search(queries_bf16, k, out=(message.scores, message.ids))
    else:
        scores, ids = (search_fn or index.search)(queries_bf16, k_local)
    if not multi:
        return scores, ids
    if k_local < k:
        # tiny shard: pad so that every rank gathers the same shape.  Pads score -inf and carry DISTINCT ids (per rank and
        # slot), so the merge's rank-by-counting stays a permutation and every output slot is written even when the whole
        # corpus holds fewer than k rows (the tail is then (-inf, pad id) -- pass n_total to clamp k instead).
        pad = k - k_local
        rank = dist.get_rank(group)
        scores = torch.cat([scores, torch.full((scores.shape[0], pad), -float("inf"), device=scores.device)], 1)
        pad_ids = torch.iinfo(torch.int64).max - (rank * k + torch.arange(pad, dtype=torch.int64, device=ids.device))
        ids = torch.cat([ids, pad_ids.expand(ids.shape[0], pad)], 1)
    gs, gi = all_gather_topk(scores, ids, group, message)
    return merge_gathered(gs, gi, merge_fn)
