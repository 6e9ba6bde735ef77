"""Multi-GPU search: one process per GPU, corpus row-sharded, queries replicated, per-shard fused
top-k, then ONE exchange step -- a single all-gather of the packed shard message over RCCL/xGMI --
and a merge on every rank.  (The reference scores on GPU 0 only: SURVEY 2a/8e, scripts/ms_marco_eval.py:205-218.)

shard_bounds() gives contiguous row blocks; the message carries u32 LOCAL rows + the shard's row offset in its header
(8 bytes per entry on the wire), scores are canonical (shard-independent), so the merged result is identical to a
single-GPU search.

No host round trip before the collective: the per-shard search is asynchronous (CCR_SEARCH_ASYNC) and leaves the number of
queries it flagged in the message HEADER, on the stream.  After the all-gather every rank holds every rank's header, so all
ranks take the same branch: lists final (the usual case: no rank flagged a query) -> merge; otherwise every rank completes its
search (ccr_search_finish: a no-op where nothing was flagged) and ALL ranks repeat the all-gather -- the second collective is
matched by construction.

SHORT LISTS (large k: ranking() keeps 1001, scripts/ms_marco_eval.py:230).  The select / canonical re-score stage of a shard's search
does not shrink with the shard -- every rank would return the top-k of ITS rows although a shard of exchangeable rows holds only
k / R +- sqrt(k (1/R)(1 - 1/R)) of the global top-k.  So every rank searches and sends its canonical top-k_list, k_list =
short_list_length(k, R) ~ k / R + 6 sigma (196 instead of 1001 at R = 8), and the merge VERIFIES the shortcut exactly: a shard's list is
its exact top-k_list, hence if its last entry is not among the merged top-k none of its unsent rows is.  Queries for which some list was
consumed to its end (a corpus in topical order can put most of a query's top-k into one shard) are repeated with full lists by all
ranks together -- every rank derives the same query set from the same gathered bytes, so the repeat is a matched collective too."""
import ctypes
import math
import os

import torch
import torch.distributed as dist

from . import _lib, ops

HEADER_WORDS = _lib.SHARD_HEADER_BYTES // 4   # the header as int32 words: magic, n_flagged, k_valid, n_covered, row_offset (2), n_rows (2)
PAD_ID = torch.iinfo(torch.int64).max           # padding slot (r, p) carries id PAD_ID - (r k + p): distinct, above every real id


def shard_bounds(n_rows, world_size, rank, weights=None):
    """Contiguous row block of `rank`: [lo, hi).  weights=None: equal row counts.  weights = one non-negative cost per row (token
    counts: the encoder's time per passage is proportional to its tokens, and the reference's DataParallel scatters every batch
    evenly, scripts/al_0_rank.py:70-74,92): the blocks carry equal WEIGHT instead -- boundary r is the number of leading rows whose
    total weight stays within r / world of the whole, so a corpus whose passage length follows the row order (sorted by length, one
    source after another) still gives every rank the same encode time.  Every rank must pass the same weights."""
    if weights is None:
        base, rem = divmod(n_rows, world_size)
        lo = rank * base + min(rank, rem)
        return lo, lo + base + (1 if rank < rem else 0)
    cuts = weighted_cuts(weights, world_size)
    assert len(cuts) == world_size + 1 and cuts[-1] == n_rows, "weights: one per row"
    return int(cuts[rank]), int(cuts[rank + 1])


def weighted_cuts(weights, world_size):
    """[0 = c_0 <= c_1 <= ... <= c_world = n]: rank r takes rows [c_r, c_r+1) (shard_bounds with weights)."""
    import numpy as np
    w = np.asarray(weights, dtype=np.float64)
    assert w.ndim == 1 and (w >= 0).all(), "weights: non-negative, one per row"
    cum = np.cumsum(w)
    total = float(cum[-1]) if len(w) else 0.0
    if total <= 0.0:      # nothing to balance: equal row counts
        return [shard_bounds(len(w), world_size, r)[0] for r in range(world_size)] + [len(w)]
    inner = np.searchsorted(cum, total * np.arange(1, world_size) / world_size, side="right")
    cuts = [0] + [int(c) for c in inner] + [len(w)]
    if len(w) >= world_size:      # one heavy row must not leave a rank without rows (an empty shard cannot fill an exchange message)
        for r in range(1, world_size):
            cuts[r] = min(max(cuts[r], cuts[r - 1] + 1), len(w) - (world_size - r))
    return cuts


def agreed_cuts(n_rows, world_size, weights=None, group=None, device=None):
    """The world + 1 shard boundaries EVERY rank uses: rank 0 computes them and broadcasts them as int64 (one tiny collective).
    Computing them per rank is not enough once the blocks follow weights: a float cumsum + searchsorted over weights that differ by
    one ulp between ranks (another BLAS / thread count behind the polyfit, another tokenizer build, a caller's own array) moves a
    boundary on one rank only -- rank r's hi != rank r + 1's lo, rows silently dropped or encoded twice under wrong row offsets
    (round-5 advisor, medium).  Every rank still VALIDATES what it received: monotone, [0 .. n_rows], world + 1 entries.
    device: where the collective's tensor lives (a cuda device under RCCL; None = cpu for gloo)."""
    if weights is None:
        local = [shard_bounds(n_rows, world_size, r)[0] for r in range(world_size)] + [int(n_rows)]
    else:
        local = weighted_cuts(weights, world_size)
    if world_size <= 1 or not (dist.is_available() and dist.is_initialized()):
        return [int(c) for c in local]
    # the collective's tensor lives where the backend moves bytes: a cuda device under RCCL ("nccl"), the host under gloo
    if dist.get_backend(group) == "nccl":
        device = device if device is not None and torch.device(device).type == "cuda" else torch.device("cuda", torch.cuda.current_device())
    else:
        device = None
    t = torch.tensor(local, dtype=torch.int64, device=device if device is not None else "cpu")
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    cuts = [int(c) for c in t.cpu().tolist()]
    assert len(cuts) == world_size + 1 and cuts[0] == 0 and cuts[-1] == int(n_rows) and all(a <= b for a, b in zip(cuts, cuts[1:])), \
        f"shard cuts {cuts} do not tile [0, {n_rows})"
    return cuts


def largest_share(cuts):
    """Largest fraction of the corpus rows one rank holds (1 / world for equal row counts)."""
    n = int(cuts[-1])
    return max(b - a for a, b in zip(cuts, cuts[1:])) / n if n > 0 else 1.0


def short_list_length(k, world, sigmas=6.0, share=None):
    """Entries per query a rank sends in the short-list exchange: the share of a global top-k that one of `world` shards of
    exchangeable rows holds is Binomial(k, share); mean + 6 sigma + 8 is exceeded by some shard with probability < 1e-8 per query.
    share: the LARGEST row share of a rank (largest_share(cuts); default 1 / world = equal row counts) -- token-balanced cuts give
    unequal row counts (3.1x on a length-sorted corpus), and the rank with the most rows holds the most of a query's top-k."""
    if world <= 1:
        return int(k)
    share = 1.0 / world if share is None else min(1.0, max(float(share), 1.0 / world))
    return int(min(k, math.ceil(k * share + sigmas * math.sqrt(k * share * (1.0 - share))) + 8))


SHORT_LIST_MAX_REPEAT_FRACTION = 0.05   # more queries than this repeated with full lists -> later exchanges of this (k, world) send full lists
_SUSPENDED = {}                         # (k, world) -> the exchange statistics that suspended the short lists


def short_lists_possible(k, world, share=None):
    kl = short_list_length(k, world, share=share)
    return world > 1 and kl < k and world * kl * 12 <= ops.SHORT_LIST_LDS_BYTES


def short_lists_pay(k, world, share=None):
    """Use the short-list exchange?  When it cuts the lists by at least a quarter, the R lists of a query fit the merge kernel's LDS, and
    it has not been SUSPENDED for this (k, world): the k / R + 6 sigma budget assumes exchangeable rows, and a corpus in topical order
    (adjacent passages of one document) concentrates a query's top-k in few shards -- every flagged query then costs a second search
    and a second collective.  An exchange that had to repeat more than SHORT_LIST_MAX_REPEAT_FRACTION of its queries suspends the
    shortcut (note_short_list_outcome: every rank sees the same gathered flags, so every rank switches at the same step).
    CCREC_SHORT_LISTS=0 / 1 switches it off / forces it where it is possible (the A/B knob)."""
    kl = short_list_length(k, world, share=share)
    possible = short_lists_possible(k, world, share)
    env = os.environ.get("CCREC_SHORT_LISTS", "").strip()
    if env in ("0", "1"):
        return possible and env == "1"
    return possible and 4 * kl <= 3 * k and (int(k), int(world)) not in _SUSPENDED


def note_short_list_outcome(k, world, n_q, repeated):
    """Called by every short-list exchange with the number of queries it had to repeat with full lists."""
    if n_q > 0 and repeated > SHORT_LIST_MAX_REPEAT_FRACTION * n_q:
        _SUSPENDED.setdefault((int(k), int(world)), {"queries": int(n_q), "repeated": int(repeated)})


def short_lists_suspended(k, world):
    return _SUSPENDED.get((int(k), int(world)))


def resume_short_lists():
    """Forget every suspension (a new corpus)."""
    _SUSPENDED.clear()


def exchange_list_length(k, world, short_lists=None, blocked=False, share=None):
    """Entries per query and rank of an exchange for top-k -- from quantities that are IDENTICAL on every rank (k, world, the flags,
    the module's suspension state, the largest row share of the AGREED cuts), never from a rank's own shard size: ranks that
    disagreed would post collectives of different sizes."""
    if blocked or world <= 1:
        return int(k)
    short = short_lists_pay(k, world, share) if short_lists is None else (bool(short_lists) and short_lists_possible(k, world, share))
    return short_list_length(k, world, share=share) if short else int(k)


class ShardMessage:
    """One rank's packed exchange message (include/ccr_retrieval.h: ccr_shard_header | scores [n_q,k] fp32 | rows [n_q,k] u32
    local) and the gathered copy of all ranks' messages: ONE all_gather_into_tensor moves everything (xGMI collectives are
    latency-bound at this size).  `send` is what ccr_search_shard / ccr_shard_message_fill write; after gather(), `recv` holds
    the R messages back to back for ccr_merge_shard_messages."""

    def __init__(self, n_q, k, device, world):
        self.n_q, self.k, self.world = int(n_q), int(k), int(world)
        n = self.n_q * self.k
        self.rows_at = (_lib.SHARD_HEADER_BYTES + n * 4 + 15) // 16 * 16
        self.nbytes = ops.shard_message_bytes(self.n_q, self.k)
        self.send = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.recv = torch.zeros(self.world * self.nbytes, dtype=torch.uint8, device=device)
        hb = _lib.SHARD_HEADER_BYTES
        self.header = self.send[:hb].view(torch.int32)
        self.scores = self.send[hb:hb + n * 4].view(torch.float32).view(self.n_q, self.k)
        self.rows = self.send[self.rows_at:self.rows_at + n * 4].view(torch.int32).view(self.n_q, self.k)   # u32 bit patterns
        per_rank = self.recv.view(self.world, self.nbytes)
        self.all_headers = per_rank[:, :hb].view(torch.int32)                                                  # [R, 8]
        self.all_scores = per_rank[:, hb:hb + n * 4].view(torch.float32).view(self.world, self.n_q, self.k)
        self.all_rows = per_rank[:, self.rows_at:self.rows_at + n * 4].view(torch.int32).view(self.world, self.n_q, self.k)
        self.is_cuda = self.send.is_cuda
        # headers land here (pinned) from a side stream, so reading them never waits for work enqueued after the collective
        self.headers_host = torch.empty(self.world, HEADER_WORDS, dtype=torch.int32)
        self.count_host = torch.zeros(1, dtype=torch.int32)      # short lists: queries the merge flagged (same route as the headers)
        if self.is_cuda:
            self.headers_host = self.headers_host.pin_memory()
            self.count_host = self.count_host.pin_memory()

    # ---- writing
    def fill(self, scores, ids, row_offset, n_rows):
        """Ordinary results ([n_q, k_valid] scores, int64 GLOBAL ids, k_valid <= k) -> the send buffer.  Device tensors go
        through ccr_shard_message_fill; host tensors (the gloo tests' CPU stand-in for an index) are laid out here."""
        k_valid = scores.shape[1]
        assert scores.shape[0] == self.n_q and k_valid <= self.k
        if self.is_cuda:
            ops.shard_message_fill(self.send, self.n_q, self.k, scores, ids, row_offset, n_rows)
            return
        hdr = _lib.ShardHeader(_lib.SHARD_MAGIC, 0, k_valid, 0, int(row_offset), int(n_rows))
        self.header.copy_(torch.frombuffer(bytearray(bytes(hdr)), dtype=torch.int32))
        self.scores.zero_()
        self.rows.zero_()
        local = ids.to(torch.int64) - int(row_offset)
        assert k_valid == 0 or (int(local.min()) >= 0 and int(local.max()) < 2 ** 32), "ids outside the shard's row range"
        self.scores[:, :k_valid] = scores
        self.rows[:, :k_valid] = ((local + 2 ** 31) % 2 ** 32 - 2 ** 31).to(torch.int32)   # u32 bit patterns

    # ---- exchange
    def gather(self, group=None):
        dist.all_gather_into_tensor(self.recv, self.send, group=group)

    def gather_async(self, group=None):
        """Start the all-gather on the communication stream and return its work handle: kernels enqueued afterwards on the
        compute stream (the next step's pack and search) overlap with it.  The message must not be rewritten before the
        exchange has been completed (double-buffer it)."""
        return dist.all_gather_into_tensor(self.recv, self.send, group=group, async_op=True)

    # ---- reading
    @staticmethod
    def parse_headers(words):
        """[R, 8] int32 host tensor -> list of dicts."""
        out = []
        for row in words.tolist():
            raw = (ctypes.c_int32 * HEADER_WORDS)(*row)
            h = _lib.ShardHeader.from_buffer_copy(raw)
            assert h.magic == _lib.SHARD_MAGIC, f"not a shard message (magic {h.magic:#x})"
            out.append({f: getattr(h, f) for f, _ in _lib.ShardHeader._fields_})
        return out

    def decoded(self):
        """The gathered messages as ([R, n_q, k] fp32 scores, [R, n_q, k] int64 global ids), padding slots as the merge kernel
        sees them: (-inf, PAD_ID - (r k + p)).  Host-side view for tests and for merge hooks; the product path merges the
        packed bytes directly (merge())."""
        hdrs = self.parse_headers(self.all_headers.cpu())
        scores = self.all_scores.clone()
        ids = self.all_rows.to(torch.int64).bitwise_and(0xFFFFFFFF)
        slot = torch.arange(self.k, dtype=torch.int64, device=ids.device)
        for r, h in enumerate(hdrs):
            ids[r] += h["row_offset"]
            if h["k_valid"] < self.k:
                scores[r, :, h["k_valid"]:] = -float("inf")
                ids[r, :, h["k_valid"]:] = (PAD_ID - (r * self.k + slot))[h["k_valid"]:]
        return scores, ids

    def merge(self, merge_fn=None):
        """Global ([n_q, k] fp32, [n_q, k] int64) from the gathered messages.  merge_fn: test hook taking decoded()."""
        if merge_fn is not None:
            return merge_fn(*self.decoded())
        return ops.merge_shard_messages(self.recv, self.world, self.n_q, self.k)

    def merge_short(self, k_out, headers, short_merge_fn=None):
        """Short-list exchange: the k_out best of the R x k entries per query + the verification flags (ops.merge_short_lists).
        short_merge_fn: test hook (decoded scores, decoded ids, truncated [R] bools, k_out) -> (scores, ids, flags [n_q], count)."""
        if short_merge_fn is not None:
            truncated = [h["n_rows"] > h["k_valid"] for h in headers]
            return short_merge_fn(*self.decoded(), truncated, k_out)
        return ops.merge_short_lists(self.recv, self.world, self.n_q, self.k, k_out)


class ShardExchange:
    """One exchange in flight: submit() starts it behind the (asynchronous) search, result() completes it.  Between the two the
    caller may enqueue the next step's pack and search: nothing in here waits for work enqueued after the collective.
    k_out > message.k: the message holds SHORT lists (module docstring); the merge + verification then runs on the SIDE stream right
    behind the collective and its 4-byte flag count travels to pinned memory the way the headers do, so result() reads both after
    waiting for the side stream's event only -- no host read behind work of a later step (r4 read count.item() on the compute stream:
    the host could not enqueue step i + 2 before step i + 1's search had finished).  Queries that fail the verification are repeated
    with full lists -- `queries` (this exchange's query rows) and `index` are needed for that.
    host_syncs counts the times result() had to synchronise with the COMPUTE stream (0 on the pipelined path)."""

    def __init__(self, message, index=None, group=None, merge_fn=None, k_out=None, queries=None, search_fn=None, short_merge_fn=None):
        self.message, self.index, self.group, self.merge_fn = message, index, group, merge_fn
        self.k_out = message.k if k_out is None else int(k_out)
        assert self.k_out >= message.k
        self.queries, self.search_fn, self.short_merge_fn = queries, search_fn, short_merge_fn
        self.work = self.event = None
        self.repeated = False
        self.fallback_queries = 0      # short lists: queries repeated with full lists
        self.headers = None
        self.wait_ms = 0.0             # host time result() spent waiting for the collective (+ the header copy) to arrive
        self.host_syncs = 0
        self.side_merge = None         # (scores, ids, flags, count) of the merge the side stream ran

    def submit(self, _work=None):
        """_work (tests): a stand-in for the collective's work handle over an already gathered `recv` -- exercises the stream-ordered
        (RCCL) path of submit() / result() where no second GPU exists."""
        m = self.message
        self.work = _work if _work is not None else m.gather_async(self.group)
        if m.is_cuda and (_work is not None or dist.get_backend(self.group) == "nccl"):
            # a side stream waits for the collective and copies the R headers to pinned memory; result() waits for THAT event
            side = _side_stream(m.send.device)
            short = self.k_out > m.k and self.short_merge_fn is None and self.merge_fn is None
            outs = None
            if short:   # outputs allocated on the caller's stream (its allocator pool), written by the side stream behind the collective
                dev = m.send.device
                outs = (torch.empty(m.n_q, self.k_out, dtype=torch.float32, device=dev), torch.empty(m.n_q, self.k_out, dtype=torch.int64, device=dev),
                        torch.zeros(max(1, m.n_q), dtype=torch.int32, device=dev)[:m.n_q], torch.zeros(1, dtype=torch.int32, device=dev))
                side.wait_stream(torch.cuda.current_stream(dev))     # ... after the two zero fills
                for t in outs:                                       # written by the side stream: an exchange dropped between submit() and
                    t.record_stream(side)                            # result() must not hand these blocks back to the pool while the merge is pending
            with torch.cuda.stream(side):
                self.work.wait()
                m.headers_host.copy_(m.all_headers, non_blocking=True)
                if short:
                    # (if some rank's lists turn out not to be final -- headers: n_flagged > n_covered -- this merge is discarded in result())
                    self.side_merge = ops.merge_short_lists(m.recv, m.world, m.n_q, m.k, self.k_out, out=outs)
                    m.count_host.copy_(self.side_merge[3], non_blocking=True)
                self.event = torch.cuda.Event()
                self.event.record(side)
        return self

    def result(self):
        import time
        m = self.message
        t0 = time.perf_counter()
        if self.event is not None:
            self.event.synchronize()                              # the collective + the header copy (+ the short-list merge), nothing later
            torch.cuda.current_stream(m.send.device).wait_event(self.event)   # later kernels read `recv` / the merged lists behind it
            words = m.headers_host
        else:
            self.work.wait()
            words = m.all_headers.cpu()
            self.host_syncs += 1
        self.wait_ms = (time.perf_counter() - t0) * 1e3
        self.headers = m.parse_headers(words)
        if self.index is not None and getattr(self.index, "_deferred", None) is not None:
            self.index.finish()    # its own event is long complete: fills last_stats(); re-does the queries the search flagged (rare)
        if any(h["n_flagged"] > h["n_covered"] for h in self.headers):
            # some rank's lists were not final when they were exchanged.  Every rank sees the same headers, so every rank is
            # here: the flagged ranks have completed their lists in finish() above, all ranks repeat the collective.
            self.repeated = True
            self.side_merge = None
            m.header[1:2].zero_()
            m.gather(self.group)
        if self.k_out == m.k:
            return m.merge(self.merge_fn)
        if self.side_merge is not None:
            scores, ids, flags, _ = self.side_merge
            self.fallback_queries = int(m.count_host[0])          # arrived in pinned memory with the event above
        else:
            scores, ids, flags, count = m.merge_short(self.k_out, self.headers, self.short_merge_fn)
            self.fallback_queries = int(count.item())             # (gloo / test hooks / a repeated collective: a synchronous path anyway)
            self.host_syncs += 1
        note_short_list_outcome(self.k_out, m.world, m.n_q, self.fallback_queries)
        if self.fallback_queries:
            # a list was consumed to its end for these queries: all ranks (same flags from the same bytes) repeat them with full lists
            which = flags.nonzero().squeeze(1)
            again = self.queries[which].contiguous() if self.queries is not None else None
            s2, i2 = sharded_search(self.index, again, self.k_out, self.group, merge_fn=self.merge_fn, search_fn=self.search_fn,
                                    short_lists=False)
            scores[which], ids[which] = s2, i2
            self.host_syncs += 1
        return scores, ids


_SIDE = {}


def _side_stream(device):
    key = torch.device(device).index
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def submit_sharded_search(index, queries_bf16, k, group=None, message=None, merge_fn=None, short_lists=None, share=None):
    """Asynchronous per-shard search straight into a packed message + the all-gather behind it -> ShardExchange (call
    .result() for the merged lists).  short_lists: None = short_lists_pay(k, world) (and the shard holds k_list rows); a `message`
    built for k_list < k entries selects the short-list exchange by itself.  Needs min(k, k_list) <= index.n_rows (sharded_search
    handles tiny shards)."""
    world = dist.get_world_size(group)
    n_q = queries_bf16.shape[0]
    if message is not None:
        k_list = message.k
    else:
        k_list = exchange_list_length(k, world, short_lists, share=share)
        message = ShardMessage(n_q, k_list, queries_bf16.device, world)
    assert (message.n_q, message.world) == (n_q, world) and message.k <= k
    # (never re-size the lists from this rank's own shard: the ranks' collectives must agree -- tiny shards go through sharded_search)
    assert message.k <= index.n_rows, f"shard of {index.n_rows} rows cannot fill lists of {message.k}: use sharded_search"
    index.search_shard(queries_bf16, message.k, message.send, defer=True)
    return ShardExchange(message, index, group, merge_fn, k_out=k, queries=queries_bf16).submit()


def sharded_search(index, queries_bf16, k, group=None, merge_fn=None, search_fn=None, message=None, block=None, n_total=None,
                   short_lists=None, short_merge_fn=None, share=None):
    """index: this rank's CorpusIndex (built with global_row_offset = its shard's first row).
    message: optional reusable ShardMessage(n_q, k or short_list_length(k, world), device, world) -- the search then writes straight into it.
    block: (ptr, idx) CSR of per-query blocked GLOBAL row ids (the same on every rank; each shard applies its own part).
    n_total: rows of the whole corpus; k is clamped to it (a corpus smaller than k cannot fill k ranks).
    short_lists: None = automatic (short_lists_pay), False = always full lists, True = short lists wherever k_list < k.
    share: the largest row share of a rank under the cuts every rank agreed on (largest_share(agreed_cuts(...))): sizes the short lists
    for unequal shards; None = equal row counts.
    search_fn / merge_fn / short_merge_fn: test hooks (CPU stand-ins for the per-shard search and the merges)."""
    if n_total is not None:
        k = min(k, int(n_total))
    k_local = min(k, index.n_rows)
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    if not multi:
        if block is not None and search_fn is None:
            return index.search_blocked(queries_bf16, k_local, block[0], block[1])
        return (search_fn or index.search)(queries_bf16, k_local)
    world = dist.get_world_size(group)
    n_q = queries_bf16.shape[0] if queries_bf16 is not None else -1     # (None: a test hook that ignores its query argument)
    if message is not None:
        k_list = message.k
    else:
        k_list = exchange_list_length(k, world, short_lists, blocked=block is not None, share=share)
    direct = block is None and search_fn is None and k_list <= index.n_rows and 0 < n_q <= ops.MAX_QUERIES_PER_SEARCH
    if direct:   # the kernel writes the exchange message itself, no host round trip (larger batches are searched in pieces below)
        if message is None:
            message = ShardMessage(n_q, k_list, queries_bf16.device, world)
        return submit_sharded_search(index, queries_bf16, k, group, message, merge_fn).result()
    # blocked lists, tiny shards (fewer rows than entries asked for: the message pads with (-inf, distinct ids), so every output slot is
    # written even when the whole corpus holds fewer than k rows -- pass n_total to clamp k instead) and test hooks: ordinary results -> message
    k_search = min(k_list, index.n_rows)
    if block is not None and search_fn is None:
        scores, ids = index.search_blocked(queries_bf16, k_search, block[0], block[1])
    else:
        scores, ids = (search_fn or index.search)(queries_bf16, k_search)
    if message is None:
        message = ShardMessage(scores.shape[0], k_list, scores.device, world)
    lo = index.offset
    message.fill(scores, ids, lo, index.n_rows)
    return ShardExchange(message, index, group, merge_fn, k_out=k, queries=queries_bf16, search_fn=search_fn,
                         short_merge_fn=short_merge_fn).submit().result()
