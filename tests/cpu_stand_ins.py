"""CPU stand-ins for the device ops that the N > 1 CONTROL FLOW touches (bench.Workload, ccrec_amd.dist): oracle-backed, so that the
world-8 protocol -- message sizes, header parsing of eight ranks, short lists at k_list = 196, the repeat / suspension branches, the
bench's pipelined step loop and its exchange record -- can be rehearsed over gloo on a box without a GPU.

TEST INFRASTRUCTURE: only tests/ import this module.  Nothing here is a product path (the product has no CPU fallback)."""
import ctypes

import numpy as np
import torch

from ccrec_amd import _lib
from ccrec_amd import dist as cdist
from oracle import oracle as orc

STAT_KEYS = [f for f, _ in _lib.SearchStats._fields_]


def _bits(t):
    """bf16 torch tensor -> uint16 numpy (the oracle's representation)."""
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def pack_bf16(x, normalize=False, out=None, return_norms=False, norm_bounds=None):
    bits = orc.normalize_pack_bf16(x.numpy()) if normalize else orc.pack_bf16(x.numpy())
    packed = torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16)
    if out is not None:
        out.copy_(packed)
        packed = out
    if norm_bounds is not None:
        norm_bounds.copy_(torch.from_numpy(orc.row_norms_bf16(bits).astype(np.float32) * 1.001))
    return packed


class OracleIndex:
    """ops.CorpusIndex with the oracle behind it: canonical top-k of the local rows."""
    calls = []          # (n_q, k) of every search of this process

    def __init__(self, corpus_bf16, global_row_offset=0, norm_bounds=None, workspace=None):
        self.D = _bits(corpus_bf16)
        self.n_rows, self.dim = self.D.shape
        self.offset = int(global_row_offset)
        self.workspace = workspace
        self._deferred = None

    def search(self, queries_bf16, k, flags=0, out=None, defer=False):
        OracleIndex.calls.append((int(queries_bf16.shape[0]), int(k)))
        ids, sc = orc.canonical_search(_bits(queries_bf16), self.D, k)
        s, i = torch.from_numpy(sc), torch.from_numpy(ids + self.offset)
        self._deferred = (s, i) if defer else None
        return s, i

    def search_shard(self, queries_bf16, k, message, defer=False, flags=0):
        s, i = self.search(queries_bf16, k)
        n_q = s.shape[0]
        hdr = _lib.ShardHeader(_lib.SHARD_MAGIC, 0, k, 0, self.offset, self.n_rows)
        hb = _lib.SHARD_HEADER_BYTES
        message[:hb].copy_(torch.frombuffer(bytearray(bytes(hdr)), dtype=torch.uint8))
        rows_at = (hb + n_q * k * 4 + 15) // 16 * 16
        message[hb:hb + n_q * k * 4].view(torch.float32).view(n_q, k).copy_(s)
        local = (i - self.offset).to(torch.int64)
        message[rows_at:rows_at + n_q * k * 4].view(torch.int32).view(n_q, k).copy_(((local + 2 ** 31) % 2 ** 32 - 2 ** 31).to(torch.int32))
        self._deferred = (queries_bf16, message) if defer else None

    def finish(self):
        self._deferred = None

    def last_stats(self):
        st = dict.fromkeys(STAT_KEYS, 0)
        st.update(ms_main=1.0, ms_total=1.0, main_launches=1, sublists=8)
        return st


def _gathered(gathered, world, n_q, k):
    m = cdist.ShardMessage(n_q, k, "cpu", world)
    m.recv.copy_(gathered)
    return m, m.parse_headers(m.all_headers)


def merge_shard_messages(gathered, world, n_q, k):
    m, _ = _gathered(gathered, world, n_q, k)
    gs, gi = m.decoded()
    s, i = orc.merge_topk(gs.numpy(), gi.numpy())
    return torch.from_numpy(s), torch.from_numpy(i)


def merge_short_lists(gathered, world, n_q, k_list, k_out, out=None):
    m, hdrs = _gathered(gathered, world, n_q, k_list)
    gs, gi = m.decoded()
    truncated = [h["n_rows"] > h["k_valid"] for h in hdrs]
    s, i, flags = orc.merge_short_lists(gs.numpy(), gi.numpy(), truncated, k_out)
    return torch.from_numpy(s), torch.from_numpy(i), torch.from_numpy(flags.astype(np.int32)), torch.tensor([int(flags.sum())], dtype=torch.int32)


def install():
    """Patch ccrec_amd.ops (and the two torch.cuda calls the bench's loop makes) in THIS process."""
    from ccrec_amd import ops
    ops.pack_bf16 = pack_bf16
    ops.CorpusIndex = OracleIndex
    ops.merge_shard_messages = merge_shard_messages
    ops.merge_short_lists = merge_short_lists
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda *a, **k: None
    OracleIndex.calls = []
    return OracleIndex
