"""Encoder-side fusion: length-sorted variable-length batches, pooled + packed straight into the
resident bf16 shard, each rank encoding only its own corpus rows.

What the reference does (and what this replaces):
  * `tokenizer_kw` pads every text to max_length (src/ccrec/models/item_tower.py:27-33, CCREC_MAX_LENGTH 200), or
    to the longest text of a corpus-order batch of 512 x n_gpus (scripts/al_0_rank.py:73-84): most encoder flops
    go into padding tokens.  Here texts are tokenised once without padding, sorted by token count and cut into
    batches under a token budget; a batch is padded only to its own longest text (rounded up to `pad_multiple`).
  * the pooled fp32 batch is copied to the host and vstack-ed (scripts/ms_marco_eval.py:141-149).  Here
    ccr_meanpool_pack_bf16_ex pools, optionally L2-normalises, rounds to bf16 and scatters every row to its
    corpus position inside the shard, and accumulates the max row norm the index needs -- one kernel per batch.
  * torch.nn.DataParallel splits every batch over the GPUs of one process (src/ccrec/util/data_parallel.py:8-20,
    scripts/al_0_rank.py:92).  Here one process per GPU encodes the contiguous row block dist.shard_bounds gives it
    and searches it as a shard (dist.sharded_search merges over RCCL).

Masked mean pooling skips padding positions and sums the real tokens in order, so the pooled row is independent
of the padded length whenever the encoder's hidden states for the real tokens are (BERT attention masks make them
so up to the fp32 reduction order of its GEMMs/softmax: ~1e-6 relative).
"""
import itertools
import math
import os

import numpy as np
import torch

from . import ops
from .dist import shard_bounds


def plan_batches(lengths, max_tokens=65536, max_batch=512, pad_multiple=8):
    """lengths: token count per text.  -> list of (index array, padded length): texts sorted by length (stable),
    cut so that batch_size * padded_length <= max_tokens and batch_size <= max_batch.  Every index appears once."""
    lengths = np.asarray(lengths, dtype=np.int64)
    assert lengths.ndim == 1 and (lengths.size == 0 or lengths.min() >= 1)
    order = np.argsort(lengths, kind="stable")
    batches, lo, n = [], 0, lengths.size
    while lo < n:
        hi = lo
        padded = 0
        while hi < n and hi - lo < max_batch:
            cand = int(math.ceil(int(lengths[order[hi]]) / pad_multiple) * pad_multiple)   # ascending: the newest is the longest
            if hi > lo and (hi - lo + 1) * cand > max_tokens:
                break
            padded = cand
            hi += 1
        batches.append((order[lo:hi], padded))
        lo = hi
    return batches


def _rust_backend(tokenizer, max_length):
    """A private copy of a HF fast tokenizer's Rust object with truncation set and padding off, or None.  Calling it directly
    (encode_batch_fast) skips the Python wrapper's per-text dict building: 1.5 + 1.6 s instead of 5.6 s per 65 K passages."""
    backend = getattr(tokenizer, "backend_tokenizer", None)
    if backend is None or not hasattr(backend, "to_str"):
        return None
    try:
        from tokenizers import Tokenizer
        own = Tokenizer.from_str(backend.to_str())
        own.enable_truncation(max_length=max_length)
        own.no_padding()
        return own
    except Exception:
        return None


def _tokenize_unpadded(tokenizer, texts, max_length, backend=None):
    """-> list of token-id sequences (input ids incl. special tokens), truncated to max_length, no padding."""
    if backend is not None:
        batch = backend.encode_batch_fast if hasattr(backend, "encode_batch_fast") else backend.encode_batch
        return [e.ids for e in batch(list(texts))]      # the Rust batch encoder runs on all cores with the GIL released
    kw = dict(truncation=True, padding=False, max_length=max_length, return_tensors=None)
    try:    # HF tokenizers: skip the outputs nobody reads (every extra list of 136 M Python ints per 1 M passages costs seconds)
        enc = tokenizer(list(texts), return_attention_mask=False, return_token_type_ids=False, **kw)
    except TypeError:
        enc = tokenizer(list(texts), **kw)
    ids = enc["input_ids"]
    if torch.is_tensor(ids):   # a tokenizer that always pads: strip by its attention mask
        mask = enc["attention_mask"] if "attention_mask" in enc else tokenizer(list(texts), **kw)["attention_mask"]
        return [row[m.bool()].tolist() for row, m in zip(ids, mask)]
    return ids


class _TokenizerWorkers:
    """A few tokenizer worker PROCESSES (ccrec_amd/_tokenize_worker.py, started as children over pipes; they import neither torch
    nor this package).  tokenize(texts) -> (flat int32 ids, int32 lengths); callable from several threads at once -- a call takes
    a free worker from a queue, so the round trips (pipe I/O releases the GIL) of different chunks overlap."""

    def __init__(self, n, tokenizer_json, max_length):
        import json
        import queue
        import subprocess
        import sys
        here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_tokenize_worker.py")
        env = dict(os.environ, TOKENIZERS_PARALLELISM="true")
        self.procs = [subprocess.Popen([sys.executable, "-u", here], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
                      for _ in range(n)]
        head = json.dumps({"tokenizer_json": tokenizer_json, "max_length": int(max_length)}).encode("utf-8")
        self.free = queue.Queue()
        for p in self.procs:
            p.stdin.write(len(head).to_bytes(8, "little") + head)
            p.stdin.flush()
            self.free.put(p)

    @staticmethod
    def _read(f, n):
        buf = f.read(n)
        if buf is None or len(buf) != n:
            raise RuntimeError("tokenizer worker ended unexpectedly")
        return buf

    def tokenize(self, texts):
        import pickle
        p = self.free.get()
        try:
            blob = pickle.dumps(list(texts), protocol=pickle.HIGHEST_PROTOCOL)
            p.stdin.write(len(blob).to_bytes(8, "little"))
            p.stdin.write(blob)
            p.stdin.flush()
            n = int.from_bytes(self._read(p.stdout, 8), "little")
            lengths = np.frombuffer(self._read(p.stdout, 4 * n), dtype=np.int32)
            t = int.from_bytes(self._read(p.stdout, 8), "little")
            flat = np.frombuffer(self._read(p.stdout, 4 * t), dtype=np.int32)
        except BaseException:
            p.kill()                 # a worker that failed mid-message is out of step with the protocol: never reuse it
            self.free.put(p)         # (callers waiting on the queue get it, fail on its closed pipes and raise too -- loudly, not a hang)
            raise
        self.free.put(p)
        return flat, lengths.astype(np.int64)

    def close(self):
        for p in self.procs:
            try:
                p.stdin.write((0).to_bytes(8, "little"))
                p.stdin.flush()
                p.stdin.close()
            except Exception:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()
        self.procs = []


def _flatten(token_lists):
    """list of id sequences -> (flat int64 array, lengths, start offsets): ONE pass over the Python ints of a chunk; the batch
    arrays are then cut out of the flat array with vectorised numpy indexing (no per-row Python)."""
    lengths = np.fromiter(map(len, token_lists), dtype=np.int64, count=len(token_lists))
    flat = np.fromiter(itertools.chain.from_iterable(token_lists), dtype=np.int64, count=int(lengths.sum()))
    return flat, lengths, np.cumsum(lengths) - lengths


class LengthSortedEncoder:
    """texts -> packed bf16 rows through `tower` (a NaiveItemTower: .cls_model is the HF encoder).

    tokenizer: HF-style callable; pad id from tokenizer.pad_token_id (0 if absent).
    max_tokens / max_batch: batch budget (padded tokens / texts); pad_multiple: padded lengths are rounded up to it.
    chunk_texts: the corpus is taken in chunks of this many texts -- tokenise -> sort by length -> plan -> build the padded id
    arrays of chunk c + 1 on a host thread WHILE the GPU encodes chunk c (HF fast tokenizers release the GIL inside
    their Rust batch encoder), instead of tokenising the whole shard before the first GPU batch.  Length-sorting inside a
    chunk of 64 K texts keeps the padding within a fraction of a percent of the global sort's.
    """

    def __init__(self, tower, tokenizer, max_length=None, max_tokens=65536, max_batch=512, pad_multiple=8, chunk_texts=65536,
                 host_threads=4, host_processes=0, output_step="mean_pooling", fused="auto"):
        self.tower, self.tokenizer = tower, tokenizer
        self.max_length = int(max_length if max_length is not None else os.environ.get("CCREC_MAX_LENGTH", 200))
        self.max_tokens, self.max_batch, self.pad_multiple = int(max_tokens), int(max_batch), int(pad_multiple)
        self.chunk_texts = max(1, int(chunk_texts))
        self.host_threads = max(1, int(host_threads))
        self.pad_id = int(getattr(tokenizer, "pad_token_id", 0) or 0)
        # the tower's output steps (src/ccrec/models/item_tower.py:133-147): masked mean pooling (fused pool + pack), the CLS row
        # (cls | mu | mean) or LayerNorm(CLS) (mean_layer_norm, the reference's CCREC_EMBEDDING_TYPE default); "embedding" reads the env
        if output_step == "embedding":
            output_step = os.environ["CCREC_EMBEDDING_TYPE"]
        assert output_step in ("mean_pooling", "cls", "mu", "mean", "mean_layer_norm"), output_step
        self.output_step = output_step
        self._backend = _rust_backend(tokenizer, self.max_length)
        # host_processes > 0 (HF fast tokenizers only): the chunks are tokenised in that many worker processes, so the Python-list
        # building of the tokenizer's results no longer competes for the GIL with the thread that launches the GPU kernels
        self.host_processes = int(host_processes) if self._backend is not None else 0
        self._workers = None
        self._packed_batches = False
        self.stats = {}
        # fused: run the encoder layer by layer on this library's attention / add + LayerNorm kernels (fused_bert.FusedBertEncoder)
        # instead of the torch module.  "auto": whenever the model is one the kernels cover (a BertModel with 64-wide heads ...) AND the
        # caller asks for reduced precision (an autocast context around encode(), or CCREC_FUSED_ENCODER=1: fused_bert.wanted);
        # True: required (ValueError if the model is not covered); False: always the module's own forward.
        from . import fused_bert
        self.fused = fused if fused in (True, False) else "auto"
        self._fused = fused_bert.for_model(tower.cls_model) if self.fused is not False else None
        if self._fused is not None and self.max_length > 512:
            self._fused = None          # the attention kernel holds one head's keys and values in LDS: 512 tokens at most
            if self.fused is True:
                raise ValueError(f"LengthSortedEncoder(fused=True): max_length {self.max_length} > 512 tokens")
        if self.fused is True and self._fused is None:
            raise ValueError(f"LengthSortedEncoder(fused=True): {fused_bert.unsupported_reason(tower.cls_model)}")

    def close(self):
        if self._workers is not None:
            self._workers.close()
            self._workers = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _batch_arrays(self, flat, lengths, starts, idx, padded):
        """Padded [B, padded] int64 id / mask arrays of one batch + its int32 token counts (host, pinned when a GPU is present)."""
        lens = lengths[idx]
        pin = torch.cuda.is_available()
        ids_t = torch.full((len(idx), padded), self.pad_id, dtype=torch.int64, pin_memory=pin)
        mask_t = torch.zeros((len(idx), padded), dtype=torch.int64, pin_memory=pin)
        total = int(lens.sum())
        r_of = np.repeat(np.arange(len(idx)), lens)
        c_of = np.arange(total) - np.repeat(np.cumsum(lens) - lens, lens)
        ids_t.numpy()[r_of, c_of] = flat[np.repeat(starts[idx], lens) + c_of]
        mask_t.numpy()[r_of, c_of] = 1
        lens_t = torch.empty(len(idx), dtype=torch.int32, pin_memory=pin)
        lens_t.numpy()[:] = lens
        return ids_t, mask_t, lens_t

    def _batch_arrays_packed(self, flat, lengths, starts, idx, padded):
        """One batch as a PACKED token array for the kernel forward (fused_bert.forward_packed): token ids and in-sequence positions
        [T] int64, sequence starts / token counts [B] int32 (host, pinned when a GPU is present), and the longest sequence."""
        lens = lengths[idx].astype(np.int64)
        pin = torch.cuda.is_available()
        total = int(lens.sum())
        first = np.cumsum(lens) - lens
        c_of = np.arange(total, dtype=np.int64) - np.repeat(first, lens)
        ids_t = torch.empty(total, dtype=torch.int64, pin_memory=pin)
        pos_t = torch.empty(total, dtype=torch.int64, pin_memory=pin)
        ids_t.numpy()[:] = flat[np.repeat(starts[idx], lens) + c_of]
        pos_t.numpy()[:] = c_of
        start_t = torch.empty(len(idx), dtype=torch.int32, pin_memory=pin)
        lens_t = torch.empty(len(idx), dtype=torch.int32, pin_memory=pin)
        start_t.numpy()[:] = first
        lens_t.numpy()[:] = lens
        return ids_t, pos_t, start_t, lens_t, int(lens.max())

    def _prepare(self, texts):
        """Host side of one chunk: tokenise without padding, plan length-sorted batches, build their padded arrays.
        -> (batches [(index array, ids, mask, token counts)] -- or [(index array, *_batch_arrays_packed)] for the kernel forward --,
        real tokens, padded tokens, seconds spent)."""
        import time
        t0 = time.perf_counter()
        if self._workers is not None:
            flat, lengths = self._workers.tokenize(texts)
            starts = np.cumsum(lengths) - lengths
        else:
            flat, lengths, starts = _flatten(_tokenize_unpadded(self.tokenizer, texts, self.max_length, self._backend))
        plan = plan_batches(lengths, self.max_tokens, self.max_batch, self.pad_multiple) if lengths.size else []
        build = self._batch_arrays_packed if self._packed_batches else self._batch_arrays
        batches = [(idx, *build(flat, lengths, starts, idx, padded)) for idx, padded in plan]
        return batches, int(lengths.sum()), int(sum(len(idx) * pl for idx, pl in plan)), time.perf_counter() - t0

    @torch.no_grad()
    def encode(self, texts, sim="dot", out=None, row_offset=0, norm_bounds=None, out_f32=None):
        """Encode `texts` into rows [row_offset, row_offset + len(texts)) of `out` (bf16 [>=N, dim] cuda; allocated if
        None).  sim "cos" L2-normalises before rounding.  norm_bounds: fp32 [>=N] tensor indexed like `out`'s rows, see
        ops.pack_bf16.  out_f32: optional fp32
        [>=N, dim] tensor that also receives the un-rounded pooled rows (tests / the cls|mean_layer_norm consumers).
        Returns the bf16 tensor.  self.stats afterwards: texts, batches, real / padded tokens, host seconds of the
        tokenise + plan thread, seconds the GPU loop waited for it, GPU seconds of the batches (events) and the wall time."""
        import time
        from concurrent.futures import ThreadPoolExecutor
        ops.require_gpu()
        tower = self.tower
        tower.eval()
        device = tower.cls_model.device
        from . import fused_bert
        half = fused_bert.kernel_dtype(self.fused) if self._fused is not None else None   # the caller's autocast type (fp16 under the reference's autocast())
        fused = self._fused if half is not None else None
        if fused is not None:
            fused.refresh(half)     # 16-bit weight copies follow the module (fine-tuning between two ranking steps)
        self._packed_batches = fused is not None     # the kernel forward takes packed token arrays: no padding row is ever computed
        texts = texts if isinstance(texts, (list, tuple)) else list(texts)
        n, chunk = len(texts), self.chunk_texts
        st = {"texts": n, "batches": 0, "real_tokens": 0, "padded_tokens": 0, "fixed_length_tokens": n * self.max_length,
              "chunks": (n + chunk - 1) // chunk, "host_prepare_s": 0.0, "gpu_wait_for_host_s": 0.0, "gpu_busy_s": 0.0,
              "fused_layers": fused is not None, "layer_dtype": None if half is None else str(half).replace("torch.", "")}
        wall0 = time.perf_counter()
        spans = []          # (start, end) events of every chunk's GPU work
        if self.host_processes > 0 and self._workers is None and n > chunk:
            self._workers = _TokenizerWorkers(self.host_processes, self._backend.to_str(), self.max_length)
        ahead = self.host_threads     # chunks being prepared while one encodes (the Rust tokenizer and numpy release the GIL)
        with ThreadPoolExecutor(max_workers=ahead) as pool:
            # chunk boundaries: full chunks, except a ramp at the start of a long corpus (1/8, 1/4, 1/2 of a chunk) so that the
            # first GPU batch waits for the tokenisation of 8 K texts, not of 64 K
            bounds, c = [], 0
            ramp = [chunk // 8, chunk // 4, chunk // 2] if n > 2 * chunk and chunk >= 1024 else []
            while c < n:
                size = ramp.pop(0) if ramp else chunk
                bounds.append((c, min(n, c + size)))
                c += size
            st["chunks"] = len(bounds)
            starts = [lo for lo, _ in bounds]
            futs = [pool.submit(self._prepare, texts[lo:hi]) for lo, hi in bounds[:ahead]]
            for ci, c0 in enumerate(starts):
                t0 = time.perf_counter()
                batches, real, padded_tokens, secs = futs[ci].result()
                futs[ci] = None
                st["gpu_wait_for_host_s"] += time.perf_counter() - t0
                if ci == 0:
                    st["first_chunk_wait_s"] = time.perf_counter() - t0
                if ci + ahead < len(starts):   # keep `ahead` chunks in preparation while this one encodes
                    lo, hi = bounds[ci + ahead]
                    futs.append(pool.submit(self._prepare, texts[lo:hi]))
                st["batches"] += len(batches)
                st["real_tokens"] += real
                st["padded_tokens"] += padded_tokens
                st["host_prepare_s"] += secs
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for idx, *arrays in batches:
                    rows = torch.as_tensor(np.asarray(idx, dtype=np.int64) + int(row_offset) + c0)
                    if fused is not None:
                        ids_t, pos_t, start_t, lens_t, longest = arrays
                        seq_start, seq_len = start_t.to(device, non_blocking=True), lens_t.to(device, non_blocking=True)
                        hidden = fused.forward_packed(ids_t.to(device, non_blocking=True), pos_t.to(device, non_blocking=True),
                                                      seq_start, seq_len, longest,                       # [T, dim] fp32, or the CLS rows [B, dim]
                                                      cls_only=self.output_step != "mean_pooling", dtype=half)
                        if out is None:
                            out = torch.empty(n + row_offset, hidden.shape[-1], dtype=torch.bfloat16, device=hidden.device)
                        if self.output_step == "mean_pooling":
                            ops.meanpool_pack_packed(hidden, seq_start, seq_len, normalize=(sim == "cos"), out_bf16=out, out_f32=out_f32,
                                                     dst_rows=rows, norm_bounds=norm_bounds)
                            continue
                        first = hidden                                                                     # the CLS rows
                        mask = torch.ones(len(idx), 1, dtype=torch.int64, device=device)
                    else:
                        ids_t, mask_t, lens_t = arrays
                        inputs = {"input_ids": ids_t.to(device, non_blocking=True), "attention_mask": mask_t.to(device, non_blocking=True)}
                        hidden = tower.cls_model(**inputs).last_hidden_state
                        if out is None:
                            out = torch.empty(n + row_offset, hidden.shape[-1], dtype=torch.bfloat16, device=hidden.device)
                        mask = inputs["attention_mask"]
                        first = hidden[:, 0]
                    if self.output_step != "mean_pooling":
                        # CLS-type outputs through the same scatter + pack kernel: a one-token "sequence" [B, 1, dim] with a mask
                        # of ones pools to the row itself (sum of one element / 1)
                        if self.output_step == "mean_layer_norm":
                            first = tower.standard_layer_norm(first)
                        hidden, mask = first.unsqueeze(1).contiguous(), mask[:, :1]
                    ops.meanpool_pack(hidden, mask, normalize=(sim == "cos"), want_f32=False, out_bf16=out,
                                      out_f32=out_f32, dst_rows=rows, norm_bounds=norm_bounds)
                e1.record()
                spans.append((e0, e1))
        if out is None:   # no texts
            out = torch.empty(row_offset, 0, dtype=torch.bfloat16, device=device)
        if spans:
            spans[-1][1].synchronize()
            st["gpu_busy_s"] = sum(a.elapsed_time(b) for a, b in spans) * 1e-3
        st["wall_s"] = time.perf_counter() - wall0
        self.stats = st
        return out


def encode_local_shard(encoder, corpus_ids, corpus, sim, rank, world, norm_bounds=None, weights=None, cuts=None):
    """This rank's contiguous block of the corpus (dist.shard_bounds; weights: equal-weight blocks; cuts: the world + 1 boundaries every
    rank agreed on -- dist.agreed_cuts -- take precedence) -> (bf16 shard [hi-lo, dim], lo, hi)."""
    lo, hi = (int(cuts[rank]), int(cuts[rank + 1])) if cuts is not None else shard_bounds(len(corpus_ids), world, rank, weights)
    shard = encoder.encode([corpus[c] for c in corpus_ids[lo:hi]], sim=sim, norm_bounds=norm_bounds)
    return shard, lo, hi


def token_weights(texts, max_length, tokens_per_word=1.3, special_tokens=2, tokenizer=None, sample=512):
    """A per-text estimate of the encoder's token count that every rank can compute for the WHOLE corpus in a second or two (no
    tokeniser run over it): words x tokens_per_word + the special tokens, clipped to max_length -- the quantity the encode time of a
    passage is proportional to.  (Exact counts exist only after tokenisation, which every rank does for its own rows only.)
    tokenizer: the two constants are fitted on `sample` evenly spaced texts run through it (the same texts on every rank, so the same
    weights on every rank)."""
    import numpy as np
    words = np.fromiter((t.count(" ") + 1 if t else 0 for t in texts), dtype=np.float64, count=len(texts))
    if tokenizer is not None and len(texts) >= 2:
        idx = np.unique(np.linspace(0, len(texts) - 1, min(len(texts), int(sample))).astype(np.int64))
        try:
            ids = tokenizer([texts[i] for i in idx], truncation=True, padding=False, max_length=int(max_length))["input_ids"]
            toks = np.array([len(r) for r in ids], np.float64)
        except Exception:                               # a tokeniser without this call form: keep the default constants
            toks = None
        if toks is not None:
            free = toks < max_length                    # truncated texts say nothing about the slope
            if free.sum() >= 2 and np.ptp(words[idx][free]) > 0:
                tokens_per_word, special_tokens = np.polyfit(words[idx][free], toks[free], 1)
    return np.clip(words * tokens_per_word + special_tokens, 1.0, float(max_length))


def ranking_sharded(corpus, queries, encoder, block_dict=None, rank=0, world=1, group=None, keep=None, with_tensors=False, lazy=False,
                    balance="tokens"):
    """Multi-GPU form of ms_marco_eval.ranking (scripts/ms_marco_eval.py:189-235): every rank encodes and indexes its
    own corpus rows, all ranks encode the (small) query set, per-shard fused top-k, one all-gather, merge.
    Returns the same rank-ordered {qid: {pid: score}} on every rank.  block_dict: each shard scores its own blocked rows
    -1e6 (ccr_search_blocked, lists of any length) before the exchange, so the merged list is the reference's.
    with_tensors: return (profile, row ids [Q, keep] int64, scores [Q, keep]) -- the device tensors behind the dicts.
    lazy: the profile is a ranking_profile.RankingProfile over those tensors (inner dicts built when a query is read).
    balance: how the corpus rows are cut into the ranks' contiguous blocks.  "tokens" (default): equal ESTIMATED TOKENS per rank
    (token_weights: a x words + b, clipped to max_length, a and b fitted on 512 evenly spaced texts run through the tokeniser; the encode is ~97 % of the step and its time follows the tokens, so a corpus whose passage length follows the row
    order would otherwise leave the step waiting for its slowest rank; the reference's DataParallel splits every batch evenly,
    scripts/al_0_rank.py:70-74,92); "rows": equal row counts; or one weight per corpus row.  The search only needs each shard's
    global_row_offset, so the result does not depend on the cut."""
    from .ms_marco_eval import KEEP, Retriever
    from .dist import sharded_search
    queries_ids, corpus_ids = list(queries.keys()), list(corpus.keys())
    sim = "cos" if os.environ["CCREC_SIM_TYPE"] == "cos" else "dot"
    keep = KEEP if keep is None else keep
    q_bf16 = encoder.encode([queries[q] for q in queries_ids], sim=sim)
    weights = None
    if world > 1 and not (isinstance(balance, str) and balance == "rows"):
        weights = (token_weights([corpus[c] for c in corpus_ids], encoder.max_length, tokenizer=encoder.tokenizer)
                   if isinstance(balance, str) else balance)
    # ONE set of boundaries for all ranks: rank 0's, broadcast as int64 (weights that differ by an ulp between ranks would otherwise move
    # a boundary on one rank only: rows dropped or encoded twice).  The short lists are sized from the largest row share of these cuts.
    from .dist import agreed_cuts, largest_share, resume_short_lists
    cuts = agreed_cuts(len(corpus_ids), world, weights, group=group, device=q_bf16.device if q_bf16.is_cuda else None)
    lo0, hi0 = cuts[rank], cuts[rank + 1]
    bounds = torch.empty(max(hi0 - lo0, 1), dtype=torch.float32, device=q_bf16.device)   # norm bound of every packed row
    shard, lo, hi = encode_local_shard(encoder, corpus_ids, corpus, sim, rank, world, norm_bounds=bounds, cuts=cuts)
    bounds = bounds if hi > lo else None
    if world == 1:
        return Retriever(corpus_ids, shard, norm_bounds=bounds).ranking_profile(queries_ids, q_bf16, block_dict, keep, with_tensors, lazy)
    index = ops.CorpusIndex(shard, global_row_offset=lo, norm_bounds=bounds)
    resume_short_lists()     # a new corpus: a suspension earned on another corpus's row order does not carry over (same call on every rank)
    n = len(corpus_ids)
    block = None
    if block_dict is not None:   # every rank passes the whole lists; a shard applies the ids that fall inside it
        print("using block_dict")
        from .ms_marco_eval import block_csr
        pos = {pid: i for i, pid in enumerate(corpus_ids)}
        lists = []
        for qid in queries_ids:
            rows = [pos.get(pid, -1) for pid in block_dict[qid]]
            assert -1 not in rows, "block id not found"
            lists.append(rows)
        block = block_csr(lists, n)
    scores, ids = sharded_search(index, q_bf16, min(n, keep), group=group, block=block, n_total=n, share=largest_share(cuts))
    from .ranking_profile import RankingProfile
    profile = RankingProfile(queries_ids, corpus_ids, ids, scores)
    if not lazy:
        profile = profile.to_dict()
    return (profile, ids, scores) if with_tensors else profile
