"""The item tower's encoder forward, layer by layer on this library's kernels (encoder-side fusion).

The reference encodes with transformers' BertModel under autocast (src/ccrec/models/item_tower.py:122
`cls_model(**inputs).last_hidden_state`; scripts/al_0_rank.py:92-101,125).  Run as torch modules on one MI355X that forward
spends 38 % of its GPU time in the projections (hipBLASLt, already at library speed), 22 % in the attention call (57 TFLOP/s at
these sequence lengths) and 30 % in separate residual-add / LayerNorm / dtype-cast passes (profiles/r03_encode_kernel_stats.csv).
FusedBertEncoder keeps the projections as library GEMMs (torch.nn.functional.linear on 16-bit weights: one stacked Q|K|V
projection instead of three) and replaces the rest with ccr_attention_half and ccr_add_layernorm_half (csrc/ccr_encoder.hip).

Arithmetic: what autocast does in the reference's layer -- 16-bit projection operands and outputs, fp32 attention scores and
softmax, fp32 residual stream and LayerNorm -- IN THE AUTOCAST CONTEXT'S OWN 16-BIT TYPE (kernel_dtype): fp16 under the reference's
`torch.cuda.amp.autocast()` (scripts/al_0_rank.py:8,125: the CUDA default), bf16 under autocast(dtype=torch.bfloat16).  The hidden
states agree with the module forward under the same autocast to that type's rounding (the tests compare both with the fp32 forward).
Inference only (eval mode, no dropout, no autograd).

Only what the kernels cover is accepted (unsupported_reason): a BertModel or DistilBertModel encoder (post-LayerNorm layers, absolute
positions, exact GELU), head width 64, hidden size a multiple of 256, at most 512 tokens, right-padded batches.  Every other encoder keeps
running as its own torch module -- LengthSortedEncoder picks per model."""
import os
import threading

import torch
import torch.nn.functional as F

from . import ops


class _Arch:
    """Where a supported architecture keeps the pieces of its (post-LayerNorm, GELU) encoder layer."""

    def __init__(self, heads, hidden, activation, stack, embeddings, type_table, layer_parts):
        self.heads, self.hidden, self.activation = heads, hidden, activation
        self.stack = stack                    # the module whose parameters are the layers' weights
        self.embeddings, self.type_table = embeddings, type_table
        self.layer_parts = layer_parts        # layer module -> (q, k, v, attention out, LayerNorm 1, ffn in, ffn out, LayerNorm 2)


def _describe(model):
    """-> _Arch for a transformers BertModel or DistilBertModel (the reference's encoders: facebook/contriever and bert-base-uncased are
    BertModels, src/ccrec/models/bbpr.py:334, scripts/al_0_rank.py:120; distilbert-base-uncased is its default model_name,
    bbpr.py:50, bert_mt.py:35), or a string saying why the class is not covered."""
    name, cfg = type(model).__name__, getattr(model, "config", None)
    if name == "BertModel" and hasattr(model, "embeddings") and hasattr(model, "encoder"):
        if getattr(cfg, "position_embedding_type", None) not in (None, "absolute"):
            return f"position_embedding_type {cfg.position_embedding_type!r}"
        if getattr(cfg, "is_decoder", False) or getattr(cfg, "add_cross_attention", False):
            return "decoder / cross-attention layers"
        e = model.embeddings
        return _Arch(int(cfg.num_attention_heads), int(cfg.hidden_size), getattr(cfg, "hidden_act", "gelu"), model.encoder, e,
                     e.token_type_embeddings.weight,
                     lambda m: (m.attention.self.query, m.attention.self.key, m.attention.self.value, m.attention.output.dense,
                                m.attention.output.LayerNorm, m.intermediate.dense, m.output.dense, m.output.LayerNorm))
    if name == "DistilBertModel" and hasattr(model, "embeddings") and hasattr(model, "transformer"):
        return _Arch(int(cfg.n_heads), int(cfg.dim), getattr(cfg, "activation", "gelu"), model.transformer, model.embeddings, None,
                     lambda m: (m.attention.q_lin, m.attention.k_lin, m.attention.v_lin, m.attention.out_lin, m.sa_layer_norm,
                                m.ffn.lin1, m.ffn.lin2, m.output_layer_norm))
    return f"{name} is not a BertModel or DistilBertModel"


def unsupported_reason(model):
    """None when FusedBertEncoder can run `model` (a transformers BertModel or DistilBertModel); otherwise why not."""
    arch = _describe(model)
    if isinstance(arch, str):
        return arch
    heads, hidden = arch.heads, arch.hidden
    if hidden % heads or hidden // heads != 64:
        return f"head width {hidden / heads:g} (the attention kernel is built for 64)"
    if hidden % 256 or hidden > 2048:
        return f"hidden size {hidden} (the LayerNorm kernel takes multiples of 256 up to 2048)"
    if arch.activation != "gelu":
        return f"activation {arch.activation!r} (exact GELU only)"
    return None


_ENV_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16, "half": torch.float16}


def kernel_dtype(explicit="auto"):
    """The 16-bit operand type an inference forward should run the layer kernels in, or None = run the torch module.

    Inside a CUDA autocast context it is the context's own type (torch.get_autocast_dtype): torch.float16 under the reference's
    `with torch.cuda.amp.autocast():` (scripts/al_0_rank.py:125), torch.bfloat16 under autocast(dtype=torch.bfloat16); any other
    autocast type keeps the module.  Outside autocast the caller asked for an fp32 forward and gets the fp32 module -- unless
    explicit is True or CCREC_FUSED_ENCODER=1 demands the kernels, which then run in CCREC_FUSED_ENCODER_DTYPE (bf16 | fp16, default bf16).
    explicit False or CCREC_FUSED_ENCODER=0: always None."""
    env = os.environ.get("CCREC_FUSED_ENCODER", "").strip()
    if explicit is False or (explicit is not True and env == "0"):
        return None
    if torch.cuda.is_available() and torch.is_autocast_enabled("cuda"):
        dtype = torch.get_autocast_dtype("cuda")
        return dtype if dtype in (torch.float16, torch.bfloat16) else None
    if explicit is True or env == "1":
        name = os.environ.get("CCREC_FUSED_ENCODER_DTYPE", "bf16").strip().lower()
        assert name in _ENV_DTYPES, f"CCREC_FUSED_ENCODER_DTYPE={name!r}: bf16 or fp16"
        return _ENV_DTYPES[name]
    return None


def wanted(explicit="auto"):
    """Should an inference forward take the kernel path?  (kernel_dtype says in which 16-bit type.)"""
    return kernel_dtype(explicit) is not None


_SLOT = "_ccr_fused_encoder"        # the encoder lives in its model's __dict__: model <-> encoder is an ordinary cycle the GC collects
_SLOT_LOCK = threading.Lock()      # DataParallel runs its replicas' forwards on threads


def for_model(model):
    """The FusedBertEncoder of `model` (one per module object, built on first use), or None if the kernels do not cover it.
    torch's replicate() copies a module's __dict__ shallowly, so a DataParallel replica arrives holding its ORIGINAL's encoder:
    an encoder that belongs to another module object is replaced by one of the replica's own (its weights sit on the replica's device)."""
    if not isinstance(model, torch.nn.Module):      # a stand-in encoder (tests, wrappers): never covered
        return None
    with _SLOT_LOCK:
        enc = model.__dict__.get(_SLOT, False)
        if enc is False or (enc is not None and enc.model is not model):
            enc = FusedBertEncoder(model) if unsupported_reason(model) is None else None
            model.__dict__[_SLOT] = enc
        return enc


def prefix_lengths(attention_mask):
    """int32 [B] token counts if every row of the 0/1 mask is ones followed by zeros with at least one token (right padding: what
    the tokenizers of the reference's models produce); None otherwise.  One small reduction + a host read per batch."""
    m = attention_mask
    if m.dim() != 2 or m.shape[1] == 0 or m.shape[1] > 512:
        return None
    lengths = m.sum(dim=1)
    L = m.shape[1]
    ok = ((m != 0) == (torch.arange(L, device=m.device)[None, :] < lengths[:, None])).all() & (lengths > 0).all()
    return lengths.to(torch.int32) if bool(ok) else None


class _Layer:
    __slots__ = ("wqkv", "bqkv", "wo", "bo", "g1", "b1", "eps1", "wi", "bi", "wo2", "bo2", "g2", "b2", "eps2")


class FusedBertEncoder:
    """16-bit (bf16 or fp16: the `dtype` of each call) forward of a transformers BertModel / DistilBertModel: forward(input_ids [B, L]
    right-padded, lengths [B]) -> last hidden state fp32 [B, L, hidden]; forward_packed(...) over a packed token array.  Holds 16-bit
    copies of the projection weights, one set per type used (rebuilt when the model's parameters change: fine-tuning between two
    ranking steps, load_state_dict, .to(device))."""

    def __init__(self, model):
        reason = unsupported_reason(model)
        if reason is not None:
            raise ValueError(f"FusedBertEncoder: {reason}")
        self.model = model
        arch = _describe(model)
        self.heads, self.hidden = arch.heads, arch.hidden
        self.has_token_types = arch.type_table is not None
        self._layers, self._signature, self._no_types = {}, None, None

    def __getstate__(self):      # pickled with its model (torch.save(model)): without the 16-bit weight copies
        return {"model": self.model, "heads": self.heads, "hidden": self.hidden, "has_token_types": self.has_token_types}

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._layers, self._signature, self._no_types = {}, None, None

    def _params_signature(self):
        return tuple((p.data_ptr(), p._version, p.device) for p in _describe(self.model).stack.parameters())

    def refresh(self, dtype=None):
        """Drop the 16-bit weight copies if the module's parameters changed since they were made (-> True), and build the set of
        `dtype` (torch.bfloat16 / torch.float16) if one is named and missing."""
        sig = self._params_signature()
        changed = sig != self._signature
        if changed:
            self._layers, self._signature = {}, sig
        if dtype is not None and dtype not in self._layers:
            self._layers[dtype] = self._build_layers(dtype)
        return changed

    def _build_layers(self, half):
        assert half in (torch.bfloat16, torch.float16), half
        layers = []
        with torch.no_grad():
            arch = _describe(self.model)
            for mod in arch.stack.layer:
                q, k, v, so, ln1, ff, out, ln2 = arch.layer_parts(mod)
                l = _Layer()
                l.wqkv = torch.cat([q.weight, k.weight, v.weight]).to(half).contiguous()
                l.bqkv = torch.cat([q.bias, k.bias, v.bias]).to(half).contiguous()
                l.wo, l.bo = so.weight.to(half).contiguous(), so.bias.to(half).contiguous()
                l.g1, l.b1, l.eps1 = ln1.weight.float().contiguous(), ln1.bias.float().contiguous(), ln1.eps
                l.wi, l.bi = ff.weight.to(half).contiguous(), ff.bias.to(half).contiguous()
                l.wo2, l.bo2 = out.weight.to(half).contiguous(), out.bias.to(half).contiguous()
                l.g2, l.b2, l.eps2 = ln2.weight.float().contiguous(), ln2.bias.float().contiguous(), ln2.eps
                layers.append(l)
        return layers

    def _embed(self, token_ids, positions, token_types, dtype):
        """The embedding block (BertEmbeddings.forward: (word + type) + position, LayerNorm; DistilBERT's Embeddings: word + position,
        LayerNorm) over flat int64 index vectors [T] -> (fp32 [T, hidden], its copy in `dtype`): one kernel on the module's own fp32
        tables (ccr_embed_layernorm; an all-zero type row stands in where the architecture has no token types), or the same sum
        in torch when the tables are not plain fp32 (a half-precision checkpoint)."""
        arch = _describe(self.model)
        e = arch.embeddings
        token_ids, positions = token_ids.long(), positions.long()          # (tokenizers may hand over int32 ids)
        token_types = None if token_types is None else token_types.long()
        word, pos, ln = e.word_embeddings.weight, e.position_embeddings.weight, e.LayerNorm
        types = arch.type_table
        if types is None:
            assert token_types is None, "this architecture has no token types"
            if self._no_types is None or self._no_types.device != word.device:
                self._no_types = torch.zeros(1, self.hidden, dtype=torch.float32, device=word.device)
            types = self._no_types
        tables = (word, pos, types, ln.weight, ln.bias)
        if all(t.dtype == torch.float32 and t.is_contiguous() for t in tables):
            return ops.embed_layernorm(word, pos, types, token_ids.contiguous(), positions.contiguous(),
                                       None if token_types is None else token_types.contiguous(), ln.weight, ln.bias, ln.eps, dtype=dtype)
        x = word[token_ids].float() + types[token_types if token_types is not None else torch.zeros_like(token_ids)].float()
        h = F.layer_norm(x + pos[positions].float(), (self.hidden,), ln.weight.float(), ln.bias.float(), ln.eps).contiguous()
        return h, h.to(dtype)

    def _layers_forward(self, h, hb, seq_start, lengths, max_len, pad_len, cls_rows=None):
        """The encoder layers over token rows h [T, hidden] fp32 / hb (its 16-bit copy, whose dtype selects the weight set); sequence s = rows seq_start[s] .. -> last
        hidden state [T, hidden] fp32.

        cls_rows ([n_seq] int64 rows of the sequences' first tokens): the caller reads only those rows of the last hidden state (the
        tower's cls | mu | mean | mean_layer_norm output steps, src/ccrec/models/item_tower.py:133-136 -- mean_layer_norm is the
        reference's CCREC_EMBEDDING_TYPE default).  The LAST layer then needs every token's keys and values but only the first
        tokens' attention output, so its output projection, both LayerNorms, the FFN and the GELU run on n_seq rows instead of T
        (same values: every one of those operations is row-wise) -> [n_seq, hidden]."""
        layers = self._layers[hb.dtype]
        last = len(layers) - 1
        for i, l in enumerate(layers):
            qkv = F.linear(hb, l.wqkv, l.bqkv)
            ctx = ops.attention(qkv, seq_start, lengths, self.heads, max_len=max_len, pad_len=pad_len, scale=0.125)
            if i == last and cls_rows is not None:
                ctx, h = ctx[cls_rows].contiguous(), h[cls_rows].contiguous()
            h, hb = ops.add_layernorm(F.linear(ctx, l.wo, l.bo), h, l.g1, l.b1, l.eps1)
            mid = ops.gelu_(F.linear(hb, l.wi, l.bi))      # exact GELU, in place (torch's bits)
            h, hb = ops.add_layernorm(F.linear(mid, l.wo2, l.bo2), h, l.g2, l.b2, l.eps2, want_bf16=i != last)
        if last < 0 and cls_rows is not None:
            h = h[cls_rows].contiguous()
        return h

    @torch.no_grad()
    def forward_packed(self, token_ids, positions, seq_start, lengths, max_len, token_type_ids=None, cls_only=False, dtype=torch.bfloat16):
        """The forward over a packed token array that the caller built itself (no padded batch ever exists): token_ids / positions
        [T] int64 (position of each token inside its sequence), seq_start / lengths [n_seq] int32 (cuda), max_len = the longest
        sequence (host int).  -> fp32 [T, hidden]; pool it with ops.meanpool_pack_packed.  cls_only: -> [n_seq, hidden], the last
        hidden state of every sequence's first token only (_layers_forward).  dtype: the layer's 16-bit operand type (kernel_dtype())."""
        ops.require_gpu()
        assert not self.model.training, "FusedBertEncoder is an inference forward: call model.eval() first"
        assert token_ids.is_cuda and token_ids.dim() == 1 and positions.shape == token_ids.shape and 1 <= int(max_len) <= 512
        if dtype not in self._layers:      # (the per-batch caller, LengthSortedEncoder.encode, refreshes once per corpus)
            self.refresh(dtype)
        with torch.autocast("cuda", enabled=False):
            h, hb = self._embed(token_ids, positions, token_type_ids, dtype)
            return self._layers_forward(h, hb, seq_start, lengths, int(max_len), 0, seq_start.long() if cls_only else None)

    @torch.no_grad()
    def forward(self, input_ids, lengths, token_type_ids=None, packed=None, lengths_host=None, cls_only=False, dtype=torch.bfloat16):
        """input_ids [B, L] int64 (cuda, right-padded), lengths [B] int32 (cuda): real tokens per row, 1 .. L.
        -> fp32 [B, L, hidden]; rows of padding tokens hold finite values nobody reads (the pooling masks them).

        packed: run the layers on the REAL tokens only (a packed token array: the projections, LayerNorms and GELU skip the
        padding, the attention kernel takes row offsets + lengths) and scatter the result back into a zero-filled [B, L, hidden].
        That is what makes the reference-style batches cheap -- corpus-order batches padded to their longest text
        (scripts/al_0_rank.py:76-81) or to max_length (src/ccrec/models/item_tower.py:27-33) are 30-90 % padding.  None: packed
        when more than a tenth of the batch is padding (needs the lengths on the host: `lengths_host`, or one device read).
        cls_only: the caller reads only hidden[:, 0] -> fp32 [B, 1, hidden] (the last layer runs on the first tokens' rows only).
        dtype: the layer's 16-bit operand type (kernel_dtype(): the caller's autocast type)."""
        ops.require_gpu()
        model = self.model
        assert not model.training, "FusedBertEncoder is an inference forward: call model.eval() first"
        assert input_ids.is_cuda and input_ids.dim() == 2
        B, L = input_ids.shape
        assert L <= 512 and lengths.dtype == torch.int32 and lengths.is_cuda and lengths.numel() == B
        if dtype not in self._layers:
            self.refresh(dtype)
        dev = input_ids.device
        if packed is None or packed:
            lens_h = lengths.cpu() if lengths_host is None else torch.as_tensor(lengths_host)
            total, longest = int(lens_h.sum()), int(lens_h.max()) if B else 0
            if packed is None:
                packed = total < 0.9 * B * L
        with torch.autocast("cuda", enabled=False):
            if packed:
                keep = (torch.arange(L, device=dev)[None, :] < lengths[:, None]).flatten().nonzero().squeeze(1)   # rows of the real tokens
                assert keep.numel() == total, "lengths_host does not match lengths"
                h, hb = self._embed(input_ids.flatten()[keep], keep % L, None if token_type_ids is None else token_type_ids.flatten()[keep], dtype)
                max_len, pad_len = max(longest, 1), 0
                seq_start = (torch.cumsum(lengths, 0, dtype=torch.int32) - lengths).contiguous()
            else:
                h, hb = self._embed(input_ids.flatten(), torch.arange(L, device=dev).repeat(B),
                                    None if token_type_ids is None else token_type_ids.flatten(), dtype)
                max_len, pad_len = L, L
                seq_start = torch.arange(B, dtype=torch.int32, device=dev) * L
            h = self._layers_forward(h, hb, seq_start, lengths, max_len, pad_len, seq_start.long() if cls_only else None)
            if cls_only:
                return h.view(B, 1, self.hidden)
            if packed:
                full = torch.zeros(B * L, self.hidden, dtype=torch.float32, device=dev)
                full[keep] = h
                h = full
        return h.view(B, L, self.hidden)
