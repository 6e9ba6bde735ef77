"""SURVEY 8 f2, encoder-side fusion: the attention and add + LayerNorm kernels of the item tower's encoder layer
(csrc/ccr_encoder.hip) and the layer-by-layer forward built on them (ccrec_amd/fused_bert.py).

These are floating-point kernels, so the reference is plain PyTorch in fp32 on the same (16-bit-rounded) operands, with the
tolerance written in each test: two ulps of the 16-bit type for the attention output (probabilities are rounded to it for the P.V
product, the output is rounded once more), fp32 rounding for LayerNorm, and for the whole encoder "no further from the fp32
module forward than the module's own forward under the same autocast is" (src/ccrec/models/item_tower.py:122,
scripts/al_0_rank.py:125).  Both 16-bit types run: fp16 is what the reference's `torch.cuda.amp.autocast()` means, bf16 is the
type of autocast(dtype=torch.bfloat16) (tests/test_gpu_encoder_fp16.py holds the fp16 tower / top-k / golden checks)."""

import numpy as np
import pytest
import torch

from conftest import ROOT, PKG  # noqa: F401

pytestmark = pytest.mark.gpu

# |kernel - fp32 reference| of the attention output, per 16-bit type: atol, rtol = two ulps of the value + the rounding of the
# probabilities (relative 2^-9 for bf16, 2^-11 for fp16, summed with weights <= 1)
ATT_TOL = {torch.bfloat16: (1.5e-2, 1.6e-2), torch.float16: (2e-3, 2e-3)}
HALVES = [torch.bfloat16, torch.float16]


def _attention_reference(qkv, starts, lens, H):
    """fp32 softmax(Q K^T / 8) V per (sequence, head) on the bf16 operands."""
    T = qkv.shape[0]
    out = torch.zeros(T, H * 64, dtype=torch.float32, device=qkv.device)
    x = qkv.float().view(T, 3, H, 64)
    for s0, n in zip(starts, lens):
        q, k, v = (x[s0:s0 + n, i].transpose(0, 1) for i in range(3))          # [H, n, 64]
        p = torch.softmax(q @ k.transpose(1, 2) * 0.125, dim=-1)
        out[s0:s0 + n] = (p @ v).transpose(0, 1).reshape(n, H * 64)
    return out


@pytest.mark.parametrize("lens,H,padded", [
    ([1, 2, 31, 32, 33], 2, True),
    ([63, 64, 65, 127, 128, 129], 3, True),
    ([200, 17, 136, 136, 5], 12, True),
    ([512, 300, 257, 1], 2, True),
    ([1, 2, 31, 32, 33, 64, 65, 100], 4, False),
    ([129, 255, 256, 384], 2, False),
])
@pytest.mark.parametrize("half", HALVES)
def test_attention_matches_fp32_reference(lens, H, padded, half):
    from ccrec_amd import ops
    atol, rtol = ATT_TOL[half]
    torch.manual_seed(sum(lens) + H)
    L = max(lens)
    if padded:
        pad = (L + 7) // 8 * 8
        starts = [i * pad for i in range(len(lens))]
        T = pad * len(lens)
    else:
        pad = 0
        starts = list(np.cumsum([0] + lens[:-1]))
        T = sum(lens)
    # scores with a real spread (softmax far from uniform), values O(1); padding rows hold NaN to prove nothing reads them as keys
    qkv = (torch.randn(T, 3 * H * 64, device="cuda") * 1.5).to(half)
    if padded:
        live = torch.zeros(T, dtype=torch.bool, device="cuda")
        for s0, n in zip(starts, lens):
            live[s0:s0 + n] = True
        qkv[~live] = float("nan")
    seq_start = torch.tensor(starts, dtype=torch.int32, device="cuda")
    seq_len = torch.tensor(lens, dtype=torch.int32, device="cuda")
    out = torch.full((T, H * 64), 7.0, dtype=half, device="cuda")
    ops.attention(qkv, seq_start, seq_len, H, max_len=L, pad_len=pad, out=out)
    ref = _attention_reference(torch.nan_to_num(qkv.float()).to(half), starts, lens, H)
    got = out.float()
    assert torch.isfinite(got).all()
    for s0, n in zip(starts, lens):
        torch.testing.assert_close(got[s0:s0 + n], ref[s0:s0 + n], atol=atol, rtol=rtol)
        if padded:
            assert (got[s0 + n:s0 + pad] == 0).all(), "padding rows must be zeros"
    err = (got - ref).abs().max().item()
    assert err < (5e-2 if half == torch.bfloat16 else 8e-3)


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("half", HALVES)
def test_attention_fuzz_against_fp32_reference(seed, half):
    """Random batches: 1-40 sequences of 1-512 tokens, 1 / 2 / 12 / 16 heads, right-padded (pad_len >= the longest, also beyond the
    512-token kernel limit's neighbourhood) or packed; same tolerance as above."""
    from ccrec_amd import ops
    rs = np.random.RandomState(1000 + seed)
    H = int(rs.choice([1, 2, 12, 16]))
    n = int(rs.randint(1, 41))
    top = int(rs.choice([8, 33, 70, 130, 200, 260, 512]))
    lens = [int(v) for v in rs.randint(1, top + 1, n)]
    L = max(lens)
    padded = bool(rs.randint(0, 2))
    if padded:
        pad = min(512, L + int(rs.randint(0, 20)))
        starts = [i * pad for i in range(n)]
        T = pad * n
    else:
        pad = 0
        starts = [int(v) for v in np.cumsum([0] + lens[:-1])]
        T = sum(lens)
    torch.manual_seed(seed)
    qkv = (torch.randn(T, 3 * H * 64, device="cuda") * float(rs.choice([0.5, 1.5, 3.0]))).to(half)
    out = torch.full((T, H * 64), -3.0, dtype=half, device="cuda")
    ops.attention(qkv, torch.tensor(starts, dtype=torch.int32, device="cuda"), torch.tensor(lens, dtype=torch.int32, device="cuda"),
                  H, max_len=L, pad_len=pad, out=out)
    ref = _attention_reference(qkv, starts, lens, H)
    got = out.float()
    for s0, m in zip(starts, lens):
        torch.testing.assert_close(got[s0:s0 + m], ref[s0:s0 + m], atol=ATT_TOL[half][0], rtol=ATT_TOL[half][1])
        if padded:
            assert (got[s0 + m:s0 + pad] == 0).all()


def test_attention_empty_and_overlong_entries_are_memory_safe():
    """seq_len lives in device memory, so the kernel guards it itself: 0 (or negative) = an empty sequence whose padding rows get
    zeros and whose row range is never read; an entry above max_len is cut to max_len (the LDS image holds max_len keys)."""
    from ccrec_amd import ops
    H, pad = 2, 40
    torch.manual_seed(0)
    qkv = torch.randn(4 * pad, 3 * H * 64, device="cuda").to(torch.bfloat16)
    start = torch.arange(4, dtype=torch.int32, device="cuda") * pad
    lens = torch.tensor([0, 17, -3, 40], dtype=torch.int32, device="cuda")
    out = torch.full((4 * pad, H * 64), 5.0, dtype=torch.bfloat16, device="cuda")
    ops.attention(qkv, start, lens, H, max_len=40, pad_len=pad, out=out)
    ref = _attention_reference(qkv, [pad, 3 * pad], [17, 40], H)
    got = out.float()
    assert (got[:pad] == 0).all() and (got[2 * pad:3 * pad] == 0).all()
    torch.testing.assert_close(got[pad:pad + 17], ref[pad:pad + 17], atol=1.5e-2, rtol=1.6e-2)
    torch.testing.assert_close(got[3 * pad:], ref[3 * pad:], atol=1.5e-2, rtol=1.6e-2)
    # an entry longer than max_len: attended as its first max_len tokens, no access beyond them
    lens2 = torch.tensor([40, 17, 40, 40], dtype=torch.int32, device="cuda")
    out2 = torch.full((4 * pad, H * 64), 5.0, dtype=torch.bfloat16, device="cuda")
    ops.attention(qkv, start, lens2, H, max_len=24, pad_len=pad, out=out2)
    ref2 = _attention_reference(qkv, [0, pad, 2 * pad, 3 * pad], [24, 17, 24, 24], H)
    torch.testing.assert_close(out2.float()[:24], ref2[:24], atol=1.5e-2, rtol=1.6e-2)
    torch.testing.assert_close(out2.float()[pad:pad + 17], ref2[pad:pad + 17], atol=1.5e-2, rtol=1.6e-2)
    assert (out2.float()[24:pad] == 0).all()


def test_attention_rejects_bad_shapes():
    from ccrec_amd import ops, _lib
    qkv = torch.zeros(16, 3 * 64, dtype=torch.bfloat16, device="cuda")
    s = torch.zeros(1, dtype=torch.int32, device="cuda")
    n = torch.full((1,), 16, dtype=torch.int32, device="cuda")
    with pytest.raises(_lib.CcrError):
        ops.attention(qkv, s, n, 1, max_len=513)
    with pytest.raises(_lib.CcrError):
        ops.attention(qkv, s, n, 1, max_len=0)
    with pytest.raises(AssertionError):
        ops.attention(qkv, s, n, 2, max_len=16)     # rows are 192 wide, not 3 x 2 x 64


@pytest.mark.parametrize("rows,dim", [(1, 256), (7, 768), (1000, 768), (333, 1024), (5, 2048)])
@pytest.mark.parametrize("half", HALVES)
def test_add_layernorm_matches_torch(rows, dim, half):
    from ccrec_amd import ops
    torch.manual_seed(rows + dim)
    x = torch.randn(rows, dim, device="cuda").to(half)
    res = torch.randn(rows, dim, device="cuda") * 3 + 0.5
    gamma = torch.rand(dim, device="cuda") + 0.5
    beta = torch.randn(dim, device="cuda")
    f32, b16 = ops.add_layernorm(x, res, gamma, beta, 1e-12)
    ref = torch.nn.functional.layer_norm(x.float() + res, (dim,), gamma, beta, 1e-12)
    torch.testing.assert_close(f32, ref, atol=2e-5, rtol=2e-5)       # fp32 reduction order only
    assert b16.dtype == half and torch.equal(b16, f32.to(half))   # the 16-bit copy is the rounded fp32 row, bit for bit
    only_b16 = ops.add_layernorm(x, None, gamma, beta, 1e-5, want_f32=False)
    assert only_b16[0] is None
    ref2 = torch.nn.functional.layer_norm(x.float(), (dim,), gamma, beta, 1e-5)
    torch.testing.assert_close(only_b16[1].float(), ref2, atol=2e-2, rtol=8e-3)


def _bert(hidden=256, heads=4, layers=2, inter=512, seed=0, scale=1.0):
    from transformers import BertConfig, BertModel
    torch.manual_seed(seed)
    m = BertModel(BertConfig(vocab_size=600, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                             intermediate_size=inter, max_position_embeddings=512)).cuda().eval()
    if scale != 1.0:   # the default init (std 0.02) leaves every softmax near uniform: widen the attention projections
        with torch.no_grad():
            for layer in m.encoder.layer:
                layer.attention.self.query.weight.mul_(scale)
                layer.attention.self.key.weight.mul_(scale)
    return m


def _batch(lens, L, vocab=600, seed=0):
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(len(lens), L, dtype=torch.int64)
    mask = torch.zeros(len(lens), L, dtype=torch.int64)
    for r, n in enumerate(lens):
        ids[r, :n] = torch.randint(1, vocab, (n,), generator=g)
        mask[r, :n] = 1
    return ids.cuda(), mask.cuda(), torch.tensor(lens, dtype=torch.int32, device="cuda")


@pytest.mark.parametrize("hidden,heads,layers,scale", [(256, 4, 2, 1.0), (256, 4, 3, 25.0), (768, 12, 2, 12.0)])
def test_fused_forward_is_as_close_to_fp32_as_the_autocast_module(hidden, heads, layers, scale):
    from ccrec_amd.fused_bert import FusedBertEncoder, unsupported_reason
    model = _bert(hidden, heads, layers, hidden * 2, seed=hidden + layers, scale=scale)
    assert unsupported_reason(model) is None
    lens = [40, 1, 17, 33, 64, 65, 128, 130, 97, 200]
    ids, mask, lengths = _batch(lens, 200)
    with torch.no_grad():
        ref = model(input_ids=ids, attention_mask=mask).last_hidden_state                       # fp32 module forward
        with torch.autocast("cuda", dtype=torch.bfloat16):
            amp = model(input_ids=ids, attention_mask=mask).last_hidden_state.float()           # what the reference's layer computes
    got = FusedBertEncoder(model).forward(ids, lengths)
    assert got.dtype == torch.float32 and got.shape == ref.shape
    live = mask.bool()
    e_amp = (amp - ref)[live].abs()
    e_got = (got - ref)[live].abs()
    # tolerance: at most 1.5 x the autocast module's own distance from fp32 (max and mean), plus an absolute floor of 1e-3
    assert e_got.max().item() <= 1.5 * e_amp.max().item() + 1e-3, (e_got.max().item(), e_amp.max().item())
    assert e_got.mean().item() <= 1.5 * e_amp.mean().item() + 1e-4, (e_got.mean().item(), e_amp.mean().item())
    cos = torch.nn.functional.cosine_similarity(got[live], ref[live], dim=-1)
    cos_amp = torch.nn.functional.cosine_similarity(amp[live], ref[live], dim=-1)
    assert cos.min().item() > cos_amp.min().item() - 1e-3 and cos.min().item() > 0.995, (cos.min().item(), cos_amp.min().item())
    assert torch.isfinite(got).all()


def test_fused_forward_follows_weight_updates_and_rejects_other_encoders():
    from ccrec_amd.fused_bert import FusedBertEncoder, unsupported_reason
    model = _bert(256, 4, 1, 512)
    ids, mask, lengths = _batch([9, 30], 32)
    enc = FusedBertEncoder(model)
    a = enc.forward(ids, lengths).clone()
    assert enc.refresh() is False
    with torch.no_grad():
        model.encoder.layer[0].output.dense.weight.add_(0.05)      # an optimiser step between two ranking steps
    assert enc.refresh() is True
    b = enc.forward(ids, lengths)
    assert (a - b).abs().max().item() > 1e-3
    with torch.no_grad():
        ref = model(input_ids=ids, attention_mask=mask).last_hidden_state
    assert (b - ref)[mask.bool()].abs().max().item() < 5e-2
    # head width 32: not covered -> a reason, and the constructor refuses
    other = _bert(256, 8, 1, 512)
    assert "head width" in unsupported_reason(other)
    with pytest.raises(ValueError):
        FusedBertEncoder(other)
    model.train()
    with pytest.raises(AssertionError):
        enc.forward(ids, lengths)


class _WordTokenizer:
    """Whitespace tokenizer with the HF call shape (padding=False -> lists)."""
    pad_token_id = 0

    def __call__(self, texts, truncation=True, padding=True, max_length=64, return_tensors="pt"):
        ids = [[1] + [2 + (sum(map(ord, w)) * 7 % 500) for w in t.split()][: max_length - 2] + [3] for t in texts]
        assert padding is False
        return {"input_ids": ids, "attention_mask": [[1] * len(r) for r in ids]}


def test_length_sorted_encoder_runs_the_layer_kernels_and_agrees_with_the_modules():
    """LengthSortedEncoder(fused="auto") takes the kernel forward for a model the kernels cover; the packed rows agree with the
    module forward's to bf16 rounding of the hidden states (cosine >= 0.9995; identical top-1 neighbours on a small search)."""
    from ccrec_amd.encode import LengthSortedEncoder
    from ccrec_amd.item_tower import NaiveItemTower
    model = _bert(256, 4, 2, 512, seed=3, scale=10.0)
    tower = NaiveItemTower(model, torch.nn.LayerNorm(256, elementwise_affine=False)).cuda()
    rs = np.random.RandomState(0)
    words = [f"w{i}" for i in range(300)]
    texts = [" ".join(rs.choice(words, rs.randint(1, 60))) for _ in range(700)]
    tok = _WordTokenizer()
    fused = LengthSortedEncoder(tower, tok, max_length=64, max_tokens=4096, fused="auto")
    plain = LengthSortedEncoder(tower, tok, max_length=64, max_tokens=4096, fused=False)
    assert fused._fused is not None and plain._fused is None
    f32_a = torch.zeros(700, 256, device="cuda")
    f32_b = torch.zeros(700, 256, device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a = fused.encode(texts, out_f32=f32_a)
        b = plain.encode(texts, out_f32=f32_b)
    assert fused.stats["fused_layers"] is True and plain.stats["fused_layers"] is False
    cos = torch.nn.functional.cosine_similarity(f32_a, f32_b, dim=1)
    assert cos.min().item() >= 0.9995, cos.min().item()
    assert a.dtype == torch.bfloat16 and a.shape == b.shape
    for step in ("cls", "mean_layer_norm"):
        fa = torch.zeros(700, 256, device="cuda")
        fb = torch.zeros(700, 256, device="cuda")
        with torch.autocast("cuda", dtype=torch.bfloat16):
            LengthSortedEncoder(tower, tok, max_length=64, max_tokens=4096, output_step=step).encode(texts, out_f32=fa)
            LengthSortedEncoder(tower, tok, max_length=64, max_tokens=4096, output_step=step, fused=False).encode(texts, out_f32=fb)
        assert torch.nn.functional.cosine_similarity(fa, fb, dim=1).min().item() >= 0.9995
    # a model outside the kernels' coverage: "auto" keeps the module forward, True refuses
    other = NaiveItemTower(_bert(256, 8, 1, 512), torch.nn.LayerNorm(256, elementwise_affine=False)).cuda()
    assert LengthSortedEncoder(other, tok, max_length=64)._fused is None
    with pytest.raises(ValueError):
        LengthSortedEncoder(other, tok, max_length=64, fused=True)


def test_item_tower_forward_takes_the_kernels_only_under_autocast_with_right_padded_batches(monkeypatch):
    """The drop-in path (al_0_rank.py:98-101: tower(**tokenizer(texts), output_step=...)) runs the layer kernels when -- and only
    when -- gradients are off, the model is in eval mode, the caller asked for reduced precision (autocast) and the batch is
    right-padded; everything else is the module forward, bit for bit."""
    from ccrec_amd import fused_bert
    from ccrec_amd.item_tower import NaiveItemTower
    monkeypatch.delenv("CCREC_FUSED_ENCODER", raising=False)
    model = _bert(256, 4, 2, 512, seed=5, scale=8.0)
    tower = NaiveItemTower(model, torch.nn.LayerNorm(256, elementwise_affine=False)).cuda().eval()
    ids, mask, lengths = _batch([12, 40, 1, 33, 64], 64)
    calls = []
    real = fused_bert.FusedBertEncoder.forward
    monkeypatch.setattr(fused_bert.FusedBertEncoder, "forward", lambda self, *a, **k: (calls.append(1), real(self, *a, **k))[1])

    with torch.no_grad():
        plain = tower(input_ids=ids, attention_mask=mask, output_step="mean_pooling")          # fp32, no autocast: the module
        assert not calls
        with torch.autocast("cuda", dtype=torch.bfloat16):
            fast = tower(input_ids=ids, attention_mask=mask, output_step="mean_pooling")
            assert len(calls) == 1
            monkeypatch.setenv("CCREC_FUSED_ENCODER", "0")
            amp = tower(input_ids=ids, attention_mask=mask, output_step="mean_pooling")        # the module under autocast
            assert len(calls) == 1
            monkeypatch.delenv("CCREC_FUSED_ENCODER")
            # left padding: not a prefix mask -> module forward, same bits as the module run above on the same inputs
            lmask = torch.flip(mask, dims=[1])
            lids = torch.flip(ids, dims=[1])
            tower(input_ids=lids, attention_mask=lmask, output_step="mean_pooling")
            assert len(calls) == 1
            # extra model inputs (position_ids) -> module forward
            tower(input_ids=ids, attention_mask=mask, position_ids=torch.arange(64, device="cuda")[None].expand(5, -1), output_step="mean_pooling")
            assert len(calls) == 1
        monkeypatch.setenv("CCREC_FUSED_ENCODER", "1")                                         # forced, also outside autocast
        forced = tower(input_ids=ids, attention_mask=mask, output_step="mean_pooling")
        assert len(calls) == 2
    cos = torch.nn.functional.cosine_similarity
    assert cos(fast.float(), plain.float(), dim=1).min().item() > 0.9995
    assert cos(fast.float(), amp.float(), dim=1).min().item() > 0.9995
    assert torch.equal(forced, fast)
    # gradients on (the training forward, bbpr.py:130-141): always the module
    monkeypatch.delenv("CCREC_FUSED_ENCODER")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = tower(input_ids=ids, attention_mask=mask, output_step="mean_pooling")
    assert out.requires_grad and len(calls) == 2


def test_prefix_lengths():
    from ccrec_amd.fused_bert import prefix_lengths
    m = torch.tensor([[1, 1, 0, 0], [1, 1, 1, 1], [1, 0, 0, 0]], device="cuda")
    assert prefix_lengths(m).tolist() == [2, 4, 1]
    assert prefix_lengths(torch.tensor([[1, 0, 1, 0]], device="cuda")) is None       # a hole
    assert prefix_lengths(torch.tensor([[0, 1, 1, 1]], device="cuda")) is None       # left padding
    assert prefix_lengths(torch.tensor([[0, 0, 0, 0], [1, 1, 0, 0]], device="cuda")) is None   # an empty row
    assert prefix_lengths(torch.ones(2, 513, dtype=torch.long, device="cuda")) is None


def test_packed_forward_equals_the_padded_forward_on_the_real_tokens():
    """packed=True runs the layers on the real tokens only and scatters back: same hidden states as the padded forward (GEMM row
    blocks differ, so to fp32 summation order of a bf16 pipeline: 2 bf16 ulps of the LayerNorm-scale values), zeros on padding;
    packed=None picks it for a batch that is mostly padding."""
    from ccrec_amd.fused_bert import FusedBertEncoder
    model = _bert(256, 4, 2, 512, seed=11, scale=10.0)
    enc = FusedBertEncoder(model)
    lens = [3, 60, 1, 17, 33, 9]
    ids, mask, lengths = _batch(lens, 64)
    types = torch.zeros_like(ids)
    types[:, 5:] = 1
    for tt in (None, types):
        a = enc.forward(ids, lengths, token_type_ids=tt, packed=False)
        b = enc.forward(ids, lengths, token_type_ids=tt, packed=True)
        c = enc.forward(ids, lengths, token_type_ids=tt, lengths_host=lens)           # 123 of 384 tokens are real: packed
        live = mask.bool()
        torch.testing.assert_close(b[live], a[live], atol=3e-2, rtol=2e-2)
        assert torch.nn.functional.cosine_similarity(b[live], a[live], dim=-1).min().item() > 0.9995
        assert (b[~live] == 0).all() and torch.equal(b, c)
    with torch.no_grad():
        ref = model(input_ids=ids, attention_mask=mask, token_type_ids=types).last_hidden_state
    assert (b - ref)[live].abs().max().item() < 6e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("normalize", [False, True])
def test_packed_pooling_gives_the_padded_pooling_bits(dtype, normalize):
    """ccr_meanpool_pack_bf16_packed over a packed token array == ccr_meanpool_pack_bf16_ex over the same tokens in a right-padded
    [B, L, dim] batch, bit for bit (fp32 rows, bf16 rows, norm bounds, scattered destination rows)."""
    from ccrec_amd import ops
    torch.manual_seed(3)
    lens = [5, 1, 24, 17, 9, 24]
    B, L, d = len(lens), 24, 768
    padded = torch.zeros(B, L, d, device="cuda", dtype=dtype)
    mask = torch.zeros(B, L, dtype=torch.int64, device="cuda")
    pieces = []
    for b, n in enumerate(lens):
        x = (torch.randn(n, d, device="cuda") * 0.4).to(dtype)
        padded[b, :n] = x
        mask[b, :n] = 1
        pieces.append(x)
    packed = torch.cat(pieces).contiguous()
    seq_len = torch.tensor(lens, dtype=torch.int32, device="cuda")
    seq_start = (torch.cumsum(seq_len, 0, dtype=torch.int32) - seq_len).contiguous()
    rows = torch.tensor([7, 0, 3, 9, 1, 4], device="cuda")
    outs = []
    for form in ("padded", "packed"):
        b16 = torch.zeros(10, d, dtype=torch.bfloat16, device="cuda")
        f32 = torch.zeros(10, d, dtype=torch.float32, device="cuda")
        nb = torch.zeros(10, dtype=torch.float32, device="cuda")
        if form == "padded":
            ops.meanpool_pack(padded, mask, normalize=normalize, out_bf16=b16, out_f32=f32, dst_rows=rows, norm_bounds=nb)
        else:
            ops.meanpool_pack_packed(packed, seq_start, seq_len, normalize=normalize, out_bf16=b16, out_f32=f32, dst_rows=rows, norm_bounds=nb)
        outs.append((b16, f32, nb))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert outs[0][1][7].abs().sum().item() > 0


@pytest.mark.parametrize("half", HALVES)
def test_embed_layernorm_gives_the_module_bits(half):
    """ccr_embed_layernorm == transformers' BertEmbeddings.forward in eval mode: the same fp32 sum order, LayerNorm to fp32
    reduction order (2e-5); the bf16 copy is the rounded fp32 row; an index outside its table is clamped, not read."""
    from ccrec_amd import ops
    model = _bert(768, 12, 1, 1536, seed=2)
    e = model.embeddings
    g = torch.Generator().manual_seed(1)
    T = 3000
    ids = torch.randint(0, 600, (T,), generator=g).cuda()
    pos = torch.randint(0, 512, (T,), generator=g).cuda()
    types = torch.randint(0, 2, (T,), generator=g).cuda()
    with torch.no_grad():
        e.LayerNorm.weight.uniform_(0.5, 1.5)
        e.LayerNorm.bias.normal_()
        for tt in (types, None):
            ref = e(input_ids=ids[None], token_type_ids=(types if tt is not None else torch.zeros_like(ids))[None], position_ids=pos[None])[0]
            f32, b16 = ops.embed_layernorm(e.word_embeddings.weight, e.position_embeddings.weight, e.token_type_embeddings.weight,
                                           ids, pos, tt, e.LayerNorm.weight, e.LayerNorm.bias, e.LayerNorm.eps, dtype=half)
            torch.testing.assert_close(f32, ref, atol=2e-5, rtol=2e-5)
            assert b16.dtype == half and torch.equal(b16, f32.to(half))
        bad = ids.clone()
        bad[0], bad[1] = 10 ** 9, -5
        f32, _ = ops.embed_layernorm(e.word_embeddings.weight, e.position_embeddings.weight, e.token_type_embeddings.weight,
                                     bad, pos, None, e.LayerNorm.weight, e.LayerNorm.bias, e.LayerNorm.eps)
        assert torch.isfinite(f32).all()


def test_encoder_slot_follows_replicas_and_does_not_pin_the_model():
    """fused_bert.for_model keeps the encoder in the model's own __dict__: a shallow copy of the module (what torch's replicate()
    makes) gets an encoder of its own, a dead model is collectable (no registry holds it), and pickling the model drops the weight copies."""
    import copy
    import gc
    import io
    import weakref
    from ccrec_amd import fused_bert
    model = _bert(256, 4, 1, 512)
    enc = fused_bert.for_model(model)
    assert enc is not None and fused_bert.for_model(model) is enc
    replica = copy.copy(model)                       # __dict__ copied shallowly, as _replicate_for_data_parallel does
    replica.__dict__ = dict(model.__dict__)
    assert replica.__dict__["_ccr_fused_encoder"] is enc
    enc2 = fused_bert.for_model(replica)
    assert enc2 is not enc and enc2.model is replica and fused_bert.for_model(model) is enc
    ids, mask, lengths = _batch([5, 9], 16)
    enc.forward(ids, lengths)
    buf = io.BytesIO()
    torch.save(model, buf)
    assert buf.tell() < 2.5 * sum(p.numel() * 4 for p in model.parameters())      # no bf16 weight copies inside
    ref = weakref.ref(model)
    del model, enc, enc2, replica
    gc.collect()
    assert ref() is None
    assert fused_bert.for_model(_bert(256, 8, 1, 512)) is None


@pytest.mark.parametrize("half", HALVES)
def test_gelu_matches_torch_bit_for_bit(half):
    """ccr_gelu_half (in place) == torch.nn.functional.gelu on a bf16 / fp16 tensor (the exact erf form BERT uses), same bits: both
    evaluate 0.5 x (1 + erf(x / sqrt 2)) in fp32 and round once."""
    from ccrec_amd import ops
    torch.manual_seed(0)
    x = (torch.randn(4096, 3072, device="cuda") * 3).to(half)
    x[0, :8] = torch.tensor([0.0, -0.0, 1e-30, -1e-30, 50.0, -50.0, float("inf"), float("-inf")], device="cuda").to(half)
    ref = torch.nn.functional.gelu(x)
    got = ops.gelu_(x.clone())
    same = (got.view(torch.int16) == ref.view(torch.int16)) | (torch.isnan(got) & torch.isnan(ref))
    assert same.all(), int((~same).sum())


def test_kernel_forward_is_deterministic_run_to_run():
    """Two encodes of the same texts give the same packed rows bit for bit (no atomics in the layer kernels; the library GEMMs are
    deterministic for a fixed shape) -- a resumed ranking step finds the rows it saved (also checked at 20 K passages x BERT-base)."""
    from ccrec_amd.encode import LengthSortedEncoder
    from ccrec_amd.item_tower import NaiveItemTower
    tower = NaiveItemTower(_bert(256, 4, 2, 512, seed=9, scale=6.0), torch.nn.LayerNorm(256, elementwise_affine=False)).cuda()
    rs = np.random.RandomState(4)
    words = [f"w{i}" for i in range(300)]
    texts = [" ".join(rs.choice(words, rs.randint(1, 60))) for _ in range(900)]
    enc = LengthSortedEncoder(tower, _WordTokenizer(), max_length=64, max_tokens=4096, fused=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a = enc.encode(texts).clone()
        b = enc.encode(texts)
    assert torch.equal(a, b)


def test_distilbert_takes_the_same_kernels():
    """DistilBertModel (the reference's default model_name, src/ccrec/models/bbpr.py:50, bert_mt.py:35) is the same post-LayerNorm GELU
    layer under other attribute names and without token types: same parity bar as BertModel (no further from the fp32 module forward
    than its autocast forward), padded and packed, and through the tower's drop-in forward."""
    from transformers import DistilBertConfig, DistilBertModel
    from ccrec_amd import fused_bert
    from ccrec_amd.item_tower import NaiveItemTower
    torch.manual_seed(21)
    model = DistilBertModel(DistilBertConfig(vocab_size=600, dim=256, n_layers=3, n_heads=4, hidden_dim=512, max_position_embeddings=512)).cuda().eval()
    with torch.no_grad():
        for layer in model.transformer.layer:
            layer.attention.q_lin.weight.mul_(8.0)
            layer.attention.k_lin.weight.mul_(8.0)
    assert fused_bert.unsupported_reason(model) is None
    lens = [40, 1, 17, 33, 64, 65, 128, 130, 97, 200]
    ids, mask, lengths = _batch(lens, 200)
    with torch.no_grad():
        ref = model(input_ids=ids, attention_mask=mask).last_hidden_state
        with torch.autocast("cuda", dtype=torch.bfloat16):
            amp = model(input_ids=ids, attention_mask=mask).last_hidden_state.float()
    enc = fused_bert.for_model(model)
    assert enc is not None and enc.has_token_types is False
    live = mask.bool()
    e_amp = (amp - ref)[live].abs()
    for packed in (False, True):
        got = enc.forward(ids, lengths, packed=packed)
        e_got = (got - ref)[live].abs()
        assert e_got.max().item() <= 1.5 * e_amp.max().item() + 1e-3, (packed, e_got.max().item(), e_amp.max().item())
        assert e_got.mean().item() <= 1.5 * e_amp.mean().item() + 1e-4
    tower = NaiveItemTower(model, torch.nn.LayerNorm(256, elementwise_affine=False)).cuda().eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        fast = tower(input_ids=ids, attention_mask=mask, output_step="mean_pooling")
        os_env = __import__("os").environ
        os_env["CCREC_FUSED_ENCODER"] = "0"
        try:
            slow = tower(input_ids=ids, attention_mask=mask, output_step="mean_pooling")
        finally:
            del os_env["CCREC_FUSED_ENCODER"]
    assert torch.nn.functional.cosine_similarity(fast.float(), slow.float(), dim=1).min().item() > 0.9995


def test_cls_only_last_layer_equals_the_full_forward_first_rows():
    """cls_only=True runs the last layer's output projection / LayerNorms / FFN on the sequences' first tokens only: the same rows as
    the full forward's hidden[:, 0] (row-wise operations; GEMM row blocks differ -> bf16-pipeline noise), padded, packed and through the
    tower's cls / mean_layer_norm output steps (src/ccrec/models/item_tower.py:133-136)."""
    from ccrec_amd import fused_bert
    from ccrec_amd.item_tower import NaiveItemTower
    model = _bert(256, 4, 3, 512, seed=13, scale=9.0)
    enc = fused_bert.for_model(model)
    lens = [3, 60, 1, 17, 33, 9, 64]
    ids, mask, lengths = _batch(lens, 64)
    full = enc.forward(ids, lengths, packed=False)
    for packed in (False, True):
        got = enc.forward(ids, lengths, packed=packed, cls_only=True)
        assert got.shape == (len(lens), 1, 256)
        torch.testing.assert_close(got[:, 0], full[:, 0], atol=3e-2, rtol=2e-2)
        assert torch.nn.functional.cosine_similarity(got[:, 0], full[:, 0], dim=-1).min().item() > 0.9995
    tower = NaiveItemTower(model, torch.nn.LayerNorm(256, elementwise_affine=False)).cuda().eval()
    import os
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for step in ("cls", "mean_layer_norm"):
            fast = tower(input_ids=ids, attention_mask=mask, output_step=step)
            os.environ["CCREC_FUSED_ENCODER"] = "0"
            try:
                slow = tower(input_ids=ids, attention_mask=mask, output_step=step)
            finally:
                del os.environ["CCREC_FUSED_ENCODER"]
            assert fast.shape == slow.shape == (len(lens), 256)
            assert torch.nn.functional.cosine_similarity(fast.float(), slow.float(), dim=1).min().item() > 0.9995


def test_golden_tower_on_a_real_encoder(golden_dir):
    """Golden g16 (the reference's own NaiveItemTower + a real BertModel, fp32 CPU): this package's tower reproduces it with the module
    forward in fp32 (1e-4: fp32 GEMM order on another device) and with the kernel forward under autocast(bf16) to bf16 activation rounding
    (weights are bf16-exact in the fixture): |error| <= 4 bf16 ulps of the largest value, cosine >= 0.9998 per row.  The fp16 form
    (the reference's own autocast type) is in tests/test_gpu_encoder_fp16.py."""
    import json
    from transformers import BertConfig, BertModel
    from ccrec_amd import fused_bert
    from ccrec_amd.item_tower import NaiveItemTower
    g = np.load(f"{golden_dir}/g16_item_tower_bert.npz")
    model = BertModel(BertConfig(**json.loads(str(g["config"])))).eval()
    state = {k[2:]: torch.from_numpy(g[k].astype(np.int16)).view(torch.bfloat16).float() for k in g.files if k.startswith("w_")}
    model.load_state_dict(state, strict=False)
    tower = NaiveItemTower(model, torch.nn.LayerNorm(256, elementwise_affine=False)).cuda().eval()
    ids, mask = torch.from_numpy(g["ids"]).cuda(), torch.from_numpy(g["mask"]).cuda()
    assert fused_bert.unsupported_reason(model) is None
    for step in ("mean_pooling", "cls", "mean_layer_norm"):
        ref = torch.from_numpy(g["out_" + step]).cuda()
        with torch.no_grad():
            plain = tower(input_ids=ids, attention_mask=mask, output_step=step).float()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                fast = tower(input_ids=ids, attention_mask=mask, output_step=step).float()
        torch.testing.assert_close(plain, ref, atol=1e-4, rtol=1e-4)
        # absolute bound in ulps of the 16-bit type the layer ran in (2^-8 relative for bf16) of the largest fp32 output
        ulp = 2.0 ** -8 * ref.abs().max().item()
        assert (fast - ref).abs().max().item() <= 4 * ulp, (step, (fast - ref).abs().max().item() / ulp)
        assert torch.nn.functional.cosine_similarity(fast, ref, dim=1).min().item() >= 0.9998, step


def test_length_sorted_encoder_edge_cases_on_the_kernel_path():
    """No texts, one text, one-token texts, a preallocated shard with a row offset, identical texts: the packed-array path keeps the
    contract of the padded one (rows land at row_offset + i; equal texts give equal rows)."""
    from ccrec_amd.encode import LengthSortedEncoder
    from ccrec_amd.item_tower import NaiveItemTower
    tower = NaiveItemTower(_bert(256, 4, 1, 512, seed=1), torch.nn.LayerNorm(256, elementwise_affine=False)).cuda()
    enc = LengthSortedEncoder(tower, _WordTokenizer(), max_length=64, max_tokens=512, fused=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        empty = enc.encode([])
        assert empty.shape[0] == 0
        one = enc.encode(["w1 w2 w3"])
        assert one.shape == (1, 256) and torch.isfinite(one.float()).all()
        texts = ["w1 w2 w3", "", "w9", "w1 w2 w3", "w4 " * 100]
        shard = torch.zeros(9, 256, dtype=torch.bfloat16, device="cuda")
        bounds = torch.zeros(9, device="cuda")
        out = enc.encode(texts, out=shard, row_offset=3, norm_bounds=bounds)
    assert out.data_ptr() == shard.data_ptr()
    assert (shard[:3] == 0).all() and (shard[8:] == 0).all()
    assert torch.equal(shard[3], shard[6])                                          # the same text twice in a batch: the same row
    assert torch.nn.functional.cosine_similarity(shard[3].float(), one[0].float(), dim=0).item() > 0.9999   # ... and (to GEMM order) in another batch
    assert torch.isfinite(shard.float()).all() and (bounds[3:8] > 0).all()
    assert (bounds[3:8] >= shard[3:8].float().norm(dim=1) * 0.999).all()
