"""The encoder kernel path in the REFERENCE'S OWN autocast type.  The reference encodes under `with torch.cuda.amp.autocast():`
(scripts/al_0_rank.py:8,125), whose CUDA default is fp16: transformers' BertLayer then runs fp16 projections, an fp16 attention call
and an fp16 GELU around an fp32 residual sum + LayerNorm (src/ccrec/models/item_tower.py:122).  The layer kernels
(csrc/ccr_encoder.hip) have an fp16 instantiation of every piece, selected by the caller's autocast type (fused_bert.kernel_dtype);
these tests run the drop-in tower and the length-sorted encoder under the DEFAULT `torch.autocast("cuda")` on golden g16's weights
(the reference's tower around a real BertModel) and on larger random-init encoders.

Bars (floating point, so each tolerance is written where it is used): the kernel forward is no further from the fp32 outputs
(golden g16's, produced by the reference itself) than 1.5 x the module's own fp16-autocast forward is; absolute errors are stated in
fp16 ulps (2^-11 relative) of the largest output; and the top-k ids of a search over kernel-path embeddings equal those over
module-path embeddings at every rank that the module's scores separate by more than 1e-3 (the north-star's score tolerance)."""
import json

import numpy as np
import pytest
import torch

from conftest import ROOT, PKG  # noqa: F401

pytestmark = pytest.mark.gpu

FP16_ULP = 2.0 ** -11


def _g16_tower(golden_dir):
    from transformers import BertConfig, BertModel
    from ccrec_amd.item_tower import NaiveItemTower
    g = np.load(f"{golden_dir}/g16_item_tower_bert.npz")
    model = BertModel(BertConfig(**json.loads(str(g["config"])))).eval()
    state = {k[2:]: torch.from_numpy(g[k].astype(np.int16)).view(torch.bfloat16).float() for k in g.files if k.startswith("w_")}
    model.load_state_dict(state, strict=False)
    tower = NaiveItemTower(model, torch.nn.LayerNorm(256, elementwise_affine=False)).cuda().eval()
    return g, tower


class _TinyTokenizer:
    """Whitespace tokenizer into golden g16's 64-id vocabulary, with the HF call shapes the tower (padding=True -> tensors) and the
    length-sorted encoder (padding=False -> lists) use."""
    pad_token_id = 0

    def __call__(self, texts, truncation=True, padding=True, max_length=32, return_tensors="pt"):
        ids = [[1] + [4 + (sum(map(ord, w)) * 7 % 60) for w in t.split()][: max_length - 2] + [2] for t in texts]
        if padding is False:
            return {"input_ids": ids, "attention_mask": [[1] * len(r) for r in ids]}
        L = max(len(r) for r in ids)
        out = torch.zeros(len(ids), L, dtype=torch.int64)
        mask = torch.zeros(len(ids), L, dtype=torch.int64)
        for r, row in enumerate(ids):
            out[r, :len(row)] = torch.tensor(row)
            mask[r, :len(row)] = 1
        return {"input_ids": out, "attention_mask": mask}


def _texts(n, seed, longest=28):
    rs = np.random.RandomState(seed)
    words = [f"w{i}" for i in range(200)]
    return [" ".join(rs.choice(words, rs.randint(1, longest + 1))) for _ in range(n)]


def test_kernel_dtype_is_the_autocast_contexts_own_type(monkeypatch):
    """fused_bert.kernel_dtype: fp16 under the reference's autocast() (old and new spelling), bf16 only when the caller names it,
    nothing outside autocast unless CCREC_FUSED_ENCODER=1 demands the kernels (then CCREC_FUSED_ENCODER_DTYPE, default bf16)."""
    from ccrec_amd import fused_bert
    monkeypatch.delenv("CCREC_FUSED_ENCODER", raising=False)
    monkeypatch.delenv("CCREC_FUSED_ENCODER_DTYPE", raising=False)
    assert fused_bert.kernel_dtype() is None and not fused_bert.wanted()
    with torch.autocast("cuda"):
        assert fused_bert.kernel_dtype() is torch.float16 and fused_bert.wanted()
        assert fused_bert.kernel_dtype(False) is None
        monkeypatch.setenv("CCREC_FUSED_ENCODER", "0")
        assert fused_bert.kernel_dtype() is None
        assert fused_bert.kernel_dtype(True) is torch.float16      # an explicit request outranks the environment
        monkeypatch.delenv("CCREC_FUSED_ENCODER")
    with pytest.warns(FutureWarning), torch.cuda.amp.autocast():         # the reference's spelling (al_0_rank.py:8,125)
        assert fused_bert.kernel_dtype() is torch.float16
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert fused_bert.kernel_dtype() is torch.bfloat16
        assert fused_bert.kernel_dtype(True) is torch.bfloat16
    with torch.autocast("cuda", enabled=False):
        assert fused_bert.kernel_dtype() is None
    monkeypatch.setenv("CCREC_FUSED_ENCODER", "1")
    assert fused_bert.kernel_dtype() is torch.bfloat16
    monkeypatch.setenv("CCREC_FUSED_ENCODER_DTYPE", "fp16")
    assert fused_bert.kernel_dtype() is torch.float16
    with torch.autocast("cuda", dtype=torch.bfloat16):                   # inside autocast the context's type wins
        assert fused_bert.kernel_dtype() is torch.bfloat16


def _errors(x, ref):
    e = (x - ref).abs()
    cos = torch.nn.functional.cosine_similarity(x, ref, dim=-1)
    return e.max().item(), e.mean().item(), cos.min().item()


@pytest.mark.parametrize("step", ["mean_pooling", "cls", "mean_layer_norm"])
def test_tower_under_the_references_autocast_is_as_close_to_golden_g16_as_the_fp16_module(golden_dir, monkeypatch, step):
    """NaiveItemTower(**inputs, output_step) under `torch.autocast("cuda")` (fp16, the reference's context) on golden g16's weights
    and inputs: the kernel forward (fp16 instantiations) vs the module forward under the same context, both against g16's fp32
    outputs (the reference's tower, CPU).  Bar: max and mean |error| <= 1.5 x the module's (+ 0.05 / 0.01 ulp), cosine no lower than
    the module's - 1e-6, and an absolute bound of ONE fp16 ulp (2^-11 relative) of the largest output -- measured on MI355X:
    0.09-0.16 ulp for both paths (the layer's roundings average out in the 256-wide sums behind them)."""
    from ccrec_amd import fused_bert
    monkeypatch.delenv("CCREC_FUSED_ENCODER", raising=False)
    g, tower = _g16_tower(golden_dir)
    ids, mask = torch.from_numpy(g["ids"]).cuda(), torch.from_numpy(g["mask"]).cuda()
    ref = torch.from_numpy(g["out_" + step]).cuda()
    seen = []
    real = fused_bert.FusedBertEncoder.forward
    monkeypatch.setattr(fused_bert.FusedBertEncoder, "forward",
                        lambda self, *a, **k: (seen.append(k.get("dtype")), real(self, *a, **k))[1])
    with torch.no_grad(), torch.autocast("cuda"):
        fast = tower(input_ids=ids, attention_mask=mask, output_step=step).float()
        assert seen == [torch.float16], seen                      # the kernels ran, in the context's type
        monkeypatch.setenv("CCREC_FUSED_ENCODER", "0")
        slow = tower(input_ids=ids, attention_mask=mask, output_step=step).float()
        assert len(seen) == 1
    k_max, k_mean, k_cos = _errors(fast, ref)
    m_max, m_mean, m_cos = _errors(slow, ref)
    ulp = FP16_ULP * ref.abs().max().item()
    print(f"g16 {step}: kernels max {k_max / ulp:.2f} ulp mean {k_mean / ulp:.3f} ulp cos {k_cos:.7f} | "
          f"module max {m_max / ulp:.2f} ulp mean {m_mean / ulp:.3f} ulp cos {m_cos:.7f}")
    assert k_max <= 1.5 * m_max + 0.05 * ulp, (k_max / ulp, m_max / ulp)
    assert k_mean <= 1.5 * m_mean + 0.01 * ulp, (k_mean / ulp, m_mean / ulp)
    assert k_cos >= m_cos - 1e-6 and k_cos >= 0.999999, (k_cos, m_cos)
    assert k_max <= 1.0 * ulp, k_max / ulp


def test_length_sorted_encoder_under_the_references_autocast_runs_fp16_kernels(golden_dir, monkeypatch):
    """LengthSortedEncoder.encode under `torch.autocast("cuda")` on golden g16's weights: the packed-array kernel forward runs in fp16
    (stats["layer_dtype"]) and its pooled fp32 rows are as close to the tower's fp32 module forward as the fp16-autocast module path's
    are (1.5 x rule, cosine), over 600 texts of 3-30 tokens."""
    from ccrec_amd.encode import LengthSortedEncoder
    monkeypatch.delenv("CCREC_FUSED_ENCODER", raising=False)
    _, tower = _g16_tower(golden_dir)
    tok = _TinyTokenizer()
    texts = _texts(600, seed=5)
    n = len(texts)
    ref = torch.zeros(n, 256, device="cuda")
    LengthSortedEncoder(tower, tok, max_length=32, max_tokens=2048, fused=False).encode(texts, out_f32=ref)          # fp32 modules
    fast, slow = torch.zeros(n, 256, device="cuda"), torch.zeros(n, 256, device="cuda")
    kern = LengthSortedEncoder(tower, tok, max_length=32, max_tokens=2048)
    mod = LengthSortedEncoder(tower, tok, max_length=32, max_tokens=2048, fused=False)
    with torch.autocast("cuda"):
        kern.encode(texts, out_f32=fast)
        mod.encode(texts, out_f32=slow)
    assert kern.stats["fused_layers"] is True and kern.stats["layer_dtype"] == "float16"
    assert mod.stats["fused_layers"] is False and mod.stats["layer_dtype"] is None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        kern.encode(texts[:50])
    assert kern.stats["layer_dtype"] == "bfloat16"
    k_max, k_mean, k_cos = _errors(fast, ref)
    m_max, m_mean, m_cos = _errors(slow, ref)
    ulp = FP16_ULP * ref.abs().max().item()
    print(f"length-sorted g16: kernels max {k_max / ulp:.2f} ulp mean {k_mean / ulp:.3f} cos {k_cos:.7f} | module max {m_max / ulp:.2f} "
          f"mean {m_mean / ulp:.3f} cos {m_cos:.7f}")
    assert k_max <= 1.5 * m_max + 0.05 * ulp and k_mean <= 1.5 * m_mean + 0.01 * ulp, (k_max / ulp, m_max / ulp, k_mean / ulp, m_mean / ulp)
    assert k_cos >= m_cos - 1e-6 and k_cos >= 0.999999, (k_cos, m_cos)
    assert k_max <= 1.0 * ulp, k_max / ulp


@pytest.mark.parametrize("hidden,heads,layers,scale", [(256, 4, 3, 25.0), (768, 12, 2, 12.0)])
def test_fp16_kernel_forward_is_as_close_to_fp32_as_the_fp16_autocast_module(hidden, heads, layers, scale):
    """Hidden states of larger random-init encoders with peaked softmaxes (attention projections widened): FusedBertEncoder(dtype=fp16)
    vs the module under autocast(fp16), both against the fp32 module forward -- the 1.5 x rule of
    test_fused_forward_is_as_close_to_fp32_as_the_autocast_module, in the reference's type."""
    from test_gpu_encoder_kernels import _bert, _batch
    from ccrec_amd.fused_bert import FusedBertEncoder
    model = _bert(hidden, heads, layers, hidden * 2, seed=hidden + layers, scale=scale)
    lens = [40, 1, 17, 33, 64, 65, 128, 130, 97, 200]
    ids, mask, lengths = _batch(lens, 200)
    with torch.no_grad():
        ref = model(input_ids=ids, attention_mask=mask).last_hidden_state
        with torch.autocast("cuda"):
            amp = model(input_ids=ids, attention_mask=mask).last_hidden_state.float()
    enc = FusedBertEncoder(model)
    live = mask.bool()
    e_amp = (amp - ref)[live].abs()
    for packed in (False, True):
        got = enc.forward(ids, lengths, packed=packed, dtype=torch.float16)
        assert got.dtype == torch.float32 and torch.isfinite(got).all()
        e_got = (got - ref)[live].abs()
        assert e_got.max().item() <= 1.5 * e_amp.max().item() + 2e-4, (packed, e_got.max().item(), e_amp.max().item())
        assert e_got.mean().item() <= 1.5 * e_amp.mean().item() + 2e-5, (packed, e_got.mean().item(), e_amp.mean().item())
    # ... and fp16 is closer to fp32 than the bf16 instantiation on the same inputs (11 vs 8 significand bits)
    e_bf = (enc.forward(ids, lengths, dtype=torch.bfloat16) - ref)[live].abs()
    assert e_got.mean().item() < 0.5 * e_bf.mean().item(), (e_got.mean().item(), e_bf.mean().item())
    assert set(enc._layers) == {torch.float16, torch.bfloat16}


def test_top_k_ids_from_kernel_embeddings_equal_the_module_paths_where_scores_are_separated(golden_dir, monkeypatch):
    """The ranking consequence.  3 000 passages + 64 queries through golden g16's tower under the reference's autocast(), once on the
    layer kernels and once as torch modules, cosine scores (CCREC_SIM_TYPE=cos: O(1) scores, so the north-star's 1e-3 is meaningful)
    in fp32 from the fp32 pooled rows, top-20 of each.  Bar: max |score difference| <= 2e-5 (measured 2.9e-6), hence at every rank
    where the module's score is more than 1e-3 -- and, ten times stricter, more than 1e-4 -- away from both neighbours (rank 21
    included) the kernel path holds the same passage at the same rank; the top-20 SETS agree except for passages within 1e-4 of
    the 20th score."""
    from ccrec_amd.encode import LengthSortedEncoder
    monkeypatch.delenv("CCREC_FUSED_ENCODER", raising=False)
    _, tower = _g16_tower(golden_dir)
    tok = _TinyTokenizer()
    corpus, queries = _texts(3000, seed=11), _texts(64, seed=12, longest=12)
    k = 20
    emb = {}
    for name, fused in (("kernels", "auto"), ("modules", False)):
        enc = LengthSortedEncoder(tower, tok, max_length=32, max_tokens=4096, fused=fused)
        d, q = torch.zeros(len(corpus), 256, device="cuda"), torch.zeros(len(queries), 256, device="cuda")
        with torch.autocast("cuda"):
            enc.encode(corpus, out_f32=d)
            enc.encode(queries, out_f32=q)
        assert enc.stats["fused_layers"] is (fused == "auto")
        emb[name] = (torch.nn.functional.normalize(q.double(), dim=1) @ torch.nn.functional.normalize(d.double(), dim=1).T).float()
    s_k, s_m = emb["kernels"], emb["modules"]
    diff = (s_k - s_m).abs().max().item()
    assert diff <= 2e-5, diff
    top_m, idx_m = s_m.topk(k + 1, dim=1)
    _, idx_k = s_k.topk(k + 1, dim=1)
    gap_next = top_m[:, :-1] - top_m[:, 1:]                                   # [Q, k]: score(r) - score(r + 1)
    gap_prev = torch.cat([torch.full_like(gap_next[:, :1], float("inf")), gap_next[:, :-1]], dim=1)
    same = idx_k[:, :k] == idx_m[:, :k]
    bound = {}
    for tau in (1e-3, 1e-4):
        separated = (gap_next > tau) & (gap_prev > tau)
        bound[tau] = separated.float().mean().item()
        assert bool(same[separated].all()), (tau, int((~same & separated).sum()))
    print(f"max |cos score (kernels) - cos score (modules)| = {diff:.2e}; ranks separated by > 1e-3: {bound[1e-3]:.2f}, by > 1e-4: "
          f"{bound[1e-4]:.2f}; identical ranks overall: {same.float().mean().item():.3f}")
    assert bound[1e-3] > 0.05 and bound[1e-4] > 0.25, bound            # the test must bind
    # set agreement: a passage of the module's top-k missing from the kernel path's top-k sits within 1e-4 of the cut
    cut = top_m[:, k - 1]
    for qi in range(len(queries)):
        missing = set(idx_m[qi, :k].tolist()) - set(idx_k[qi, :k].tolist())
        for p in missing:
            assert s_m[qi, p].item() - cut[qi].item() <= 1e-4, (qi, p)


def test_fp16_weight_copies_follow_the_module_and_both_types_coexist():
    from test_gpu_encoder_kernels import _bert, _batch
    from ccrec_amd.fused_bert import FusedBertEncoder
    model = _bert(256, 4, 1, 512)
    ids, mask, lengths = _batch([9, 30], 32)
    enc = FusedBertEncoder(model)
    a16 = enc.forward(ids, lengths, dtype=torch.float16).clone()
    abf = enc.forward(ids, lengths, dtype=torch.bfloat16).clone()
    assert enc.refresh() is False and set(enc._layers) == {torch.float16, torch.bfloat16}
    assert enc._layers[torch.float16][0].wqkv.dtype == torch.float16
    with torch.no_grad():
        model.encoder.layer[0].output.dense.weight.mul_(3.0)          # (a uniform shift would vanish in the LayerNorm behind it)
    assert enc.refresh() is True and not enc._layers                   # every set dropped; each is rebuilt on its next use
    b16 = enc.forward(ids, lengths, dtype=torch.float16)
    assert (a16 - b16).abs().max().item() > 1e-2
    assert (a16 - abf).abs().max().item() < 5e-2
    with pytest.raises(AssertionError):
        enc.forward(ids, lengths, dtype=torch.float32)
