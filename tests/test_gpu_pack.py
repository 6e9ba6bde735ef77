"""GPU parity: pack / normalise / mean-pool kernels through the C ABI vs the oracle and the golden vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def test_pack_matches_golden_and_torch(golden_dir):
    from ccrec_amd import ops
    g = np.load(os.path.join(golden_dir, "g9_pack_bf16.npz"))
    x = torch.from_numpy(g["x"]).cuda()
    out = ops.pack_bf16(x)
    assert np.array_equal(_bits(out), g["bits"])          # RNE incl. +-0, inf, subnormals, exact half-way cases
    big = torch.randn(1001, 776, device="cuda") * 3        # rows*dim not a multiple of 8 -> tail kernel
    assert torch.equal(ops.pack_bf16(big).view(torch.int16), big.to(torch.bfloat16).view(torch.int16))
    nan = torch.tensor([[float("nan"), 1.0, -float("nan"), 2.0] * 2], device="cuda")
    assert torch.isnan(ops.pack_bf16(nan).float()).cpu().tolist() == [[True, False, True, False] * 2]


@pytest.mark.parametrize("rows,dim", [(7, 768), (300, 1024), (65, 64), (5, 772), (9, 301), (3, 5)])
def test_normalize_pack_bit_exact_vs_oracle(rows, dim):
    """(772, 301, 5: widths that are no multiple of 8 are stored zero-padded to the next one; the canonical sum of squares treats
    a partial last chunk as zero-padded, so the bits of the first `dim` columns are the oracle's.)"""
    from ccrec_amd import ops
    g = torch.Generator().manual_seed(rows * dim)
    x = torch.randn(rows, dim, generator=g) * torch.rand(rows, 1, generator=g) * 5
    x[0] = 0  # zero row: x / max(0, 1e-12) = 0
    out, norms = ops.pack_bf16(x.cuda(), normalize=True, return_norms=True)
    assert out.shape == (rows, ops.padded_dim(dim)) and not _bits(out)[:, dim:].any()
    assert np.array_equal(_bits(out)[:, :dim], orc.normalize_pack_bf16(x.numpy()))
    assert np.array_equal(norms.cpu().numpy(), orc.row_norms(x.numpy()))
    plain = ops.pack_bf16(x.cuda())                                  # and the plain pack of an odd width
    assert np.array_equal(_bits(plain)[:, :dim], orc.pack_bf16(x.numpy())) and not _bits(plain)[:, dim:].any()
    # and it is the reference's F.normalize up to bf16 rounding
    ref = torch.nn.functional.normalize(x, p=2, dim=1)
    np.testing.assert_allclose(out.float().cpu().numpy()[:, :dim], ref.numpy(), atol=4e-3, rtol=8e-3)


def test_pack_into_shard_slice():
    from ccrec_amd import ops
    shard = torch.zeros(10, 768, dtype=torch.bfloat16, device="cuda")
    x = torch.randn(4, 768, device="cuda")
    ops.pack_bf16(x, out=shard[3:7])
    assert torch.equal(shard[3:7], x.to(torch.bfloat16)) and float(shard[:3].abs().sum()) == 0


def test_meanpool_golden(golden_dir):
    from ccrec_amd import ops
    g = np.load(os.path.join(golden_dir, "g6_item_tower.npz"))
    hidden, mask = torch.from_numpy(g["hidden"]).cuda(), torch.from_numpy(g["mask"]).cuda()
    f32, b16 = ops.meanpool_pack(hidden, mask)
    # fp32 adds in token order == the oracle's order: bit-exact; reference (torch sum) within fp32 noise
    assert np.array_equal(f32.cpu().numpy(), orc.meanpool(g["hidden"], g["mask"]))
    np.testing.assert_allclose(f32.cpu().numpy(), g["mean_pooling"], rtol=2e-6, atol=2e-6)
    assert np.array_equal(_bits(b16), orc.pack_bf16(f32.cpu().numpy()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_meanpool_ragged_dtypes(dtype):
    from ccrec_amd import ops
    g = torch.Generator().manual_seed(5)
    B, L, d = 9, 37, 1024
    hidden = torch.randn(B, L, d, generator=g).to(dtype)
    mask = (torch.arange(L)[None, :] < torch.randint(1, L + 1, (B, 1), generator=g)).long()
    f32, b16 = ops.meanpool_pack(hidden.cuda(), mask.cuda())
    ref = orc.meanpool(hidden.float().numpy(), mask.numpy())
    assert np.array_equal(f32.cpu().numpy(), ref)
    _, b16n = ops.meanpool_pack(hidden.cuda(), mask.cuda(), normalize=True, want_f32=False)
    refn = ref / np.maximum(np.linalg.norm(ref.astype(np.float64), axis=1, keepdims=True), 1e-12)
    np.testing.assert_allclose(b16n.float().cpu().numpy(), refn, atol=4e-3, rtol=8e-3)


def test_item_tower_mirror(golden_dir):
    """NaiveItemTower with a fake encoder returning the golden hidden state."""
    import types
    from ccrec_amd.item_tower import NaiveItemTower
    g = np.load(os.path.join(golden_dir, "g6_item_tower.npz"))
    hidden = torch.from_numpy(g["hidden"]).cuda()

    class FakeCls(torch.nn.Module):
        device = torch.device("cuda")

        def forward(self, **inputs):
            return types.SimpleNamespace(last_hidden_state=hidden)

    tower = NaiveItemTower(FakeCls(), torch.nn.LayerNorm(768, elementwise_affine=False).cuda())
    inputs = {"input_ids": torch.ones(3, 16, dtype=torch.long), "attention_mask": torch.from_numpy(g["mask"])}
    with torch.no_grad():
        np.testing.assert_allclose(tower(**inputs, output_step="mean_pooling").cpu().numpy(), g["mean_pooling"], rtol=2e-6, atol=2e-6)
        np.testing.assert_array_equal(tower(**inputs, output_step="cls").cpu().numpy(), g["cls"])
        np.testing.assert_allclose(tower(**inputs, output_step="mean_layer_norm").cpu().numpy(), g["mean_layer_norm"], rtol=1e-5, atol=1e-5)
        packed = tower(**inputs, output_step="mean_pooling_bf16")
        assert packed.dtype == torch.bfloat16 and np.array_equal(_bits(packed), orc.pack_bf16(orc.meanpool(g["hidden"], g["mask"])))
        with pytest.raises(NotImplementedError):
            tower(**inputs, output_step="nonsense")


def test_pack_norm_bounds_and_index_with_norms():
    """ccr_pack_bf16_ex leaves an upper bound of every packed row's norm, batch by batch; an index built
    from them (no pass over the shard) must search identically."""
    from ccrec_amd import ops
    g = torch.Generator().manual_seed(17)
    x = torch.randn(5000, 768, generator=g) * torch.rand(5000, 1, generator=g) * 3
    shard = torch.empty(5000, 768, dtype=torch.bfloat16, device="cuda")
    nb = torch.empty(5000, device="cuda")
    for lo in range(0, 5000, 1024):            # batch by batch, as generate_embeddings does
        ops.pack_bf16(x[lo:lo + 1024].cuda(), out=shard[lo:lo + 1024], norm_bounds=nb[lo:lo + 1024])
    assert torch.equal(shard.view(torch.int16), x.cuda().to(torch.bfloat16).view(torch.int16))
    true = orc.row_norms_bf16(_bits(shard)).astype(np.float64)
    got = nb.cpu().numpy().astype(np.float64)
    assert np.all(true <= got) and np.all(got <= true * 1.001)
    nbn = torch.empty(5000, device="cuda")
    ops.pack_bf16(x.cuda(), normalize=True, norm_bounds=nbn)
    assert 1.0 <= float(nbn.min()) and float(nbn.max()) <= 1.01
    q = ops.pack_bf16(torch.randn(40, 768, generator=g).cuda())
    a = ops.CorpusIndex(shard).search(q, 50, 2)
    b = ops.CorpusIndex(shard, norm_bounds=nb).search(q, 50, 2)
    assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0])
