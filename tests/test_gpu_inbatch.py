"""GPU parity of the in-batch-negative contrastive loss (bbpr.py:205-212) vs the oracle and the golden
vectors produced by the reference's own training_and_validation_step."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _bf(x):
    return torch.from_numpy(np.asarray(x, np.float32)).to(torch.bfloat16)


@pytest.mark.parametrize("tag,B,sim", [("b8_dot", 8, "dot"), ("b32_dot", 32, "dot"), ("b8_cos", 8, "cos"), ("b32_cos", 32, "cos")])
def test_loss_and_grads_vs_reference_golden(golden_dir, tag, B, sim):
    from ccrec_amd.bbpr_loss import multiple_nrl_loss
    g = np.load(os.path.join(golden_dir, "g7_contrastive.npz"))
    E = torch.from_numpy(g[f"{tag}_E"]).cuda().requires_grad_(True)
    loss = multiple_nrl_loss(E[:B], E[B:2 * B], E[2 * B:], inv_temperature=20.0, sim_type=sim)
    loss.backward()
    # bf16 operands (the reference ran fp32): logits differ by ~inv_T * 2^-9 * |q||k|
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-3 * max(1.0, abs(float(g[f"{tag}_loss"])))
    ref = g[f"{tag}_grad"]
    got = E.grad.cpu().numpy()
    assert np.abs(got - ref).max() < 1e-2 * np.abs(ref).max()
    # against the oracle on the SAME bf16-rounded operands: tight
    Eb = E.detach()
    if sim == "cos":
        Eb = torch.nn.functional.normalize(Eb, p=2, dim=1)
    Eb = Eb.to(torch.bfloat16).float().cpu().numpy()
    l2, dQ, dP, dN = orc.inbatch_ce(Eb[:B], Eb[B:2 * B], Eb[2 * B:], 20.0, "dot")
    assert abs(float(loss) - l2) < 2e-5 * max(1.0, abs(l2))


@pytest.mark.parametrize("B,dim", [(1024, 768), (100, 64), (33, 1024), (1, 16)])
def test_fwd_bwd_vs_oracle(B, dim):
    from ccrec_amd import ops
    g = torch.Generator().manual_seed(B)
    q, p, n = (_bf(torch.randn(B, dim, generator=g) / dim ** 0.5) for _ in range(3))
    qc, pc, nc = (t.cuda().float().requires_grad_(True) for t in (q, p, n))
    inv_t = 20.0
    loss = ops.inbatch_ce(qc, pc, nc, inv_t)
    (loss * 3.0).backward()           # grad_out != 1
    l2, dQ, dP, dN = orc.inbatch_ce(q.float().numpy(), p.float().numpy(), n.float().numpy(), inv_t, "dot")
    assert abs(float(loss) - l2) < 2e-5 * max(1.0, abs(l2))
    for got, ref in ((qc.grad, dQ), (pc.grad, dP), (nc.grad, dN)):
        np.testing.assert_allclose(got.cpu().numpy(), 3.0 * ref, rtol=2e-4, atol=3e-6 * max(1e-6, np.abs(ref).max()) * 100)
    # deterministic: same bits on a second run
    loss2 = ops.inbatch_ce(qc.detach(), pc.detach(), nc.detach(), inv_t)
    assert float(loss2) == float(loss)


@pytest.mark.parametrize("tag,B,sim", [("b8_dot", 8, "dot"), ("b32_cos", 32, "cos")])
def test_training_step_mirror_vs_reference_golden(golden_dir, tag, B, sim):
    """The whole training_and_validation_step (bbpr.py:149-214, multiple_nrl): fixture g7 was produced by calling the
    reference's method with a table-lookup forward; the mirror is set up the same way (SURVEY 8b signature)."""
    from ccrec_amd.bbpr_loss import MultipleNrlStep, compute_user_to_negatives
    g = np.load(os.path.join(golden_dir, "g7_contrastive.npz"))
    os.environ["CCREC_SIM_TYPE"] = sim
    os.environ["CCREC_BBPR_INV_TEMPERATURE"] = "20"
    E = torch.from_numpy(g[f"{tag}_E"]).cuda().requires_grad_(True)
    # hard negatives from a sparse prior, as compute_user_to_negatives builds them: user u -> item B + u (value 1)
    prior = torch.sparse_coo_tensor(torch.stack([torch.arange(B), B + torch.arange(B)]), torch.ones(B), (B, 2 * B))
    negs = compute_user_to_negatives(prior)
    assert negs == {u: [B + u] for u in range(B)}
    step = MultipleNrlStep(lambda ptr: E[ptr], torch.arange(0, B), torch.arange(B, 3 * B), negs)
    batch = torch.stack([torch.arange(B), torch.arange(B), torch.ones(B, dtype=torch.long)], 1)
    loss = step.training_and_validation_step(batch, 0)
    loss.backward()
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-3 * max(1.0, abs(float(g[f"{tag}_loss"])))
    ref = g[f"{tag}_grad"]
    assert np.abs(E.grad.cpu().numpy() - ref).max() < 1e-2 * np.abs(ref).max()
    assert negs == {u: [B + u] for u in range(B)}          # round robin over a single negative leaves the list as it was


def test_bert_mt_step_weighting_and_one_optimizer_step():
    """_BertMT.training_and_validation_step (bert_mt.py:105-113): alpha / ft_cycles times the multiple_nrl loss; one optimiser
    step (stock torch AdamW: the reference's optimiser set-up, bert_mt.py:115-134, is out of scope) through the HIP loss moves the table."""
    import os
    from ccrec_amd.bbpr_loss import BertMTStep, MultipleNrlStep
    os.environ["CCREC_SIM_TYPE"] = "dot"
    os.environ["CCREC_BBPR_INV_TEMPERATURE"] = "5"
    torch.manual_seed(0)
    table = torch.nn.Embedding(40, 64).cuda()
    with torch.no_grad():
        table.weight.mul_(0.2)
    i_to_ptr = torch.arange(0, 20)
    j_to_ptr = torch.arange(20, 40)
    negs = lambda: {u: [(u + 3) % 20, (u + 7) % 20] for u in range(20)}   # noqa: E731
    batch = torch.stack([torch.arange(16), (torch.arange(16) * 3) % 20, torch.ones(16)], 1).long()
    fwd = lambda ptr: table(ptr.cuda())                                   # noqa: E731
    base = MultipleNrlStep(fwd, i_to_ptr, j_to_ptr, negs())(batch)
    mt = BertMTStep(fwd, i_to_ptr, j_to_ptr, negs(), alpha=0.25, ct_cycles=3, ft_cycles=2)((batch, None))
    assert abs(float(mt) - 0.25 / 2 * float(base)) < 1e-6
    opt = torch.optim.AdamW(table.parameters(), lr=1e-2, weight_decay=0.01)
    before = table.weight.detach().clone()
    opt.zero_grad()
    mt.backward()
    opt.step()
    assert float((table.weight - before).abs().max()) > 0


def test_pack3_bits_gradient_determinism_and_odd_inputs():
    """ccr_inbatch_pack3_bf16 = torch's .to(bfloat16) bits for the three blocks in one launch (incl. NaN, inf, denormals, ties to even);
    two backward passes of the same step give the same gradient bits; inputs the one-launch pack cannot take (fp16 / non-contiguous)
    go through the copy path and agree with the fp32 contiguous call on the same values."""
    import ctypes
    from ccrec_amd import _lib, ops
    lib = ops.require_gpu()
    B, dim = 70, 48
    g = torch.Generator().manual_seed(1)
    blocks = [torch.randn(B, dim, generator=g) for _ in range(3)]
    blocks[0][0, :6] = torch.tensor([float("nan"), float("inf"), -float("inf"), 1e-40, 1.00390625, 1.01171875])   # the last two: ties
    dev = [b.cuda().contiguous() for b in blocks]
    out = torch.empty(3, B, dim, dtype=torch.bfloat16, device="cuda")
    vp = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    _lib.check(lib.ccr_inbatch_pack3_bf16(vp(dev[0]), vp(dev[1]), vp(dev[2]), B, dim, vp(out), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "pack3")
    ref = torch.stack([d.to(torch.bfloat16) for d in dev])
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))

    q, p, n = (torch.randn(B, dim, generator=g).cuda() * dim ** -0.5 for _ in range(3))
    grads = []
    for _ in range(2):
        a, b, c = (t.clone().requires_grad_(True) for t in (q, p, n))
        ops.inbatch_ce(a, b, c, 20.0).backward()
        grads.append(torch.stack([a.grad, b.grad, c.grad]))
    assert torch.equal(grads[0].view(torch.int32), grads[1].view(torch.int32))
    # fp16 inputs / a transposed view: the same rounded operands through the copy path
    qh = q.to(torch.bfloat16).float()
    a, b, c = qh.half().requires_grad_(True), p.t().contiguous().t().requires_grad_(True), n.clone().requires_grad_(True)
    assert not b.is_contiguous()
    ops.inbatch_ce(a, b, c, 20.0).backward()
    a2, b2, c2 = (t.clone().requires_grad_(True) for t in (qh.half().float(), p, n))
    ops.inbatch_ce(a2, b2, c2, 20.0).backward()
    assert a.grad.dtype == torch.float16 and torch.allclose(a.grad.float(), a2.grad, rtol=1e-2, atol=1e-6)
    assert torch.equal(b.grad, b2.grad) and torch.equal(c.grad, c2.grad)


def test_backward_rejects_a_workspace_that_is_not_its_forwards():
    """The backward reads the forward's logits from the workspace.  A workspace no forward (or another shape's / temperature's forward)
    has filled carries no matching stamp: the gradients come back NaN -- loudly wrong, not plausible garbage -- and a matching one works."""
    import ctypes
    from ccrec_amd import _lib, ops
    lib = ops.require_gpu()
    B, dim = 64, 32
    g = torch.Generator().manual_seed(2)
    qb, pb, nb = (torch.randn(B, dim, generator=g).cuda().to(torch.bfloat16) for _ in range(3))
    loss, lse = torch.empty(1, device="cuda"), torch.empty(B, device="cuda")
    grads = torch.empty(3, B, dim, device="cuda")
    need = int(lib.ccr_inbatch_ce_workspace_bytes(B, dim))
    ws, scratch = (torch.zeros(need, dtype=torch.uint8, device="cuda") for _ in range(2))
    vp = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def bwd(w, inv_t):
        grads.zero_()
        _lib.check(lib.ccr_inbatch_ce_bwd(vp(qb), vp(pb), vp(nb), vp(lse), B, dim, inv_t, 1.0, vp(grads[0]), vp(grads[1]), vp(grads[2]), vp(w), need, stream), "bwd")
        torch.cuda.synchronize()
        return grads.clone()

    _lib.check(lib.ccr_inbatch_ce_fwd(vp(qb), vp(pb), vp(nb), B, dim, 20.0, vp(loss), vp(lse), vp(ws), need, stream), "fwd")
    good = bwd(ws, 20.0)
    assert torch.isfinite(good).all() and good.abs().max() > 0
    assert torch.isnan(bwd(scratch, 20.0)).all()          # never written by a forward
    assert torch.isnan(bwd(ws, 10.0)).all()               # another temperature's logits
    _lib.check(lib.ccr_inbatch_ce_fwd(vp(qb), vp(pb), vp(nb), B, dim, 10.0, vp(loss), vp(lse), vp(ws), need, stream), "fwd")
    assert torch.isnan(bwd(ws, 20.0)).all()               # the workspace has been reused by another forward since
    assert torch.isfinite(bwd(ws, 10.0)).all()


def test_fragment_major_forward_equals_the_row_major_forward(monkeypatch):
    """B % 32 == 0 and dim % 128 == 0: the fp32 entry point packs a fragment-major copy of [Q ; P ; N] and the forward reads its operands
    as contiguous KiB blocks; CCR_INBATCH_ROWMAJOR=1 keeps the row-major loads.  Same contraction order: the same loss and gradient bits."""
    from ccrec_amd import ops
    g = torch.Generator().manual_seed(5)
    q, p, n = (torch.randn(96, 256, generator=g).cuda() * 256 ** -0.5 for _ in range(3))
    out = []
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("CCR_INBATCH_ROWMAJOR", env)
        a, b, c = (t.clone().requires_grad_(True) for t in (q, p, n))
        loss = ops.inbatch_ce(a, b, c, 20.0)
        loss.backward()
        out.append((loss.detach().clone(), torch.stack([a.grad, b.grad, c.grad])))
    assert torch.equal(out[0][0].view(torch.int32), out[1][0].view(torch.int32))
    assert torch.equal(out[0][1].view(torch.int32), out[1][1].view(torch.int32))
