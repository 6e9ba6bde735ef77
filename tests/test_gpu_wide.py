"""The 256 x 384 form of the main pass (gemm_topk16w_kernel; planner's choice, CCR_WIDE = 0 / 1 pins it): exact at every shape --
partly filled query blocks, a partial last corpus tile, one K step per tile, phased launches with re-tightened thresholds, estimated
thresholds, flooded candidate lists -- and bit-equal to the 256 x 256 kernel's results."""
import os
from contextlib import contextmanager

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

DENSE, FUSED = 1, 2


def _bf16(bits):
    return torch.from_numpy(bits.view(np.int16)).view(torch.bfloat16).cuda()


def _rand_bits(n, d, seed, scale=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, d, generator=g) * (scale if scale is not None else d ** -0.5)
    return orc.pack_bf16(x.numpy())


@contextmanager
def _knobs(**kv):
    """Environment knobs are read ONCE, when an index is created."""
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update({k: str(v) for k, v in kv.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _index(Db, wide, **kv):
    from ccrec_amd import ops
    with _knobs(CCR_WIDE=wide, **kv):
        return ops.CorpusIndex(_bf16(Db), global_row_offset=4321)


@pytest.mark.parametrize("n,nq,d,k", [
    (20000, 300, 768, 100),    # one partly filled block (300 of 384 rows), partial last corpus tile
    (30001, 513, 1024, 257),   # two blocks, the second with 129 rows
    (25600, 384, 64, 10),      # exactly one block, whole tiles, two K steps per tile
    (9000, 130, 32, 5),        # ONE K step per tile
    (40000, 1153, 96, 50),     # four blocks, the last with a single row
    (40000, 770, 128, 1001),   # large k
    (300, 200, 32, 10),        # ONE full corpus tile + 44 rows, one K step: an item is a single K step (more ranges than tiles)
    (513, 385, 64, 5),         # two tiles and one row; a second query block of one row
    (2048, 384, 32, 16),       # eight whole tiles, one whole block
])
def test_wide_main_pass_equals_the_oracle(n, nq, d, k):
    Db, Qb = _rand_bits(n, d, n + 3), _rand_bits(nq, d, nq + 5)
    index = _index(Db, 1)
    s, i = index.search(_bf16(Qb), k, FUSED)
    st = index.last_stats()
    print(st)
    assert st["path"] == 1 and st["main_tile_queries"] == 384, st
    ref_i, ref_s = orc.canonical_search(Qb, Db, k, None)
    assert np.array_equal(i.cpu().numpy() - 4321, ref_i), f"ids differ in {np.sum(i.cpu().numpy() - 4321 != ref_i)} places"
    assert np.array_equal(s.cpu().numpy().view(np.uint32), ref_s.view(np.uint32))


@pytest.mark.parametrize("n,nq,d,k,extra", [
    (400_000, 700, 768, 100, {}),                          # phased launches: thresholds re-tightened between them
    (400_000, 700, 768, 100, {"CCR_PROGRESSIVE": 0}),      # single launch
    (300_000, 1100, 256, 1001, {}),                        # estimated thresholds (large k)
    (250_003, 3452, 128, 100, {}),                         # nine blocks, several items per workgroup, partial last tile
    (200_000, 390, 768, 10, {"CCR_QGROUPS": 2}),           # two blocks as two query groups over the XCDs
    (150_000, 3072, 768, 100, {}),                         # eight blocks: the planner's own two groups of four (4.7 MiB would not fit one L2)
])
def test_wide_main_pass_equals_the_narrow_tile_and_the_dense_path(n, nq, d, k, extra):
    Db, Qb = _rand_bits(n, d, n + 11), _rand_bits(nq, d, nq + 13)
    Q = _bf16(Qb)
    wide = _index(Db, 1, **extra)
    s, i = wide.search(Q, k, FUSED)
    st = wide.last_stats()
    print(st)
    assert st["path"] == 1 and st["main_tile_queries"] == 384 and st["n_fallback"] == 0, st
    if "CCR_PROGRESSIVE" in extra:
        assert st["main_launches"] == 1, st
    ref = _index(Db, 0, **extra)
    s0, i0 = ref.search(Q, k, FUSED)
    st0 = ref.last_stats()
    assert st0["path"] == 1 and st0["main_tile_queries"] == 256, st0
    assert torch.equal(i, i0) and torch.equal(s.view(torch.int32), s0.view(torch.int32))
    s2, i2 = wide.search(Q, k, DENSE)
    assert torch.equal(i, i2) and torch.equal(s.view(torch.int32), s2.view(torch.int32))
    sub = np.r_[0:4, nq // 2:nq // 2 + 4, nq - 4:nq]
    ref_i, ref_s = orc.canonical_search(Qb[sub], Db, k, None)
    assert np.array_equal(i.cpu().numpy()[sub] - 4321, ref_i) and np.array_equal(s.cpu().numpy()[sub].view(np.uint32), ref_s.view(np.uint32))


def test_planner_takes_the_wide_tile_where_its_padding_pays():
    """3 452 queries: nine blocks of 384 (3 456 columns) instead of fourteen of 256 (3 584); 512 queries stay on two 256-blocks; a small
    batch keeps the streaming kernel; query blocks that no XCD's L2 can hold keep the 256 x 256 kernel (its blocks split into groups);
    a dim that is no multiple of 32 keeps the 256 x 256 kernel with its zero-filled last K step."""
    from ccrec_amd import ops
    Db = _rand_bits(70_000, 64, 1)
    index = ops.CorpusIndex(_bf16(Db))
    for nq, want in ((3452, 384), (300, 384), (512, 256), (1000, 256), (1100, 384), (64, 0)):
        index.search(_bf16(_rand_bits(nq, 64, nq)), 10, FUSED)
        st = index.last_stats()
        assert st["path"] == 1 and st["main_tile_queries"] == want, (nq, st)
    # 768-dim rows: nine blocks (5 MiB of query rows per XCD) still pay, eleven -- a prime count cannot be split over the XCDs -- do not
    index = ops.CorpusIndex(_bf16(_rand_bits(40_000, 768, 4)))
    for nq, want in ((3452, 384), (4096, 256), (1100, 384)):
        index.search(_bf16(_rand_bits(nq, 768, nq)), 10, FUSED)
        st = index.last_stats()
        assert st["path"] == 1 and st["main_tile_queries"] == want, (nq, st)
    Db = _rand_bits(30_000, 72, 2)
    index = _index(Db, 1)
    Qb = _rand_bits(300, 72, 3)
    s, i = index.search(_bf16(Qb), 10, FUSED)
    assert index.last_stats()["main_tile_queries"] == 256
    ref_i, ref_s = orc.canonical_search(Qb, Db, 10, None)
    assert np.array_equal(i.cpu().numpy() - 4321, ref_i) and np.array_equal(s.cpu().numpy().view(np.uint32), ref_s.view(np.uint32))


def test_wide_main_pass_on_a_corpus_in_topical_order_retries_exactly():
    """Clustered rows in cluster order flood the sub-lists of the queries of that cluster: overflowing lists flag the query, the retry
    pass (256 x 256 tiles, the whole candidate area) and the dense path finish it -- the same answers as without the wide tile."""
    g = torch.Generator().manual_seed(21)
    n, d, nq, k, nc = 120_000, 128, 400, 100, 40
    centres = torch.randn(nc, d, generator=g)
    lab = torch.arange(n) * nc // n                      # rows sorted by cluster
    D = centres[lab] + 0.05 * torch.randn(n, d, generator=g)
    Q = centres[torch.arange(nq) % nc] + 0.05 * torch.randn(nq, d, generator=g)
    Db, Qb = orc.pack_bf16(D.numpy()), orc.pack_bf16(Q.numpy())
    wide, ref = _index(Db, 1), _index(Db, 0)
    s, i = wide.search(_bf16(Qb), k, FUSED)
    st = wide.last_stats()
    print(st)
    assert st["main_tile_queries"] == 384
    s0, i0 = ref.search(_bf16(Qb), k, FUSED)
    assert torch.equal(i, i0) and torch.equal(s.view(torch.int32), s0.view(torch.int32))
    sub = np.r_[0:6, 200:206]
    ref_i, ref_s = orc.canonical_search(Qb[sub], Db, k, None)
    assert np.array_equal(i.cpu().numpy()[sub] - 4321, ref_i) and np.array_equal(s.cpu().numpy()[sub].view(np.uint32), ref_s.view(np.uint32))
