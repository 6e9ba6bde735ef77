"""Seeded fuzz of the search planner and kernels: random (rows, queries, dim, k, offset) shapes with ragged
sizes; the default path, the forced fused path and the exact dense path must agree bit for bit, and small
cases are checked against the CPU oracle.  Includes data with heavy exact ties (quantised embeddings)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _case(seed):
    rs = np.random.RandomState(seed)
    n = int(rs.choice([rs.randint(260, 3000), rs.randint(3000, 60000), rs.randint(60000, 220000)]))
    nq = int(rs.choice([1, rs.randint(2, 40), rs.randint(40, 700)]))
    d = int(rs.choice([8, 24, 32, 40, 64, 96, 128, 160, 200, 300, 384, 768, 1024]))   # 300: not a multiple of 8 (zero-padded by the pack)
    k = int(min(n, rs.choice([1, rs.randint(2, 30), rs.randint(30, 300), rs.randint(300, 1300)])))
    off = int(rs.choice([0, 7, 1 << 33]))
    quant = bool(rs.rand() < 0.3)
    wild = bool(rs.rand() < 0.35)   # rows / queries with norms up to 1000x the rest (the filter margins are per tile / per row)
    return n, nq, d, k, off, quant, wild


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("CCR_FUZZ_SEEDS", "24")))))   # soak: CCR_FUZZ_SEEDS=400
def test_random_shapes_all_paths_agree(seed):
    from ccrec_amd import ops
    n, nq, d, k, off, quant, wild = _case(seed)
    g = torch.Generator().manual_seed(seed)
    D = torch.randn(n, d, generator=g) / d ** 0.5
    Q = torch.randn(nq, d, generator=g) / d ** 0.5
    if quant:   # coarse grid -> many exactly equal scores
        D, Q = torch.round(D * 8) / 8, torch.round(Q * 8) / 8
    if wild:
        rows = torch.randint(0, n, (max(1, n // 5000),), generator=g)
        D[rows] *= 10.0 ** (3.0 * torch.rand(rows.numel(), 1, generator=g))
        Q[torch.randint(0, nq, (1,), generator=g)] *= 64.0
    nb = torch.empty(n, device="cuda") if seed % 2 else None   # odd seeds: the pack kernel's norm bounds; even: the index's own pass
    Db, Qb = ops.pack_bf16(D.cuda(), norm_bounds=nb), ops.pack_bf16(Q.cuda())
    index = ops.CorpusIndex(Db, global_row_offset=off, norm_bounds=nb)
    s0, i0 = index.search(Qb, k, 0)
    st0 = index.last_stats()
    s1, i1 = index.search(Qb, k, 1)
    assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32)), (n, nq, d, k, st0)
    if n >= 256:   # every width runs the fused kernels (the last 32-element K step zero-filled when dim % 32 != 0)
        s2, i2 = index.search(Qb, k, 2)
        st2 = index.last_stats()
        if st2["path"] == 1:
            assert torch.equal(i2, i1) and torch.equal(s2.view(torch.int32), s1.view(torch.int32)), (n, nq, d, k, st2)
    assert Db.shape[1] == (d + 7) // 8 * 8 and (d % 8 == 0 or not Db[:, d:].any())          # odd widths: zero tail
    if n * nq <= 3_000_000:
        bits = lambda t: t.view(torch.int16).cpu().numpy().view(np.uint16)   # noqa: E731
        ref_i, ref_s = orc.canonical_search(bits(Qb)[:, :d], bits(Db)[:, :d], k)                # the oracle sees the unpadded rows
        assert np.array_equal(i1.cpu().numpy() - off, ref_i) and np.array_equal(s1.cpu().numpy(), ref_s)


@pytest.mark.parametrize("seed", [3, 5, 11, 17, 21])
def test_mfma_32x32_variant_agrees(seed, monkeypatch):
    """The main pass runs on v_mfma_f32_16x16x32_bf16 by default; CCR_MFMA16=0 selects the 32x32x16 kernel (4 instead of
    8 candidate sub-lists per (range, query)).  Both must return the canonical result."""
    from ccrec_amd import ops
    rs = np.random.RandomState(seed)
    n, nq, d, k = int(rs.randint(20_000, 180_000)), int(rs.randint(1, 600)), int(rs.choice([64, 384, 768])), int(rs.randint(1, 400))
    g = torch.Generator().manual_seed(seed)
    Db = ops.pack_bf16((torch.randn(n, d, generator=g) / d ** 0.5).cuda())
    Qb = ops.pack_bf16((torch.randn(nq, d, generator=g) / d ** 0.5).cuda())
    s_ref, i_ref = ops.CorpusIndex(Db).search(Qb, k, 1)           # exact dense path
    for variant in ("0", "1"):
        monkeypatch.setenv("CCR_MFMA16", variant)
        index = ops.CorpusIndex(Db)                                 # the plan is cached per index: a fresh one per variant
        s, i = index.search(Qb, k, 2)
        assert index.last_stats()["path"] == 1
        assert torch.equal(i, i_ref) and torch.equal(s.view(torch.int32), s_ref.view(torch.int32)), (variant, n, nq, d, k)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_medium_shapes_multi_phase_plans_agree_with_dense(seed):
    """Shapes large enough for the planner's multi-phase main pass (several rounds of work items, per-phase candidate
    capacities, item-granular phase ends): the fused result of EVERY query block's first rows must equal the exact dense path
    bit for bit.  Odd seeds use the topically sorted clustered corpus of bench.py (overflow -> retry path)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import gen_rows
    from ccrec_amd import ops
    rs = np.random.RandomState(100 + seed)
    n = int(rs.randint(600_000, 2_000_000))
    nq = int(rs.choice([300, 1500, 3000]))
    d = int(rs.choice([128, 768]))
    k = int(rs.choice([10, 100, 400, 1001]))
    data = "sorted" if seed % 2 else "gaussian"
    D = ops.pack_bf16(gen_rows(n, d, 1234, "cuda", data))
    Q = ops.pack_bf16(gen_rows(nq, d, 4321, "cuda", data))
    index = ops.CorpusIndex(D, global_row_offset=int(rs.choice([0, 1 << 34])))
    s, i = index.search(Q, k)
    st = index.last_stats()
    assert st["path"] == 1, st
    pick = torch.cat([torch.arange(b, min(nq, b + 6)) for b in range(0, nq, 256)]).cuda()   # rows of every query block
    s1, i1 = index.search(Q[pick], k, 1)
    assert torch.equal(i[pick], i1) and torch.equal(s[pick].view(torch.int32), s1.view(torch.int32)), (n, nq, d, k, data, st)


@pytest.mark.parametrize("d", [8, 24, 40, 200, 300, 301, 50, 7])
@pytest.mark.parametrize("sim", ["dot", "cos"])
def test_any_embedding_width_takes_the_fused_path(d, sim):
    """The reference takes any factor width (rime_lite score_array.py:320-339, bbpr.py:536-540).  Widths that are not multiples
    of 32 run the fused MFMA kernels with a zero-filled last K step; widths that are not multiples of 8 are zero-padded by the
    pack kernel (its normalising form keeps the canonical sum order).  Fused == fp64 dense == oracle, bit for bit; the MFMA score
    matrix agrees with the canonical one to fp32 accumulation noise."""
    from ccrec_amd import ops
    n, nq, k = 70_000, 300, 100
    g = torch.Generator().manual_seed(d)
    D = torch.randn(n, d, generator=g) / d ** 0.5
    Q = torch.randn(nq, d, generator=g) / d ** 0.5
    nb = torch.empty(n, device="cuda")
    Db, Qb = ops.pack_bf16(D.cuda(), normalize=(sim == "cos"), norm_bounds=nb), ops.pack_bf16(Q.cuda(), normalize=(sim == "cos"))
    pd = ops.padded_dim(d)
    assert Db.shape == (n, pd) and Qb.shape == (nq, pd)
    bits = lambda t: t.view(torch.int16).cpu().numpy().view(np.uint16)   # noqa: E731
    ref_D = orc.pack(D.numpy(), sim)
    assert np.array_equal(bits(Db)[:, :d], ref_D) and not bits(Db)[:, d:].any()      # packed rows == oracle's, tail zero
    for index in (ops.CorpusIndex(Db, norm_bounds=nb), ops.CorpusIndex(Db)):
        s2, i2 = index.search(Qb, k, 2)
        st = index.last_stats()
        assert st["path"] == 1 and st["n_fallback"] == 0, st                           # the fused MFMA path, nothing flagged
        s1, i1 = index.search(Qb, k, 1)
        assert torch.equal(i2, i1) and torch.equal(s2.view(torch.int32), s1.view(torch.int32))
    ref_i, ref_s = orc.canonical_search(orc.pack(Q.numpy(), sim)[:40], ref_D, k)
    assert np.array_equal(i2.cpu().numpy()[:40], ref_i) and np.array_equal(s2.cpu().numpy()[:40].view(np.uint32), ref_s.view(np.uint32))
    small = ops.CorpusIndex(Db[:3000].contiguous())                                    # margin path of a small corpus: MFMA score rows
    s3, i3 = small.search(Qb[:50], 20)
    s4, i4 = small.search(Qb[:50], 20, 1)
    assert torch.equal(i3, i4) and torch.equal(s3.view(torch.int32), s4.view(torch.int32))
    mf, can = small.scores(Qb[:50], "mfma"), small.scores(Qb[:50], "canonical")
    assert float((mf - can).abs().max()) < 2e-6 * max(1.0, float(can.abs().max()))


@pytest.mark.parametrize("case", ["planner_k1001", "forced_k100", "tiny_rank", "sorted_forced", "clustered_tiny_rank"])
def test_estimated_thresholds_are_verified_and_exact(case, monkeypatch):
    """Large k filters under ESTIMATED thresholds (the r-th largest sampled group maximum, r = max(48, 3 k fs): about 3 k rows pass
    over the whole corpus in one launch, no bound, no re-tightening); the select stage verifies L >= tau per query and a query
    that fails is retried under the valid bound its candidates give.  Exact in every case: the planner's own choice at k = 1001;
    the mode forced at k = 100; a rank pinned so low (2) that the estimate is far too high and most queries fail the check; a
    corpus in topical order, where the sample holds some queries' whole cluster tile; clustered data with the tiny rank."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import gen_rows
    from ccrec_amd import ops
    n, nq, d, k, data = 400_000, 600, 128, 1001, "gaussian"
    if case == "forced_k100":
        k = 100
        monkeypatch.setenv("CCR_OPTIMISTIC", "1")
    elif case == "tiny_rank":
        k = 300
        monkeypatch.setenv("CCR_OPTIMISTIC", "1")
        monkeypatch.setenv("CCR_OPT_RANK", "2")
    elif case == "sorted_forced":
        data = "sorted"
        monkeypatch.setenv("CCR_OPTIMISTIC", "1")
    elif case == "clustered_tiny_rank":
        data, k = "clustered", 500
        monkeypatch.setenv("CCR_OPTIMISTIC", "1")
        monkeypatch.setenv("CCR_OPT_RANK", "3")
    nb = torch.empty(n, device="cuda")
    D = ops.pack_bf16(gen_rows(n, d, 1234, "cuda", data), norm_bounds=nb)
    Q = ops.pack_bf16(gen_rows(nq, d, 4321, "cuda", data))
    index = ops.CorpusIndex(D, norm_bounds=nb)            # the knobs are read when the index is created
    s, i = index.search(Q, k, 2)
    st = index.last_stats()
    print(case, st)
    assert st["path"] == 1 and st["opt_rank"] > 0 and st["main_launches"] == 1, st
    if "tiny_rank" in case:
        assert st["opt_rank"] <= 3 and st["n_fallback"] > nq // 2 and st["n_retried"] + st["n_dense"] >= st["n_fallback"], st
    elif case in ("planner_k1001", "forced_k100"):
        assert st["n_fallback"] == 0 and k <= st["n_candidates"] / nq < 8000, st      # iid: nobody fails, about rank / fs rows pass
    monkeypatch.setenv("CCR_OPTIMISTIC", "0")
    ref = ops.CorpusIndex(D, norm_bounds=nb)
    s0, i0 = ref.search(Q, k, 2)                          # conservative thresholds (round-2 behaviour)
    assert ref.last_stats()["opt_rank"] == 0
    s1, i1 = ref.search(Q[:64], k, 1)                     # fp64 dense
    assert torch.equal(i, i0) and torch.equal(s.view(torch.int32), s0.view(torch.int32))
    assert torch.equal(i[:64], i1) and torch.equal(s[:64].view(torch.int32), s1.view(torch.int32))


@pytest.mark.parametrize("n,nq,d,k", [(70_001, 1, 768, 100), (70_001, 16, 768, 100), (70_001, 17, 768, 10), (300_007, 33, 768, 1001),
                                      (300_007, 64, 768, 100), (40_000, 64, 1024, 300), (40_000, 5, 32, 7), (123_457, 40, 256, 64),
                                      (9_000, 64, 768, 1), (300_007, 65, 768, 100), (123_457, 128, 768, 1001), (70_001, 100, 256, 10),
                                      (40_000, 128, 1024, 100), (3_000, 90, 768, 50), (300_007, 80, 768, 100), (70_001, 96, 256, 1001)])
def test_streaming_main_pass_of_small_batches(n, nq, d, k, monkeypatch):
    """n_q <= 64: the first main pass is the streaming kernel (csrc/ccr_narrow.hip: query rows resident in LDS, the corpus straight
    into the MFMA operand registers, no barrier; one range x two atomically filled sub-lists per query).  Same canonical bits as the
    tile kernels (CCR_NARROW=0) and as the exact dense path; rows that are no multiple of 16 (tail group), every query-tile count
    (16 / 32 / 64 rows resident), dim 32 ... 1024 (64 x 1 024-wide rows do not fit the LDS image: the planner keeps the tile kernels
    there), exact ties, and a norm-outlier row (per-tile margins).  65 .. 128 queries (r5): two groups of <= 64 on paired workgroups
    that walk the same rows (a 3 000-row corpus: fewer 128-row blocks than workgroups) -- or, up to 96 queries where six query
    tiles fit the LDS (dim <= 768), ONE group with a short staging list per query."""
    from ccrec_amd import ops
    g = torch.Generator().manual_seed(n + nq + d)
    D = torch.randn(n, d, generator=g) / d ** 0.5
    Q = torch.randn(nq, d, generator=g) / d ** 0.5
    D[n // 3] = D[5]                                 # exact tie: the lower row wins
    D[n - 1] = D[n // 2]                             # ... with a row of the tail group
    D[7] *= 40.0                                     # a row whose norm is far above its tile's others
    Db, Qb = ops.pack_bf16(D.cuda()), ops.pack_bf16(Q.cuda())
    monkeypatch.delenv("CCR_NARROW", raising=False)
    index = ops.CorpusIndex(Db, global_row_offset=11)
    s, i = index.search(Qb, k, 2)
    st = index.last_stats()
    expect_narrow = not (d == 1024 and nq > 32)
    assert st["path"] == 1 and (st["ranges"], st["sublists"]) == ((1, 2) if expect_narrow else (st["ranges"], 8)), st
    assert (st["ranges"] == 1) == expect_narrow and st["n_fallback"] == 0, st
    s_ref, i_ref = index.search(Qb, k, 1)           # exact dense path
    assert torch.equal(i, i_ref) and torch.equal(s.view(torch.int32), s_ref.view(torch.int32)), st
    monkeypatch.setenv("CCR_NARROW", "0")
    tiles = ops.CorpusIndex(Db, global_row_offset=11)
    s2, i2 = tiles.search(Qb, k, 2)
    assert tiles.last_stats()["sublists"] in (4, 8) and tiles.last_stats()["ranges"] >= 8
    assert torch.equal(i2, i) and torch.equal(s2.view(torch.int32), s.view(torch.int32))
    if n * nq <= 3_000_000:
        bits = lambda t: t.view(torch.int16).cpu().numpy().view(np.uint16)   # noqa: E731
        ref_i, ref_s = orc.canonical_search(bits(Qb), bits(Db), k)
        assert np.array_equal(i.cpu().numpy() - 11, ref_i) and np.array_equal(s.cpu().numpy(), ref_s)


def test_streaming_main_pass_overflowing_lists_fall_back_to_the_retry():
    """A flooded small batch: 30 000 copies of one row that every query prefers -- the estimated / sampled threshold lets all of them
    pass, a workgroup's 64-record staging list overflows (records then go straight to the candidate area) and the mass tie around the
    cut sends the queries down the exact path: still the canonical bits."""
    from ccrec_amd import ops
    n, nq, d, k = 200_000, 8, 256, 50
    g = torch.Generator().manual_seed(1)
    D = torch.randn(n, d, generator=g) / d ** 0.5
    Q = torch.randn(nq, d, generator=g) / d ** 0.5
    D[50_000:80_000] = Q.sum(0) * 2.0
    Db, Qb = ops.pack_bf16(D.cuda()), ops.pack_bf16(Q.cuda())
    index = ops.CorpusIndex(Db)
    s, i = index.search(Qb, k, 2)
    st = index.last_stats()
    assert st["ranges"] == 1 and st["sublists"] == 2
    s_ref, i_ref = index.search(Qb, k, 1)
    assert torch.equal(i, i_ref) and torch.equal(s.view(torch.int32), s_ref.view(torch.int32)), st
    assert i[0, 0].item() == 50_000 and i[0, k - 1].item() == 50_000 + k - 1          # the tie is cut in row order
