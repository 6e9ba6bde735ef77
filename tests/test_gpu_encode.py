"""length-sorted variable-length encode, pooled + packed straight into the shard
(ccr_meanpool_pack_bf16_ex), per-rank corpus shards.  Parity: the scatter/max-norm kernel is bit-exact against
the oracle's pooling; the length-sorted encoder equals the reference's fixed max_length padding
(item_tower.py:27-33) up to the encoder's fp32 reduction-order noise; a 2-rank sharded ranking equals the
single-process one exactly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, PKG
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


class ToyTokenizer:
    """Whitespace tokenizer; honours padding=False / "max_length" / True like the HF call the reference makes."""
    pad_token_id = 0

    def __call__(self, texts, truncation=True, padding=True, max_length=32, return_tensors="pt"):
        ids = [[1] + [2 + (sum(map(ord, w)) * 7 % 500) for w in t.split()][: max_length - 2] + [3] for t in texts]
        if padding is False:
            return {"input_ids": ids, "attention_mask": [[1] * len(r) for r in ids]}
        L = max_length if padding == "max_length" else max(len(r) for r in ids)
        input_ids = torch.zeros(len(ids), L, dtype=torch.long)
        mask = torch.zeros(len(ids), L, dtype=torch.long)
        for r, row in enumerate(ids):
            input_ids[r, : len(row)] = torch.tensor(row)
            mask[r, : len(row)] = 1
        return {"input_ids": input_ids, "attention_mask": mask}


def _tower(seed=0):
    from transformers import BertConfig, BertModel
    from ccrec_amd.item_tower import NaiveItemTower
    torch.manual_seed(seed)
    cfg = BertConfig(vocab_size=512, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                     max_position_embeddings=64)
    return NaiveItemTower(BertModel(cfg).eval(), torch.nn.LayerNorm(64, elementwise_affine=False)).cuda()


def _texts(n, seed, lo=2, hi=28):
    rs = np.random.RandomState(seed)
    words = [f"w{i}" for i in range(300)]
    return [" ".join(rs.choice(words, rs.randint(lo, hi))) for _ in range(n)]


@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_meanpool_scatter_and_norm_bounds_bit_exact(normalize, dtype):
    from ccrec_amd import ops
    B, L, d, rows_total = 37, 24, 768, 100
    g = torch.Generator().manual_seed(5)
    hidden = (torch.randn(B, L, d, generator=g) * 0.3).to(dtype)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    mask[3] = 0
    mask[3, 5] = 1                      # a non-prefix mask: position 5 only
    rows = torch.randperm(rows_total, generator=g)[:B]
    out = torch.zeros(rows_total, d, dtype=torch.bfloat16, device="cuda")
    out32 = torch.zeros(rows_total, d, dtype=torch.float32, device="cuda")
    nb = torch.full((rows_total,), -1.0, dtype=torch.float32, device="cuda")
    ops.meanpool_pack(hidden.cuda(), mask.cuda(), normalize=normalize, out_bf16=out, out_f32=out32, dst_rows=rows, norm_bounds=nb)
    ref32 = orc.meanpool(hidden.float().numpy(), mask.numpy())
    got16 = out.view(torch.int16).cpu().numpy().view(np.uint16)
    assert np.array_equal(out32.cpu().numpy()[rows.numpy()].view(np.uint32), ref32.view(np.uint32))
    if normalize:   # the kernel's fp64 norm reduction order is its own: compare values (as test_gpu_pack does)
        refn = ref32 / np.maximum(np.linalg.norm(ref32.astype(np.float64), axis=1, keepdims=True), 1e-12)
        np.testing.assert_allclose(orc.unpack_bf16(got16[rows.numpy()]), refn, atol=4e-3, rtol=8e-3)
    else:
        assert np.array_equal(got16[rows.numpy()], orc.pack_bf16(ref32))
    untouched = np.setdiff1d(np.arange(rows_total), rows.numpy())
    assert not got16[untouched].any() and not out32.cpu().numpy()[untouched].any()
    ref16 = got16[rows.numpy()]
    true_norms = np.sqrt((orc.unpack_bf16(ref16).astype(np.float64) ** 2).sum(1))   # a bound per packed row, at its destination
    got_nb = nb.cpu().numpy()
    assert np.all(true_norms <= got_nb[rows.numpy()]) and np.all(got_nb[rows.numpy()] <= true_norms * 1.001 + 1e-30)
    assert np.all(got_nb[untouched] == -1.0)


def test_plan_and_encode_match_fixed_padding():
    """Length-sorted batches vs the reference's tokenizer_kw (padding="max_length"): same pooled rows up to the fp32
    noise of the encoder at a different padded length; far fewer padded tokens."""
    from ccrec_amd.encode import LengthSortedEncoder
    tower, tok = _tower(), ToyTokenizer()
    texts = _texts(333, 1)
    enc = LengthSortedEncoder(tower, tok, max_length=32, max_tokens=1024, max_batch=64)
    out32 = torch.zeros(len(texts), 64, dtype=torch.float32, device="cuda")
    nb = torch.empty(len(texts), dtype=torch.float32, device="cuda")
    got = enc.encode(texts, sim="dot", out_f32=out32, norm_bounds=nb)
    st = enc.stats
    assert st["texts"] == 333 and st["real_tokens"] <= st["padded_tokens"] < 0.75 * st["fixed_length_tokens"]
    assert st["batches"] > 333 // 64           # the token budget, not only max_batch, cut the batches
    # reference path: fixed padding to max_length, corpus order, pooled by the same kernel
    ref32, ref16 = [], []
    with torch.no_grad():
        for lo in range(0, len(texts), 50):
            toks = tok(texts[lo:lo + 50], padding="max_length", max_length=32)
            ref32.append(tower(**{k: v.cuda() for k, v in toks.items()}, output_step="mean_pooling"))
            ref16.append(tower(**{k: v.cuda() for k, v in toks.items()}, output_step="mean_pooling_bf16"))
    ref32, ref16 = torch.cat(ref32), torch.cat(ref16)
    torch.testing.assert_close(out32, ref32, rtol=1e-4, atol=1e-5)        # tolerance: fp32 reduction order inside BERT
    diff = (got.view(torch.int16).int() - ref16.view(torch.int16).int()).abs()
    assert int(diff.max()) <= 1 and float((diff != 0).float().mean()) < 0.02   # bf16 rows: equal bits, rare 1-ulp flips
    packed_norms = got.float().norm(dim=1)
    assert torch.all(packed_norms <= nb * 1.000001) and torch.all(nb <= packed_norms * 1.001 + 1e-30)


def test_chunked_pipeline_equals_one_chunk():
    """The corpus is tokenised / planned chunk by chunk on a host thread while the GPU encodes the previous chunk: every text
    lands in its own row exactly once and the rows agree with a single-chunk run up to the encoder's fp32 noise at a different
    batch composition (bf16 rows: equal bits, rare 1-ulp flips)."""
    from ccrec_amd.encode import LengthSortedEncoder
    tower, tok = _tower(), ToyTokenizer()
    texts = _texts(1000, 5)
    one = LengthSortedEncoder(tower, tok, max_length=32, max_tokens=1024, max_batch=64, chunk_texts=10_000)
    many = LengthSortedEncoder(tower, tok, max_length=32, max_tokens=1024, max_batch=64, chunk_texts=130)   # 8 chunks, ragged last
    nb1, nb2 = torch.full((1003,), -1.0, device="cuda"), torch.full((1003,), -1.0, device="cuda")
    a = one.encode(texts, sim="cos", norm_bounds=nb1, row_offset=3)
    b = many.encode(texts, sim="cos", norm_bounds=nb2, row_offset=3)
    assert one.stats["chunks"] == 1 and many.stats["chunks"] == 8 and many.stats["texts"] == 1000
    assert many.stats["real_tokens"] == one.stats["real_tokens"] and many.stats["batches"] >= one.stats["batches"]
    assert many.stats["wall_s"] > 0 and many.stats["gpu_busy_s"] > 0 and many.stats["host_prepare_s"] > 0
    assert a.shape == b.shape == (1003, 64)
    diff = (a[3:].view(torch.int16).int() - b[3:].view(torch.int16).int()).abs()
    assert int(diff.max()) <= 1 and float((diff != 0).float().mean()) < 0.02
    assert torch.all(nb2[:3] == -1.0) and torch.all(nb2[3:] > 0) and torch.allclose(nb1[3:], nb2[3:], rtol=1e-2)
    assert many.encode([], sim="dot").shape[0] == 0 and many.stats["chunks"] == 0


@pytest.mark.parametrize("step", ["cls", "mean_layer_norm"])
def test_cls_output_steps_through_the_length_sorted_encoder(step):
    """The tower's other output steps (item_tower.py:133-136: h[:, 0] and LayerNorm(h[:, 0]), the reference's default
    CCREC_EMBEDDING_TYPE) through the length-sorted encoder: packed rows == pack(the tower's own output at fixed padding) up to
    the encoder's fp32 noise at a different padded length."""
    from ccrec_amd import ops
    from ccrec_amd.encode import LengthSortedEncoder
    tower, tok = _tower(), ToyTokenizer()
    texts = _texts(257, 7)
    enc = LengthSortedEncoder(tower, tok, max_length=32, max_tokens=1024, max_batch=64, output_step=step)
    out32 = torch.zeros(len(texts), 64, dtype=torch.float32, device="cuda")
    got = enc.encode(texts, sim="dot", out_f32=out32)
    ref = []
    with torch.no_grad():
        for lo in range(0, len(texts), 50):
            toks = tok(texts[lo:lo + 50], padding="max_length", max_length=32)
            ref.append(tower(**{k: v.cuda() for k, v in toks.items()}, output_step=step))
    ref = torch.cat(ref)
    torch.testing.assert_close(out32, ref.float(), rtol=1e-4, atol=1e-5)
    diff = (got.view(torch.int16).int() - ops.pack_bf16(ref.float()).view(torch.int16).int()).abs()
    assert int(diff.max()) <= 1 and float((diff != 0).float().mean()) < 0.02


def test_tokenizer_worker_processes_give_the_same_rows():
    """host_processes > 0: a HF fast tokenizer's chunks are tokenised in worker processes (ccrec_amd/_tokenize_worker.py, no torch
    import) -- same token ids as the in-process Rust backend and the HF call, hence the same packed rows."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from bench_encode import fast_tokenizer
    from transformers import BertConfig, BertModel
    from ccrec_amd.encode import LengthSortedEncoder
    from ccrec_amd.item_tower import NaiveItemTower
    tok = fast_tokenizer(2000)
    rs = np.random.RandomState(3)
    texts = [" ".join(f"w{j}" for j in rs.randint(0, 2000, rs.randint(3, 60))) for _ in range(700)]
    torch.manual_seed(0)
    cfg = BertConfig(vocab_size=2048, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128)
    tower = NaiveItemTower(BertModel(cfg).eval(), torch.nn.LayerNorm(64, elementwise_affine=False)).cuda()
    a = LengthSortedEncoder(tower, tok, max_length=48, max_tokens=2048, chunk_texts=100, host_threads=2)
    b = LengthSortedEncoder(tower, tok, max_length=48, max_tokens=2048, chunk_texts=100, host_threads=3, host_processes=2)
    ra, rb = a.encode(texts, sim="dot"), b.encode(texts, sim="dot")
    assert b._workers is not None and len(b._workers.procs) == 2 and a._workers is None
    assert a.stats["real_tokens"] == b.stats["real_tokens"] and torch.equal(ra.view(torch.int16), rb.view(torch.int16))
    ref = tok(texts[:50], truncation=True, max_length=48)["input_ids"]
    flat, lengths = b._workers.tokenize(texts[:50])
    assert lengths.tolist() == [len(r) for r in ref] and flat.tolist() == [t for r in ref for t in r]
    b.close()
    assert b._workers is None


def test_ranking_sharded_single_rank_equals_ranking_api():
    from ccrec_amd.encode import LengthSortedEncoder, ranking_sharded
    os.environ["CCREC_SIM_TYPE"] = "cos"
    tower, tok = _tower(), ToyTokenizer()
    corpus = {f"p{j}": t for j, t in enumerate(_texts(500, 2))}
    queries = {f"q{i}": t for i, t in enumerate(_texts(7, 3, 2, 9))}
    enc = LengthSortedEncoder(tower, tok, max_length=32, max_tokens=2048)
    block = {q: [f"p{(7 * i + j) % 500}" for j in range(3)] for i, q in enumerate(queries)}
    prof = ranking_sharded(corpus, queries, enc, block_dict=block, keep=50)
    # the oracle on the encoder outputs the product packed (fp32 rows recovered by a second encode into out_f32 would
    # not be bit-reproducible; the packed bf16 rows ARE the inputs of the search)
    q16 = enc.encode(list(queries.values()), sim="cos")
    d16 = enc.encode(list(corpus.values()), sim="cos")
    Qb = q16.view(torch.int16).cpu().numpy().view(np.uint16)
    Db = d16.view(torch.int16).cpu().numpy().view(np.uint16)
    blocked = [[int(p[1:]) for p in block[q]] for q in queries]
    ref_i, ref_s = orc.canonical_search(Qb, Db, 50, block=blocked)
    got_i = np.array([[int(p[1:]) for p in prof[q]] for q in queries])
    got_s = np.array([list(prof[q].values()) for q in queries], np.float32)
    assert np.array_equal(got_i, ref_i) and np.array_equal(got_s.view(np.uint32), ref_s.view(np.uint32))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    sys.path[:0] = [ROOT, PKG]
    os.environ["CCREC_SIM_TYPE"] = "dot"
    from ccrec_amd.encode import LengthSortedEncoder, ranking_sharded
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    tower, tok = _tower(), ToyTokenizer()
    corpus = {f"p{j}": t for j, t in enumerate(_texts(901, 4))}
    queries = {f"q{i}": t for i, t in enumerate(_texts(5, 5, 2, 9))}
    enc = LengthSortedEncoder(tower, tok, max_length=32, max_tokens=4096)
    prof = ranking_sharded(corpus, queries, enc, rank=rank, world=world, keep=40)
    torch.save(prof, os.path.join(out_dir, f"prof{rank}.pt"))
    # the whole rank step in the same two-process layout: every rank returns the same requests, rank 0 alone writes the files
    from ccrec_amd.al_step import run_rank_step
    os.environ["CCREC_DISPLAY_LENGTH"] = "40"
    big = {f"p{j}": t for j, t in enumerate(_texts(1300, 6))}
    qrels = {q: {"p3": 1} for q in queries}
    bm25 = {q: {"p5": 2.0, "p6": 1.0, "p7": 0.5} for q in queries}
    res = run_rank_step(tower, tok, big, queries, qrels, list(queries)[:3], 0, os.path.join(out_dir, "step"), ranking_profile_bm25=bm25,
                        encoder_kw={"max_length": 32, "max_tokens": 4096}, autocast=False, rank=rank, world=world)
    res["requests"]["request_orig"].to_csv(os.path.join(out_dir, f"orig{rank}.csv"), index=False)
    torch.save(dict(res["mrr"]), os.path.join(out_dir, f"mrr{rank}.pt"))
    # ... and with the BM25 ranking computed inside (ranking_profile_bm25=None): rank 0 alone runs it -- on ITS device, from a worker
    # thread -- and builds the requests; the other rank returns the merged profile and MRR without them
    import threading
    from ccrec_amd import al_step
    seen = []
    real = al_step.ranking_bm25
    al_step.ranking_bm25 = lambda *a, **k: (seen.append((threading.current_thread() is threading.main_thread(), torch.cuda.current_device())),
                                            real(*a, **k))[1]
    res2 = run_rank_step(tower, tok, big, queries, qrels, list(queries)[:3], 0, os.path.join(out_dir, "step_bm25"),
                         encoder_kw={"max_length": 32, "max_tokens": 4096}, autocast=False, rank=rank, world=world)
    al_step.ranking_bm25 = real
    assert seen == ([(False, torch.cuda.current_device())] if rank == 0 else []), seen
    assert (res2["requests"] is not None) == (rank == 0) and res2["mrr"] == res["mrr"]
    if rank == 0:
        res2["requests"]["request_orig"].to_csv(os.path.join(out_dir, "orig_bm25.csv"), index=False)
    dist.barrier()
    dist.destroy_process_group()


def test_ranking_sharded_two_ranks_equal_one(tmp_path):
    """Two processes (gloo, both on this one GPU) each encode + index half of the corpus; the merged profile must equal
    the single-process profile: the shard batches differ, so compare ids exactly and scores to the encoder noise."""
    import torch.multiprocessing as mp
    from ccrec_amd.encode import LengthSortedEncoder, ranking_sharded
    port = _free_port()
    mp.spawn(_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = torch.load(tmp_path / "prof0.pt"), torch.load(tmp_path / "prof1.pt")
    assert p0 == p1                                            # every rank holds the same merged result
    assert open(tmp_path / "orig0.csv").read() == open(tmp_path / "orig1.csv").read() == open(tmp_path / "step" / "data_iteration_0" / "request_orig.csv").read()
    assert torch.load(tmp_path / "mrr0.pt") == torch.load(tmp_path / "mrr1.pt")
    assert sorted(os.listdir(tmp_path / "step" / "data_iteration_0")) == ["id_track.pt", "ranking_profile.pt", "request_orig.csv", "request_perm.csv"]
    assert sorted(os.listdir(tmp_path / "step_bm25" / "data_iteration_0")) == ["id_track.pt", "ranking_profile.pt", "request_orig.csv", "request_perm.csv"]
    assert (tmp_path / "orig_bm25.csv").is_file()
    os.environ["CCREC_SIM_TYPE"] = "dot"
    tower, tok = _tower(), ToyTokenizer()
    corpus = {f"p{j}": t for j, t in enumerate(_texts(901, 4))}
    queries = {f"q{i}": t for i, t in enumerate(_texts(5, 5, 2, 9))}
    single = ranking_sharded(corpus, queries, LengthSortedEncoder(tower, tok, max_length=32, max_tokens=4096), keep=40)
    for q in queries:
        a, b = single[q], p0[q]
        assert len(a) == len(b) == 40
        sa, sb = np.array(list(a.values())), np.array(list(b.values()))
        np.testing.assert_allclose(sa, sb, rtol=1e-3, atol=1e-3)
        # ids agree wherever neighbouring scores are separated by more than the encoder noise
        ia, ib = list(a), list(b)
        for r in range(40):
            if ia[r] != ib[r]:
                assert abs(sa[r] - sb[r]) < 1e-3 and ib[r] in ia[max(0, r - 3):r + 4]


def _balance_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    sys.path[:0] = [ROOT, PKG]
    os.environ["CCREC_SIM_TYPE"] = "dot"
    from ccrec_amd.encode import LengthSortedEncoder, ranking_sharded
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    tower, tok = _tower(), ToyTokenizer()
    texts = sorted(_texts(1500, 8), key=lambda t: t.count(" "))            # the corpus in order of passage length: 2 .. 27 words
    corpus = {f"p{j:04d}": t for j, t in enumerate(texts)}
    queries = {f"q{i}": t for i, t in enumerate(_texts(5, 5, 2, 9))}
    exact = [min(t.count(" ") + 1 + 2, 32) for t in texts]                 # the toy tokeniser's token counts
    out = {}
    for name, balance in (("rows", "rows"), ("estimate", "tokens"), ("exact", exact)):
        enc = LengthSortedEncoder(tower, tok, max_length=32, max_tokens=4096)
        prof = ranking_sharded(corpus, queries, enc, rank=rank, world=world, keep=30, balance=balance)
        out[name] = (enc.stats["real_tokens"], prof)                      # the corpus is encoded last: its statistics are the ones left
    torch.save(out, os.path.join(out_dir, f"balance{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_ranking_sharded_cuts_a_length_sorted_corpus_by_tokens(tmp_path):
    """Three ranks (gloo, one GPU) on a corpus sorted by passage length.  Equal ROW counts give the last rank ~2.6x the tokens of the first
    (its encode, 97 % of the step, sets the step time); cut by the per-row token counts every rank encodes the same tokens within 2 %
    (exact weights; and the a x words + b estimate fitted on 512 sampled texts, exact for a whitespace tokeniser) -- and the merged profile is the
    same whichever way the rows were cut (the search only needs each shard's row offset)."""
    import torch.multiprocessing as mp
    mp.spawn(_balance_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    res = [torch.load(tmp_path / f"balance{r}.pt") for r in range(3)]
    tok = {name: np.array([res[r][name][0] for r in range(3)], np.float64) for name in ("rows", "estimate", "exact")}
    assert tok["rows"].max() / tok["rows"].min() > 2.0, tok
    assert tok["exact"].max() / tok["exact"].min() < 1.02, tok
    assert tok["estimate"].max() / tok["estimate"].min() < 1.02, tok
    assert tok["rows"].sum() == tok["exact"].sum() == tok["estimate"].sum()
    base = res[0]["rows"][1]
    for r in range(3):
        for name in ("rows", "estimate", "exact"):
            prof = res[r][name][1]
            assert list(prof) == list(base)
            for q in base:     # the batches differ with the cut: ids wherever the scores are separated by more than the encoder noise
                sa, sb = np.array(list(base[q].values())), np.array(list(prof[q].values()))
                np.testing.assert_allclose(sa, sb, rtol=1e-3, atol=1e-3)
