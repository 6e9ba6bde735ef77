"""One active-learning step, end to end, on the device path: the sequence scripts/al_0_rank.py runs at module level
(:107-218) as a function -- rank (cached), report MRR, build the labelling request.

    ranking_profile.pt  <- encode (length-sorted) + fused search      [encode.ranking_sharded / cached on disk]
    MRR@{1,5,10,100}    <- ccr_rank_metrics on the id tensor           [evaluation.rank_metrics]
    request_orig.csv, request_perm.csv, id_track.pt                    [al_request.build_requests]
`ranking_profile_bm25` is an input of the reference (parse_al_args loads it from disk, al_commons.py:55-58);
pass it in, or leave it None to compute it with bm25.ranking_bm25.
"""
import os

import torch

from . import evaluation, ranking_profile
from .al_request import build_requests
from .bm25 import ranking_bm25
from .encode import LengthSortedEncoder, ranking_sharded


def _bm25_on_device(device, corpus, queries):
    """The lexical ranking on a worker thread: torch's current device is thread-local, so the thread selects its process's GPU first
    (one process per GPU: without this every rank's BM25 search would land on cuda:0)."""
    with torch.cuda.device(device):
        return ranking_bm25(corpus, queries, lazy=True)


def run_rank_step(tower, tokenizer, corpus, queries, qrels, step_qids, step, results_dir, ranking_profile_bm25=None,
                  block_dict=None, landing_image=None, n_repeats=3, repeat_seed=42, encoder_kw=None, autocast=True,
                  compat_profile=True, rank=0, world=1, group=None, balance="tokens"):
    """-> {"ranking_profile", "mrr", "requests", "timings"}; files are written to results_dir/data_iteration_{step}/.
    timings: wall seconds of the stages (rank = encode + search; save; mrr; bm25; requests) and the corpus encoder's own statistics.
    autocast: rank inside `torch.autocast("cuda")` -- the reference's fp16 context (scripts/al_0_rank.py:125); the layer kernels run
    in the context's type.  compat_profile: ranking_profile.pt in the reference's own form, the nested {qid: {pid: score}} dict
    (al_0_rank.py:127: a RESULTS_DIR shared with the reference's scripts resumes from it); False = the tensor form (a few MB, loads
    under torch.load's weights_only default).  It is written right after the ranking, before MRR and the request files, so a failure
    further down does not lose the encode + search.
    rank / world / group: one process per GPU (torch.distributed initialised by the caller): every rank encodes and indexes its own
    block of the corpus, all ranks hold the same merged profile and MRR afterwards.  Only rank 0 writes files, and only rank 0
    computes the BM25 ranking when none is passed in (the other ranks then return "requests": None).  balance: how the corpus is cut
    into the ranks' blocks (encode.ranking_sharded: "tokens" = equal estimated tokens per rank, "rows", or one weight per row)."""
    import time
    t_start = time.perf_counter()
    timings = {}
    work = os.path.join(results_dir, f"data_iteration_{step}")
    os.makedirs(work, exist_ok=True)
    path = os.path.join(work, "ranking_profile.pt")
    corpus_ids = list(corpus)
    # the qrels in corpus-row space do not depend on the search: prepared BEFORE it, so that nothing of O(corpus) Python work
    # sits between the search and the request builder
    wanted = {p for rel_q in qrels.values() for p in rel_q}
    pos = {pid: i for i, pid in enumerate(corpus_ids) if pid in wanted}
    ids = None
    fresh = not os.path.isfile(path)
    makes_requests = rank == 0 or ranking_profile_bm25 is not None
    bm25_job = bm25_pool = None
    if ranking_profile_bm25 is None and rank == 0:
        # the lexical leg does not depend on the encoder: its text analysis runs in worker processes (bm25.BM25.fit) and its device
        # search takes about a second, so it is started now and collected after the dense ranking -- beside the GPU encode
        from concurrent.futures import ThreadPoolExecutor
        bm25_pool = ThreadPoolExecutor(max_workers=1)
        t_bm25 = time.perf_counter()
        bm25_job = bm25_pool.submit(_bm25_on_device, torch.cuda.current_device(), corpus, queries)
    try:
        if not fresh:                                              # al_0_rank.py:115-118: resume from the saved profile
            profile = ranking_profile.load(path)                   # (the tensor form or the reference's nested dict)
        else:
            # (a HF fast tokenizer is run in four worker processes beside the GPU loop: 18 k instead of 15 k passages/s)
            kw = {"host_processes": 4} if encoder_kw is None else dict(encoder_kw)
            kw.setdefault("max_length", int(os.environ.get("CCREC_MAX_LENGTH", 512)))      # al_0_rank.py:78
            encoder = LengthSortedEncoder(tower, tokenizer, **kw)
            t0 = time.perf_counter()
            try:
                with torch.autocast("cuda", enabled=bool(autocast)):
                    profile, ids, _ = ranking_sharded(corpus, queries, encoder, block_dict=block_dict, with_tensors=True, lazy=True,
                                                      rank=rank, world=world, group=group, balance=balance)
                torch.cuda.synchronize()
                timings["rank_s"] = time.perf_counter() - t0
                timings["corpus_encoder"] = dict(encoder.stats)      # (the corpus is encoded last: its statistics are the ones left)
            finally:
                encoder.close()
            if rank == 0:
                t0 = time.perf_counter()
                profile.save(path, compat=compat_profile)
                timings["save_s"] = time.perf_counter() - t0
        qids = list(profile)
        if ids is None:     # resumed: the id tensor comes back from the file (the fresh path keeps the search's own tensor)
            qids, ids, _ = evaluation.profile_to_tensors(profile, corpus_ids)
        rel = [[pos[p] for p, r in qrels.get(q, {}).items() if r > 0 and p in pos] for q in qids]
        kmax = ids.shape[1]
        t0 = time.perf_counter()
        mrr = evaluation.rank_metrics(ids.cuda(), rel, tuple(k for k in (1, 5, 10, 100) if k <= kmax), n_qrels=len(qrels))
        timings["mrr_s"] = time.perf_counter() - t0
        if rank == 0:
            for name, value in mrr.items():
                print(name, ":", value)
        if bm25_job is not None:
            t0 = time.perf_counter()
            ranking_profile_bm25 = bm25_job.result()
            timings["bm25_wait_s"] = time.perf_counter() - t0       # what was NOT hidden behind the dense ranking
            timings["bm25_since_start_s"] = time.perf_counter() - t_bm25
        bm25_job = None
    finally:
        if bm25_pool is not None:
            # normal path: the job's result has been collected above.  Error path (bm25_job still set): the exception must not wait
            # for the whole BM25 job -- drop what has not started and let the worker thread finish on its own
            bm25_pool.shutdown(wait=bm25_job is None, cancel_futures=True)
    requests = None
    if makes_requests:
        t0 = time.perf_counter()
        requests = build_requests(profile, ranking_profile_bm25, step_qids, corpus, queries, step, n_repeats=n_repeats,
                                  repeat_seed=repeat_seed, landing_image=landing_image, out_dir=work if rank == 0 else None)
        timings["requests_s"] = time.perf_counter() - t0
    timings["total_s"] = time.perf_counter() - t_start
    return {"ranking_profile": profile, "mrr": mrr, "requests": requests, "timings": timings}
