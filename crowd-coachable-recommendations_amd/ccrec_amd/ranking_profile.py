"""A lazy `ranking_profile`: the {qid: {pid: score}} mapping of scripts/ms_marco_eval.py:231-235, backed by the [Q, k] id /
score tensors of the search instead of Q x k Python (pid, float) pairs.

The consumers of the reference read a few hundred queries per active-learning step (scripts/al_0_rank.py:165-201 walks
`qids_split[STEP]` only; scripts/al_oracle_agent.py:145-153 the same) and BEIR's MRR re-reads values (al_0_rank.py:130-133);
building 3.5 M pairs per NQ-sized profile cost 1.3 s around a 19-ms search.  Here an inner dict is built when its query is
read, in the reference's order (dict iteration order = rank order), and the file form is the tensors themselves:

    save(path)              {"format": FORMAT, "query_ids", "pids" (the referenced passages only), "rows", "scores"}
                            -- plain containers + tensors: loads under torch.load's weights_only default
    save(path, compat=True) the reference's nested dict (what torch.save(ranking_profile, ...) wrote, al_0_rank.py:127)
    load(path)              either form -> RankingProfile (a nested dict is wrapped without copying its inner dicts)
"""
from collections.abc import Mapping

import numpy as np
import torch

FORMAT = "ccrec_amd.ranking_profile.v1"


class RankingProfile(Mapping):
    """Read-only Mapping qid -> {pid: score} (rank-ordered).  len(), iteration order (query order), `in`, [] , .get(),
    .keys() / .items() / .values() behave like the nested dict; inner dicts are built on first access and kept.

    query_ids: list; corpus_ids: sequence or numpy object array indexed by the entries of `rows`;
    rows [Q, k] int64 and scores [Q, k] fp32: host numpy arrays (torch tensors are converted)."""

    def __init__(self, query_ids, corpus_ids, rows, scores):
        self.query_ids = list(query_ids)
        self._cid = corpus_ids if isinstance(corpus_ids, np.ndarray) and corpus_ids.dtype == object else _object_array(corpus_ids)
        self.rows = _host(rows, np.int64)
        self.scores = _host(scores, np.float32)
        assert self.rows.shape == self.scores.shape and self.rows.ndim == 2 and self.rows.shape[0] == len(self.query_ids)
        self._where = None      # qid -> row of the tensors (built on the first lookup)
        self._cache = {}

    # ---- Mapping
    def __len__(self):
        return len(self.query_ids)

    def __iter__(self):
        return iter(self.query_ids)

    def _index(self):
        if self._where is None:
            self._where = {q: i for i, q in enumerate(self.query_ids)}
        return self._where

    def __contains__(self, qid):
        return qid in self._index()

    def __getitem__(self, qid):
        hit = self._cache.get(qid)
        if hit is not None:
            return hit
        i = self._index()[qid]            # KeyError like a dict
        inner = dict(zip(self._cid[self.rows[i]].tolist(), self.scores[i].tolist()))
        self._cache[qid] = inner
        return inner

    # ---- cheap reads that build no dict
    def top(self, qid, n=None):
        """The first n passage ids of a query in rank order (all of them: n=None)."""
        i = self._index()[qid]
        r = self.rows[i] if n is None else self.rows[i, :n]
        return self._cid[r].tolist()

    def tensors(self):
        """(query ids, rows [Q, k] int64 into the corpus order, scores [Q, k] fp32) -- what evaluation.rank_metrics takes."""
        return self.query_ids, torch.from_numpy(self.rows), torch.from_numpy(self.scores)

    def to_dict(self):
        """The reference's nested dict (3.5 M pairs at the NQ shape: the cost this class exists to avoid)."""
        keys = self._cid[self.rows]
        return {q: dict(zip(kr, sr)) for q, kr, sr in zip(self.query_ids, keys.tolist(), self.scores.tolist())}

    def __eq__(self, other):
        if isinstance(other, Mapping):
            return len(self) == len(other) and all(q in other and self[q] == other[q] for q in self.query_ids)
        return NotImplemented

    __hash__ = None

    # ---- file forms
    def state(self):
        """Self-contained tensor form: only the passages some list refers to are named (bitmap + prefix sum, no sort)."""
        n = len(self._cid)
        used = np.zeros(n + 1, dtype=bool)
        used[self.rows.ravel()] = True
        used[n] = False
        remap = np.cumsum(used[:n], dtype=np.int64) - 1
        pids = self._cid[np.flatnonzero(used[:n])].tolist()
        if pids and all(type(p) is str for p in pids) and not any("\n" in p for p in pids):
            pids = {"joined": "\n".join(pids), "count": len(pids)}   # ONE string instead of a million pickled ones (3 x faster to write)
        return {"format": FORMAT, "query_ids": self.query_ids, "pids": pids, "rows": torch.from_numpy(remap[self.rows]),
                "scores": torch.from_numpy(self.scores)}

    def save(self, path, compat=False):
        torch.save(self.to_dict() if compat else self.state(), path)

    def __reduce__(self):   # pickling (e.g. a bare torch.save(profile, path)) writes the tensor form, not 3.5 M pairs
        st = self.state()
        return (_rebuild, (st["query_ids"], st["pids"], st["rows"].numpy(), st["scores"].numpy()))


def _pids(p):
    if isinstance(p, dict):
        out = p["joined"].split("\n")
        assert len(out) == p["count"]
        return out
    return p


def _rebuild(query_ids, pids, rows, scores):
    return RankingProfile(query_ids, _pids(pids), rows, scores)


def _object_array(seq):
    seq = list(seq)
    return np.fromiter(seq, dtype=object, count=len(seq))   # (np.array(list_of_tuples) would build a 2-d array: ids may be any hashable)


def _host(t, dtype):
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(t, dtype=dtype)


def from_state(obj):
    """The object torch.load returned -> a Mapping: the tensor form becomes a RankingProfile, a nested dict is returned as is."""
    if isinstance(obj, RankingProfile):
        return obj
    if isinstance(obj, dict) and obj.get("format") == FORMAT:
        return RankingProfile(obj["query_ids"], _pids(obj["pids"]), obj["rows"], obj["scores"])
    return obj


def load(path):
    return from_state(torch.load(path))


def as_tensors(profile, corpus_ids=None, positions=None):
    """Any ranking_profile -> (query ids, rows [Q, k] int64 into `corpus_ids` order, scores [Q, k]) on the host.
    A RankingProfile built over the same corpus order hands its tensors over; a nested dict (or a profile loaded from a file,
    whose rows index only the passages it names) is mapped through `positions` ({pid: row}, built from corpus_ids if absent)."""
    if isinstance(profile, RankingProfile) and corpus_ids is not None and len(profile._cid) == len(corpus_ids) \
            and (profile._cid is corpus_ids or _same_ids(profile._cid, corpus_ids)):
        return profile.tensors()
    if positions is None:
        positions = {pid: i for i, pid in enumerate(corpus_ids)}
    if isinstance(profile, RankingProfile):
        table = np.fromiter((positions[p] for p in profile._cid.tolist()), dtype=np.int64, count=len(profile._cid))
        return profile.query_ids, torch.from_numpy(table[profile.rows]), torch.from_numpy(profile.scores)
    qids = list(profile)
    rows = torch.tensor([[positions[p] for p in profile[q]] for q in qids], dtype=torch.int64)
    scores = torch.tensor([list(profile[q].values()) for q in qids], dtype=torch.float32)
    return qids, rows, scores


def _same_ids(arr, seq):
    if isinstance(seq, np.ndarray):
        return bool((arr == seq).all())
    n = len(arr)
    probe = range(0, n, max(1, n // 64))     # spot check (a full comparison would be the O(N) Python loop this avoids)
    return all(arr[i] == seq[i] for i in probe) and (n == 0 or arr[n - 1] == seq[n - 1])
