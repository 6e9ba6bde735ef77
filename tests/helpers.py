"""Shared comparison helpers for rank lists (tie-aware)."""
import numpy as np


def canonicalise(ids, scores):
    """Re-order each row by (score desc, id asc) -- the reference leaves tie order unspecified."""
    out_i = np.empty_like(ids)
    out_s = np.empty_like(scores)
    for q in range(ids.shape[0]):
        o = np.lexsort((ids[q], -scores[q].astype(np.float64)))
        out_i[q], out_s[q] = ids[q][o], scores[q][o]
    return out_i, out_s


def assert_rank_close(ids, scores, ref_ids, ref_scores, tol, truncated=False):
    """ids/scores: ours (canonical order). ref_*: reference order (ties arbitrary, fp32 MKL sums).

    * scores agree rank by rank within tol;
    * ids agree at every rank whose reference score is separated from both neighbours by > 2 tol;
    * as sets, ids agree except for members whose score is within 2 tol of the cut (if truncated).
    """
    assert ids.shape == ref_ids.shape, (ids.shape, ref_ids.shape)
    np.testing.assert_allclose(scores, ref_scores, atol=tol, rtol=0)
    for q in range(ids.shape[0]):
        rs = ref_scores[q].astype(np.float64)
        gap_prev = np.r_[np.inf, rs[:-1] - rs[1:]]
        gap_next = np.r_[rs[:-1] - rs[1:], np.inf if not truncated else 0.0]
        clear = (gap_prev > 2 * tol) & (gap_next > 2 * tol)
        bad = clear & (ids[q] != ref_ids[q])
        assert not bad.any(), f"query {q}: id mismatch at clear ranks {np.nonzero(bad)[0][:10]}"
        diff = set(ids[q].tolist()) ^ set(ref_ids[q].tolist())
        if diff:
            assert truncated, f"query {q}: id sets differ {sorted(diff)[:10]}"
            cut = rs[-1]
            ours = dict(zip(ids[q].tolist(), scores[q].tolist()))
            refd = dict(zip(ref_ids[q].tolist(), ref_scores[q].tolist()))
            for j in diff:
                s = ours.get(j, refd.get(j))
                assert abs(s - cut) <= 2 * tol, f"query {q}: id {j} score {s} far from cut {cut}"
