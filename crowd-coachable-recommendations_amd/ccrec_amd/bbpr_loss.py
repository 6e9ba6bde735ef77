"""The 'multiple_nrl' in-batch-negative objective of src/ccrec/models/bbpr.py:187-214 with the score
GEMMs + softmax-CE in HIP (ccr_inbatch_ce_fwd/bwd).  Embeddings are the three encoder outputs
(query, positive, hard negative); the round-robin negative picking (:188-193) stays host Python."""
import os

import torch

from . import ops


def pick_round_robin_negatives(user_to_negs, users):
    """bbpr.py:188-193: pop the front negative of each user and re-append it."""
    nj = []
    for user in users:
        u = int(user)
        neg = user_to_negs[u].pop(0)
        nj.append(neg)
        user_to_negs[u].append(neg)
    return nj


def multiple_nrl_loss(qid_emb, pos_emb, neg_emb, inv_temperature=None, sim_type=None):
    """scores = cat(Q P^T, Q N^T) * inv_temperature; CrossEntropyLoss()(scores, arange(B))  (bbpr.py:205-212).
    cos: rows are L2-normalised first by torch (autograd handles that Jacobian), then the HIP loss."""
    if sim_type is None:
        sim_type = os.environ["CCREC_SIM_TYPE"]
    if inv_temperature is None:
        inv_temperature = float(os.environ["CCREC_BBPR_INV_TEMPERATURE"])
    if sim_type == "cos":
        qid_emb = torch.nn.functional.normalize(qid_emb, p=2, dim=1)
        pos_emb = torch.nn.functional.normalize(pos_emb, p=2, dim=1)
        neg_emb = torch.nn.functional.normalize(neg_emb, p=2, dim=1)
    return ops.inbatch_ce(qid_emb, pos_emb, neg_emb, inv_temperature)


def compute_user_to_negatives(tr_prior_score):
    """bbpr.py:216-227: {user: [hard negatives]} from the sparse prior (entries with value >= 1.0 are negatives; every
    user that appears gets a list)."""
    coo = tr_prior_score.coalesce() if not tr_prior_score.is_coalesced() else tr_prior_score
    users, negs = coo.indices().tolist()
    out = {}
    for u, j, v in zip(users, negs, coo.values().tolist()):
        out.setdefault(u, [])
        if v >= 1.0:
            out[u].append(j)
    return out


class MultipleNrlStep:
    """training_and_validation_step of _BertBPR for objective == "multiple_nrl" (bbpr.py:149-152,187-214) as a callable:
    batch [B,3] = (i, j, w) -> scalar loss with autograd through `forward`.

    forward: item pointer tensor -> embeddings [n, dim] (the item tower on self.all_inputs[ptr] in the reference);
    i_to_ptr / j_to_ptr: user / item index -> item pointer; user_to_negs: round-robin hard-negative lists (mutated)."""

    def __init__(self, forward, i_to_ptr, j_to_ptr, user_to_negs):
        self.forward, self.i_to_ptr, self.j_to_ptr, self.user_to_negs = forward, i_to_ptr, j_to_ptr, user_to_negs

    def __call__(self, batch, batch_idx=0):
        i, j, _w = batch.T
        i, j = i.to(int), j.to(int)
        with torch.no_grad():
            nj = pick_round_robin_negatives(self.user_to_negs, i)
        qid_emb = self.forward(self.i_to_ptr[i.ravel()]).reshape([*i.shape, -1])
        pos_emb = self.forward(self.j_to_ptr[j.ravel()]).reshape([*j.shape, -1])
        neg_emb = self.forward(self.j_to_ptr[nj]).reshape([*j.shape, -1])
        return multiple_nrl_loss(qid_emb, pos_emb, neg_emb)

    training_and_validation_step = __call__


class BertMTStep(MultipleNrlStep):
    """training_and_validation_step of _BertMT (src/ccrec/models/bert_mt.py:105-113): the fine-tune loss of the base step
    weighted alpha / ft_cycles; the corpus-tuning (VAE) term is identically zero for contriever models there
    ((1 - alpha) / ct_cycles * 0).  batch = (ijw, inputs) as the reference's CombinedLoader hands it over."""

    def __init__(self, forward, i_to_ptr, j_to_ptr, user_to_negs, alpha=1.0, ct_cycles=1, ft_cycles=1):
        super().__init__(forward, i_to_ptr, j_to_ptr, user_to_negs)
        self.alpha, self.ct_cycles, self.ft_cycles = float(alpha), ct_cycles, ft_cycles

    def __call__(self, batch, batch_idx=0):
        ijw = batch[0] if isinstance(batch, (tuple, list)) else batch
        ft_loss = super().__call__(ijw, batch_idx)
        return (1 - self.alpha) / self.ct_cycles * 0.0 + self.alpha / self.ft_cycles * ft_loss.mean()

    training_and_validation_step = __call__
