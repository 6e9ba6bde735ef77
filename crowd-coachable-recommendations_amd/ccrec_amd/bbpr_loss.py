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
