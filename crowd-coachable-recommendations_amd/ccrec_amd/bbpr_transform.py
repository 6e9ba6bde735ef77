"""Ranker-API twin of the retrieval path: src/ccrec/models/bbpr.py:466-550 (BertBPR.get_all_embeddings /
BertBPR.transform).  The reference encodes every item title into a host fp32 [n_items, 768] matrix, indexes
user rows, fills a dense host [n_users, n_items] score matrix chunk by chunk and wraps it in a
LazyDenseMatrix.  Here the embeddings are packed to bf16 on device as they are produced and transform()
returns a LOW-RANK lazy score (user rows x item rows); nothing of size n_users x n_items exists unless a
caller asks for .as_tensor().  rime_util._assign_topk consumes it through the fused top-k."""
import os

import numpy as np
import torch

from . import ops


def get_all_embeddings(item_titles, tokenizer, tokenizer_kw, model, batch_size, output_step="embedding", sim_type=None):
    """bbpr.py:466-483: encode all titles in batches of `batch_size` -> packed bf16 [n_items, dim] on device
    (cos: rows L2-normalised by the pack kernel)."""
    ops.require_gpu()
    if sim_type is None:
        sim_type = os.environ["CCREC_SIM_TYPE"]
    if output_step == "embedding":
        output_step = os.environ["CCREC_EMBEDDING_TYPE"]
    all_texts = list(item_titles)
    num = len(all_texts)
    num_batches = int(np.ceil(num / batch_size))
    out = None
    with torch.no_grad():
        for step in range(num_batches):
            if (step % 100) == 0:
                print("Processing", step, "|", num_batches)
            text_batch = all_texts[step * batch_size:(step + 1) * batch_size]
            tokens = tokenizer(text_batch, **tokenizer_kw)
            emb = torch.as_tensor(model(**tokens, output_step=output_step)).cuda()
            if out is None:
                out = torch.empty(num, emb.shape[1], dtype=torch.bfloat16, device=emb.device)
            ops.pack_bf16(emb, normalize=(sim_type == "cos"), out=out[step * batch_size:step * batch_size + len(text_batch)])
    return out


class LowRankScore:
    """Lazy [n_users, n_items] score = U V^T over packed bf16 rows (duck-types the .shape / .T / as_tensor
    surface of rime_lite's LazyScoreBase that the evaluation code touches)."""

    def __init__(self, user_bf16, item_bf16):
        assert user_bf16.dtype == torch.bfloat16 and item_bf16.dtype == torch.bfloat16
        self.user, self.item = user_bf16.contiguous(), item_bf16.contiguous()
        self.shape = (self.user.shape[0], self.item.shape[0])

    def __len__(self):
        return self.shape[0]

    @property
    def T(self):
        return LowRankScore(self.item, self.user)

    def as_tensor(self, device=None):
        """Dense scores (this materialises n_users x n_items fp32): canonical fp64-ordered values for small problems,
        the MFMA tile kernel (same bf16 rows, fp32 accumulation) above 4e9 multiply-adds."""
        big = self.shape[0] * self.shape[1] * self.user.shape[1] > 4e9 and self.user.shape[1] % 64 == 0
        s = ops.CorpusIndex(self.item).debug_scores(self.user, canonical=not big)
        return s if device is None else s.to(device)

    def numpy(self):
        return self.as_tensor().cpu().numpy()

    def topk(self, k):
        """-> (scores [n_users, k], item rows [n_users, k]) in canonical order."""
        return ops.CorpusIndex(self.item).search(self.user, k)


def transform(all_emb_bf16, i_to_ptr, j_to_ptr):
    """bbpr.py:526-550 without the dense matrix: user rows = all_emb[i_to_ptr], item rows = all_emb[j_to_ptr]."""
    i = torch.as_tensor(np.asarray(i_to_ptr), dtype=torch.long, device=all_emb_bf16.device)
    j = torch.as_tensor(np.asarray(j_to_ptr), dtype=torch.long, device=all_emb_bf16.device)
    return LowRankScore(all_emb_bf16[i], all_emb_bf16[j])
