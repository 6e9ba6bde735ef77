"""Mirror of the ranking half of scripts/al_0_rank.py:69-127 (twin: scripts/al_oracle_agent.py:83-129):
build the text -> embedding closure around the item tower and call ranking(); cache/reuse
ranking_profile.pt exactly like the script does.

Differences that stay inside the contract:
  * the tower may be asked for packed bf16 rows directly (output_step mean_pooling_bf16[_cos]) -- the fused
    pool+pack kernel -- when CCREC_EMBEDDING_TYPE is mean_pooling; ranking() re-packs bf16 input bit-exactly;
  * `tokenizer` may be passed in (offline images cannot AutoTokenizer.from_pretrained(model_name)).
"""
import os
import warnings

import torch

from .replica_cache import DataParallel
from .ms_marco_eval import ranking


def generate_ranking_profile(model, model_name, corpus, queries, block_dict=None, tokenizer=None):
    """al_0_rank.py:69-105.  Returns {qid: {pid: score}} in rank order."""
    batch_size = 512
    _gpu_ids = [i for i in range(torch.cuda.device_count())]
    if torch.cuda.device_count() > 0:
        batch_size = batch_size * len(_gpu_ids)

    tokenizer_kw = {
        "truncation": True,
        "padding": True,
        "max_length": int(os.environ.get("CCREC_MAX_LENGTH", 512)),
        "return_tensors": "pt",
    }
    if tokenizer is None:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(model_name)
    model = (model.item_tower if hasattr(model, "item_tower")
             else model.model.item_tower if hasattr(model, "model") else model)
    model.eval()
    model = DataParallel(model.cuda(), device_ids=_gpu_ids).cache_replicas()

    embedding_type = os.environ["CCREC_EMBEDDING_TYPE"]
    if embedding_type != "mean_pooling":
        warnings.warn(f"{embedding_type} != mean_pooling for contriever models")

    def embedding_func(x):
        tokens = tokenizer(x, **tokenizer_kw)
        return model(**tokens, output_step=embedding_type)

    return ranking(corpus, queries, embedding_func, batch_size, block_dict)


def cached_ranking_profile(path_to_ranking_profile, make_model, model_name, corpus, queries, block_dict=None,
                           tokenizer=None, previous_state_dict=None):
    """al_0_rank.py:115-127: reuse ranking_profile.pt if present, else build the model (loading the previous
    AL step's state-dict.pth when given), rank under autocast and save."""
    if os.path.isfile(path_to_ranking_profile):
        return torch.load(path_to_ranking_profile)
    model = make_model()
    if previous_state_dict is not None:
        model.item_tower.load_state_dict(torch.load(previous_state_dict))
    with torch.autocast("cuda"):
        ranking_profile = generate_ranking_profile(model, model_name, corpus, queries, block_dict, tokenizer)
    torch.save(ranking_profile, path_to_ranking_profile)
    return ranking_profile
