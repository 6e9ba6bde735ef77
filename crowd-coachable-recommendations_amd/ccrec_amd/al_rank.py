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

from .ms_marco_eval import ranking
from .replica_cache import DataParallel

PER_GPU_BATCH = 512   # al_0_rank.py:70-74: 512 texts per visible GPU and batch


def unwrap_item_tower(model):
    """The encoder that maps token tensors to embeddings: `model.item_tower`, or `model.model.item_tower` for the
    ranker wrappers (BertBPR / BertMT keep the Lightning module in .model), or the object itself (al_0_rank.py:84-90)."""
    for owner in (model, getattr(model, "model", None)):
        tower = getattr(owner, "item_tower", None) if owner is not None else None
        if tower is not None:
            return tower
    return model


class TextEmbedder:
    """The text -> embedding closure the script hands to ranking() (al_0_rank.py:76-81, 98-101), as an object:
    tokenise with dynamic padding (`padding=True`), truncate at CCREC_MAX_LENGTH, run the encoder replicas."""

    def __init__(self, runner, tokenizer, output_step, max_length):
        self.runner, self.tokenizer, self.output_step = runner, tokenizer, output_step
        self.tokenizer_kw = dict(padding=True, truncation=True, max_length=max_length, return_tensors="pt")

    def __call__(self, texts):
        return self.runner(**self.tokenizer(texts, **self.tokenizer_kw), output_step=self.output_step)


def generate_ranking_profile(model, model_name, corpus, queries, block_dict=None, tokenizer=None, length_sorted=None):
    """al_0_rank.py:69-105.  Returns {qid: {pid: score}} in rank order.

    length_sorted (default: the environment variable CCREC_LENGTH_SORTED == "1"): encode through encode.LengthSortedEncoder --
    texts tokenised once without padding, sorted by length, batched under a token budget, pooled and packed straight into the
    resident shard (no host copy per batch), the next chunk tokenised while the GPU encodes -- instead of the script's corpus-order
    batches of 512 padded to their longest text.  Same contract and ranking rule; the embeddings differ from the padded batches'
    by the encoder's own summation-order noise.  One process per GPU (this process's current device), not DataParallel."""
    if tokenizer is None:   # needs the hub or a local cache; offline callers pass their tokenizer in
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(model_name)
    gpus = list(range(torch.cuda.device_count()))
    tower = unwrap_item_tower(model)
    tower.eval()
    if length_sorted is None:
        length_sorted = os.environ.get("CCREC_LENGTH_SORTED", "0") == "1"
    if length_sorted:
        from .encode import LengthSortedEncoder, ranking_sharded
        output_step = os.environ["CCREC_EMBEDDING_TYPE"]
        if output_step != "mean_pooling":
            warnings.warn(f"{output_step} != mean_pooling for contriever models")
        encoder = LengthSortedEncoder(tower.cuda(), tokenizer, max_length=int(os.environ.get("CCREC_MAX_LENGTH", 512)), output_step=output_step)
        try:
            return ranking_sharded(corpus, queries, encoder, block_dict=block_dict)
        finally:
            encoder.close()
    runner = DataParallel(tower.cuda(), device_ids=gpus).cache_replicas()
    output_step = os.environ["CCREC_EMBEDDING_TYPE"]
    if output_step != "mean_pooling":
        warnings.warn(f"{output_step} != mean_pooling for contriever models")
    embed = TextEmbedder(runner, tokenizer, output_step, int(os.environ.get("CCREC_MAX_LENGTH", 512)))
    return ranking(corpus, queries, embed, PER_GPU_BATCH * max(1, len(gpus)), block_dict)


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
