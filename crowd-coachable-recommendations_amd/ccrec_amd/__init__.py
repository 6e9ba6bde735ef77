"""ccrec_amd -- MI355X-native retrieval hot path behind the ccrec / rime_lite call surface.

Mirrors (reference file:line):
  ms_marco_eval.generate_embeddings / cos_sim / ranking   scripts/ms_marco_eval.py:123-162,189-235
  item_tower.ItemTowerBase / NaiveItemTower               src/ccrec/models/item_tower.py:8-151
  replica_cache.DataParallel.cache_replicas               src/ccrec/util/data_parallel.py:8-20
  rime_util._assign_topk                                  src/rime_lite/util/__init__.py:117-155
  bbpr_loss.multiple_nrl_loss / MultipleNrlStep           src/ccrec/models/bbpr.py:149-227
  bbpr_transform.get_all_embeddings / transform           src/ccrec/models/bbpr.py:466-550
  rime_util.evaluate_item_rec / evaluate_assigned         src/rime_lite/metrics/__init__.py:52-89
  al_rank.generate_ranking_profile                        scripts/al_0_rank.py:69-127
  encode.LengthSortedEncoder / ranking_sharded            (variable-length encode, per-rank shards: SURVEY 8 f2)
  al_request.build_requests / generate_train_data         scripts/al_0_rank.py:136-218, scripts/al_oracle_agent.py:134-186
  bm25.BM25 / ranking_bm25                                scripts/bm_25.py, scripts/ms_marco_eval.py:165-186
  evaluation.rank_metrics                                 scripts/al_0_rank.py:130-133 (BEIR evaluate_custom, mrr)
  al_step.run_rank_step                                   scripts/al_0_rank.py:107-218 as one call
Config: the CCREC_* environment variables of src/ccrec/__init__.py:8-25 (same names, defaults, options).
"""
import os
import warnings
from typing import NamedTuple, Optional, Tuple


class _EnvVar(NamedTuple):
    name: str
    default: str
    choices: Optional[Tuple[str, ...]]   # None = free-form value


# same variable names, defaults and allowed values as the reference's import-time configuration
# (src/ccrec/__init__.py:8-25); only the entries that reach this path are kept
_ENV = (
    _EnvVar("CCREC_EMBEDDING_TYPE", "mean_layer_norm", ("cls", "mu", "mean", "mean_pooling", "mean_layer_norm")),
    _EnvVar("CCREC_MAX_LENGTH", "256", None),
    _EnvVar("CCREC_SIM_TYPE", "cos", ("cos", "dot")),
    _EnvVar("CCREC_TRAIN_MAIN", "bmt_main", ("bmt_main", "bbpr_main")),
    _EnvVar("CCREC_TRAINING_PRECISION", "32", ("32", "bf16")),
    _EnvVar("CCREC_BBPR_INV_TEMPERATURE", "20", None),
    _EnvVar("CCREC_DISPLAY_LENGTH", "250", None),
    _EnvVar("CCREC_NON_BLOCKING", "1", ("0", "1")),
)
env_defaults = [(v.name, v.default, list(v.choices) if v.choices else None) for v in _ENV]   # reference-shaped view


def init_env_defaults(verbose=False):
    """Fill in missing CCREC_* variables and validate the present ones (AssertionError on a value outside the
    allowed set, like the reference's import-time check, src/ccrec/__init__.py:28-48)."""
    for var in _ENV:
        value = os.environ.setdefault(var.name, var.default)
        assert var.choices is None or value in var.choices, f"{var.name}={value} not in {list(var.choices)}"
        if verbose:
            print(f"{var.name}={value}")
    if os.environ["CCREC_SIM_TYPE"] == "dot" and float(os.environ["CCREC_BBPR_INV_TEMPERATURE"]) >= 20:
        warnings.warn("dot similarity works best with small inv_temperature")


init_env_defaults()

from . import _lib  # noqa: E402
from .ops import (  # noqa: E402,F401
    CorpusIndex, apply_block, colsum_bf16, inbatch_ce, meanpool, meanpool_pack, merge_topk, pack_bf16, require_gpu,
)
