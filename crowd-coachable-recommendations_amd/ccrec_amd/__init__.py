"""ccrec_amd -- MI355X-native retrieval hot path behind the ccrec / rime_lite call surface.

Mirrors (reference file:line):
  ms_marco_eval.generate_embeddings / cos_sim / ranking   scripts/ms_marco_eval.py:123-162,189-235
  item_tower.ItemTowerBase / NaiveItemTower               src/ccrec/models/item_tower.py:8-151
  replica_cache.DataParallel.cache_replicas               src/ccrec/util/data_parallel.py:8-20
  rime_util._assign_topk                                  src/rime_lite/util/__init__.py:117-155
  bbpr_loss.multiple_nrl_loss                             src/ccrec/models/bbpr.py:187-214
Config: the CCREC_* environment variables of src/ccrec/__init__.py:8-25 (same names, defaults, options).
"""
import os
import warnings

env_defaults = [
    ("CCREC_EMBEDDING_TYPE", "mean_layer_norm", ["cls", "mu", "mean", "mean_pooling", "mean_layer_norm"]),
    ("CCREC_MAX_LENGTH", "256", None),
    ("CCREC_SIM_TYPE", "cos", ["cos", "dot"]),
    ("CCREC_TRAIN_MAIN", "bmt_main", ["bmt_main", "bbpr_main"]),
    ("CCREC_TRAINING_PRECISION", "32", ["32", "bf16"]),
    ("CCREC_BBPR_INV_TEMPERATURE", "20", None),
    ("CCREC_DISPLAY_LENGTH", "250", None),
    ("CCREC_NON_BLOCKING", "1", ["0", "1"]),
]


def init_env_defaults(verbose=False):
    """Same names / defaults / validation as ccrec/__init__.py:28-48 (the options assert included)."""
    for name, default, options in env_defaults:
        val = os.environ.setdefault(name, default)
        if options is not None:
            assert val in options, f"{name}={val} not in {options}"
        if verbose:
            print(f"{name}={val}; options: {options}")
    if os.environ["CCREC_SIM_TYPE"] == "dot" and float(os.environ["CCREC_BBPR_INV_TEMPERATURE"]) >= 20:
        warnings.warn("dot similarity works best with small inv_temperature")


init_env_defaults()

from . import _lib  # noqa: E402
from .ops import (  # noqa: E402,F401
    CorpusIndex, apply_block, inbatch_ce, meanpool_pack, merge_topk, pack_bf16, require_gpu,
)
