"""Item-tower surface of the reference (src/ccrec/models/item_tower.py) over the fused pooling kernel.

Kept from the reference, because callers depend on it:
  * class names ItemTowerBase / NaiveItemTower and their constructor arguments (:13, :102);
  * forward(cls=None, text=None, input_step="inputs", output_step="embedding", **inputs) (:34-44, :107-114) with
    the output steps cls | mu | mean (CLS row), mean_layer_norm (LayerNorm of the CLS row), mean_pooling (masked
    token mean, un-normalised, :137-147), "embedding" = os.environ["CCREC_EMBEDDING_TYPE"], NotImplementedError else;
  * to_map_fn(input_step, output_step, data_parallel=False, sample_param=0) -> no_grad batch function (:46-89);
  * tokenizer_kw defaults: truncation, padding="max_length", max_length=CCREC_MAX_LENGTH (200), return_tensors="pt".
New here: the mean pooling runs in ccr_meanpool_pack_bf16 (one pass over the hidden states), and the extra output
steps "mean_pooling_bf16" / "mean_pooling_bf16_cos" hand back the packed bf16 rows the retrieval index consumes.
The HF encoder itself stays PyTorch-ROCm.
"""
import collections.abc
import os
import warnings

import torch

from . import ops
from .replica_cache import DataParallel

_CLS_STEPS = ("cls", "mu", "mean")
_POOL_STEPS = {"mean_pooling": (False, False), "mean_pooling_bf16": (True, False), "mean_pooling_bf16_cos": (True, True)}


def _tokenizer_defaults():
    return dict(truncation=True, padding="max_length", max_length=int(os.environ.get("CCREC_MAX_LENGTH", 200)),
                return_tensors="pt")


class _BatchMapper:
    """The callable to_map_fn returns: batch (dict | tensor | list of str) -> {output_step: ndarray}."""

    def __init__(self, tower, runner, input_step, output_step, to_device):
        self.tower, self.runner = tower, runner
        self.input_step, self.output_step, self.to_device = input_step, output_step, to_device

    def _as_inputs(self, batch):
        if not isinstance(batch, collections.abc.Mapping):
            if self.input_step not in ("cls", "text"):
                return NotImplemented
            batch = {self.input_step: batch}
        if self.input_step == "text":
            batch = self.tower.tokenizer(batch["text"], **self.tower.tokenizer_kw)
        if self.to_device:
            batch = {name: value.cuda() for name, value in batch.items()}
        return batch

    @torch.no_grad()
    def __call__(self, batch):
        step = "inputs" if self.input_step == "text" else self.input_step
        out = self.runner(**self._as_inputs(batch), input_step=step, output_step=self.output_step)
        return {self.output_step: out.float().cpu().numpy()}


class ItemTowerBase(torch.nn.Module):
    """text -> inputs -> cls -> embedding; a tokenizer is needed for the text entry points."""

    def __init__(self, *module_list, tokenizer=None, tokenizer_kw={}):
        super().__init__()
        self.module_list = module_list
        self.tokenizer = tokenizer
        self.tokenizer_kw = {**_tokenizer_defaults(), **tokenizer_kw}

    @property
    def device(self):
        return self.module_list[-1].device

    def text_to_inputs(self, text):
        return self.tokenizer(text, **self.tokenizer_kw)

    def forward(self, cls=None, text=None, input_step="inputs", output_step="embedding", **inputs):
        raise NotImplementedError(f"{type(self).__name__} does not support {input_step}->{output_step} forward")

    def to_map_fn(self, input_step, output_step, data_parallel=False, sample_param=0):
        """Puts the tower in eval mode (and on the GPUs when data_parallel) as a side effect, like the reference."""
        if input_step == "text" and self.tokenizer is None:
            raise AssertionError("map_fn with text input requires tokenizer attribute")
        self.eval()
        if hasattr(self, "set_sample_param"):
            self.set_sample_param(sample_param)
        runner = DataParallel(self.cuda()).cache_replicas() if data_parallel else self
        return _BatchMapper(self, runner, input_step, output_step, to_device=bool(data_parallel))


class NaiveItemTower(ItemTowerBase):
    """HF encoder + (CLS | LayerNorm(CLS) | masked mean pooling)."""

    def __init__(self, cls_model, standard_layer_norm, **kw):
        super().__init__(cls_model, standard_layer_norm, **kw)
        self.cls_model = cls_model
        self.standard_layer_norm = standard_layer_norm

    def _encode(self, inputs, cls_only=False):
        """-> last hidden state [B, L, hidden]; with cls_only (the caller reads only [:, 0]) the kernel forward may return [B, 1, hidden]."""
        on_model = {name: value.to(self.cls_model.device) for name, value in inputs.items()}
        hidden = self._encode_on_kernels(on_model, cls_only)
        return hidden if hidden is not None else self.cls_model(**on_model).last_hidden_state

    def _encode_on_kernels(self, inputs, cls_only=False):
        """Inference forward on the library's layer kernels (fused_bert) when all of this holds: no gradients, eval mode, the
        caller asks for reduced precision (autocast -- al_0_rank.py:125 -- or CCREC_FUSED_ENCODER=1), the encoder is a BertModel the
        kernels cover, and the batch is plain right-padded token ids.  None = run the module."""
        from . import fused_bert
        dtype = fused_bert.kernel_dtype("auto")      # the caller's autocast type: fp16 under the reference's autocast(), al_0_rank.py:125
        if dtype is None or torch.is_grad_enabled() or getattr(self.cls_model, "training", True):
            return None
        if not set(inputs) <= {"input_ids", "attention_mask", "token_type_ids"} or "input_ids" not in inputs or "attention_mask" not in inputs:
            return None
        if not inputs["input_ids"].is_cuda:
            return None
        enc = fused_bert.for_model(self.cls_model)
        if enc is None or ("token_type_ids" in inputs and not enc.has_token_types):
            return None
        lengths = fused_bert.prefix_lengths(inputs["attention_mask"])
        if lengths is None:
            return None
        enc.refresh(dtype)      # the weight copies follow the module's parameters (fine-tuning between two ranking steps)
        return enc.forward(inputs["input_ids"], lengths, inputs.get("token_type_ids"), cls_only=cls_only, dtype=dtype)

    def forward(self, cls=None, text=None, input_step="inputs", output_step="embedding", **inputs):
        if input_step == "text":
            inputs, input_step = self.text_to_inputs(text=text), "inputs"
        hidden = None
        if input_step == "inputs":
            step = os.environ.get("CCREC_EMBEDDING_TYPE", "") if output_step == "embedding" else output_step
            hidden = self._encode(inputs, cls_only=step in _CLS_STEPS or step == "mean_layer_norm")
            cls = hidden[:, 0]
        else:
            cls = cls.to(self.device)

        if output_step == "embedding":
            output_step = os.environ["CCREC_EMBEDDING_TYPE"]
            warnings.warn(f"{self.__class__} inferring output_step from CCREC_EMBEDDING_TYPE as {output_step}")

        if output_step in _CLS_STEPS:
            return cls
        if output_step == "mean_layer_norm":
            return self.standard_layer_norm(cls)
        if output_step in _POOL_STEPS:
            assert hidden is not None, "cannot create mean pooling from cls"
            packed, cosine = _POOL_STEPS[output_step]
            if not packed and torch.is_grad_enabled() and hidden.requires_grad:
                # training forward (bbpr.py:130-141 calls the tower with gradients on): same kernel, with a backward
                return ops.meanpool(hidden, inputs["attention_mask"])
            pooled_f32, pooled_bf16 = ops.meanpool_pack(hidden, inputs["attention_mask"], normalize=cosine,
                                                        want_f32=not packed, want_bf16=packed)
            return pooled_bf16 if packed else pooled_f32
        raise NotImplementedError(f"{type(self).__name__} does not support {input_step}->{output_step} forward")
