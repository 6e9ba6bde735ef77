"""Encoder wrapper mirror of src/ccrec/models/item_tower.py: the HF encoder stays PyTorch-ROCm, the
pooling + pack that follows it is the fused HIP kernel.

  ItemTowerBase   item_tower.py:8-96    (text -> inputs -> cls -> embedding; tokenizer_kw defaults :17-22)
  NaiveItemTower  item_tower.py:99-151  (output_step cls|mu|mean -> CLS row, mean_layer_norm -> LayerNorm(CLS),
                                         mean_pooling -> masked mean over tokens, un-normalised :137-147)
Extra (not in the reference): output_step="mean_pooling_bf16" / forward_packed() return the packed bf16
rows the retrieval index consumes, so fp32 [B,768] never round-trips through HBM twice.
"""
import collections
import os
import warnings

import torch

from . import ops
from .data_parallel import DataParallel


class ItemTowerBase(torch.nn.Module):
    def __init__(self, *module_list, tokenizer=None, tokenizer_kw={}):
        super().__init__()
        self.module_list = module_list
        self.tokenizer = tokenizer
        _default_tokenizer_kw = {
            "truncation": True,
            "padding": "max_length",
            "max_length": int(os.environ.get("CCREC_MAX_LENGTH", 200)),
            "return_tensors": "pt",
        }
        self.tokenizer_kw = {**_default_tokenizer_kw, **tokenizer_kw}

    @property
    def device(self):
        return self.module_list[-1].device

    def text_to_inputs(self, text):
        return self.tokenizer(text, **self.tokenizer_kw)

    def forward(self, cls=None, text=None, input_step="inputs", output_step="embedding", **inputs):
        raise NotImplementedError(f"{self.__class__.__name__} does not support {input_step}->{output_step} forward")

    def to_map_fn(self, input_step, output_step, data_parallel=False, sample_param=0):
        """item_tower.py:46-89: a no_grad function mapping a batch dict to {output_step: ndarray}."""
        assert self.tokenizer is not None or input_step != "text", "map_fn with text input requires tokenizer attribute"
        self.eval()
        if hasattr(self, "set_sample_param"):
            self.set_sample_param(sample_param)

        def wrap_dict(x):
            if isinstance(x, collections.abc.Mapping):
                return x
            if input_step == "cls":
                return {"cls": x}
            if input_step == "text":
                return {"text": x}
            return NotImplemented

        if input_step == "text":
            step = "inputs"
            tokenizer, tokenizer_kw = self.tokenizer, self.tokenizer_kw
            wrap_text = lambda x: tokenizer(x["text"], **tokenizer_kw)  # noqa: E731
        else:
            step = input_step
            wrap_text = lambda x: x  # noqa: E731

        model = self
        if data_parallel:
            model = DataParallel(self.cuda()).cache_replicas()
            wrap_device = lambda x: {k: v.cuda() for k, v in x.items()}  # noqa: E731
        else:
            wrap_device = lambda x: x  # noqa: E731

        return torch.no_grad()(
            lambda x: {output_step: model(**wrap_device(wrap_text(wrap_dict(x))), input_step=step,
                                          output_step=output_step).float().cpu().numpy()})


class NaiveItemTower(ItemTowerBase):
    """standard_layer_norm on top of the CLS token, or masked mean pooling."""

    def __init__(self, cls_model, standard_layer_norm, **kw):
        super().__init__(cls_model, standard_layer_norm, **kw)
        self.cls_model = cls_model
        self.standard_layer_norm = standard_layer_norm

    def forward(self, cls=None, text=None, input_step="inputs", output_step="embedding", **inputs):
        if input_step == "text":
            inputs = self.text_to_inputs(text=text)
            input_step = "inputs"

        if input_step == "inputs":
            inputs = {k: v.to(self.cls_model.device) for k, v in inputs.items()}
            last_hidden_state = self.cls_model(**inputs).last_hidden_state
            cls = last_hidden_state[:, 0]
        else:  # cls
            cls = cls.to(self.device)

        if output_step == "embedding":
            output_step = os.environ["CCREC_EMBEDDING_TYPE"]
            warnings.warn(f"{self.__class__} inferring output_step from CCREC_EMBEDDING_TYPE as {output_step}")

        if output_step in ["cls", "mu", "mean"]:
            return cls
        elif output_step == "mean_layer_norm":
            return self.standard_layer_norm(cls)
        elif output_step in ("mean_pooling", "mean_pooling_bf16", "mean_pooling_bf16_cos"):
            assert input_step != "cls", "cannot create mean pooling from cls"
            mask = inputs["attention_mask"]
            want_bf16 = output_step != "mean_pooling"
            pooled, packed = ops.meanpool_pack(last_hidden_state, mask, normalize=output_step.endswith("_cos"),
                                               want_f32=not want_bf16, want_bf16=want_bf16)
            return packed if want_bf16 else pooled  # unnormalized fp32, as item_tower.py:147

        raise NotImplementedError(f"{self.__class__.__name__} does not support {input_step}->{output_step} forward")
