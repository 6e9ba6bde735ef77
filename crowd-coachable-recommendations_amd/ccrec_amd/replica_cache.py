"""Inference-only DataParallel that broadcasts the encoder to the GPUs ONCE.

API kept from src/ccrec/util/data_parallel.py:8-20: `DataParallel(module, device_ids).cache_replicas()` returns the
wrapper itself, and every later forward reuses the stored copies instead of re-broadcasting ~438 MB of BERT weights
per batch.  This exists for in-process compatibility with the reference's scripts; the MI355X-native multi-GPU
layout is one process per GPU over a row-sharded corpus (ccrec_amd.dist), which has no replica broadcast at all.
"""
import torch
from torch.nn.parallel import replicate as _broadcast_module


class DataParallel(torch.nn.DataParallel):
    _by_device = None

    def cache_replicas(self):
        print("caching replicas")
        if self.device_ids:  # nothing to copy on a CPU-only host
            copies = _broadcast_module(self.module, self.device_ids, detach=True)  # detached: used under no_grad only
            self._by_device = dict(zip(self.device_ids, copies))
        return self

    def replicate(self, module, device_ids):
        """Called by torch's DataParallel.forward with the devices that received a chunk of the batch."""
        if self._by_device is None:
            return super().replicate(module, device_ids)
        return [self._by_device[d] for d in device_ids]
