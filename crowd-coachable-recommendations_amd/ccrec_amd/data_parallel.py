"""src/ccrec/util/data_parallel.py:8-20: torch.nn.DataParallel that replicates the module ONCE and
reuses the replicas on every forward (inference only).  Kept for drop-in compatibility inside one
process; the MI355X-native multi-GPU layout is one process per GPU with a row-sharded corpus
(ccrec_amd.dist), where no replica broadcast exists at all."""
from torch.nn.parallel.data_parallel import DataParallel as _DataParallel
from torch.nn.parallel.replicate import replicate as _replicate


class DataParallel(_DataParallel):
    def cache_replicas(self):
        print("caching replicas")
        if self.device_ids:
            self._replicas = _replicate(self.module, self.device_ids, detach=True)  # detach: no_grad use only
        return self

    def replicate(self, module, device_ids):
        if hasattr(self, "_replicas"):
            return [self._replicas[self.device_ids.index(d)] for d in device_ids]
        return super().replicate(module, device_ids)
