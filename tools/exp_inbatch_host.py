#!/usr/bin/env python3
"""Where the 0.2 ms of the in-batch step go on the HOST: wall time of each Python stage of bench.inbatch_side_run's `ours()` without any
synchronisation inside the loop (the GPU side is ~0.13 ms per step), 200 steps."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
from ccrec_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
B, d = 1024, 768
g = torch.Generator(device=dev).manual_seed(0)
q, p, n = (torch.randn(B, d, device=dev, generator=g) * d ** -0.5 for _ in range(3))
acc = {"clone": 0.0, "forward": 0.0, "backward": 0.0}
for it in range(260):
    if it == 60:
        torch.cuda.synchronize()
        acc = {k: 0.0 for k in acc}
        t_all = time.perf_counter()
    t0 = time.perf_counter()
    a, b, c = (t.clone().requires_grad_(True) for t in (q, p, n))
    t1 = time.perf_counter()
    loss = ops.inbatch_ce(a, b, c, 20.0)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    acc["clone"] += t1 - t0
    acc["forward"] += t2 - t1
    acc["backward"] += t3 - t2
torch.cuda.synchronize()
total = (time.perf_counter() - t_all) / 200 * 1e3
print({k: round(v / 200 * 1e3, 4) for k, v in acc.items()}, "host sum ms", round(sum(acc.values()) / 200 * 1e3, 4), "| wall per step ms", round(total, 4))
