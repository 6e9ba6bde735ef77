#!/usr/bin/env python3
"""Race screen for the LDS-DMA ring protocol: repeat the same NQ-shaped search many times and require bit-identical
ids and scores every time (a RAW/WAR slip on the ring shows up as a rare wrong tile).  Also runs the 32x32x16 kernel."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]


def main():
    from ccrec_amd import ops
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    g = torch.Generator(device="cuda").manual_seed(7)
    n, nq, d, k = 1_340_000, 3452, 768, 100
    D = ops.pack_bf16(torch.randn(n, d, device="cuda", generator=g) / d ** 0.5)
    Q = ops.pack_bf16(torch.randn(nq, d, device="cuda", generator=g) / d ** 0.5)
    for variant in ("1", "0"):
        os.environ["CCR_MFMA16"] = variant
        index = ops.CorpusIndex(D)
        s0, i0 = index.search(Q, k)
        bad = 0
        for it in range(iters):
            s, i = index.search(Q, k)
            if not (torch.equal(i, i0) and torch.equal(s.view(torch.int32), s0.view(torch.int32))):
                bad += 1
            if it % 50 == 0:
                print(f"mfma16={variant} iteration {it}: mismatches so far {bad}", flush=True)
        print(f"mfma16={variant}: {iters} repeats, {bad} mismatches, fallbacks {index.last_stats()['n_fallback']}", flush=True)
        assert bad == 0


if __name__ == "__main__":
    main()
