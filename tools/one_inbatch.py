#!/usr/bin/env python3
"""The bench line's configs[4] side run alone (bench.inbatch_side_run: B = 1024, d = 768 forward + backward): JSON on stdout.
Under rocprofv3 --kernel-trace --stats it gives the per-kernel times of the step."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]

if __name__ == "__main__":
    import bench
    torch.cuda.set_device(0)
    print(json.dumps(bench.inbatch_side_run(torch.device("cuda", 0))), flush=True)
