#!/usr/bin/env python3
"""Timing-only ablations of the 16x16x32 main pass.  Runs ON THE GPU BOX against the DIAGNOSTIC library
(tools/build_diag.sh -> crowd-coachable-recommendations_amd/lib_diag/libccr_hip.so; its CCR_GEMM_DBG modes return WRONG results and
exist in no shipped build).  One process per variant (the knob is read at index creation; the library is loaded once per process),
NQ shape, CCR_PROGRESSIVE=0 (ONE launch: comparable items) and the default three-launch plan.

  python3 tools/exp_main_pass_ablation.py [outfile]        # driver
  python3 tools/exp_main_pass_ablation.py --one            # one variant (environment set by the driver)
"""
import os
import re
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "crowd-coachable-recommendations_amd")
PITCH = [(1152, f"thresholds +inf, DMA addresses generated with a row pitch of {p} elements ({2 * p} B) instead of 768 (1536 B)", {"CCR_DBG_PITCH": str(p)})
         for p in (768, 784, 800, 832, 896, 1024)]
POLICY = [(2176, "thresholds +inf, cache policy sc1 on every DMA piece"), (4224, "thresholds +inf, sc0 on every piece"), (8320, "thresholds +inf, sc0 sc1 on every piece")]
# --pitch2: the honest form -- the index is given the first n * 768 / pitch rows of the n-row array (CCR_DBG_ALLOC_ROWS = n), so that NO
# address is clamped; each pitch is compared with the plain kernel on the same number of rows
N_FULL = 2681468
PITCH2 = []
for _p in (768, 832, 896, 1024):
    _rows = N_FULL * 768 // _p // 256 * 256
    PITCH2.append((128, f"{_rows} rows at the array's own pitch (baseline of the next row)", {"ABL_USE_ROWS": str(_rows)}))
    PITCH2.append((1152, f"{_rows} rows, DMA addresses with a row pitch of {2 * _p} B (no clamping: the array holds {N_FULL} rows)",
                   {"ABL_USE_ROWS": str(_rows), "CCR_DBG_PITCH": str(_p), "CCR_DBG_ALLOC_ROWS": str(N_FULL), "CCR_DBG_ALLOC_Q": str(3452 * _p // 768 + 8)}))
VARIANTS = [
    (0, "production kernel (hits recorded)"),
    (128, "complete kernel, thresholds +inf (no hit): BASELINE of the rows below"),
    (384, "thresholds +inf, every DMA piece reads whole 128-byte lines (8 rows x 128 B instead of 16 rows x 64 B; same bytes, same pieces)"),
    (640, "thresholds +inf, the traffic of a K-TILED layout: every piece ONE contiguous KiB, a sub-stage's 16 + 16 KiB contiguous (same bytes, same pieces)"),
    (4, "no DMA at all (ring zero-filled once)"),
    (32, "corpus-only DMA (query region of the ring left zero; 2 pieces per wave and sub-stage)"),
    (64, "query-only DMA (corpus region left zero; 2 pieces per wave and sub-stage)"),
    (8, "DMA + LDS reads + barriers + filter trees, no MFMA"),
    (40, "corpus-only DMA, no MFMA"),
    (12, "LDS reads + barriers + filter trees only (no DMA, no MFMA)"),
]


def one():
    sys.path[:0] = [ROOT, PKG]
    import torch
    from ccrec_amd import _lib
    _lib.LIB_PATH = os.path.join(PKG, "lib_diag", "libccr_hip.so")
    from ccrec_amd import ops
    n, nq, d, k = int(os.environ.get("ABL_ROWS", 2681468)), int(os.environ.get("ABL_QUERIES", 3452)), 768, 100
    g = torch.Generator(device="cuda").manual_seed(1234)
    D = torch.empty(n, d, dtype=torch.bfloat16, device="cuda")
    for lo in range(0, n, 1 << 19):
        hi = min(n, lo + (1 << 19))
        D[lo:hi] = (torch.randn(hi - lo, d, generator=g, device="cuda") / d ** 0.5).to(torch.bfloat16)
    nq_alloc = int(os.environ.get("CCR_DBG_ALLOC_Q", nq))      # --pitch2: the query array holds more rows than are searched (no clamping)
    Q = (torch.randn(max(nq, nq_alloc), d, generator=g, device="cuda") / d ** 0.5).to(torch.bfloat16)[:nq]
    use = int(os.environ.get("ABL_USE_ROWS", n))      # --pitch2: the index sees the first `use` rows of the n-row array
    ix = ops.CorpusIndex(D[:use])
    dbg = int(os.environ.get("CCR_GEMM_DBG", "0"))
    ms = []
    for it in range(8):
        ix.search(Q, k)
        torch.cuda.synchronize()
        if dbg == 0:
            ms.append(ix.last_stats()["ms_main"])
    if dbg == 0:
        for m in ms:
            print(f"[ccr diag] CCR_GEMM_DBG=0 main pass {m:.4f} ms", file=sys.stderr)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(ROOT, "gpurun_out", "r06_main_pass_ablation.txt")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    lines = ["# tools/exp_main_pass_ablation.py: gemm_topk16_kernel<EPI_FILTER>, NQ 2,681,468 x 768 x 3,452 queries, top-100, one box, diagnostic library",
             "# main-pass time by the library's own HIP events (ms), median of the last 5 of 8 searches per process"]
    for plan, env_plan in (("single launch (CCR_PROGRESSIVE=0)", {"CCR_PROGRESSIVE": "0"}), ("default plan (three launches, thresholds re-tightened)", {})):
        lines.append(f"## {plan}")
        base = None
        for dbg, what, *more in (PITCH2 if "--pitch2" in sys.argv else [v for v in VARIANTS if v[0] in (0, 128)] + (PITCH if "--pitch" in sys.argv else POLICY) if ("--pitch" in sys.argv or "--policy" in sys.argv or "--pitch2" in sys.argv) else VARIANTS):
            env = dict(os.environ, CCR_GEMM_DBG=str(dbg), CCR_MFMA16="1", **env_plan, **(more[0] if more else {}))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True, timeout=900)
            vals = [float(m) for m in re.findall(r"main pass ([0-9.]+) ms", r.stderr)]
            if r.returncode != 0 or len(vals) < 6:
                lines.append(f"dbg={dbg:3d}  FAILED rc={r.returncode} {r.stderr[-300:]!r}")
                continue
            v = sorted(vals[-5:])[2]
            if dbg == 128:
                base = v
            rel = f"  ({v / base:5.3f} of baseline)" if base and dbg not in (0, 128) else ""
            if dbg == 128:
                base = v
            lines.append(f"dbg={dbg:3d}  {v:8.3f} ms{rel}   {what}")
            print(lines[-1], flush=True)
    open(out, "w").write("\n".join(lines) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    one() if "--one" in sys.argv else main()
