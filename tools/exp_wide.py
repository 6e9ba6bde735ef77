#!/usr/bin/env python3
"""Same-box A/B of the main pass on 256 x 384 tiles (CCR_WIDE, csrc/ccr_fused.hip gemm_topk16w_kernel) against the 256 x 256 kernel, at the
NQ shape.  Runs ON THE GPU BOX, one process per variant.  Variants with CCR_GEMM_DBG=128 (thresholds +inf: the complete kernel without a
single hit) need the DIAGNOSTIC library (tools/build_diag.sh) and return no results; every other variant must return the ids and score
bits of the first one.

  python3 tools/exp_wide.py [outfile] [--rows N --queries Q --k K] [--set NAME]"""
import os
import re
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "crowd-coachable-recommendations_amd")


def arg(name, dflt, conv=int):
    return conv(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else dflt


def one():
    sys.path[:0] = [ROOT, PKG]
    import torch
    from ccrec_amd import _lib
    diag = os.environ.get("CCR_GEMM_DBG", "0") != "0"
    if diag:
        _lib.LIB_PATH = os.path.join(PKG, "lib_diag", "libccr_hip.so")
    from ccrec_amd import ops
    n, nq, d, k = arg("--rows", 2681468), arg("--queries", 3452), arg("--dim", 768), arg("--k", 100)
    g = torch.Generator(device="cuda").manual_seed(1234)
    D = torch.empty(n, d, dtype=torch.bfloat16, device="cuda")
    for lo in range(0, n, 1 << 19):
        hi = min(n, lo + (1 << 19))
        D[lo:hi] = (torch.randn(hi - lo, d, generator=g, device="cuda") / d ** 0.5).to(torch.bfloat16)
    Q = (torch.randn(nq, d, generator=g, device="cuda") / d ** 0.5).to(torch.bfloat16)
    ix = ops.CorpusIndex(D)
    ms, tot = [], []
    for it in range(8):
        s, i = ix.search(Q, k)
        torch.cuda.synchronize()
        st = ix.last_stats()
        ms.append(st["ms_main"])
        tot.append(st["ms_total"])
    if diag:
        print("RESULT diag")
        return
    ref = os.environ["QD_REF"]
    same = "ref"
    if os.path.exists(ref):
        rs, ri = torch.load(ref)
        same = "same" if (torch.equal(ri, i.cpu()) and torch.equal(rs.view(torch.int32), s.cpu().view(torch.int32))) else "DIFFERENT"
    else:
        torch.save((s.cpu(), i.cpu()), ref)
    print(f"RESULT main {sorted(ms[-5:])[2]:.4f} total {sorted(tot[-5:])[2]:.4f} tile_q {st['main_tile_queries']} ranges {st['ranges']} launches {st['main_launches']} "
          f"rank {st['opt_rank']} sample {st['sample_tiles']} cand {st['n_candidates']} fallback {st['n_fallback']} {same}")


_NH = {"CCR_PROGRESSIVE": "0", "CCR_OPTIMISTIC": "0"}
SETS = {
    "ablate": [dict(_NH, CCR_GEMM_DBG=str(v)) for v in (128, 132, 136, 140, 160, 192, 168, 128)],
    "narrow": [{"CCR_WIDE": "0"}, {"CCR_WIDE": "0", "CCR_PROGRESSIVE": "0"}, {"CCR_WIDE": "0"}],
    "ab": [{"CCR_WIDE": "0"}, {}, {"CCR_WIDE": "0"}, {}],
    "order": [{}, {"CCR_ITEM_SWAP": "1"}, {"CCR_QGROUPS": "1", "CCR_WIDE": "0", "CCR_PROGRESSIVE": "0"}, {}],
    "plans": [{"CCR_WIDE": "0"}, {}, {"CCR_OPTIMISTIC": "0"}, {"CCR_PROGRESSIVE": "0", "CCR_OPTIMISTIC": "0"}, {"CCR_WIDE": "0", "CCR_PROGRESSIVE": "0"},
              {"CCR_RANGES": "128"}, {"CCR_RANGES": "256"}, {"CCR_RANGES": "128", "CCR_OPTIMISTIC": "0"}, {"CCR_WIDE": "0"}],
    "nohit": [{"CCR_WIDE": "0", "CCR_GEMM_DBG": "128", "CCR_PROGRESSIVE": "0"}, {"CCR_GEMM_DBG": "128", "CCR_PROGRESSIVE": "0", "CCR_OPTIMISTIC": "0"},
              {"CCR_GEMM_DBG": "128", "CCR_PROGRESSIVE": "0", "CCR_OPTIMISTIC": "0", "CCR_RANGES": "256"},
              {"CCR_GEMM_DBG": "128", "CCR_PROGRESSIVE": "0", "CCR_OPTIMISTIC": "0", "CCR_RANGES": "64"},
              {"CCR_WIDE": "0", "CCR_GEMM_DBG": "128", "CCR_PROGRESSIVE": "0"}],
}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(ROOT, "gpurun_out", "r06_wide_ab.txt")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    ref = os.path.join(ROOT, "gpurun_out", "wide_ref.pt")
    if os.path.exists(ref):
        os.remove(ref)
    which = arg("--set", "plans", str)
    extra = [a for a in sys.argv[1:] if a != out]
    lines = [f"# tools/exp_wide.py {' '.join(extra)}: main pass / whole search by the library's HIP events (ms), median of the last 5 of 8 searches, one box"]
    for env_v in SETS[which]:
        env = dict(os.environ, QD_REF=ref, **env_v)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"] + extra, env=env, capture_output=True, text=True, timeout=900)
        m = re.search(r"RESULT (.*)", r.stdout)
        name = " ".join(f"{k}={v}" for k, v in env_v.items()) or "(defaults)"
        if m and m.group(1) == "diag":
            t = sorted(float(x) for x in re.findall(r"\[ccr diag\] CCR_GEMM_DBG=\d+ main pass ([0-9.]+) ms", r.stderr)[-5:])
            lines.append(f"{name:70s} no-hit main pass {t[len(t) // 2]:.4f} ms")
        else:
            lines.append(f"{name:70s} " + (m.group(1) if m else f"FAILED rc={r.returncode} {r.stderr[-400:]!r}"))
        print(lines[-1], flush=True)
    open(out, "w").write("\n".join(lines) + "\n")
    if os.path.exists(ref):
        os.remove(ref)


if __name__ == "__main__":
    one() if "--one" in sys.argv else main()
