#!/usr/bin/env python3
"""Same-box A/B of the query-direct forms of the 16x16x32 main pass (CCR_QDIRECT = 0 ring / 1 / 3 / 4 / 5, csrc/ccr_fused.hip
gemm_topk16q_kernel; since the measurement the kernel is compiled into the DIAGNOSTIC library only: tools/build_diag.sh first).  Runs ON THE GPU BOX.  One process per variant; every variant must return the ids and
score bits of variant 0 (the canonical results do not depend on the main pass).

  python3 tools/exp_qdirect.py [outfile] [--rows N --queries Q --k K]"""
import os
import re
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "crowd-coachable-recommendations_amd")


def arg(name, dflt):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else dflt


def one():
    sys.path[:0] = [ROOT, PKG]
    import torch
    from ccrec_amd import _lib
    if os.environ.get("QD_DIAG_LIB", "1") == "1":     # the query-direct kernels are compiled into the DIAGNOSTIC library only (tools/build_diag.sh)
        _lib.LIB_PATH = os.path.join(PKG, "lib_diag", "libccr_hip.so")
    from ccrec_amd import ops
    n, nq, d, k = arg("--rows", 2681468), arg("--queries", 3452), 768, arg("--k", 100)
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
    ref = os.environ["QD_REF"]
    same = "ref"
    if os.path.exists(ref):
        rs, ri = torch.load(ref)
        same = "same" if (torch.equal(ri, i.cpu()) and torch.equal(rs.view(torch.int32), s.cpu().view(torch.int32))) else "DIFFERENT"
    else:
        torch.save((s.cpu(), i.cpu()), ref)
    print(f"RESULT main {sorted(ms[-5:])[2]:.4f} total {sorted(tot[-5:])[2]:.4f} fallback {st['n_fallback']} launches {st['main_launches']} {same}")


def main():
    out = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(ROOT, "gpurun_out", "r06_qdirect_ab.txt")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    ref = os.path.join(ROOT, "gpurun_out", "qd_ref.pt")
    if os.path.exists(ref):
        os.remove(ref)
    extra = [a for a in sys.argv[1:] if a != out]
    lines = [f"# tools/exp_qdirect.py {' '.join(extra)}: main pass / whole search by the library's HIP events (ms), median of the last 5 of 8 searches, one box"]
    for plan, env_plan in (("default plan", {}), ("single launch (CCR_PROGRESSIVE=0)", {"CCR_PROGRESSIVE": "0"})):
        lines.append(f"## {plan}")
        for qd in (0, 1, 3, 4, 5, 0):
            env = dict(os.environ, CCR_QDIRECT=str(qd), QD_REF=ref, **env_plan)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"] + extra, env=env, capture_output=True, text=True, timeout=900)
            m = re.search(r"RESULT (.*)", r.stdout)
            lines.append(f"CCR_QDIRECT={qd}  " + (m.group(1) if m else f"FAILED rc={r.returncode} {r.stderr[-400:]!r}"))
            print(lines[-1], flush=True)
    open(out, "w").write("\n".join(lines) + "\n")
    os.remove(ref)


if __name__ == "__main__":
    one() if "--one" in sys.argv else main()
