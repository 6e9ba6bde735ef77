"""Shared test helpers: comparison of rank lists (tie-aware); child processes that must not hang silently."""
import os
import signal
import subprocess
import time

import numpy as np
import pytest


def canonicalise(ids, scores):
    """Re-order each row by (score desc, id asc) -- the reference leaves tie order unspecified."""
    out_i = np.empty_like(ids)
    out_s = np.empty_like(scores)
    for q in range(ids.shape[0]):
        o = np.lexsort((ids[q], -scores[q].astype(np.float64)))
        out_i[q], out_s[q] = ids[q][o], scores[q][o]
    return out_i, out_s


def assert_rank_close(ids, scores, ref_ids, ref_scores, tol, truncated=False):
    """ids/scores: ours (canonical order). ref_*: reference order (ties arbitrary, fp32 MKL sums).

    * scores agree rank by rank within tol;
    * ids agree at every rank whose reference score is separated from both neighbours by > 2 tol;
    * as sets, ids agree except for members whose score is within 2 tol of the cut (if truncated).
    """
    assert ids.shape == ref_ids.shape, (ids.shape, ref_ids.shape)
    np.testing.assert_allclose(scores, ref_scores, atol=tol, rtol=0)
    for q in range(ids.shape[0]):
        rs = ref_scores[q].astype(np.float64)
        gap_prev = np.r_[np.inf, rs[:-1] - rs[1:]]
        gap_next = np.r_[rs[:-1] - rs[1:], np.inf if not truncated else 0.0]
        clear = (gap_prev > 2 * tol) & (gap_next > 2 * tol)
        bad = clear & (ids[q] != ref_ids[q])
        assert not bad.any(), f"query {q}: id mismatch at clear ranks {np.nonzero(bad)[0][:10]}"
        diff = set(ids[q].tolist()) ^ set(ref_ids[q].tolist())
        if diff:
            assert truncated, f"query {q}: id sets differ {sorted(diff)[:10]}"
            cut = rs[-1]
            ours = dict(zip(ids[q].tolist(), scores[q].tolist()))
            refd = dict(zip(ref_ids[q].tolist(), ref_scores[q].tolist()))
            for j in diff:
                s = ours.get(j, refd.get(j))
                assert abs(s - cut) <= 2 * tol, f"query {q}: id {j} score {s} far from cut {cut}"


# ---------------------------------------------------------------------------------------------- child processes
def _proc_state(pid):
    """What the kernel says a process and its threads are doing: state + wait channel of every thread (readable without root)."""
    lines = []
    try:
        for tid in sorted(os.listdir(f"/proc/{pid}/task"), key=int):
            base = f"/proc/{pid}/task/{tid}"
            try:
                comm = open(base + "/comm").read().strip()
                state = [ln for ln in open(base + "/status").read().splitlines() if ln.startswith("State:")][0]
                wchan = open(base + "/wchan").read().strip()
                lines.append(f"  tid {tid} {comm}: {state} wchan={wchan}")
            except OSError:
                pass
    except OSError:
        lines.append(f"  pid {pid}: gone")
    return lines


def _children_of(pid):
    try:
        out = subprocess.run(["ps", "-o", "pid=", "--ppid", str(pid)], capture_output=True, text=True).stdout.split()
        kids = [int(x) for x in out]
    except Exception:
        kids = []
    return kids + [g for c in kids for g in _children_of(c)]


def run_child_with_evidence(cmd, env, tmp_path, tag, limit=240):
    """Run a bench.py child in its own process group.  A child that overruns `limit` is a FAILURE with evidence: the Python
    stacks of every rank (SIGUSR1 -> faulthandler, written to files that survive the kill), the kernel-side state and wait
    channel of every thread of every process of the group, and the child's stderr so far.  The whole group is killed then
    (a killed launcher alone would leave its ranks holding the GPU)."""
    dump_dir = tmp_path / f"{tag}_stacks"
    dump_dir.mkdir(exist_ok=True)
    env = dict(env, CCR_BENCH_WATCHDOG_DIR=str(dump_dir))
    err_path, out_path = tmp_path / f"{tag}.stderr", tmp_path / f"{tag}.stdout"
    with open(err_path, "w") as ferr, open(out_path, "w") as fout:
        proc = subprocess.Popen(cmd, stdout=fout, stderr=ferr, env=env, start_new_session=True)
        t0 = time.time()
        try:
            proc.wait(timeout=limit)
        except subprocess.TimeoutExpired:
            pids = [proc.pid] + _children_of(proc.pid)
            evidence = [f"{tag}: child still running after {time.time() - t0:.0f} s: {' '.join(cmd)}"]
            for pid in pids:
                evidence.append(f"pid {pid}: {open(f'/proc/{pid}/cmdline').read().replace(chr(0), ' ')[:200] if os.path.exists(f'/proc/{pid}/cmdline') else 'gone'}")
                evidence += _proc_state(pid)
            for pid in pids:
                if not _children_of(pid):                  # the ranks (leaves): launchers have no handler and would just die
                    try:
                        os.kill(pid, signal.SIGUSR1)       # faulthandler: every thread's Python stack into the dump file
                    except OSError:
                        pass
            time.sleep(3)
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except OSError:
                pass
            try:
                proc.wait(timeout=30)
            except subprocess.TimeoutExpired:
                evidence.append("child did not die within 30 s of SIGKILL (uninterruptible)")
            for f in sorted(dump_dir.iterdir()):
                evidence.append(f"--- {f.name}\n{f.read_text()[-6000:]}")
            evidence.append(f"--- stderr\n{err_path.read_text()[-6000:]}")
            pytest.fail("\n".join(evidence))
    stderr, stdout = err_path.read_text(), out_path.read_text()
    if proc.returncode != 0:   # includes the in-child watchdog (exit code 1 after its stack dump)
        dumps = "".join(f"--- {f.name}\n{f.read_text()[-6000:]}\n" for f in sorted(dump_dir.iterdir()))
        pytest.fail(f"{tag}: exit code {proc.returncode}\n{dumps}--- stderr\n{stderr[-6000:]}")
    return stdout, stderr




# ---------------------------------------------------------------------------------------------- goldens g18 / g19 (shared with tools/make_golden.py)
class GoldenTokenizer:
    """Deterministic whitespace tokenizer with the HF call shapes the reference uses (padding=True | "max_length" | False, truncation,
    max_length, return_tensors="pt"): id = 4 + crc32(word) % (vocab - 4); [CLS] = 1, [SEP] = 2, [PAD] = 0.  (No Python hash(): the golden
    generator and the tests must tokenise alike in every process.)"""
    pad_token_id = 0

    def __init__(self, vocab=64):
        self.vocab = int(vocab)

    def __call__(self, texts, truncation=True, padding=True, max_length=32, return_tensors="pt", **kw):
        import zlib
        import torch
        ids = [[1] + [4 + zlib.crc32(w.encode()) % (self.vocab - 4) for w in t.split()][: max_length - 2] + [2] for t in texts]
        if padding is False:
            return {"input_ids": ids, "attention_mask": [[1] * len(r) for r in ids]}
        L = max_length if padding == "max_length" else max(len(r) for r in ids)
        out = torch.zeros(len(ids), L, dtype=torch.int64)
        mask = torch.zeros(len(ids), L, dtype=torch.int64)
        for r, row in enumerate(ids):
            out[r, :len(row)] = torch.tensor(row)
            mask[r, :len(row)] = 1
        return {"input_ids": out, "attention_mask": mask}


def numpy_seeded_bert(cfg, seed):
    """A transformers BertModel whose every parameter is drawn from numpy's RandomState(seed) in named_parameters() order (the same
    bits on every host and torch build -- torch's own CPU normal sampler depends on the vector width): weights N(0, 0.05) with the
    query / key projections x 6 (softmaxes that are not uniform), LayerNorm weights U(0.6, 1.4), biases N(0, 0.05); every value
    rounded to a bf16-exact fp32.  eval() mode."""
    import torch
    from transformers import BertConfig, BertModel
    model = BertModel(BertConfig(**cfg)).eval()
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            shape = tuple(prm.shape)
            if "LayerNorm.weight" in name:
                v = rs.uniform(0.6, 1.4, shape)
            else:
                v = rs.standard_normal(shape) * 0.05
                if "attention.self.query.weight" in name or "attention.self.key.weight" in name:
                    v = v * 6.0
            prm.copy_(torch.from_numpy(v.astype(np.float32)).to(torch.bfloat16).float())
    return model


G18_CFG = dict(vocab_size=64, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, max_position_embeddings=40)
G19_CFG = dict(vocab_size=64, hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=128, max_position_embeddings=24)


def golden_texts(n, seed, longest=20, words=150):
    rs = np.random.RandomState(seed)
    vocab = [f"w{i}" for i in range(words)]
    return [" ".join(rs.choice(vocab, rs.randint(1, longest + 1))) for _ in range(n)]
