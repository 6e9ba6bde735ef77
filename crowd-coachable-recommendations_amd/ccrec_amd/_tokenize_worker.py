"""Tokenizer worker process of encode.LengthSortedEncoder (host_processes > 0).  A standalone script -- it imports neither torch
nor this package -- started with `python -u _tokenize_worker.py` and driven over its stdin / stdout:

    parent -> worker   8-byte little-endian length L, then L bytes
                         first message : JSON {"tokenizer_json": <tokenizers.Tokenizer.to_str()>, "max_length": int}
                         later messages: pickle of a list of str (one chunk of texts);  L = 0 ends the worker
    worker -> parent   8-byte count n, n int32 token counts, 8-byte total T, T int32 token ids (all texts back to back)

Why a process: turning the Rust tokenizer's Encoding objects into Python lists (`.ids`) holds the GIL for ~1.6 s per 64 K
passages, beside the thread that launches the encoder's GPU kernels; in a worker it costs the launch thread nothing.
"""
import itertools
import json
import pickle
import struct
import sys


def _read_exact(f, n):
    buf = bytearray()
    while len(buf) < n:
        part = f.read(n - len(buf))
        if not part:
            raise EOFError
        buf += part
    return bytes(buf)


def main():
    import numpy as np
    from tokenizers import Tokenizer
    fin, fout = sys.stdin.buffer, sys.stdout.buffer
    (n,) = struct.unpack("<Q", _read_exact(fin, 8))
    cfg = json.loads(_read_exact(fin, n).decode("utf-8"))
    tk = Tokenizer.from_str(cfg["tokenizer_json"])
    tk.enable_truncation(max_length=int(cfg["max_length"]))
    tk.no_padding()
    batch = tk.encode_batch_fast if hasattr(tk, "encode_batch_fast") else tk.encode_batch
    while True:
        try:
            (n,) = struct.unpack("<Q", _read_exact(fin, 8))
        except EOFError:
            return
        if n == 0:
            return
        texts = pickle.loads(_read_exact(fin, n))
        encs = batch(texts)
        lengths = np.fromiter(map(len, encs), dtype=np.int32, count=len(encs))
        flat = np.fromiter(itertools.chain.from_iterable(e.ids for e in encs), dtype=np.int32, count=int(lengths.sum()))
        fout.write(struct.pack("<Q", lengths.size))
        fout.write(lengths.tobytes())
        fout.write(struct.pack("<Q", flat.size))
        fout.write(flat.tobytes())
        fout.flush()


if __name__ == "__main__":
    main()
