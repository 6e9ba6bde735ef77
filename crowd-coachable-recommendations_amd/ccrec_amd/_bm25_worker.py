"""Text-analysis worker process of bm25.BM25.fit (processes > 0).  A standalone script -- it imports neither torch nor this
package -- started with `python -u _bm25_worker.py` and driven over its stdin / stdout:

    parent -> worker   8-byte little-endian length L, then L bytes = pickle of a list of str (one chunk of documents); L = 0 ends it
    worker -> parent   8-byte length, then a pickle of (vocab, lengths, indptr, terms, counts):
                         vocab    sorted list of the chunk's distinct tokens
                         lengths  int32 [docs]   tokens per document (duplicates counted)
                         indptr   int64 [docs+1], terms int32 [nnz] (ids into `vocab`, ascending inside a document), counts int32 [nnz]

The analysis is scikit-learn's default (what the reference's TfidfVectorizer uses, scripts/bm_25.py:11): lower-case, tokens of two
or more word characters (`(?u)\\b\\w\\w+\\b`).  Why a process: it is pure-Python work (37 s per million 135-word passages in one
interpreter), and it can run beside the GPU encode of the same step without taking the GIL from the thread that launches kernels.
"""
import pickle
import re
import struct
import sys

_TOKEN = re.compile(r"(?u)\b\w\w+\b")


def _read_exact(f, n):
    buf = bytearray()
    while len(buf) < n:
        part = f.read(n - len(buf))
        if not part:
            raise EOFError
        buf += part
    return bytes(buf)


def analyse_chunk(texts):
    import numpy as np
    docs = [_TOKEN.findall(t.lower()) for t in texts]
    vocab = sorted({w for d in docs for w in d})
    voc = {w: i for i, w in enumerate(vocab)}
    lengths = np.fromiter((len(d) for d in docs), dtype=np.int32, count=len(docs))
    flat = np.fromiter((voc[w] for d in docs for w in d), dtype=np.int64, count=int(lengths.sum()))
    owner = np.repeat(np.arange(len(docs), dtype=np.int64), lengths)
    key = owner * max(1, len(vocab)) + flat            # (document, term) pairs: sort by document then term, run-length encode
    key.sort()
    uniq, counts = np.unique(key, return_counts=True)
    rows, terms = uniq // max(1, len(vocab)), uniq % max(1, len(vocab))
    indptr = np.zeros(len(docs) + 1, np.int64)
    np.add.at(indptr, rows + 1, 1)
    return vocab, lengths, np.cumsum(indptr), terms.astype(np.int32), counts.astype(np.int32)


def main():
    fin, fout = sys.stdin.buffer, sys.stdout.buffer
    while True:
        try:
            (n,) = struct.unpack("<Q", _read_exact(fin, 8))
        except EOFError:
            return
        if n == 0:
            return
        out = pickle.dumps(analyse_chunk(pickle.loads(_read_exact(fin, n))), protocol=pickle.HIGHEST_PROTOCOL)
        fout.write(struct.pack("<Q", len(out)))
        fout.write(out)
        fout.flush()


if __name__ == "__main__":
    main()
