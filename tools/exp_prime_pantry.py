import sys, time, os
sys.path[:0]=[".","crowd-coachable-recommendations_amd"]
import numpy as np, torch
os.environ["CCREC_SIM_TYPE"]="cos"
from ccrec_amd.ms_marco_eval import ranking, Retriever, block_csr
from ccrec_amd import ops
n, d = 9862, 768
g = torch.Generator().manual_seed(3)
E = torch.randn(n, d, generator=g) / d ** 0.5
rs = np.random.RandomState(5)
brand = np.minimum((rs.zipf(1.3, size=n) - 1), 1959)
members = {b: np.nonzero(brand == b)[0].tolist() for b in np.unique(brand)}
print("largest brands", sorted((len(v) for v in members.values()), reverse=True)[:5])
Eb = ops.pack_bf16(E.cuda(), normalize=True)
lists = [members[brand[j]] for j in range(n)]
ix = ops.CorpusIndex(Eb)
ptr, idx = block_csr(lists, n)
for _ in range(2):
    torch.cuda.synchronize(); t=time.time()
    s,i = ix.search_blocked(Eb, 1001, ptr, idx)
    torch.cuda.synchronize(); print("search_blocked 9862 x 9862, k=1001:", round((time.time()-t)*1e3,1), "ms")
for _ in range(2):
    torch.cuda.synchronize(); t=time.time()
    s,i = ix.search(Eb, 1001)
    torch.cuda.synchronize(); print("plain search k=1001:", round((time.time()-t)*1e3,1), "ms", ix.last_stats()["path"])
for _ in range(2):
    torch.cuda.synchronize(); t=time.time()
    s,i = ix.search(Eb, 10)
    torch.cuda.synchronize(); print("plain search k=10:", round((time.time()-t)*1e3,1), "ms", ix.last_stats()["path"])
