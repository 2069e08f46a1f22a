import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, reverso_amd
from reverso_amd import engine
from oracle import search as osearch
N, D, Q, k = 20000, 1024, 300, 10
rng = np.random.default_rng(N + D + Q)
gal = rng.standard_normal((N, D), dtype=np.float32); qr = rng.standard_normal((Q, D), dtype=np.float32)
qr[0] = gal[N // 2] * 3.0
dev = torch.device("cuda", 0)
G = engine.Gallery(D, N, device=0); G.add(torch.from_numpy(gal).to(dev))
s, i, c = (t.cpu().numpy() for t in G.search(torch.from_numpy(qr).to(dev), k))
rs, ri, rc = osearch.search(gal, qr, k)
bad = np.nonzero((i != ri).any(1))[0]
print("bad rows", len(bad), bad[:20])
for r in bad[:5]:
    print("row", r); print(" got", i[r], s[r]); print(" ref", ri[r], rs[r])
    miss = set(ri[r]) - set(i[r]); print(" missing", miss, [(m // 256, m % 256) for m in miss])
