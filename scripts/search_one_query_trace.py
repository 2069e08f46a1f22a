"""One-query searches over 1 M x 1024 for a kernel trace (rocprofv3 --kernel-trace): which kernels a search launches, how
long each runs and how long the stream idles between them.   python3 scripts/search_one_query_trace.py [Q]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd  # noqa: F401
from reverso_amd import engine
dev = torch.device("cuda", 0)
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N, D = 1_000_000, 1024
G = engine.Gallery(D, N, device=0)
g = torch.Generator(device=dev).manual_seed(42)
for s in range(0, N, 131072):
    G.add(torch.randn(min(131072, N - s), D, generator=g, device=dev))
q = torch.randn(Q, D, generator=g, device=dev)
for _ in range(3): G.search(q, 10)
torch.cuda.synchronize()
for _ in range(20): G.search(q, 10)
torch.cuda.synchronize()
