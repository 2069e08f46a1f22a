"""10 M x 1280 gallery (BASELINE.json configs[4] on one GPU): planted neighbours must come first."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import engine
dev = torch.device("cuda", 0)
N, D, k = 10_000_000, 1280, 10
g = torch.Generator(device=dev).manual_seed(5)
G = engine.Gallery(D, N, device=0)
t0 = time.time()
for s0 in range(0, N, 500_000):
    G.add(torch.randn(500_000, D, generator=g, device=dev))
print(f"gallery up in {time.time() - t0:.1f} s", flush=True)
ids = torch.tensor([0, 1, 4_194_303, 4_194_304, 8_388_607, 8_388_608, 9_999_998, 9_999_999, 5_000_001, 123_456], device=dev)
q = torch.cat([G.read(int(i), 1) for i in ids]) + 0.02 * torch.randn(len(ids), D, generator=g, device=dev)
for Q in (len(ids), 300):
    qq = torch.cat([q, torch.randn(Q - len(ids), D, generator=g, device=dev)]) if Q > len(ids) else q
    s, i, c = G.search(qq, k)
    ok = torch.equal(i[:len(ids), 0], ids) and bool((s[:, :-1] >= s[:, 1:]).all()) and bool((c == k).all())
    print("Q", Q, "planted first:", ok, "top scores", [round(float(x), 4) for x in s[:3, 0]])
    assert ok
print("ok")
