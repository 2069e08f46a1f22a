"""What would an ESTIMATED admission level (mean + z sigma of the pre-pass scores, the shards' trick) buy an unsharded
small-query search?  Phase 1 of the two-phase search on the whole 1 M x 1024 gallery with and without the estimate.
    python scripts/experiments/r5_est_unsharded_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, reverso_amd  # noqa: F401
from reverso_amd import engine
dev = torch.device("cuda", 0)
N, D, k = 1_000_000, 1024, 10
G = engine.Gallery(D, N, device=0)
g = torch.Generator(device=dev).manual_seed(42)
for s in range(0, N, 131072):
    G.add(torch.randn(min(131072, N - s), D, generator=g, device=dev))
ksel = engine.search_ksel(k)
out = []
for Q in (1, 64, 128, 256, 1000):
    q = torch.randn(Q, D, generator=g, device=dev)
    row = {"Q": Q}
    for tag, tot in (("plain", 0), ("estimate", N + 1024)):
        G.set_total_rows(tot)
        for _ in range(3): G.search_candidates(q, k, 8)
        torch.cuda.synchronize()
        engine.prof_reset(); engine.prof_enable(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): G.search_candidates(q, k, 8)
        e1.record(); torch.cuda.synchronize(); engine.prof_enable(False)
        prof = engine.prof_report()
        row[tag] = {"phase1_ms": round(e0.elapsed_time(e1) / 10, 4), "scan_ms": round(prof["topk_scan"]["ms"] / 10, 4)}
    out.append(row)
G.set_total_rows(0)
print(json.dumps(out))
