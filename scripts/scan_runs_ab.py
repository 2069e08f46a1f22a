"""Large query batches (8 query tiles and more) over one gallery: search time and the scan's share, for A/B runs of the scan's
work split (runs of pieces, topk256.hip) against another build (REVO_LIBRARY_PATH).   python scripts/scan_runs_ab.py [N] [Q ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import engine
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Qs = [int(a) for a in sys.argv[2:]] or [2048, 4096, 10000]
D, k = 1024, int(os.environ.get("TOPK", "10"))
G = engine.Gallery(D, N, device=0)
g = torch.Generator(device=dev).manual_seed(42)
for s in range(0, N, 125_000):
    G.add(torch.randn(min(125_000, N - s), D, generator=g, device=dev))
rows = []
for Q in Qs:
    q = torch.randn(Q, D, generator=g, device=dev)
    for _ in range(2): G.search(q, k)
    torch.cuda.synchronize()
    reps = 5
    engine.prof_reset(); engine.prof_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = G.search(q, k)
    e1.record(); torch.cuda.synchronize(); engine.prof_enable(False)
    prof = engine.prof_report()
    scan = prof["topk_scan"]["ms"] / reps
    rows.append({"Q": Q, "search_ms": round(e0.elapsed_time(e1) / reps, 4), "scan_ms": round(scan, 4),
                 "scan_TF": round(2.0 * Q * N * D / scan / 1e9, 1), "reduce_ms": round(prof["topk_reduce"]["ms"] / reps, 4),
                 "plan": G.search_plan(Q, k), "checksum": int(out[1].sum())})
print(json.dumps({"N": N, "k": k, "rows": rows}))
