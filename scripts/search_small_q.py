"""Whole-search time (every kernel of Gallery.search, events on the launch stream) for small query batches over the
BASELINE gallery (1 M x 1024) and per-class stage times: 1, 64, 128, 192, 256 queries, k = 10 and 50.
    python scripts/search_small_q.py [N] [D]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd  # noqa: F401
from reverso_amd import engine
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
G = engine.Gallery(D, N, device=0)
g = torch.Generator(device=dev).manual_seed(42)
for s in range(0, N, 131072):
    G.add(torch.randn(min(131072, N - s), D, generator=g, device=dev))
out = {"N": N, "D": D, "rows": []}
for Q in (1, 64, 128, 192, 256, 1000):
    q = torch.randn(Q, D, generator=g, device=dev)
    for k in (10, 50):
        for _ in range(3):
            G.search(q, k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            G.search(q, k)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        st = G.search_stats()
        engine.prof_reset(); engine.prof_enable(True)
        for _ in range(5):
            G.search(q, k)
        torch.cuda.synchronize(); engine.prof_enable(False)
        prof = {c: round(v["ms"] / 5, 4) for c, v in sorted(engine.prof_report().items())}
        out["rows"].append({"Q": Q, "k": k, "search_ms": round(ms, 4), "uncertified": st["uncertified"], "stage_ms": prof,
                            "hbm_frac_total": round((N * D * 2 + Q * D * 2) / (ms * 1e-3) / 8e12, 3)})
print(json.dumps(out))
