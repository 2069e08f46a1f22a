"""How many candidates per query should a k <= 16 search keep?  32 (the rule up to round 5) against 64 (REVO_KSEL=64, experiment
library) by query count: whole-search time, stage times and the number of queries whose certificate failed (one is enough for
a collecting pass over the whole gallery).   REVO_EXPERIMENTS=1 [REVO_KSEL=64] python scripts/ksel_by_q.py [N]"""
import json, os, sys
os.environ.setdefault("REVO_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd  # noqa: F401
from reverso_amd import engine
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D, k = 1024, 10
G = engine.Gallery(D, N, device=0)
g = torch.Generator(device=dev).manual_seed(42)
for s in range(0, N, 131072):
    G.add(torch.randn(min(131072, N - s), D, generator=g, device=dev))
rows = []
for Q in (64, 256, 384, 512, 1000, 2048, 4096, 10000):
    times, unc = [], []
    for seed in range(3):                       # three query sets: the failures are a property of the data
        q = torch.randn(Q, D, generator=g, device=dev)
        for _ in range(2): G.search(q, k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10 if Q <= 1000 else 3
        e0.record()
        for _ in range(reps): G.search(q, k)
        e1.record(); torch.cuda.synchronize()
        times.append(round(e0.elapsed_time(e1) / reps, 4)); unc.append(G.search_stats()["uncertified"])
    engine.prof_reset(); engine.prof_enable(True)
    for _ in range(3): G.search(q, k)
    torch.cuda.synchronize(); engine.prof_enable(False)
    prof = {c: round(v["ms"] / 3, 4) for c, v in sorted(engine.prof_report().items())}
    rows.append({"Q": Q, "search_ms": times, "uncertified": unc, "stage_ms": prof})
print(json.dumps({"N": N, "ksel": os.environ.get("REVO_KSEL", "32"), "rows": rows}))
