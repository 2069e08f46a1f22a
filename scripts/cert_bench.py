"""What the search's exactness certificate costs (include/revo.h "EXACTNESS").
For a random 1 M x 1024 gallery (BASELINE.json's) and (queries, k) pairs: time of Gallery.search with the certificate
counted but no fallback (the pre-certificate search), with certificate + fallback (the product), with every query forced
through the collecting pass, and -- few queries only -- through the fp32 brute-force pass; how many queries failed the
certificate.     python scripts/cert_bench.py [N]  ->  JSON lines + profiles-ready summary on the last line"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reverso_amd  # noqa: F401
from reverso_amd import engine

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = 1024
G = engine.Gallery(D, N, device=0, experiments=True)    # set_search_mode lives in librevo_exp.so
g = torch.Generator(device=dev).manual_seed(42)
for s in range(0, N, 131072):
    G.add(torch.randn(min(131072, N - s), D, generator=g, device=dev))


def timed(q, k, mode, iters):
    G.set_search_mode(mode)
    for _ in range(2):
        G.search(q, k)
    st = G.search_stats()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        G.search(q, k)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, st


rows = []
for (Q, k) in [(1, 10), (64, 10), (1000, 10), (10000, 10), (64, 50), (1000, 50)]:
    q = torch.randn(Q, D, generator=g, device=dev)
    iters = 20 if Q <= 64 else (5 if Q <= 1000 else 3)
    r = {"N": N, "D": D, "Q": Q, "k": k}
    for mode in ("uncertified", "certified", "collect") + (("bruteforce",) if Q <= 64 else ()):
        ms, st = timed(q, k, mode, iters if mode != "bruteforce" else 2)
        r[mode + "_ms"] = round(ms, 4)
        if mode in ("uncertified", "certified"):
            r[mode + "_failed"] = st["uncertified"]
        if mode == "collect":
            r["collected_rows_per_query"] = round(st["collected_rows"] / max(Q, 1), 1)
    rows.append(r)
    print(json.dumps(r), flush=True)
G.set_search_mode("certified")
print(json.dumps({"certificate_cost": rows}))
