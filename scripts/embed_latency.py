"""Embed latency / throughput of PE-Core-L14-336 by batch size (the UI path embeds one image per query)."""
import os, sys, time
if os.environ.get("VARIANT"):
    os.environ.setdefault("REVO_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import engine
dev = torch.device("cuda", 0)
eng = engine.VitEngine.synthetic("PE-Core-L14-336", seed=0, device=0, max_batch=64)
if os.environ.get("VARIANT"):
    from reverso_amd import _lib
    _lib.check(_lib.load().revo_op_set_variant(int(os.environ["VARIANT"], 0)))
BS = [int(b) for b in os.environ.get("BATCHES", "1,2,4,8,16,32,64").split(",")]
for B in BS:
    img = torch.randint(0, 256, (B, 3, 336, 336), device=dev, dtype=torch.uint8)
    for _ in range(3): eng.embed(img)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n): eng.embed(img)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B:3d}  {ms:8.3f} ms/forward  {B / ms * 1e3:8.1f} img/s", flush=True)
if len(sys.argv) > 1:
    engine.prof_reset(); engine.prof_enable(1)
    img = torch.randint(0, 256, (int(sys.argv[1]), 3, 336, 336), device=dev, dtype=torch.uint8)
    for _ in range(5): eng.embed(img)
    torch.cuda.synchronize(); engine.prof_enable(0)
    print({k: round(v["ms"] / 5, 4) for k, v in sorted(engine.prof_report().items())})
