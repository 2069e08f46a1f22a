"""Two half batches on two streams, each with persistent GEMMs sized for half the chip (REVO_GEMM_NSLOT=16, experiment
library), against one batch of 64 on the whole chip: do the halves' epilogue / attention phases fall into each other's main
loops?     REVO_EXPERIMENTS=1 [REVO_GEMM_NSLOT=16] python scripts/experiments/r5_two_half_batches.py"""
import json, os, sys
os.environ.setdefault("REVO_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, reverso_amd
from reverso_amd import engine
dev = torch.device("cuda", 0)
cfg = reverso_amd.get_config("PE-Core-L14-336")
g = torch.Generator(device=dev).manual_seed(1)
img = torch.randint(0, 256, (64, 3, 336, 336), generator=g, device=dev, dtype=torch.uint8)
out = {"nslot": os.environ.get("REVO_GEMM_NSLOT", "32")}

def timed(fn, reps=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 3)

if not os.environ.get("REVO_GEMM_NSLOT"):
    e64 = engine.VitEngine.synthetic(cfg, seed=0, device=0, max_batch=64)
    out["batch64_ms"] = timed(lambda: e64.embed(img))
    e64.close()
ea = engine.VitEngine.synthetic(cfg, seed=0, device=0, max_batch=32)
eb = engine.VitEngine.synthetic(cfg, seed=0, device=0, max_batch=32)
out["batch32_alone_ms"] = timed(lambda: ea.embed(img[:32]))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): ea.embed(img[:32])
    with torch.cuda.stream(s2): eb.embed(img[32:])
    cur.wait_stream(s1); cur.wait_stream(s2)
out["two_halves_two_streams_ms"] = timed(both)
print(json.dumps(out))
