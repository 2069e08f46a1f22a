"""PE-Core-G14-448 at full depth: sanity (finite, unit norm, batch invariance) and throughput."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import engine
dev = torch.device("cuda", 0)
cfg = reverso_amd.get_config("PE-Core-G14-448")
print(cfg)
t0 = time.time()
eng = engine.VitEngine.synthetic(cfg, seed=0, device=0, max_batch=32)
print(f"weights up in {time.time() - t0:.1f} s", flush=True)
g = torch.Generator(device=dev).manual_seed(1)
img = torch.randint(0, 256, (32, 3, cfg.image_size, cfg.image_size), device=dev, dtype=torch.uint8, generator=g)
e = eng.embed(img)
print("finite", bool(torch.isfinite(e).all()), "norm err", float((e.norm(dim=-1) - 1).abs().max()))
one = eng.embed(img[3:4]); three = eng.embed(img[2:5])
print("cos alone vs batch", float((one[0] * e[3]).sum()), "cos 3-batch vs batch", float((three[1] * e[3]).sum()))
print("pairwise cos between different images (should be < 1):", float((e[0] * e[1]).sum()))
for B in (1, 8, 32):
    x = img[:B]
    for _ in range(2): eng.embed(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): eng.embed(x)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"B={B}: {ms:.2f} ms/forward, {B / ms * 1e3:.1f} img/s, {cfg.flops_per_image() * B / ms / 1e9:.0f} TFLOP/s", flush=True)
engine.prof_reset(); engine.prof_enable(1)
for _ in range(3): eng.embed(img)
torch.cuda.synchronize(); engine.prof_enable(0)
print({k: round(v["ms"] / 3, 3) for k, v in sorted(engine.prof_report().items())})
