"""One-image forwards for a kernel trace: is batch 1 bound by the kernels or by the gaps between them?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import engine
dev = torch.device("cuda", 0)
eng = engine.VitEngine.synthetic("PE-Core-L14-336", seed=0, device=0, max_batch=8)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
u8 = torch.randint(0, 256, (B, 3, 336, 336), dtype=torch.uint8, device=dev)
for _ in range(5): eng.embed(u8)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 50
for _ in range(N): eng.embed(u8)
torch.cuda.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) / N * 1e3:.3f} ms per forward (wall)")
