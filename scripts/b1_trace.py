"""20 one-image forwards (for a rocprofv3 kernel trace: how much of the latency is gaps between kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import engine
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = engine.VitEngine.synthetic("PE-Core-L14-336", seed=0, device=0, max_batch=64)
img = torch.randint(0, 256, (B, 3, 336, 336), device=dev, dtype=torch.uint8)
for _ in range(5): eng.embed(img)
torch.cuda.synchronize()
for _ in range(20): eng.embed(img)
torch.cuda.synchronize()
