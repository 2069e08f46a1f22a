"""Measure what torch.matmul (hipBLASLt / rocBLAS) reaches on the tower's GEMM shapes.

Only a yardstick for the hand-written kernels: nothing on the product path calls it."""
import json
import sys
import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 36928
shapes = {"qkv": (M, 3072, 1024), "out": (M, 1024, 1024), "fc1": (M, 4096, 1024), "fc2": (M, 1024, 4096),
          "scan": (10000, 1000000 // 8, 1024)}
for name, (m, n, k) in shapes.items():
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, b.t(), out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.matmul(a, b.t(), out=c)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(json.dumps({"gemm": name, "M": m, "N": n, "K": k, "ms": round(ms, 4), "TFLOPs": round(2 * m * n * k / ms / 1e9, 1)}))
