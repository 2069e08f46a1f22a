"""A/B of the 192-row tile form of the residual GEMMs (variant bit 3 switches it off), alternated in one process.
    python scripts/gemm192_ab.py"""
import os, sys
os.environ.setdefault("REVO_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
def run(M, N, K, epi, flag, iters=30):
    a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    c = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (2, 3) else torch.bfloat16)
    bias = torch.randn(N, device=dev)
    _lib.check(lib.revo_op_set_variant(flag)); st = _lib.current_stream()
    go = lambda: _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), None, st))
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize(); _lib.check(lib.revo_op_set_variant(0))
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K, epi) in ((36928, 1024, 1024, 2), (36928, 1024, 4096, 2), (36864, 1024, 1024, 2), (36864, 1024, 4096, 2)):
    t = [[run(M, N, K, epi, f) for f in (0, 8)] for _ in range(3)]
    print(f"M={M} N={N} K={K} epi={epi}: 192-row " + " ".join(f"{x[0]:7.1f}" for x in t) + "  us | 256-row plan " + " ".join(f"{x[1]:7.1f}" for x in t) + " us", flush=True)
