import os, sys
os.environ.setdefault("REVO_EXPERIMENTS", "1")   # timing switches live in librevo_exp.so (make -C revers-o_amd/csrc exp)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
def run(M, N, K, epi, gy, iters=20):
    a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    c = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (2, 3) else torch.bfloat16)
    lib.revo_op_set_gemm_tile(256); lib.revo_op_set_gemm_debug(gy << 4); st = _lib.current_stream()
    go = lambda: _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, None, None, st))
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize(); lib.revo_op_set_gemm_debug(0)
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * M * N * K / ms / 1e9
for (M, N, K, epi) in ((36928, 3072, 1024, 0), (36928, 1024, 1024, 2), (36928, 4096, 1024, 1), (36928, 1024, 4096, 2), (8192, 8192, 8192, 0)):
    print(f"M={M} N={N} K={K} epi={epi}: " + "  ".join(f"gy{gy}: {run(M, N, K, epi, gy):7.1f}" for gy in (0, 1, 2, 4, 8)), flush=True)
