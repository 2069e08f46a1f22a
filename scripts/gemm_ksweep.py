import os, sys
os.environ.setdefault("REVO_EXPERIMENTS", "1")   # timing switches live in librevo_exp.so (make -C revers-o_amd/csrc exp)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
def run(M, N, K, epi=0, tile=256, dbg=0, iters=30):
    a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    c = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (2, 3) else torch.bfloat16)
    _lib.check(lib.revo_op_set_gemm_tile(tile)); lib.revo_op_set_gemm_debug(dbg); st = _lib.current_stream()
    go = lambda: _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, None, None, st))
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    lib.revo_op_set_gemm_debug(0)
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N) in ((4096, 4096), (8192, 4096), (256, 256)):
    for K in (64, 1024):
        print(f"M={M} N={N} K={K:5d}: " + "  ".join(f"epi{e}/dbg{d}: {run(M, N, K, e, 256, d):7.1f} us" for e in (0, 3) for d in (0, 1, 2, 3)), flush=True)
