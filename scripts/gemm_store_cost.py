"""What do the epilogue's stores cost the 256 x 256 GEMM?  Non-persistent kernel (one workgroup per tile), bf16
epilogue, experiment build: normal / no stores / no main loop, interleaved in one process."""
import os, sys
os.environ.setdefault("REVO_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reverso_amd
from reverso_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
st = _lib.current_stream()
def bench(M, N, K, flags, iters=50):
    a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    _lib.check(lib.revo_op_set_gemm_debug(flags))
    def go(): _lib.check(lib.revo_op_gemm(0, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, None, None, st))
    for _ in range(5): go()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): go()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    _lib.check(lib.revo_op_set_gemm_debug(0))
    return sorted(ts)[2]
for (M, N, K) in [(36864, 4096, 1024), (36864, 3072, 1024), (32768, 1024, 4096), (32768, 1024, 1024)]:
    fl = 2.0 * M * N * K
    r = {}
    variants = [("persistent", 0), ("per-tile", 1 << 16), ("per-tile no stores", (1 << 16) | 1), ("per-tile no main loop", (1 << 16) | 2)]
    for groups in ():
        for us in (6, 12, 18, 24):
            variants.append((f"persistent {groups} phase groups, last starts {us * (groups - 1) // groups} us late", ((us // 2) << 28) | ((groups - 2) << 2)))
    for name, flags in variants:
        ms = bench(M, N, K, flags)
        r[name] = ms
        print(f"M={M} N={N} K={K} {name:56s} {ms*1000:8.1f} us  {fl/ms/1e9:7.1f} TF", flush=True)
