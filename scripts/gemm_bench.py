"""Micro-benchmark of the bf16 MFMA GEMM kernels through the C ABI (revo_op_gemm).
    python scripts/gemm_bench.py [M N K epi]...   (default: a sweep)"""
import os, sys, json
os.environ.setdefault("REVO_EXPERIMENTS", "1")   # the tile switch lives in librevo_exp.so (make -C revers-o_amd/csrc exp)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reverso_amd
from reverso_amd import _lib

lib = _lib.load()
dev = torch.device("cuda", 0)

def run(M, N, K, epi, tile, iters=20):
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    c = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (2, 3) else torch.bfloat16)
    _lib.check(lib.revo_op_set_gemm_tile(tile))
    st = _lib.current_stream()
    def go():
        _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), None, st))
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9

TILES = tuple(int(x) for x in os.environ.get("GEMM_TILES", "128,256").split(","))
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (36928, 3072, 1024), (36928, 1024, 1024), (36928, 4096, 1024), (36928, 1024, 4096),
          (36864, 4096, 1024), (36864, 1024, 4096)]
if len(sys.argv) > 1:
    v = list(map(int, sys.argv[1:])); shapes = [tuple(v[i:i+3]) for i in range(0, len(v), 3)]
for (M, N, K) in shapes:
    for epi in (0, 1, 2, 3):
        row = []
        for tile in TILES:
            ms, tf = run(M, N, K, epi, tile)
            row.append(f"tile{tile}: {ms:7.3f} ms {tf:7.1f} TF")
        print(f"M={M} N={N} K={K} epi={epi}  " + "   ".join(row), flush=True)
