"""Does the drain of a tile's stores scale with how many CUs burst at once?  The qkv / fc1-shaped GEMM on all 256 CUs
against HALF the rows on 128 persistent workgroups (REVO_GEMM_NSLOT=16, experiment library: 16 per XCD): the same number of
tile rounds per workgroup, half the bytes per burst.  If a tile boundary is bound by HBM taking the whole chip's burst, the
half-chip run is faster per round; if by a per-CU limit, it is not.
    REVO_EXPERIMENTS=1 [REVO_GEMM_NSLOT=16] python scripts/experiments/r5_gemm_half_chip.py"""
import json, os, sys
os.environ.setdefault("REVO_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, reverso_amd  # noqa: F401
from reverso_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
half = bool(os.environ.get("REVO_GEMM_NSLOT"))
out = {"nslot": os.environ.get("REVO_GEMM_NSLOT", "32")}
for name, N, K, epi in (("qkv_plain", 3072, 1024, 0), ("fc1_gelu", 4096, 1024, 1), ("wide_k4096", 1024, 4096, 0)):
    M = 64 * 577 // (2 if half else 1) // 256 * 256            # whole tiles: 144 (72) tile rows
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    b = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev, generator=g)
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    st = _lib.current_stream()
    def go():
        _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), None, st))
    for _ in range(5): go()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): go()
        e1.record(); torch.cuda.synchronize()
        ts.append(round(e0.elapsed_time(e1) / 20 * 1e3, 1))
    tiles = (M // 256) * (N // 256)
    wgs = 128 if half else 256
    out[name] = {"M": M, "us": ts, "tiles": tiles, "rounds": round(tiles / wgs, 2), "us_per_round": round(min(ts) / (tiles / wgs), 2)}
print(json.dumps(out))
