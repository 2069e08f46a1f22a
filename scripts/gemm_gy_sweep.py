"""XCD arrangement (gy = N stripes per 8 XCDs) of the persistent body GEMMs, per layer shape, alternated in one process:
    REVO_EXPERIMENTS=1 python scripts/gemm_gy_sweep.py > gpurun_out/gemm_gy_sweep.json
The launchers' rule (gemm.hip launch_256p / launch_256q) is set from this table."""
import json
import os
import sys

os.environ.setdefault("REVO_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reverso_amd  # noqa: F401
from reverso_amd import _lib

lib = _lib.load()
dev = torch.device("cuda", 0)


def case(M, N, K, epi, S, hd):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(M, K, generator=g, device=dev)
    a = x.bfloat16()
    b = (torch.randn(N, K, generator=g, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, generator=g, device=dev)
    cs = torch.randn(S, hd // 2, 2, generator=g, device=dev)
    st = _lib.current_stream()
    if epi == 2:
        c = torch.randn(M, N, generator=g, device=dev)
        gamma = torch.rand(N, generator=g, device=dev) + 0.5
        return lambda: _lib.check(lib.revo_op_gemm(2, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), _lib.ptr(gamma), st))
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    if epi == 5:
        return lambda: _lib.check(lib.revo_op_gemm_rope(_lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), _lib.ptr(cs),
                                                        S, hd, 2 * N // 3, st))
    csum = b.float().sum(1)
    xs = x.view(M, K // 256, 256)
    mm = xs.mean(2)
    stats = torch.stack([mm, ((xs - mm[..., None]) ** 2).sum(2)], dim=-1).contiguous()
    return lambda: _lib.check(lib.revo_op_gemm_ln_in(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias),
                                                     _lib.ptr(csum), _lib.ptr(stats), K // 256, 1e-5, None, st))


shapes = [("L14 qkv (rope)", 36928, 3072, 1024, 5, 577, 64), ("L14 fc1 (gelu)", 36864, 4096, 1024, 1, 577, 64),
          ("L14 out-proj (resid)", 36928, 1024, 1024, 2, 577, 64), ("L14 fc2 (resid)", 36928, 1024, 4096, 2, 577, 64),
          ("G14 qkv (rope) batch 32", 32768, 4608, 1536, 5, 1024, 96), ("G14 fc1 (gelu) batch 32", 32768, 8960, 1536, 1, 1024, 96),
          ("G14 out-proj (resid)", 32768, 1536, 1536, 2, 1024, 96), ("G14 fc2 (resid)", 32768, 1536, 8960, 2, 1024, 96),
          ("B16 qkv (rope) batch 128", 25216, 2304, 768, 5, 197, 64), ("B16 fc1 (gelu) batch 128", 25088, 3072, 768, 1, 197, 64)]
res = {}
for name, M, N, K, epi, S, hd in shapes:
    go = case(M, N, K, epi, S, hd)
    r = {}
    for rnd in range(4):
        for gy in (0, 1, 2, 4):               # 0 = the launcher's rule
            lib.revo_op_set_variant(gy << 4)
            for _ in range(3):
                go()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(25):
                go()
            e1.record()
            torch.cuda.synchronize()
            r.setdefault("rule" if gy == 0 else f"gy={gy}", []).append(round(e0.elapsed_time(e1) / 25, 4))
    lib.revo_op_set_variant(0)
    res[name] = {"M": M, "N": N, "K": K, "column_tiles": (N + 255) // 256, "ms": r, "median_ms": {k: sorted(v)[len(v) // 2] for k, v in r.items()}}
    print(name, res[name]["median_ms"], file=sys.stderr, flush=True)
print(json.dumps(res))
