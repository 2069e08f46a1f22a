"""Queued-stores persistent GEMM (gemm256q_kernel) against the drained one (gemm256p_kernel), alternated in one process, on the
body-GEMM shapes of PE-Core-L14-336 at batch 64 as the forward calls them (folded LayerNorm; RoPE for qkv):
    REVO_EXPERIMENTS=1 python scripts/gemm_qstores_ab.py [g14] > gpurun_out/gemm_qstores_ab.json"""
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
G14 = "g14" in sys.argv


def case(name, M, N, K, epi, S=577, hd=64):
    g = torch.Generator(device=dev).manual_seed(M + N + K + epi)
    x = torch.randn(M, K, generator=g, device=dev)
    a = x.bfloat16()
    b = (torch.randn(N, K, generator=g, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, generator=g, device=dev)
    csum = b.float().sum(1)
    xs = x.view(M, K // 256, 256)
    mm = xs.mean(2)
    stats = torch.stack([mm, ((xs - mm[..., None]) ** 2).sum(2)], dim=-1).contiguous()
    cs = torch.randn(S, hd // 2, 2, generator=g, device=dev)
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    st = _lib.current_stream()
    if K // 256 > 6:
        return lambda: _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), None, st)), c
    if epi == 5:
        return lambda: _lib.check(lib.revo_op_gemm_rope(_lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), _lib.ptr(cs),
                                                        S, hd, 2 * N // 3, st)), c
    return lambda: _lib.check(lib.revo_op_gemm_ln_in(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias),
                                                     _lib.ptr(csum), _lib.ptr(stats), K // 256, 1e-5, None, st)), c


def timed(go, iters):
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


if G14:
    shapes = [("qkv_plain", 32768, 4608, 1536, 0), ("qkv_rope", 32768, 4608, 1536, 5), ("fc1_gelu", 32768, 8960, 1536, 1)]
else:
    shapes = [("qkv_plain", 36928, 3072, 1024, 0), ("qkv_rope", 36928, 3072, 1024, 5), ("fc1_gelu", 36864, 4096, 1024, 1),
              ("cube_plain", 8192, 8192, 8192, 0)]
res = {}
for name, M, N, K, epi in shapes:
    go, c = case(name, M, N, K, epi, S=1024 if G14 else 577, hd=96 if G14 else 64)
    r = {"M": M, "N": N, "K": K, "epi": epi, "ms_drained": [], "ms_queued": []}
    outs = {}
    for rnd in range(4):
        for q in (0, 1):
            lib.revo_op_set_qstores((3 if epi == 5 else 1) if q else 0)
            if rnd == 0:
                go(); torch.cuda.synchronize(); outs[q] = c.clone()
            r["ms_queued" if q else "ms_drained"].append(round(timed(go, 8 if name.startswith("cube") else 30), 4))
    lib.revo_op_set_qstores(1)
    r["bit_identical"] = bool(torch.equal(outs[0], outs[1]))
    fl = 2.0 * M * N * K
    r["tflops_drained"] = round(fl / (min(r["ms_drained"]) * 1e-3) / 1e12, 1)
    r["tflops_queued"] = round(fl / (min(r["ms_queued"]) * 1e-3) / 1e12, 1)
    res[name] = r
    print(name, r, file=sys.stderr, flush=True)
print(json.dumps(res))
