"""Where the folded LayerNorm's consumer side spends its time: the plain bf16 GEMM on normalised rows against the folded
form (bf16(x) + row statistics) on the same shape, back to back (same cache state), per body-GEMM shape.
    python scripts/ln_fold_probe.py > gpurun_out/ln_fold_probe.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reverso_amd  # noqa: F401
from reverso_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)

def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

res = {}
for name, M, N, K, epi in (("qkv", 36928, 3072, 1024, 0), ("fc1", 36928, 4096, 1024, 1)):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(M, K, device=dev, generator=g)
    xb = x.bfloat16()
    h = torch.nn.functional.layer_norm(x, (K,)).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.03).bfloat16()
    bias = torch.randn(N, device=dev, generator=g)
    csum = w.float().sum(1)
    xs = x.view(M, K // 256, 256); m = xs.mean(2); q = ((xs - m[..., None]) ** 2).sum(2)
    stats = torch.stack([m, q], -1).contiguous()
    out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    st = _lib.current_stream()
    plain = lambda: _lib.check(lib.revo_op_gemm(epi, _lib.ptr(h), K, _lib.ptr(w), K, M, N, K, _lib.ptr(out), N, _lib.ptr(bias), None, st))
    fold = lambda: _lib.check(lib.revo_op_gemm_ln_in(epi, _lib.ptr(xb), K, _lib.ptr(w), K, M, N, K, _lib.ptr(out), N, _lib.ptr(bias), _lib.ptr(csum), _lib.ptr(stats), K // 256, 1e-5, None, st))
    r = {"plain_ms": [], "folded_ms": []}
    for _ in range(3):
        r["plain_ms"].append(round(timed(plain), 4)); r["folded_ms"].append(round(timed(fold), 4))
    res[name] = r
    print(name, r, file=sys.stderr, flush=True)
print(json.dumps(res))
