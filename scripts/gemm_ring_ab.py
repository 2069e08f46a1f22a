"""A/B of the six-deep-ring 128 x 64 GEMM kernel (gemm128r_kernel) against the two-buffer form on the shapes it is for:
one to four images of PE-Core-L14-336 (577 rows each) and the leftover rows of the residual GEMMs at batch 64 (4160 rows).
    python scripts/gemm_ring_ab.py"""
import os, sys, json
os.environ.setdefault("REVO_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)

def run(M, N, K, epi, flags, iters=50):
    a = torch.randn(M, K, generator=g, device=dev).bfloat16()
    b = (torch.randn(N, K, generator=g, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, generator=g, device=dev)
    c = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (2, 3) else torch.bfloat16)
    _lib.check(lib.revo_op_set_variant(flags))
    st = _lib.current_stream()
    def go():
        _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), None, st))
    go(); torch.cuda.synchronize()
    ref = a.float() @ b.float().T + bias
    if epi == 1: ref = torch.nn.functional.gelu(ref)
    err = float((c.float() - ref).abs().max())
    c.zero_(); 
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): go()
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    _lib.check(lib.revo_op_set_variant(0))
    return e0.elapsed_time(e1) / iters * 1e3, err

rows = []
for (M, N, K, epi, what) in [(577, 3072, 1024, 0, "qkv, 1 image"), (577, 1024, 1024, 2, "out-proj, 1 image"), (577, 4096, 1024, 1, "fc1, 1 image"),
                             (577, 1024, 4096, 2, "fc2, 1 image"), (1154, 3072, 1024, 0, "qkv, 2 images"), (1154, 1024, 1024, 2, "out-proj, 2 images"),
                             (2308, 1024, 1024, 2, "out-proj, 4 images"), (2308, 4096, 1024, 1, "fc1, 4 images"),
                             (4160, 1024, 1024, 2, "out-proj leftover rows at batch 64"), (4160, 1024, 4096, 2, "fc2 leftover rows at batch 64")]:
    ring, e1 = run(M, N, K, epi, 0)
    old, e2 = run(M, N, K, epi, 1 << 19)
    nosplit, e3 = run(M, N, K, epi, (1 << 17)) if epi == 2 and K >= 2048 else (None, None)
    r = {"shape": [M, N, K], "epi": epi, "what": what, "ring_us": round(ring, 2), "two_buffer_us": round(old, 2), "max_err": [round(e1, 4), round(e2, 4)]}
    if nosplit is not None:
        r["ring_without_splitk_us"] = round(nosplit, 2)
    rows.append(r)
    print(json.dumps(r), flush=True)
