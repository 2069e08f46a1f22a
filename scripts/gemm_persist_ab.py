"""A/B of the persistent 256x256 GEMM against the one-workgroup-per-tile form (debug bit 16)."""
import os, sys
os.environ.setdefault("REVO_EXPERIMENTS", "1")   # timing switches live in librevo_exp.so (make -C revers-o_amd/csrc exp)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
def run(M, N, K, epi, iters=20):
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev, generator=g).bfloat16(); b = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev, generator=g)
    c = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (2, 3) else torch.bfloat16)
    st = _lib.current_stream()
    def go(): _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), None, st))
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9, c
for (M, N, K, epi) in [(36928, 3072, 1024, 0), (36928, 1024, 1024, 2), (36928, 4096, 1024, 1), (36928, 1024, 4096, 2), (8192, 8192, 8192, 0)]:
    out = []
    cs = []
    for flag in (1 << 16, 0, 1 << 16, 0):
        lib.revo_op_set_gemm_debug(flag)
        ms, tf, c = run(M, N, K, epi)
        out.append(f"{'tile-per-wg' if flag else 'persistent '}: {ms:.3f} ms {tf:7.1f} TF")
        cs.append(c.float().clone() if epi != 2 else None)
    same = "" if cs[0] is None else f" equal={torch.equal(cs[0], cs[1])}"
    print(f"M={M} N={N} K={K} epi={epi}  " + "   ".join(out[2:]) + same, flush=True)
