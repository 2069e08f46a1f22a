"""ln_post + pool logits at batch 64 (36 928 x 1024 fp32 in place): the batch form (four rows per wave) in one launch against
the one-row-per-wave form (forced by calls of < 4096 rows).   python scripts/ln_post_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd  # noqa: F401
from reverso_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
B, S, W, H = 64, 577, 1024, 8
x = torch.randn(B * S, W, device=dev)
w, b = torch.randn(W, device=dev), torch.randn(W, device=dev)
qk, ck = torch.randn(H, W, device=dev) * 0.05, torch.randn(H, device=dev)
out, lg = torch.zeros(B * S, W, device=dev), torch.zeros(B, H, S, device=dev)
st = _lib.current_stream()
def one(r0, n, lgp):
    _lib.check(lib.revo_op_layernorm_logits(_lib.ptr(x[r0:r0 + n]), W, _lib.ptr(w), _lib.ptr(b), 1e-5, n, W, _lib.ptr(out[r0:r0 + n]), W,
                                            _lib.ptr(qk), _lib.ptr(ck), H, S, lgp, st))
def batch():
    one(0, B * S, _lib.ptr(lg))
def rows_form():
    for i in range(0, B, 4):
        one(i * S, 4 * S, _lib.ptr(lg[i:i + 4]))
for name, fn in (("batch form", batch), ("row form x 16 launches", rows_form), ("batch form", batch), ("row form x 16 launches", rows_form)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(name, round(e0.elapsed_time(e1) / 50 * 1e3, 1), "us")
