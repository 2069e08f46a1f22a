"""Micro-benchmark + check of the body attention kernel through the C ABI."""
import os, sys
os.environ.setdefault("REVO_EXPERIMENTS", "1")   # timing switches live in librevo_exp.so (make -C revers-o_amd/csrc exp)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reverso_amd
from reverso_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
B, S, H, hd = 64, 577, 16, 64
if len(sys.argv) > 1: B, S, H = map(int, sys.argv[1:4])
W = H * hd
qkv = torch.randn(B * S, 3 * W, device=dev).bfloat16()
out = torch.zeros(B * S, W, device=dev, dtype=torch.bfloat16)
st = _lib.current_stream()
def go(): _lib.check(lib.revo_op_attention(_lib.ptr(qkv), 3 * W, _lib.ptr(out), W, B, S, H, hd, st))
nw = int(os.environ.get("ATTN_NW", "0")); lib.revo_op_set_gemm_debug(nw << 8)
for _ in range(3): go()
torch.cuda.synchronize()
times = []
for _ in range(5):          # median of 5 rounds of 100 launches: single rounds move by +-5 % with the clock
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): go()
    e1.record(); torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1) / 100)
ms = sorted(times)[2]
fl = 4.0 * B * H * S * S * hd
print(f"attention B={B} S={S} H={H}: {ms:.4f} ms  {fl/ms/1e9:.1f} TF")
x = qkv[: 2 * S].float().reshape(2, S, 3, H, hd)
q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v).transpose(1, 2).reshape(2 * S, W)
print("max err", (out[: 2 * S].float() - ref).abs().max().item())
