import os, sys
os.environ.setdefault("REVO_EXPERIMENTS", "1")   # timing switches live in librevo_exp.so (make -C revers-o_amd/csrc exp)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
torch.manual_seed(0)
for (M, N, K) in [(4352, 4096, 128), (4352, 4096, 192), (9000, 2304, 256), (36928, 3072, 1024)]:
    a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
    bias = torch.randn(N, device=dev)
    st = _lib.current_stream()
    outs = []
    for flag in (1 << 16, 0):
        lib.revo_op_set_gemm_debug(flag)
        c = torch.full((M, N), float("nan"), device=dev)
        _lib.check(lib.revo_op_gemm(3, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), None, st))
        torch.cuda.synchronize(); outs.append(c)
    d = (outs[0] - outs[1]).abs()
    bad = (d > 0) | torch.isnan(outs[1])
    print(M, N, K, "mismatch elements", int(bad.sum()), "max", float(torch.nan_to_num(d, nan=1e9).max()))
    if bad.any():
        r, c_ = torch.nonzero(bad, as_tuple=True)
        tiles = torch.unique(torch.stack([r // 256, c_ // 256], 1), dim=0)
        print("  bad tiles (m,n):", tiles[:12].tolist(), "count", len(tiles))
        rr = torch.unique(r % 256); cc = torch.unique(c_ % 256)
        print("  rows in tile:", rr[:20].tolist(), len(rr), " cols in tile:", cc[:20].tolist(), len(cc))
