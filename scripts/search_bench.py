"""Search micro-benchmark (K12/K13): time of Gallery.search and of its kernels, with the
MFMA / HBM roofline figures of SURVEY.md §8(d).   python scripts/search_bench.py [N] [Q ...]"""
import os, sys, json
os.environ.setdefault("REVO_EXPERIMENTS", "1")   # timing switches live in librevo_exp.so (make -C revers-o_amd/csrc exp)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import engine
dev = torch.device("cuda", 0)
from reverso_amd import _lib
_dbg = int(os.environ.get("SCAN_DBG", "0"))     # 1 no selection, 2 counters, 4 no slow path, 8 no global traffic, 16 no refreshes
_lib.load().revo_op_set_gemm_debug(((_dbg & 7) << 13) | ((_dbg >> 3) << 20))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Qs = [int(a) for a in sys.argv[2:]] or [1, 64, 256, 10000]
D, k = 1024, int(os.environ.get("TOPK", "10"))
G = engine.Gallery(D, N, device=0)
g = torch.Generator(device=dev).manual_seed(42)
for s in range(0, N, 131072):
    G.add(torch.randn(min(131072, N - s), D, generator=g, device=dev))
for Q in Qs:
    q = torch.randn(Q, D, generator=g, device=dev)
    for _ in range(2): G.search(q, k)
    torch.cuda.synchronize()
    iters = 10 if Q <= 256 else 3
    engine.prof_reset(); engine.prof_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): G.search(q, k)
    e1.record(); torch.cuda.synchronize()
    engine.prof_enable(False)
    prof = engine.prof_report()
    ms = e0.elapsed_time(e1) / iters
    scan = prof["topk_scan"]["ms"] / prof["topk_scan"]["launches"]
    flops = 2.0 * Q * N * D
    byts = N * D * 2 + Q * D * 2 + Q * k * 12
    if int(os.environ.get("SCAN_DBG", "0")) & 2:
        import ctypes
        st4 = (ctypes.c_int64 * 8)(); _lib.load().revo_debug_scan_stats(st4)
        print("scan stats over", iters + 2, "searches: drains", st4[0], "queued", st4[1], "retry passes", st4[2], "slow fragments", st4[3], "appended", st4[4], "refreshes", st4[5])
    print(json.dumps({"Q": Q, "N": N, "search_ms": round(ms, 4), "scan_ms": round(scan, 4),
                      "scan_TFLOPs": round(flops / scan / 1e9, 1), "scan_frac_mfma": round(flops / scan / 1e9 / 2500, 4),
                      "scan_GBs": round(byts / scan / 1e6, 1), "scan_frac_hbm": round(byts / scan / 1e6 / 8000, 4),
                      "queries_per_s": round(Q / ms * 1e3, 1),
                      "other_ms": {k2: round(v["ms"] / v["launches"], 4) for k2, v in prof.items() if k2 != "topk_scan"}}), flush=True)
