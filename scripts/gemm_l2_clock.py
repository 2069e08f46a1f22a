"""Round 6, verdict item 6: do L2-served operands buy clock?  fc1 of PE-L14 batch 64 (36 864 x 4096 x 1024, GELU, folded
LayerNorm) under two XCD walks of the persistent kernel:
  gy = 4 (shipped): an XCD owns an M stripe x a stripe of 4 column tiles -- its 2 MB of weights stay in its L2, every A row
                    block is fetched by 4 of its column tiles
  gy = 1          : an XCD owns an M stripe and walks all 16 column tiles of a row block back to back -- A is re-read from
                    that XCD's L2 16 times, the 8 MB of weights stream through every XCD
per walk: alternated wall time, the main loop's in-kernel clock (s_memtime / s_memrealtime per tile, wave 0 of every
workgroup); FETCH_SIZE comes from a separate rocprofv3 --pmc pass of this script with L2CLOCK_ONLY=<gy>.
    REVO_EXPERIMENTS=1 python scripts/gemm_l2_clock.py > gpurun_out/gemm_l2_clock.json"""
import json
import os
import sys
import time

os.environ.setdefault("REVO_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import reverso_amd  # noqa: F401
from reverso_amd import _lib

lib = _lib.load()
dev = torch.device("cuda", 0)
M, N, K, epi = 36864, 4096, 1024, 1
g = torch.Generator(device=dev).manual_seed(5)
x = torch.randn(M, K, generator=g, device=dev)
a = x.bfloat16()
b = (torch.randn(N, K, generator=g, device=dev) * 0.05).bfloat16()
bias = torch.randn(N, generator=g, device=dev)
csum = b.float().sum(1)
xs = x.view(M, K // 256, 256)
mm = xs.mean(2)
stats = torch.stack([mm, ((xs - mm[..., None]) ** 2).sum(2)], dim=-1).contiguous()
c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
st = _lib.current_stream()
go = lambda: _lib.check(lib.revo_op_gemm_ln_in(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias),
                                               _lib.ptr(csum), _lib.ptr(stats), K // 256, 1e-5, None, st))
only = os.environ.get("L2CLOCK_ONLY")
if only:
    lib.revo_op_set_variant(int(only) << 4)
    for _ in range(10):
        go()
    torch.cuda.synchronize()
    sys.exit(0)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.5:
    for _ in range(30):
        go()
    torch.cuda.synchronize()
res = {"shape": [M, N, K], "epilogue": "GELU + folded LayerNorm (gemm256q_kernel)", "walks": {}}
ms = {1: [], 2: [], 4: []}
for rnd in range(5):
    for gy in (4, 1, 2):
        lib.revo_op_set_variant(gy << 4)
        for _ in range(3):
            go()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            go()
        e1.record()
        torch.cuda.synchronize()
        ms[gy].append(round(e0.elapsed_time(e1) / 30, 4))
ITEMS = 12
for gy in (4, 1, 2):
    lib.revo_op_set_variant(gy << 4)
    lib.revo_op_set_qstores(1 | 8)                         # stamp item 2 = shader-clock cycles of the main loop
    for _ in range(20):
        go()
    buf = torch.zeros((256, ITEMS, 4), dtype=torch.int64, device=dev)
    lib.revo_debug_gemm_stamps(_lib.ptr(buf), ITEMS)
    go()
    torch.cuda.synchronize()
    lib.revo_debug_gemm_stamps(None, 0)
    lib.revo_op_set_qstores(1)
    t = buf.cpu().numpy().astype(np.float64)
    ok = t[..., 0] > 0
    ml_us = (t[..., 1] - t[..., 0])[ok] / 100.0
    ghz = t[..., 2][ok] / (ml_us * 1e3)
    res["walks"][f"gy={gy}"] = {"ms_per_launch": ms[gy], "median_ms": sorted(ms[gy])[2],
                                "tflops": round(2.0 * M * N * K / sorted(ms[gy])[2] / 1e9, 1),
                                "mainloop_us_per_tile": round(float(ml_us.mean()), 2),
                                "mainloop_clock_ghz": {"median": round(float(np.median(ghz)), 3), "p10": round(float(np.percentile(ghz, 10)), 3),
                                                       "p90": round(float(np.percentile(ghz, 90)), 3)},
                                "mainloop_cycles_per_tile": int(np.median(t[..., 2][ok]))}
lib.revo_op_set_variant(0)
print(json.dumps(res))
