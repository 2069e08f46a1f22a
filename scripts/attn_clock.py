"""The clock the chip holds under the body attention kernel (MI355X_MICROARCH.md, DVFS give-back item 6): every workgroup
stamps s_memtime (shader clock) and s_memrealtime (100 MHz) at its first and last instruction; after >= 2 s of back-to-back
launches on random data one stamped launch, median over workgroups.
    REVO_EXPERIMENTS=1 python scripts/attn_clock.py > gpurun_out/attn_clock.json"""
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
res = {}
for name, B, S, H, hd in [("L14 batch 64 (S = 577, 16 heads x 64)", 64, 577, 16, 64), ("S = 1024, 16 heads x 64", 32, 1024, 16, 64),
                          ("G14 batch 32 (S = 1024, 16 heads x 96)", 32, 1024, 16, 96)]:
    W = H * hd
    qkv = torch.randn(B * S, 3 * W, device=dev).bfloat16()
    out = torch.zeros(B * S, W, device=dev, dtype=torch.bfloat16)
    st = _lib.current_stream()
    go = lambda: _lib.check(lib.revo_op_attention(_lib.ptr(qkv), 3 * W, _lib.ptr(out), W, B, S, H, hd, st))
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 2.0:
        for _ in range(50):
            go()
        torch.cuda.synchronize()
        n += 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    nwg = 4 * B * H * ((S + 127) // 128)                  # an upper bound of the grid
    buf = torch.zeros((nwg, 2), dtype=torch.int64, device=dev)
    for _ in range(20):
        go()
    lib.revo_debug_attention_clock(_lib.ptr(buf))
    go()
    torch.cuda.synchronize()
    lib.revo_debug_attention_clock(None)
    t = buf.cpu().numpy().astype(np.float64)
    ok = t[:, 1] > 0
    ghz = t[ok, 0] / t[ok, 1] * 0.1
    life_us = t[ok, 1] / 100.0
    fl = 4.0 * B * H * S * S * hd
    res[name] = {"ms_per_launch": round(ms, 4), "tflops": round(fl / ms / 1e9, 1), "workgroups": int(ok.sum()),
                 "clock_ghz": {"median": round(float(np.median(ghz)), 3), "p10": round(float(np.percentile(ghz, 10)), 3),
                               "p90": round(float(np.percentile(ghz, 90)), 3)},
                 "workgroup_lifetime_us": {"median": round(float(np.median(life_us)), 2), "p90": round(float(np.percentile(life_us, 90)), 2)}}
    print(name, res[name], file=sys.stderr, flush=True)
print(json.dumps(res))
