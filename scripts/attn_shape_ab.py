"""MFMA-shape experiment of the body attention (round 6, verdict item 3): attn_fwd_kernel (v_mfma_f32_32x32x16_bf16) against
attn16_fwd_kernel (16x16x32), same per-wave tile, alternated in one process on random data: wall, in-kernel clock
(s_memtime / s_memrealtime per workgroup) -- SQ_WAVE_CYCLES comes from scripts/experiments/r6_attn_shape_pmc.sh.
    REVO_EXPERIMENTS=1 python scripts/attn_shape_ab.py > gpurun_out/attn_shape_ab.json"""
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
ONLY = os.environ.get("ATTN_SHAPE_ONLY")          # "0" / "1": one kernel only, few launches (the PMC passes)
res = {}
for name, B, S, H in [("L14 batch 64: S = 577, 16 heads x 64", 64, 577, 16), ("S = 1024, 16 heads x 64, batch 32", 32, 1024, 16),
                      ("B16 batch 64: S = 197, 12 heads x 64", 64, 197, 12)]:
    hd = 64
    W = H * hd
    qkv = torch.randn(B * S, 3 * W, device=dev).bfloat16()
    out = torch.zeros(B * S, W, device=dev, dtype=torch.bfloat16)
    st = _lib.current_stream()
    go = lambda: _lib.check(lib.revo_op_attention(_lib.ptr(qkv), 3 * W, _lib.ptr(out), W, B, S, H, hd, st))
    if ONLY is not None:
        lib.revo_op_set_variant((1 << 20) if ONLY == "1" else 0)
        for _ in range(20):
            go()
        torch.cuda.synchronize()
        continue
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:               # warm: the clock settles under this load
        for _ in range(50):
            go()
        torch.cuda.synchronize()
    r = {"ms": {"32x32x16": [], "16x16x32": []}, "clock_ghz": {}}
    for rnd in range(5):
        for tag, flag in (("32x32x16", 0), ("16x16x32", 1 << 20)):
            lib.revo_op_set_variant(flag)
            for _ in range(5):
                go()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                go()
            e1.record()
            torch.cuda.synchronize()
            r["ms"][tag].append(round(e0.elapsed_time(e1) / 100, 4))
    for tag, flag in (("32x32x16", 0), ("16x16x32", 1 << 20)):
        lib.revo_op_set_variant(flag)
        nwg = 4 * B * H * ((S + 127) // 128)
        buf = torch.zeros((nwg, 2), dtype=torch.int64, device=dev)
        for _ in range(50):
            go()
        lib.revo_debug_attention_clock(_lib.ptr(buf))
        go()
        torch.cuda.synchronize()
        lib.revo_debug_attention_clock(None)
        t = buf.cpu().numpy().astype(np.float64)
        ok = t[:, 1] > 0
        ghz = t[ok, 0] / t[ok, 1] * 0.1
        r["clock_ghz"][tag] = {"median": round(float(np.median(ghz)), 3), "p10": round(float(np.percentile(ghz, 10)), 3),
                               "p90": round(float(np.percentile(ghz, 90)), 3),
                               "workgroup_cycles_median": int(np.median(t[ok, 0])), "workgroup_us_median": round(float(np.median(t[ok, 1])) / 100, 2)}
    lib.revo_op_set_variant(0)
    fl = 4.0 * B * H * S * S * hd
    r["median_ms"] = {k: sorted(v)[len(v) // 2] for k, v in r["ms"].items()}
    r["tflops"] = {k: round(fl / v / 1e9, 1) for k, v in r["median_ms"].items()}
    res[name] = r
    print(name, json.dumps(r), file=sys.stderr, flush=True)
lib.revo_op_set_variant(0)
print(json.dumps(res))
