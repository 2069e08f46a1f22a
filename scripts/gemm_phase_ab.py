"""Phase groups of the persistent 256 x 256 GEMM (gemm.hip gemm256pp_kernel) against the in-step form, per body-GEMM shape:
bit-identity of the outputs, alternated timings, and -- with --stamps -- the per-workgroup time stamps of the diagnostic
build: how long main loops and epilogues take and how many workgroups are inside an epilogue at the same moment.

    REVO_EXPERIMENTS=1 python scripts/gemm_phase_ab.py [--stamps] [--variant g14] > gpurun_out/gemm_phase_ab.json
"""
import json
import os
import sys

os.environ.setdefault("REVO_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import reverso_amd  # noqa: F401
from reverso_amd import _lib

lib = _lib.load()
dev = torch.device("cuda", 0)
STAMPS = "--stamps" in sys.argv
G14 = "g14" in sys.argv


def make(M, N, K, epi):
    g = torch.Generator(device=dev).manual_seed(M + N + K + epi)
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    b = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev, generator=g)
    gamma = torch.rand(N, device=dev, generator=g) + 0.5
    c0 = torch.randn(M, N, device=dev, generator=g) if epi == 2 else None
    return a, b, bias, gamma, c0


def run(op, M, N, K, epi, groups, iters, keep=False):
    a, b, bias, gamma, c0 = op
    lib.revo_op_set_phase_groups(groups)
    st = _lib.current_stream()
    c = c0.clone() if epi == 2 else torch.zeros(M, N, device=dev, dtype=torch.bfloat16)

    def go():
        _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias),
                                    _lib.ptr(gamma) if epi == 2 else None, st))
    out = None
    if keep:
        go()
        torch.cuda.synchronize()
        out = c.clone()
    for _ in range(2):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


def stamps(op, M, N, K, epi, groups, items=12):
    """One launch with the diagnostic stamps on: returns per-piece main-loop / epilogue durations (us) and the
    concurrency of the epilogues."""
    a, b, bias, gamma, c0 = op
    lib.revo_op_set_phase_groups(groups)
    buf = torch.zeros((256, items, 4), dtype=torch.int64, device=dev)
    c = c0.clone() if epi == 2 else torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    st = _lib.current_stream()
    for _ in range(3):          # warm
        _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias),
                                    _lib.ptr(gamma) if epi == 2 else None, st))
    lib.revo_debug_gemm_stamps(_lib.ptr(buf), items)
    _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias),
                                _lib.ptr(gamma) if epi == 2 else None, st))
    torch.cuda.synchronize()
    lib.revo_debug_gemm_stamps(None, 0)
    t = buf.cpu().numpy().astype(np.float64)
    ok = t[..., 0] > 0
    t0 = t[..., 0][ok].min()
    ml = (t[..., 1] - t[..., 0])[ok] / 100.0            # 100 MHz -> us
    ep = (t[..., 2] - t[..., 1])[ok] / 100.0
    rows = t[..., 3][ok]
    full = rows >= 192
    # how many workgroups are inside an epilogue at a time: sweep over the stamps
    ev = sorted([(x, 1) for x in t[..., 1][ok]] + [(x, -1) for x in t[..., 2][ok]])
    cur, peak, area, last = 0, 0, 0.0, ev[0][0]
    for x, d in ev:
        area += cur * (x - last)
        last = x
        cur += d
        peak = max(peak, cur)
    span = (t[..., 2][ok].max() - t0) / 100.0
    return {"pieces": int(ok.sum()), "kernel_span_us": round(span, 2),
            "mainloop_us_full_tiles": {"mean": round(float(ml[full].mean()), 2), "p10": round(float(np.percentile(ml[full], 10)), 2),
                                       "p90": round(float(np.percentile(ml[full], 90)), 2)},
            "epilogue_us_full_tiles": {"mean": round(float(ep[full].mean()), 2), "p10": round(float(np.percentile(ep[full], 10)), 2),
                                       "p50": round(float(np.percentile(ep[full], 50)), 2), "p90": round(float(np.percentile(ep[full], 90)), 2)},
            "epilogue_hist_us": np.histogram(ep[full], bins=[0, 1, 2, 3, 4, 5, 6, 8, 10, 15, 25, 1e9])[0].tolist(),
            "epilogue_hist_bins_us": [0, 1, 2, 3, 4, 5, 6, 8, 10, 15, 25, "inf"],
            "workgroups_in_epilogue": {"peak": int(peak), "mean_while_any": round(area / max(sum(1 for _ in ev), 1), 2),
                                       "time_avg": round(area / (span * 100.0), 2)},
            "short_piece_mainloop_us": {str(int(r)): round(float(ml[rows == r].mean()), 2) for r in sorted(set(rows[~full].tolist()))}}


if G14:
    shapes = [("qkv", 32768, 4608, 1536, 0), ("out", 32768, 1536, 1536, 2), ("fc1", 32768, 8960, 1536, 1), ("fc2", 32768, 1536, 8960, 2)]
else:
    shapes = [("qkv", 36928, 3072, 1024, 0), ("out", 36928, 1024, 1024, 2), ("fc1", 36928, 4096, 1024, 1), ("fc2", 36928, 1024, 4096, 2),
              ("cube", 8192, 8192, 8192, 0)]
res = {}
for name, M, N, K, epi in shapes:
    op = make(M, N, K, epi)
    iters = 20 if name != "cube" else 8
    r = {"M": M, "N": N, "K": K, "epi": epi, "ms": {}, "bit_identical": {}}
    ref = None
    for rep in range(2):
        for g in (1, 2, 3, 4):
            ms, out = run(op, M, N, K, epi, g, iters, keep=(rep == 0))
            r["ms"].setdefault(str(g), []).append(round(ms, 4))
            if rep == 0:
                if g == 1:
                    ref = out
                else:
                    r["bit_identical"][str(g)] = bool(torch.equal(out, ref))
    r["tflops"] = {g: round(2.0 * M * N * K / (min(v) * 1e-3) / 1e12, 1) for g, v in r["ms"].items()}
    if STAMPS:
        r["stamps"] = {str(g): stamps(op, M, N, K, epi, g) for g in (1, 2, 4)}
    res[name] = r
    print(name, r["ms"], r["bit_identical"], file=sys.stderr, flush=True)
lib.revo_op_set_phase_groups(0)
print(json.dumps(res))
