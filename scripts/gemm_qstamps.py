"""Per-tile time stamps of the queued-stores GEMM (gemm256q_kernel) and of the drained one (gemm256p via its stamped twin),
body-GEMM shapes of PE-Core-L14-336 at batch 64:
    REVO_EXPERIMENTS=1 python scripts/gemm_qstamps.py > gpurun_out/gemm_qstamps.json
queued (wave 0 of every workgroup): main loop | epilogue set-up (statistics out of LDS, bias loads waited for, next tile's DMA
requested) | arithmetic with each quarter's stores behind it | gap to the next main loop (row-statistics merge);  the same with the
stores dropped by an empty descriptor (what the stores cost);  drained: main loop | epilogue."""
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
ITEMS = 12


def case(M, N, K, epi, S=577, hd=64):
    g = torch.Generator(device=dev).manual_seed(M + N + K + epi)
    x = torch.randn(M, K, generator=g, device=dev)
    a = x.bfloat16()
    b = (torch.randn(N, K, generator=g, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, generator=g, device=dev)
    csum = b.float().sum(1)
    xs = x.view(M, K // 256, 256)
    mm = xs.mean(2)
    stats = torch.stack([mm, ((xs - mm[..., None]) ** 2).sum(2)], dim=-1).contiguous()
    cs = torch.randn(S, hd // 2, 2, generator=g, device=dev)
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    st = _lib.current_stream()
    if epi == 5:
        return lambda: _lib.check(lib.revo_op_gemm_rope(_lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), _lib.ptr(cs),
                                                        S, hd, 2 * N // 3, st))
    return lambda: _lib.check(lib.revo_op_gemm_ln_in(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias),
                                                     _lib.ptr(csum), _lib.ptr(stats), K // 256, 1e-5, None, st))


def stamped(go, queued):
    lib.revo_op_set_qstores(queued)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    buf = torch.zeros((256, ITEMS, 4), dtype=torch.int64, device=dev)
    lib.revo_debug_gemm_stamps(_lib.ptr(buf), ITEMS)
    go()
    torch.cuda.synchronize()
    lib.revo_debug_gemm_stamps(None, 0)
    t = buf.cpu().numpy().astype(np.float64) / 100.0            # us
    ok = t[..., 0] > 0
    r = lambda v: round(float(v), 2)
    if queued:
        ml, epc, sti = (t[..., 1] - t[..., 0])[ok], (t[..., 2] - t[..., 1])[ok], (t[..., 3] - t[..., 2])[ok]
        nxt = np.zeros_like(ok)
        nxt[:, :-1] = ok[:, 1:]
        gap = (np.roll(t[..., 0], -1, axis=1) - t[..., 3])[ok & nxt]
        span = (t[..., 3][ok].max() - t[..., 0][ok].min())
        return {"tiles": int(ok.sum()), "kernel_span_us": r(span), "mainloop_us": {"mean": r(ml.mean()), "p10": r(np.percentile(ml, 10)), "p90": r(np.percentile(ml, 90))},
                "epilogue_setup_us": {"mean": r(epc.mean()), "p10": r(np.percentile(epc, 10)), "p90": r(np.percentile(epc, 90))},
                "arith_and_store_issue_us": {"mean": r(sti.mean()), "p10": r(np.percentile(sti, 10)), "p90": r(np.percentile(sti, 90))},
                "gap_to_next_mainloop_us": {"mean": r(gap.mean()), "p90": r(np.percentile(gap, 90))},
                "per_tile_us": r(ml.mean() + epc.mean() + sti.mean() + gap.mean())}
    full = t[..., 3] >= 192 * 100.0 / 100.0
    full = (buf.cpu().numpy()[..., 3] >= 192) & ok
    ml, ep = (t[..., 1] - t[..., 0])[full], (t[..., 2] - t[..., 1])[full]
    span = (t[..., 2][ok].max() - t[..., 0][ok].min())
    return {"tiles": int(ok.sum()), "kernel_span_us": r(span), "mainloop_us": {"mean": r(ml.mean()), "p10": r(np.percentile(ml, 10)), "p90": r(np.percentile(ml, 90))},
            "epilogue_us": {"mean": r(ep.mean()), "p10": r(np.percentile(ep, 10)), "p90": r(np.percentile(ep, 90))},
            "per_tile_us": r(ml.mean() + ep.mean())}


res = {}
for name, M, N, K, epi in [("qkv_plain", 36928, 3072, 1024, 0), ("fc1_gelu", 36864, 4096, 1024, 1)]:
    go = case(M, N, K, epi)
    res[name] = {"queued": stamped(go, 1), "queued_stores_dropped": stamped(go, 2), "drained": stamped(go, 0)}
    print(name, json.dumps(res[name]), file=sys.stderr, flush=True)
lib.revo_op_set_qstores(1)
print(json.dumps(res))
