"""Round 6, verdict item 8 (the one-image forward): what does a device-wide synchronisation INSIDE a kernel cost against a
kernel boundary?  `iters` stages of "every workgroup writes its slice, sync, every workgroup reads a slice written on another
XCD, sync" as one launch with grid barriers (a monotonic counter in device memory, or one flag word per workgroup that all poll; bounded spin) and as a chain of
launches, for slices of 1 KiB ... 256 KiB per workgroup.
    REVO_EXPERIMENTS=1 python scripts/gridbar_probe.py > gpurun_out/gridbar_probe.json"""
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
st = _lib.current_stream()
ITERS = 200
res = {"iters": ITERS, "what": "us per synchronisation: one launch with 2 grid barriers per stage against 2 launches per stage",
       "cases": []}
for blocks, threads in ((256, 256), (256, 512), (128, 512), (512, 256)):
    for kib in (1, 16, 64, 256):
        n16 = kib * 64
        buf = torch.zeros(blocks * n16 * 4, device=dev)
        sink = torch.zeros(blocks * threads, device=dev)
        ce = torch.zeros(640 + 33 * 32, device=dev, dtype=torch.int32)
        stamps = torch.zeros(2 * ITERS, device=dev, dtype=torch.int64)
        out = {"workgroups": blocks, "threads": threads, "slice_kib": kib}
        for mode, name in ((0, "counter_barrier"), (2, "flag_barrier"), (3, "xcd_barrier"), (1, "launch_chain")):
            ms = []
            for rnd in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(lib.revo_probe_gridbar(mode, _lib.ptr(ce), _lib.ptr(buf), n16, blocks, threads, ITERS, _lib.ptr(sink),
                                                  _lib.ptr(stamps) if mode != 1 else None, st))
                e1.record()
                torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1))
                if mode != 1:
                    err = int(ce[1].item())
                    if err:
                        out["error"] = {1: "barrier timed out", 2: "stale data read after the barrier"}.get(err, err)
            # 2 synchronisations per stage
            out[name + "_us_per_sync"] = round(min(ms[1:]) * 1e3 / (2 * ITERS), 3)
            if mode != 1:
                s = stamps.cpu().numpy().reshape(ITERS, 2)
                d = (s[:, 1] - s[:, 0])[5:] / 100.0
                out[name + "_wait_of_workgroup0_us"] = {"median": round(float(np.median(d)), 2), "p90": round(float(np.percentile(d, 90)), 2)}
        res["cases"].append(out)
        print(out, file=sys.stderr)
print(json.dumps(res, indent=1))
