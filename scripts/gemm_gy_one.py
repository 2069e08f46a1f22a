"""One forced XCD arrangement (argv[1] = N-stripes gy: 0 = the launcher's heuristic, 1, 2, 4, 8) on the two wide body
GEMMs of the headline step (qkv 36928 x 3072 x 1024 with the plain bf16 epilogue, fc1 36928 x 4096 x 1024 with GELU):
time per launch by HIP events.  Run it under `rocprofv3 --pmc FETCH_SIZE` for the fabric-side bytes of the same launches
(scripts/gemm_xcd_map.sh).  gy = 1 is "every XCD owns whole M stripes and streams all of W" (VERDICT r3 item 7)."""
import json, os, sys
os.environ.setdefault("REVO_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd  # noqa: F401
from reverso_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
gy = int(sys.argv[1]) if len(sys.argv) > 1 else 0
out = {"gy": gy}
for name, (M, N, K, epi) in {"qkv": (36928, 3072, 1024, 0), "fc1": (36928, 4096, 1024, 1)}.items():
    a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    lib.revo_op_set_gemm_debug(gy << 4); st = _lib.current_stream()
    go = lambda: _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), None, st))
    for _ in range(5): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 30
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize(); lib.revo_op_set_gemm_debug(0)
    ms = e0.elapsed_time(e1) / iters
    out[name] = {"ms": round(ms, 4), "tflops": round(2.0 * M * N * K / ms / 1e9, 1), "algorithmic_MB": round((M * K + N * K + M * N) * 2 / 1e6, 1)}
print(json.dumps(out))
