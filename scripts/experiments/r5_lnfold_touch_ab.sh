#!/bin/bash
# Round 5, no effect: re-touch bf16(x) right before the GEMM that streams it (is the consumer's extra time Infinity-Cache residency? no).
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --ingest-images 0 --search-queries 0"
for r in 1 2; do
  REVO_EXPERIMENTS=1 python bench.py $ARGS 2>/dev/null > gpurun_out/touch_off_$r.json
  REVO_EXPERIMENTS=1 REVO_LNFOLD_TOUCH=1 python bench.py $ARGS 2>/dev/null > gpurun_out/touch_on_$r.json
done
python - <<'PY'
import json
for r in (1,2):
    for t in ("off","on"):
        d=json.loads([l for l in open(f"gpurun_out/touch_{t}_{r}.json") if l.startswith("{")][-1])
        k=d["kernel_ms_per_step"]
        print(r,t,round(d["ms_per_step"],3), {c:k.get(c) for c in ("gemm_qkv","gemm_out","gemm_fc1","gemm_fc2","layernorm","touch")})
PY
