#!/bin/bash
# Round 5, same step time: two instead of three batches of old values in the plane epilogue of the residual GEMMs (the macro
# REVO_RESID8_TWO_BATCHES existed in gemm.hip for this run only: commit 07985f2's successor; re-create it to repeat).
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --ingest-images 0 --search-queries 0"
for r in 1 2 3; do
  REVO_EXPERIMENTS=1 python bench.py $ARGS 2>/dev/null > gpurun_out/b3_$r.json
  REVO_EXPERIMENTS=1 REVO_LIBRARY_PATH=revers-o_amd/librevo_var.so python bench.py $ARGS 2>/dev/null > gpurun_out/b2_$r.json
done
python - <<'PY'
import json
for r in (1,2,3):
    for t in ("b3","b2"):
        d=json.loads([l for l in open(f"gpurun_out/{t}_{r}.json") if l.startswith("{")][-1])
        k=d["kernel_ms_per_step"]
        print(r,t,round(d["ms_per_step"],3), {c:k.get(c) for c in ("gemm_qkv","gemm_out","gemm_fc1","gemm_fc2")})
PY
