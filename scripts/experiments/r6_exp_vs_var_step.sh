#!/bin/bash
# step A/B of two EXPERIMENT builds (librevo_var.so against librevo_exp.so), alternated: a variant build must be compared with the
# experiment build of the tree, not with the product library (the experiment build's launches carry extra arguments and switches)
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-calibration --ingest-images 0 --search-queries 0"
for r in 1 2; do
  REVO_EXPERIMENTS=1 REVO_LIBRARY_PATH=revers-o_amd/librevo_var.so python bench.py $ARGS 2>/dev/null > gpurun_out/step_var_$r.json
  REVO_EXPERIMENTS=1 REVO_LIBRARY_PATH=revers-o_amd/librevo_exp.so python bench.py $ARGS 2>/dev/null > gpurun_out/step_exp_$r.json
  python bench.py $ARGS 2>/dev/null > gpurun_out/step_prod_$r.json
done
python - <<'PY'
import json
for r in (1,2):
    for t in ("var","exp","prod"):
        d=json.loads([l for l in open(f"gpurun_out/step_{t}_{r}.json") if l.startswith("{")][-1])
        k=d["kernel_ms_per_step"]
        print(r,t,round(d["ms_per_step"],3), round(sum(k.values()),3), {c:k[c] for c in ("gemm_qkv","gemm_out","gemm_fc1","gemm_fc2","attention")})
PY
