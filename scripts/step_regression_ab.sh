#!/bin/bash
# A/B of the whole headline step: a previous build's library (REVO_LIBRARY_PATH) against the tree's, alternated on one box.
# The other library: build an older commit in a scratch worktree and keep the .so under _bisect/ (git-ignored, travels with gpurun):
#   git worktree add /tmp/wt <commit> && make -C /tmp/wt/revers-o_amd/csrc -j8 all && mkdir -p _bisect/prev && cp /tmp/wt/revers-o_amd/librevo.so _bisect/prev/ && git worktree remove --force /tmp/wt
OLD=${1:-_bisect/prev/librevo.so}
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-calibration --ingest-images 0 --search-queries 0"
for r in 1 2; do
  REVO_LIBRARY_PATH=$OLD python bench.py $ARGS 2>/dev/null > gpurun_out/step_old_$r.json
  python bench.py $ARGS 2>/dev/null > gpurun_out/step_new_$r.json
done
python - <<'PY'
import json
for r in (1,2):
    for t in ("old","new"):
        d=json.loads([l for l in open(f"gpurun_out/step_{t}_{r}.json") if l.startswith("{")][-1])
        k=d["kernel_ms_per_step"]
        print(r,t,round(d["ms_per_step"],3), {c:k[c] for c in ("attention","gemm_qkv","gemm_out","gemm_fc1","gemm_fc2","layernorm","topk_scan","topk_finish","topk_prepass")})
PY
