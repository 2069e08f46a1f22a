#!/bin/bash
# A/B of the scan kernel across this round's changes (admission margin, drop flags, histogram guard): the library of the
# commit before them (REVO_LIBRARY_PATH) against the tree's, alternated on one box.
# The other library: build an older commit in a scratch worktree and keep the .so under _bisect/ (git-ignored, travels with gpurun):
#   git worktree add /tmp/wt <commit> && make -C /tmp/wt/revers-o_amd/csrc -j8 all && mkdir -p _bisect/prev && cp /tmp/wt/revers-o_amd/librevo.so _bisect/prev/ && git worktree remove --force /tmp/wt
OLD=${1:-_bisect/prev/librevo.so}
for r in 1 2; do
  REVO_LIBRARY_PATH=$OLD python scripts/search_small_q.py > gpurun_out/reg_old_$r.json 2>/dev/null
  python scripts/search_small_q.py > gpurun_out/reg_new_$r.json 2>/dev/null
done
REVO_LIBRARY_PATH=$OLD python scripts/sharded_stage_bench.py > gpurun_out/reg_stage_old.json 2>/dev/null
python scripts/sharded_stage_bench.py > gpurun_out/reg_stage_new.json 2>/dev/null
python - <<'PY'
import json
for r in (1,2):
    o=json.loads(open(f"gpurun_out/reg_old_{r}.json").read().strip().splitlines()[-1]); n=json.loads(open(f"gpurun_out/reg_new_{r}.json").read().strip().splitlines()[-1])
    for a,b in zip(o["rows"],n["rows"]):
        if a["k"]==10: print(r, a["Q"], "old", a["search_ms"], a["stage_ms"].get("topk_scan"), "new", b["search_ms"], b["stage_ms"].get("topk_scan"))
for t in ("old","new"):
    d=json.loads(open(f"gpurun_out/reg_stage_{t}.json").read().strip().splitlines()[-1])
    print(t, d["one_gpu_ms"], d["one_gpu_stage_ms"].get("topk_scan"), d["per_rank_phase1_ms"], d["phase1_stage_ms"].get("topk_scan"))
PY
