#!/bin/bash
# The scan's work split for 8 query tiles and more (runs of pieces: every workgroup an equal run of the flat (query tile, gallery
# tile) list) against the previous form (whole query tiles pinned to XCDs, one piece per workgroup), alternated on one box:
# 1 M x 1024 with 2 048 / 4 096 / 10 000 queries, one shard of eight (125 k rows) in the two-phase search, and two
# workgroups per CU (REVO_SCAN_FLAT_ROUNDS=2, experiment library).
# The other library:  git worktree add /tmp/wt <commit before> && make -C /tmp/wt/revers-o_amd/csrc -j6 all && mkdir -p _bisect/head && cp /tmp/wt/revers-o_amd/librevo.so _bisect/head/ && git worktree remove --force /tmp/wt
OLD=${1:-_bisect/head/librevo.so}
for r in 1 2; do
  REVO_LIBRARY_PATH=$OLD python scripts/scan_runs_ab.py > gpurun_out/runs_old_$r.json 2>gpurun_out/runs_old_$r.err
  python scripts/scan_runs_ab.py > gpurun_out/runs_new_$r.json 2>gpurun_out/runs_new_$r.err
done
REVO_EXPERIMENTS=1 REVO_SCAN_FLAT_ROUNDS=2 python scripts/scan_runs_ab.py > gpurun_out/runs_new_w512.json 2>/dev/null
REVO_LIBRARY_PATH=$OLD python scripts/sharded_stage_bench.py > gpurun_out/runs_stage_old.json 2>/dev/null
python scripts/sharded_stage_bench.py > gpurun_out/runs_stage_new.json 2>/dev/null
REVO_EXPERIMENTS=1 REVO_SCAN_FLAT_ROUNDS=2 python scripts/sharded_stage_bench.py > gpurun_out/runs_stage_new_w512.json 2>/dev/null
python - <<'PY'
import json
def last(p):
    try: return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e: return None
for t in ("old_1","new_1","old_2","new_2","new_w512"):
    d=last(f"gpurun_out/runs_{t}.json")
    if d: print(t, [(r["Q"], r["search_ms"], r["scan_ms"], r["scan_TF"], r["reduce_ms"], r["plan"]["slices"], r["checksum"]) for r in d["rows"]])
for t in ("old","new","new_w512"):
    d=last(f"gpurun_out/runs_stage_{t}.json")
    if d: print(t, d["one_gpu_ms"], d["one_gpu_stage_ms"].get("topk_scan"), d["per_rank_phase1_ms"], d["phase1_stage_ms"], d["projected_8gpu_ms"], d["projected_speedup"], d["results_equal_unsharded"])
PY
