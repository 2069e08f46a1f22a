#!/bin/bash
# Stage times of the sharded 10 000-query search rehearsed on one GPU for 2, 4 and 8 shards (BASELINE's metric is a 1/2/4/8
# curve), then the shard's pre-pass size re-swept at 8 shards now that the admission level is an estimate (experiment
# library: REVO_NPRE).   bash scripts/sharded_stage_sweep.sh <tag>   ->  gpurun_out/<tag>_sharded_stage_{2,4,8}.json, <tag>_shard_prepass_sweep.txt
TAG=${1:-r05}
for P in 2 4 8; do
  python scripts/sharded_stage_bench.py 1000000 10000 $P > gpurun_out/${TAG}_sharded_stage_$P.json 2> gpurun_out/${TAG}_sharded_stage_$P.err || echo "P=$P failed"
done
: > gpurun_out/${TAG}_shard_prepass_sweep.txt
for NPRE in 2048 4096 8192 16384; do
  echo "NPRE $NPRE" >> gpurun_out/${TAG}_shard_prepass_sweep.txt
  REVO_EXPERIMENTS=1 REVO_NPRE=$NPRE python scripts/sharded_stage_bench.py 1000000 10000 8 2>/dev/null >> gpurun_out/${TAG}_shard_prepass_sweep.txt
done
python - <<PY
import json
for P in (2,4,8):
    d=json.load(open("gpurun_out/${TAG}_sharded_stage_%d.json"%P))
    print(P, d["one_gpu_ms"], d["per_rank_phase1_ms"], d["phase1_stage_ms"], d["finish_unbounded_ms"], d["merge_ms"], d["projected_8gpu_ms"], d["projected_speedup"], d["results_equal_unsharded"], d["uncertified_queries_sharded"])
for l in open("gpurun_out/${TAG}_shard_prepass_sweep.txt"):
    if l.startswith("NPRE"): print(l.strip())
    elif l.startswith("{"):
        d=json.loads(l); print("   ", d["per_rank_phase1_ms"], d["phase1_stage_ms"], d["projected_8gpu_ms"], d["uncertified_queries_sharded"], d["results_equal_unsharded"])
PY
