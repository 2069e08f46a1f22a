#!/bin/bash
# A/B of the sharded stage bench and the headline search: a saved build (REVO_LIBRARY_PATH) against the tree's.
#   scripts/ab_stage.sh _bisect/r4base
set -e
OLD=${1:-_bisect/r4base}
mkdir -p gpurun_out
for round in 1 2; do
  REVO_LIBRARY_PATH=$OLD/librevo.so python scripts/sharded_stage_bench.py > gpurun_out/ab_stage_old_$round.json
  python scripts/sharded_stage_bench.py > gpurun_out/ab_stage_new_$round.json
done
REVO_LIBRARY_PATH=$OLD/librevo.so python scripts/search_small_q.py > gpurun_out/ab_smallq_old.json
python scripts/search_small_q.py > gpurun_out/ab_smallq_new.json
