#!/bin/bash
# the XCD arrangement of the wide body GEMMs (qkv, fc1) against time AND fabric-side fetch.
#   1. micro: each arrangement alone, HIP-event time; 2. the same under rocprofv3 --pmc FETCH_SIZE (own pass);
#   3. in the step: bench.py with the arrangement forced (experiment library), per-class times.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/xcdmap
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for gy in 0 1 2 4 8; do
  python3 $REPO/scripts/gemm_gy_one.py $gy > $OUT/micro_gy$gy.json
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_gy$gy -o f -- python3 $REPO/scripts/gemm_gy_one.py $gy > $OUT/fetch_gy$gy.log 2>&1
  echo "gy $gy done"
done
for gy in 0 1 2 4; do
  REVO_EXPERIMENTS=1 python3 $REPO/bench.py --debug-flags $((gy << 4)) --steps 10 --warmup 3 --no-cpu-baseline --ingest-images 0 --search-queries 0 > $OUT/step_gy$gy.json 2> $OUT/step_gy$gy.err
  echo "step gy $gy done"
done
find $OUT -name "*.db" -delete 2>/dev/null || true
du -sh $OUT
