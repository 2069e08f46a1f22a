#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   kernel-trace stats, then three separate --pmc passes (never combined with tracing).
# Usage: bash scripts/collect_profiles.sh <tag>     -> gpurun_out/prof_<tag>/{trace,sq,fetch,write}
set -e
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --ingest-images 0"      # includes the 10 000-query batch of configs[3] after the timed steps
PMC_ARGS="$ARGS --no-calibration"       # (the box-speed probes loop by the wall clock: under counters they would only take longer)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ARGS > $OUT/trace.log 2>&1
echo "trace done"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -o sq -- python3 $PMC_ARGS > $OUT/sq.log 2>&1
echo "sq done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $PMC_ARGS > $OUT/fetch.log 2>&1
echo "fetch done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $PMC_ARGS > $OUT/write.log 2>&1
echo "write done"
# keep only the small summaries (the merged-back directory is capped at 64 MiB)
find $OUT -name "*.db" -delete 2>/dev/null || true
find $OUT -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null || true
du -sh $OUT
