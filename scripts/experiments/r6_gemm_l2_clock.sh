#!/bin/bash
# Round 6, item 6: wall + in-kernel clock (scripts/gemm_l2_clock.py), then FETCH_SIZE of the same launches per XCD walk (separate
# --pmc passes, never combined with tracing).   bash scripts/experiments/r6_gemm_l2_clock.sh  -> gpurun_out/l2clock/
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/l2clock
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REVO_EXPERIMENTS=1 timeout -k 10 200 python3 $REPO/scripts/gemm_l2_clock.py > $OUT/l2_clock.json 2> $OUT/l2_clock.err
for gy in 1 2 4; do
  REVO_EXPERIMENTS=1 L2CLOCK_ONLY=$gy timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_gy$gy -o fetch -- python3 $REPO/scripts/gemm_l2_clock.py > $OUT/fetch_gy$gy.log 2>&1
  python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/fetch_gy$gy/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256q" in r.get("Kernel_Name", "") and r.get("Counter_Name") == "FETCH_SIZE":
            rows.append(float(r["Counter_Value"]))
if rows:
    import statistics
    print("gy=$gy FETCH_SIZE per launch (KiB units x 2 per the gfx950 correction): median", statistics.median(rows), "-> MiB fetched", round(statistics.median(rows) * 2 / 1024, 1), "launches", len(rows))
PY
done | tee $OUT/fetch_summary.txt
find $OUT -name "*.db" -delete 2>/dev/null || true
