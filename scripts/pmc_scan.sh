#!/bin/bash
# SQ counters of the scan kernel with and without its slow path (experiment build): where does the selection's time go?
# Usage (through gpurun): bash scripts/pmc_scan.sh <rows> <queries>
set -e
N=${1:-125000}; Q=${2:-10240}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_scan
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in 0 4 1; do
  SCAN_DBG=$d timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS \
      --output-format csv -d $OUT/d$d -o p -- python3 $REPO/scripts/search_bench.py $N $Q > $OUT/d$d.log 2>&1
done
cd $REPO
python3 - <<PY
import csv, glob, collections
for d in (0, 4, 1):
    f = glob.glob("gpurun_out/pmc_scan/d%d/**/p_counter_collection.csv" % d, recursive=True)
    if not f: print("no csv for", d); continue
    acc = collections.defaultdict(float); n = 0
    for r in csv.DictReader(open(f[0])):
        if "topk_scan256" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    disp = max(1, int(sum(1 for r in csv.DictReader(open(f[0])) if "topk_scan256" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_WAVE_CYCLES")))
    print("DBG=%d dispatch-rows %d" % (d, disp), {k: round(v / disp / 1e6, 3) for k, v in sorted(acc.items())}, "(millions per dispatch)")
PY
find $OUT -name "*.db" -delete 2>/dev/null || true
