#!/bin/bash
# Round 6, item 3: SQ counters of the two attention kernels (32x32x16 vs 16x16x32), one --pmc pass each, never with tracing.
#   bash scripts/experiments/r6_attn_shape_pmc.sh  -> gpurun_out/attn_shape_pmc/
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/attn_shape_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in 0 1; do
  REVO_EXPERIMENTS=1 ATTN_SHAPE_ONLY=$k timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/k$k -o sq -- python3 $REPO/scripts/attn_shape_ab.py > $OUT/k$k.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
res = {}
for k, tag in ((0, "32x32x16"), (1, "16x16x32")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/k%d/**/*counter_collection.csv" % k, recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn" in r.get("Kernel_Name", ""):
                acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[tag] = {g: {c: round(sum(v) / len(v)) for c, v in d.items()} for g, d in acc.items()}
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(res))
PY
find $OUT -name "*.db" -delete 2>/dev/null || true
