#!/bin/bash
# SQ counters of the body attention kernel (scripts/attn_bench.py): issue mix and wait shares.
# Usage (through gpurun): bash scripts/pmc_attn.sh
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_attn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
      --output-format csv -d $OUT/a -o p -- python3 $REPO/scripts/attn_bench.py > $OUT/a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT \
      --output-format csv -d $OUT/b -o p -- python3 $REPO/scripts/attn_bench.py > $OUT/b.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob, collections
for d in "ab":
    f = glob.glob("gpurun_out/pmc_attn/%s/**/p_counter_collection.csv" % d, recursive=True)
    if not f: print("no csv for", d); continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if "attn_fwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(d, {k: round(v / n[k] / 1e6, 3) for k, v in sorted(acc.items())}, "(millions per dispatch)", dict(n))
PY
find $OUT -name "*.db" -delete 2>/dev/null || true
