#!/bin/bash
# SQ counters of the attention kernel, product form against the staggered form (REVO_ATTN_STAG=1, experiment library)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in plain stag; do
  OUT=$REPO/gpurun_out/pmc_attn_$v; mkdir -p $OUT
  if [ $v = stag ]; then export REVO_ATTN_STAG=1; else unset REVO_ATTN_STAG; fi
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
      --output-format csv -d $OUT/a -o p -- python3 $REPO/scripts/attn_bench.py > $OUT/a.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
      --output-format csv -d $OUT/b -o p -- python3 $REPO/scripts/attn_bench.py > $OUT/b.log 2>&1
done
cd $REPO
python3 - <<PY
import csv, glob, collections, json
res = {}
for v in ("plain", "stag"):
    acc = collections.defaultdict(float); n = collections.Counter()
    for d in "ab":
        for f in glob.glob("gpurun_out/pmc_attn_%s/%s/**/p_counter_collection.csv" % (v, d), recursive=True):
            for r in csv.DictReader(open(f)):
                if "attn_fwd" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    m = {k: acc[k] / n[k] for k in acc}
    wc = m.get("SQ_WAVE_CYCLES", 1.0)
    res[v] = {"per_dispatch_millions": {k: round(x / 1e6, 3) for k, x in sorted(m.items())},
              "wait_any_pct": round(100 * m.get("SQ_WAIT_ANY", 0) / wc, 1), "wait_inst_pct": round(100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 1),
              "active_inst_pct": round(100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 1),
              "valu_active_pct_of_wave_cycles": round(100 * m.get("SQ_ACTIVE_INST_VALU", 0) / wc, 1),
              "mfma_busy_cycles_millions": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1e6, 2),
              "lds_conflict_pct_of_lds_active": round(100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1), 2)}
print(json.dumps(res))
PY
find $REPO/gpurun_out/pmc_attn_plain $REPO/gpurun_out/pmc_attn_stag -name "*.db" -delete 2>/dev/null || true
