#!/bin/bash
# rocprofv3 kernel trace of the headline step with the LayerNorm fold on and off (experiment library): per-kernel durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --ingest-images 0 --search-queries 0 --timed-events 0"
for f in 1 2 0; do
  REVO_EXPERIMENTS=1 REVO_LN_FOLD=$f rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_fold$f -o t -- python3 $R/bench.py $ARGS > $R/gpurun_out/trace_fold$f.json 2> $R/gpurun_out/trace_fold$f.err
done
cd $R; find gpurun_out/trace_fold1 gpurun_out/trace_fold2 gpurun_out/trace_fold0 -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, glob, json
for f in (1, 2, 0):
    p = glob.glob(f"gpurun_out/trace_fold{f}/**/*kernel_stats.csv", recursive=True)
    rows = list(csv.DictReader(open(p[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    print("fold", f, json.loads([l for l in open(f"gpurun_out/trace_fold{f}.json") if l.startswith("{")][-1])["ms_per_step"])
    for r in rows[:14]:
        print("   %-90s calls %6s avg %9.1f us total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
