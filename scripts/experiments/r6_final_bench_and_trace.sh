REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r06f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --ingest-images 0 > $OUT/trace.log 2>&1
find $OUT -name "*.db" -delete 2>/dev/null || true
find $OUT -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null || true
cd $REPO
python bench.py > gpurun_out/r06f_bench.json 2> gpurun_out/r06f_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06f_bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['search_total']['frac'], d['roofline']['search_total']['ms_per_step'], d['search_query_batch']['scan_frac_of_mfma_peak'], d['search_query_batch']['sharded_ms'])
PY
